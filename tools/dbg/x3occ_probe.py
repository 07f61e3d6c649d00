import sys, time, torch
sys.path.insert(0, '/root/repo')
from tests.helpers import stage1_cfg, stage1_state_dict
from psnerf_amd.stage1 import NeuralNetwork
dev = torch.device('cuda:0')
cfg = stage1_cfg('bear')
net = NeuralNetwork(cfg); net.load_state_dict(stage1_state_dict(cfg, seed=11)); net.to(dev)
g = torch.Generator().manual_seed(0)
for Q in (1000, 128 * 513 + 7, 1 << 22):
    p = ((torch.rand(Q, 3, generator=g) * 2.4 - 1.2)).to(dev)
    with torch.no_grad():
        net.inference_precision = 'fp32'
        a = net.occupancy(p)
        net.inference_precision = 'bf16x6'
        b = net.occupancy(p)
        from oracle import stage1 as o1
    torch.cuda.synchronize()
    print(Q, 'max |x3 - fp32|', float((a - b).abs().max()), 'mean', float((a - b).abs().mean()), 'range', float(a.min()), float(a.max()))
    if Q <= 100000:
        onet = o1.NeuralNetwork(cfg); onet.load_state_dict(stage1_state_dict(cfg, seed=11)); onet.double()
        with torch.no_grad():
            t = onet(p.cpu().double(), only_occupancy=True).reshape(-1, 1).float()
        print('   vs float64 oracle: fp32 %.3e x3 %.3e' % (float((a.cpu() - t).abs().max()), float((b.cpu() - t).abs().max())))
with torch.no_grad():
    for prec in ('fp32', 'bf16x6'):
        net.inference_precision = prec
        for _ in range(3): net.occupancy(p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net.occupancy(p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(prec, '%.3f ms for %d points = %.1f TF (459,008 MAC per point)' % (dt * 1e3, Q, 2 * 459008 * Q / dt / 1e12))
