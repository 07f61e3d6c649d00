"""Does the stage-1 train step allocate device memory from the driver every step (caching-allocator misses), and how long does the
host need to issue a step?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from psnerf_amd.synthetic import stage1_cfg, stage1_batch
from psnerf_amd.stage1 import NeuralNetwork, Renderer, Trainer
from psnerf_amd.optim import FlatAdam
cfg = stage1_cfg('bear', **{'rendering.num_points_in': 96, 'rendering.num_points_out': 32, 'training.n_training_points': int(sys.argv[1]) if len(sys.argv) > 1 else 4096})
batch = stage1_batch(cfg, h=512, w=612, seed=0)
dev = torch.device('cuda:0')
torch.manual_seed(42)
net = NeuralNetwork(cfg); ren = Renderer(net, cfg, device=dev)
tr = Trainer(ren, FlatAdam(net.parameters(), lr=1e-4), cfg, device=dev)
bd = {k: v.to(dev) for k, v in batch.items()}
for i in range(8):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.train_step(bd, it=6000)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print('step %d: host issue %.1f ms, wall %.1f ms, driver allocs %d, frees %d, retries %d, reserved %.2f GB, peak alloc %.2f GB' % (
        i, (t1 - t0) * 1e3, (t2 - t0) * 1e3, st1['num_device_alloc'] - st0['num_device_alloc'], st1['num_device_free'] - st0['num_device_free'],
        st1['num_alloc_retries'] - st0['num_alloc_retries'], st1['reserved_bytes.all.current'] / 2**30, st1['allocated_bytes.all.peak'] / 2**30))
# back-to-back (the host runs ahead of the GPU, as a training loop does)
for rep in range(3):
    st0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.train_step(bd, it=6000)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st1 = torch.cuda.memory_stats()
    print('10 steps back to back: host issue %.1f ms/step, wall %.1f ms/step, driver allocs %d, frees %d, reserved %.2f GB' % (
        (t1 - t0) * 100, (t2 - t0) * 100, st1['num_device_alloc'] - st0['num_device_alloc'], st1['num_device_free'] - st0['num_device_free'],
        st1['reserved_bytes.all.current'] / 2**30))
