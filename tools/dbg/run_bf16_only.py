import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip, fused
torch.manual_seed(0)
dev = torch.device('cuda')
Ns, L = 29500, 104
ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
     [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(2)] + [torch.randn(1, 256, device=dev) * 0.06]
bs = [torch.randn(w.shape[0], device=dev) * 0.1 for w in ws]
packed = fused.pack_relu_mlp_bf16(ws, bs, 63, 63, 3, hip.OUT_SIGMOID)
ta = hip.pe_encode(torch.rand(Ns, 3, device=dev) - 0.5, 10, 64).to(torch.bfloat16)
tb = hip.pe_encode(torch.nn.functional.normalize(torch.randn(L, 3, device=dev), dim=-1), 10, 64).to(torch.bfloat16)
Q = Ns * L
out = torch.empty(Q, 1, device=dev)
for _ in range(3):
    packed(ta, Q, 1, Ns, tb, Ns, L, out=out)
torch.cuda.synchronize()
print('done', float(out.abs().mean()))
