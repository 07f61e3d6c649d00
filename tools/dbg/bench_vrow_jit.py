"""(Needs tools/dbg/chain_jit_kernel.patch applied to csrc/mlp_infer.hip: the experiment is not in the tree.)
The V-row backward chain (8 x 29487 rows, ReLU-mask chain with dumps) through the layer-end activation program (PSN_CHAIN_JIT=0,
child process) and the just-in-time kernel: time and bit-identity of every dump."""
import os, sys, subprocess, hashlib
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
if len(sys.argv) < 2:
    for mode in ('0', '1', '1a1', '1a2', '1a3'):
        env = dict(os.environ, PSN_CHAIN_JIT=mode[0], PSN_JIT_ABL=(mode[2] if len(mode) > 2 else '0'))
        print(subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, capture_output=True, text=True).stdout.strip())
    sys.exit(0)
import torch
from psnerf_amd import hip, fused
dev = torch.device('cuda')
torch.manual_seed(0)
for Ns in (29487, 3686):
    Q = 8 * Ns
    ws = [torch.randn(256, 126, device=dev) * 0.1] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + \
         [torch.randn(256, 382, device=dev) * 0.05] + [torch.randn(256, 256, device=dev) * 0.06 for _ in range(3)] + [torch.randn(1, 256, device=dev) * 0.06]
    H = [torch.randn(Q, 256, device=dev) for _ in range(8)]
    DZ = [torch.empty(Q, 256, device=dev) for _ in range(8)]
    g = torch.randn(Q, 1, device=dev)
    wl = ws[-1].contiguous()
    chain = fused.pack_relu_bwd(ws, 4)
    fn = lambda: chain(None, Q, a_div=1, a_mod=Q, rank_init=(g, wl), mask=[H[7 - j] for j in range(8)], save=DZ, save_row0=0)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    dig = hashlib.sha1(b''.join(t.cpu().numpy().tobytes() for t in DZ)).hexdigest()[:12]
    print('JIT=%s Ns %6d: %.3f ms  %.1f TF  dumps sha1 %s' % (sys.argv[1], Ns, ms, 2.0 * 7 * 65536 * Q / ms * 1e-9, dig))
