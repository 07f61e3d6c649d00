import sys, os, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from psnerf_amd import hip
dev = torch.device('cuda')
M = 236000
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev)
C = torch.empty(M, 256, device=dev); dW = torch.empty(256, 256, device=dev)
for _ in range(2):
    hip.gemm(A, W, out=C, trans_b=True, bias=b, epi=hip.EPI_BIAS_RELU)
    hip.gemm(A, W, out=C)
    hip.gemm(A, C, trans_a=True, out=dW, split_k=230)
torch.cuda.synchronize()
print('done')
