import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_amd import ops
dev = torch.device("cuda:0")
M, K, N = 25088, 640, 3840
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
for _ in range(3): ops.pw_fwd(A, W, N, want_stats=True)
G = torch.randn(M, N, device=dev)
for _ in range(3): ops.pw_dgrad(G, W, K)
dW = torch.zeros(N, K, device=dev)
for _ in range(3): ops.pw_wgrad(G, A, dW)
torch.cuda.synchronize()
