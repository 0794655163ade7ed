import sys, importlib
sys.path.insert(0, "/root/repo")
import torch
ops = importlib.import_module("3danimals_amd.ops")
M, K = 204800, 256
g = torch.randn(M, K, device="cuda"); w = torch.randn(K, 256, device="cuda") * 0.05; x = torch.randn(M, 256, device="cuda")
for _ in range(5):
    ops.gemm_nn_relumask(g, w, None)
    ops.gemm_nn_relumask(g, w, x)
torch.cuda.synchronize()
