"""Weight-gradient GEMM [256 x M] x [M x 256], M ~ 2e5: plain mm (as autograd issues it) vs split-K through bmm."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
dev = "cuda"
M, K, N = 204800, 256, 256
x = torch.randn(M, K, device=dev)
g = torch.randn(M, N, device=dev)

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

plain = lambda: g.t().mm(x)
def split(S):
    return lambda: torch.bmm(g.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0)
ref = plain()
print("plain untuned %.1f us" % t(plain))
for S in (4, 8, 16, 32, 64, 128):
    f = split(S)
    print("split S=%d untuned %.1f us  maxdiff %.2e (rel to %.1f)" % (S, t(f), (f() - ref).abs().max().item(), ref.abs().max().item()))
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_filename("/tmp/tune_w.csv", insert_device_ordinal=False)
torch.cuda.tunable.set_max_tuning_duration(30); torch.cuda.tunable.set_max_tuning_iterations(20)
plain()
for S in (8, 16, 32, 64):
    split(S)()
torch.cuda.tunable.tuning_enable(False)
print("plain tuned %.1f us" % t(plain))
for S in (8, 16, 32, 64):
    print("split S=%d tuned %.1f us" % (S, t(split(S))))
print(open("/tmp/tune_w.csv").read()[-1200:])
