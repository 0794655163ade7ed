"""Does a hipBLASLt bias+ReLU epilogue (torch._addmm_activation) beat tuned GEMM + separate ReLU on the MLP shapes?"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import importlib
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
dev = "cuda"
M, K, N = 204800, 256, 256
x = torch.randn(M, K, device=dev, requires_grad=True)
W = (torch.randn(N, K, device=dev) * 0.05).requires_grad_(True)
b = torch.zeros(N, device=dev, requires_grad=True)
g = torch.randn(M, N, device=dev)

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def plain():
    y = torch.relu(torch.nn.functional.linear(x, W, b)); return y
def fused():
    return torch._addmm_activation(b, x, W.t(), use_gelu=False)
def fb(f):
    def run():
        y = f(); y.backward(g); x.grad = W.grad = b.grad = None
    return run

print("same:", torch.allclose(plain(), fused(), atol=1e-5), (plain() - fused()).abs().max().item())
print("untuned  fwd plain %.1f us, fused %.1f us" % (t(plain), t(fused)))
tuning.enable(tuning=False)
print("shipped  fwd plain %.1f us, fused %.1f us" % (t(plain), t(fused)))
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(True)
torch.cuda.tunable.set_filename("/tmp/tune_x.csv", insert_device_ordinal=False)
torch.cuda.tunable.set_max_tuning_duration(30); torch.cuda.tunable.set_max_tuning_iterations(20)
with torch.no_grad():
    fused(); plain()
torch.cuda.tunable.tuning_enable(False)
print("tuned    fwd plain %.1f us, fused %.1f us" % (t(plain), t(fused)))
lin = lambda: torch.nn.functional.linear(x, W, b)
print("linear only %.1f us" % t(lin))
print(open("/tmp/tune_x.csv").read()[-1500:])
