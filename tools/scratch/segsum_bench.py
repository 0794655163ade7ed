import sys, importlib, os, time
sys.path.insert(0, "/root/repo")
import torch
ops = importlib.import_module("3danimals_amd.ops")
P, C, B = 204800, 256, 16
img = (torch.arange(P, device="cuda") * B // P).long()
y = torch.randn(P, C, device="cuda"); g = torch.randn(P, C, device="cuda")
rows = torch.randn(B, C, device="cuda", requires_grad=True)
def run():
    yy = (y.clone()).requires_grad_(True)
    out = ops.rows_add_relu_(yy * 1.0, rows, img)
    return out
out = run()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
bw = lambda: torch.autograd.grad(out, rows, g, retain_graph=True)
print("A3D_SS_ROWS", os.environ.get("A3D_SS_ROWS"), "bwd us %.1f  -> %.2f TB/s" % (t(bw), 12 * P * C / t(bw) / 1e6))
