"""csrc/gemm.hip (fp32 MFMA, ReLU adjoint in the epilogue) against rocBLAS mm + threshold_backward on the fields' shape."""
import sys, importlib, time
sys.path.insert(0, "/root/repo")
import torch
ops = importlib.import_module("3danimals_amd.ops")
tuning = importlib.import_module("3danimals_amd.gemm_tuning")
tuning.enable()
M, K = 204800, 256
g = torch.randn(M, K, device="cuda"); w = torch.randn(K, 256, device="cuda") * 0.05; x = torch.randn(M, 256, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ref = g.mm(w); refm = torch.ops.aten.threshold_backward(ref, x, 0)
out = ops.gemm_nn_relumask(g, w, x); outp = ops.gemm_nn_relumask(g, w, None)
print("plain maxdiff %.3e of %.3e; masked maxdiff %.3e; mask pattern equal %s" % ((outp - ref).abs().max().item(), ref.abs().max().item(), (out - refm).abs().max().item(), bool(((out != 0) == (refm != 0)).all())))
tm = t(lambda: g.mm(w)); tt = t(lambda: torch.ops.aten.threshold_backward(ref, x, 0)); tf = t(lambda: ops.gemm_nn_relumask(g, w, x)); tp = t(lambda: ops.gemm_nn_relumask(g, w, None))
fl = 2 * M * K * 256
print("rocBLAS mm %.1f us (%.1f TF/s) + threshold %.1f us = %.1f us | a3d masked %.1f us (%.1f TF/s) | a3d plain %.1f us (%.1f TF/s)" % (tm, fl / tm / 1e6, tt, tm + tt, tf, fl / tf / 1e6, tp, fl / tp / 1e6))
