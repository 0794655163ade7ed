"""PyTorch TunableOp wiring for the (unchanged, PyTorch) coordinate MLPs on the path.

The texture / DINO / SDF fields are plain ``nn.Linear`` stacks whose fp32 GEMMs ([~2e5 x 256] x [256 x 256]) dominate the
step.  rocBLAS' default heuristic picks kernels that reach ~40 TFLOP/s on these shapes on gfx950; PyTorch's own TunableOp
finds hipBLASLt/rocBLAS solutions at 130+ TFLOP/s.  Tuning on line is far too slow for a training loop, so the solutions are
tuned once (``tools/tune_gemms.py``, on the GPU box) and shipped as ``tunableop/gfx950_fp32.csv``; at run time tuning stays OFF
and the file is only looked up.  Shapes repeat because the render path pads its point lists to fixed buckets
(``render.POINT_BUCKET``, ``dmtet.SURFACE_BUCKET``).  If the file does not validate against the running PyTorch/ROCm build,
PyTorch ignores it and the default kernels are used -- slower, same results.
"""
import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop", "gfx950_fp32.csv")


def enable(tuning: bool = False, filename: str = None) -> bool:
    """Turn TunableOp on with the shipped results (``tuning=True`` only inside tools/tune_gemms.py)."""
    path = filename or TUNED_FILE
    if not tuning and not os.path.exists(path):
        return False
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(bool(tuning))
    if tuning:
        torch.cuda.tunable.set_filename(path, insert_device_ordinal=False)  # results are appended as they are found
    elif not torch.cuda.tunable.read_file(path):
        torch.cuda.tunable.enable(False)  # validators (PyTorch / ROCm / hipBLASLt / rocBLAS versions, arch) did not match
        return False
    return True
