"""3danimals_amd -- MI355X-native reconstruct-and-render hot path for 3DAnimals.

DMTet marching tetrahedra -> linear-blend skinning -> differentiable rasterise /
interpolate / antialias, as hand-written HIP kernels for gfx950 behind a flat C
ABI (``include/a3d.h``, ``csrc/``), wrapped in ``torch.autograd.Function``s
(``ops``) and surfaced through the reference's own module layout
(``model/geometry``, ``model/render``) plus an ``nvdiffrast.torch``-compatible
operator shim (``nvdiffrast/torch``).

The directory name starts with a digit, so import it with
``importlib.import_module("3danimals_amd")`` (or overlay ``model/`` onto the
reference tree).  There is NO CPU fallback: every compute
entry point raises if ``liba3d_hip.so`` is missing or the tensors are not on a
HIP device.
"""
from . import synthetic, tetgrid  # noqa: F401  (pure numpy/torch helpers; no native code needed)

__all__ = ["synthetic", "tetgrid"]
__version__ = "0.6.0"  # round 6 (C ABI: a3d_version() = 404)
