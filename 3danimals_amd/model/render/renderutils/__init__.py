"""The two renderutils entry points that sit on the hot path (reference model/render/renderutils/ops.py):
``xfm_points`` (:515-531) and ``prepare_shading_normal`` (:194-227), both called there with use_python=True
(render.py:72,278).  The reference's 27-kernel CUDA plugin behind use_python=False is never launched by any
config and is not provided."""
from .ops import prepare_shading_normal, xfm_points, xfm_vectors  # noqa: F401
