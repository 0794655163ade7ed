from . import light, mesh, render, util  # noqa: F401
from . import renderutils  # noqa: F401
