from . import dmtet, skinning, util  # noqa: F401
