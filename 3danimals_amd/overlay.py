"""Overlay of the MI355X path on the reference tree.

Each replaced reference module keeps its own source and gets ONE line appended at its end:

    from importlib import import_module as _im; _im("3danimals_amd.overlay").apply(globals(), "model.render.util")

``apply`` puts the public names of the mirror module (``3danimals_amd.model...``) on top of the reference module's own: functions
and classes this package implements replace the reference's, everything it does not implement (``util.save_image``,
``light.EnvironmentLight``, ``obj.load_obj``, the texture helpers ...) stays the reference's -- the reference's remaining
``texture.py`` / ``material.py`` / light-loading code keeps working.  Names the mirror only defines as stand-alone placeholders
(``_STANDALONE_ONLY`` in the mirror module) are never exported.
"""
from importlib import import_module

MODULES = ("model.geometry.dmtet", "model.geometry.skinning", "model.geometry.util", "model.render.mesh", "model.render.render",
           "model.render.util", "model.render.light", "model.render.obj", "model.render.renderutils")


def mirror(name):
    assert name in MODULES, f"{name} is not a replaced module ({MODULES})"
    return import_module("3danimals_amd." + name)


def exported_names(name):
    """The functions and classes DEFINED by the mirror module (its imports -- torch, its own ``util`` ... -- and its tuning constants
    stay private: the exported functions keep resolving their globals in the mirror module, not in the reference file)."""
    m = mirror(name)
    skip = set(getattr(m, "_STANDALONE_ONLY", ()))
    out = {}
    for k, v in vars(m).items():
        if k.startswith("_") or k in skip or not (callable(v) or isinstance(v, type)):
            continue
        if str(getattr(v, "__module__", "")).startswith(m.__name__):
            out[k] = v
    return out


def apply(namespace, name):
    """Update ``namespace`` (the ``globals()`` of the reference module ``name``) with this package's implementations."""
    names = exported_names(name)
    namespace.update(names)
    namespace["__a3d_overlay__"] = sorted(names)
    return namespace
