"""ctypes binding of liba3d_hip.so (include/a3d.h).  No fallback: if the library is missing we say so and stop."""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (A3D_LIB: another build of the same ABI -- tools/kernel_phases.py points it at the instrumented liba3d_hip_prof.so)
LIB_PATH = os.environ.get("A3D_LIB") or os.path.join(_HERE, "lib", "liba3d_hip.so")

_c_int, _c_float, _c_size_t, _p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/a3d.h (tests/test_host_cpu.py::test_library_exports_every_declared_symbol checks)
SIGNATURES = {
    "a3d_version": (_c_int, []),
    "a3d_last_error": (ctypes.c_char_p, []),
    "a3d_bw_probe_fill": (_c_int, [_p, ctypes.c_int64, _p]),
    "a3d_bw_probe_read": (_c_int, [_p, ctypes.c_int64, _p, _p]),
    "a3d_dmtet_scratch_bytes": (_c_size_t, [_c_int, _c_int]),
    "a3d_dmtet_count": (_c_int, [_p, _p, _p, _c_int, _c_int, _p, _p, _p, _c_int, _c_int, _p, _p, _c_int, _p, _c_int, _p]),
    "a3d_dmtet_count_ordered": (_c_int, [_p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _c_int, _p, _c_int, _p]),
    "a3d_dmtet_word_group_slots": (_c_int, []),
    "a3d_dmtet_word_group_bits": (_c_int, []),
    "a3d_dmtet_block_items": (_c_int, []),
    "a3d_dmtet_vertex_scratch_bytes": (_c_size_t, [_c_int]),
    "a3d_dmtet_gather_rows": (_c_int, [_p, _p, ctypes.c_int64, ctypes.c_int64, _c_int, _p, _p]),
    "a3d_dmtet_emit": (_c_int, [_p, _p, _p, _p, _c_int, _c_int, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p]),
    "a3d_dmtet_emit_sparse": (_c_int, [_p, _p, _p, _p, _c_int, _c_int, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, _p]),
    "a3d_dmtet_bwd": (_c_int, [_p, _p, _p, _p, _p, _c_int, _c_int, _p, _p, _c_int, _p]),
    "a3d_skin_fwd": (_c_int, [_p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_float, _p, _p, _p, _p]),
    "a3d_skin_bwd": (_c_int, [_p, _p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_float, _p, _p, _c_int, _p]),
    "a3d_bone_transforms_fwd": (_c_int, [_p, _c_int, _p, _p, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_bone_transforms_bwd": (_c_int, [_p, _p, _c_int, _p, _p, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_estimate_bones": (_c_int, [_p, _p]),
    "a3d_skin_pose_max_bones": (_c_int, []),
    "a3d_skin_pose_products_floats": (_c_size_t, [_c_int, _c_int]),
    "a3d_skin_pose_fwd": (_c_int, [_p, _c_int, _p, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_float, _p, _p, _p, _p, _p]),
    "a3d_skin_pose_bwd": (_c_int, [_p, _p, _c_int, _p, _c_int, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_float, _p, _p, _p, _c_int, _p]),
    "a3d_normals_adjacency": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_normals_fwd": (_c_int, [_p, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _c_int, _p]),
    "a3d_normals_fwd_pair": (_c_int, [_p, _c_int, _p, _c_int, _p, _p, _p, _c_int, _c_int, _p, _p, _p, _p, _c_int, _p]),
    "a3d_normals_bwd": (_c_int, [_p, _c_int, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _c_int, _p, _p]),
    "a3d_shade_fwd": (_c_int, [_p, _p, _c_int, _p, _p, _c_int, ctypes.c_int64, _c_int, _p, _p, _p, _p, _c_int, _p]),
    "a3d_shade_bwd": (_c_int, [_p, _p, _p, _p, _p, _c_int, _p, _c_int, _p, _c_int, ctypes.c_int64, _c_int, _p, _p, _p, _c_int, _p]),
    "a3d_shade_bwd_rows": (_c_int, [_p, _p, _p, _p, _p, ctypes.c_int64, _p, _c_int, ctypes.c_int64, _c_int, _p, _p, _c_int, ctypes.c_int64, _p]),
    "a3d_xfm_points_fwd": (_c_int, [_p, _c_int, _p, _c_int, _c_int, _c_int, _p, _p, _p]),
    "a3d_flow_delta_fwd": (_c_int, [_p, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_flow_delta_bwd": (_c_int, [_p, _c_int, _p, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_xfm_points_bwd": (_c_int, [_p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _p, _p, _c_int, _p, _c_int, _p, _c_int, _p, _c_int, _p]),
    "a3d_cover_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "a3d_cover_blocks": (_c_int, [_c_int, _c_int, _c_int]),
    "a3d_cover_groups": (_c_int, [_c_int, _c_int, _c_int]),
    "a3d_cover_group_stride": (_c_int, []),
    "a3d_cover_count": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_cover_emit": (_c_int, [_p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_rast_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "a3d_rast_bins_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "a3d_rast_resolve": (_c_int, [_p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_rast_resolve_gbuffer_fwd": (_c_int, [_p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, ctypes.c_int64, _p, _p, _p, _p, _p,
                                              _c_int, _p, _p, _c_int, _p, _p, _p, _p]),
    "a3d_rast_fwd": (_c_int, [_p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _c_int, _p, _p]),
    "a3d_dispatch_order_probe": (_c_int, [_c_int, _c_int, _p, _p]),
    "a3d_rast_bwd": (_c_int, [_p, _p, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_interp_fwd": (_c_int, [_p, _c_int, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_interp_bwd": (_c_int, [_p, _p, _c_int, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "a3d_mesh_topology": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _p, _c_int, _p]),
    "a3d_mesh_topology_finalize_max_vertices": (_c_int, []),
    "a3d_mesh_topology_finalize": (_c_int, [_p, _c_int, _c_int, _p, _p, _p, _p, _c_int, _p]),
    "a3d_gbuffer_fwd": (_c_int, [_p, _p, _p, ctypes.c_int64, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _c_int, _p, _p, _p, _p]),
    "a3d_cover_gbuffer_fwd": (_c_int, [_p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, ctypes.c_int64, _p, _p, _p, _p, _p, _c_int, _p, _p, _c_int,
                                       _p, _p, _p, _p]),
    "a3d_gbuffer_bwd": (_c_int, [_p, _p, _p, _p, ctypes.c_int64, _p, _p, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _c_int,
                                 _c_int, _p, _c_int, _p, _p, _p]),
    "a3d_gbuffer_prior_grad": (_c_int, [_p, _c_int, _c_int, _p, _p]),
    "a3d_gemm_nn_relumask": (_c_int, [_p, _p, _p, ctypes.c_int64, _c_int, _c_int, _p, _p]),
    "a3d_harmonic_embed_fwd": (_c_int, [_p, _p, _c_int, _c_int, _c_int, ctypes.c_int64, _p, _p]),
    "a3d_harmonic_embed_bwd": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, ctypes.c_int64, _p, _p]),
    "a3d_recon_losses_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "a3d_recon_losses_mask_bytes": (_c_size_t, [_c_int, _c_int, _c_int]),
    "a3d_recon_losses_columns": (_c_int, []),
    "a3d_recon_losses_fwd": (_c_int, [_p, _p, _c_int, _c_int, _p, _p, _p, _p, _p, ctypes.c_int64, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_recon_losses_bwd": (_c_int, [_p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p, ctypes.c_int64, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_flow_loss_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "a3d_flow_loss_fwd": (_c_int, [_p, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p]),
    "a3d_flow_loss_bwd": (_c_int, [_p, _p, _p, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_rows_segsum": (_c_int, [_p, _p, ctypes.c_int64, _c_int, _c_int, _p, _p]),
    "a3d_rows_add_relu_fwd": (_c_int, [_p, _p, _p, ctypes.c_int64, _c_int, _c_int, _p]),
    "a3d_rows_add_relu_bwd": (_c_int, [_p, _p, _p, ctypes.c_int64, _c_int, _c_int, _p, _p, _p]),
    "a3d_aa_hash_bytes": (_c_size_t, [_c_int]),
    "a3d_aa_shards": (_c_int, []),
    "a3d_aa_capacity": (_c_int, [_c_int, _c_int, _c_int]),
    "a3d_aa_topology": (_c_int, [_p, _c_int, _c_int, _p, _p, _p]),
    "a3d_aa_topology_from_lists": (_c_int, [_p, _c_int, _p, _p, _p, _c_int, _p]),
    "a3d_aa_analyze": (_c_int, [_p, _p, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _c_int, _p, _c_int, _p, _p, _c_int, _p]),
    "a3d_aa_fwd": (_c_int, [_p, _c_int, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_aa_bwd": (_c_int, [_p, _p, _c_int, _p, _p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "a3d_composite_aa_fwd": (_c_int, [_p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
    "a3d_mask_aa_fwd": (_c_int, [_p, _c_int, _p, _c_int, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p]),
    "a3d_mask_aa_bwd": (_c_int, [_p, _p, _c_int, _p, _c_int, _p, _p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _c_int, _p]),
    "a3d_composite_aa_bwd": (_c_int, [_p, _p, _p, ctypes.c_int64, _p, _p, _p, _c_int, _p, _c_int, _p, _c_int, _c_int, _c_int, _c_int, _c_int, _p, _p, _p]),
}



class DmtetOrder(ctypes.Structure):
    """a3d_dmtet_order of include/a3d.h (field for field; tests/test_host_cpu.py compares the two)."""

    _fields_ = [("size", ctypes.c_uint32), ("group_slots", ctypes.c_int32), ("vertex_of_rank", _p), ("edges_ranked", _p), ("edge_of_row", _p),
                ("tets_ranked", _p), ("tet_of_row", _p), ("edge_groups", _p), ("tet_groups", _p)]


class RastOpts(ctypes.Structure):
    """a3d_rast_opts of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("lists_stride", ctypes.c_int32), ("prev_rast", _p), ("cover_scratch", _p), ("aa_screen", _p),
                ("aa_count", _p), ("topo_off", _p), ("topo_adj", _p), ("topo_opp", _p), ("normals_v_a", _p), ("normals_v_b", _p),
                ("normals_off", _p), ("normals_adj", _p), ("normals_acc_a", _p), ("normals_a", _p), ("normals_acc_b", _p), ("normals_b", _p),
                ("normals_B_a", ctypes.c_int32), ("normals_B_b", ctypes.c_int32), ("bins", _p), ("bin_cap", ctypes.c_int32), ("bins_clean", ctypes.c_int32),
                ("defer_resolve", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class AaRide(ctypes.Structure):
    """a3d_aa_ride of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("clip_batch", ctypes.c_int32), ("rast", _p), ("screen", _p), ("tri", _p), ("opp", _p), ("off", _p),
                ("adj", _p), ("V", ctypes.c_int32), ("F", ctypes.c_int32), ("lists_stride", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class CaBuffer(ctypes.Structure):
    """a3d_ca_buffer of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("C", ctypes.c_int32), ("vals", _p), ("bg", _p), ("out", _p), ("g_out", _p), ("g_vals", _p),
                ("bg_batch", ctypes.c_int32), ("reserved", ctypes.c_int32), ("bg_channels", ctypes.c_int32), ("g_stride", ctypes.c_int32),
                ("g_channels", ctypes.c_int32), ("out_channels", ctypes.c_int32), ("vals_rows", ctypes.c_int64)]


class CaShade(ctypes.Structure):
    """a3d_ca_shade of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("kd_stride", ctypes.c_int32), ("gb", _p), ("par", _p), ("kd", _p), ("clear", _p),
                ("n_clear", ctypes.c_int32), ("two_sided", ctypes.c_int32), ("params", _p), ("shaded_out", _p)]


class ShadeParams(ctypes.Structure):
    """a3d_shade_params of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("rot_row_stride", ctypes.c_int32), ("rot", _p), ("view", _p), ("light", _p),
                ("rot_image_stride", ctypes.c_int64), ("view_image_stride", ctypes.c_int64), ("light_image_stride", ctypes.c_int64)]


class EstimateBonesArgs(ctypes.Structure):
    """a3d_estimate_bones_args of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("N", ctypes.c_int32), ("pos", _p), ("bones", _p), ("nearest", _p), ("ok", _p), ("V", ctypes.c_int32),
                ("n_body", ctypes.c_int32), ("n_leg", ctypes.c_int32), ("body_mode_y_plus", ctypes.c_int32), ("use_y_threshold", ctypes.c_int32),
                ("y_threshold", ctypes.c_float), ("attach", ctypes.c_int32 * 4), ("blend", ctypes.c_float * 17), ("ramp", ctypes.c_float * 9), ("workspace", _p)]


class GbAux(ctypes.Structure):
    """a3d_gb_aux of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("reserved", ctypes.c_int32), ("tex_out", _p), ("img_out", _p), ("rows", ctypes.c_int64),
                ("pad_to", ctypes.c_int64)]


class DmtetEmitOpts(ctypes.Structure):
    """a3d_dmtet_emit_opts of include/a3d.h."""

    _fields_ = [("size", ctypes.c_uint32), ("Nv", ctypes.c_int32), ("vertex_scratch", _p), ("surf_idx", _p), ("g_sdf_to_clear", _p), ("tri32", _p),
                ("topo_count", _p), ("topo_adj", _p), ("device_counts", _p), ("n_surf", ctypes.c_int32), ("topo_stride", ctypes.c_int32),
                ("use_block_lists", ctypes.c_int32), ("n_edge_blocks_listed", ctypes.c_int32), ("n_tet_blocks_listed", ctypes.c_int32),
                ("surf_bucket", ctypes.c_int32), ("surf_pts", _p)]


ABI_VERSION = 404  # a3d_version() of the library these signatures belong to (include/a3d.h)
_lib = None


class A3DError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise A3DError(
                f"{LIB_PATH} is missing: build it with `python 3danimals_amd/csrc/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback for the HIP hot path."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == ABI drift
            fn.restype, fn.argtypes = res, args
        if handle.a3d_version() != ABI_VERSION:  # same symbols, other argument lists: a stale build must not be called
            raise A3DError(f"{LIB_PATH} has ABI version {handle.a3d_version()}, this package binds version {ABI_VERSION}: rebuild it "
                           "with `python 3danimals_amd/csrc/build.py`")
        _lib = handle
    return _lib


class KernelTimer:
    """Optional HIP-event timing of C-ABI entry points, on the stream they are launched on.

    ``with KernelTimer() as t: ...`` brackets every ``call()`` with a pair of events recorded on the current torch
    stream (the stream the kernels go to); ``t.summary()`` synchronises and returns name -> (launches, mean ms).
    Used by bench.py for the live roofline figure; off (zero overhead) otherwise.
    """

    active = None

    def __init__(self):
        self.records = {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.records.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


# ---- guard mode (A3D_GUARD=1|2, a debugging aid like A3D_SYNC_CALLS): every buffer ops.py allocates for the library -- outputs, scratch,
# lists, cached key / counter buffers -- sits between two 256-byte canaries, and after EVERY entry point the stream is drained and all live
# canaries are compared (a one-element overrun of a data-dependent capacity is then reported at the call that did it, by name, with the
# buffer's shape and the line that allocated it).  Level 2 also POISONS the payload (float: NaN, integer: 0x7f7f7f7f, byte: 0x7f), so that
# a kernel that reads what nobody wrote -- the tail of a speculative capacity, a list slot past its count -- computes NaN or gathers
# far out of range on every run instead of once in fifty.  tests/test_gpu_parity.py::test_guard_mode_* run the workloads and a shape fuzz
# under it; the product path never pays for it (one flag test per call).
GUARD = int(os.environ.get("A3D_GUARD", "0") or 0)
GUARD_BYTES = 256
_GUARD_BYTE = 0xA5
_guard_live = []  # (storage of the padded buffer, payload bytes, description)
guard_stats = dict(checks=0, buffers_checked=0, allocations=0)


def set_guard(level):
    """Switch guard mode at run time (tests): patches ops' allocation calls on, or off again.  -> previous level."""
    global GUARD
    import importlib

    prev, GUARD = GUARD, int(level)
    ops = importlib.import_module(__package__ + ".ops")
    ops.torch = GuardedTorch(torch) if GUARD else torch
    if not GUARD:
        _guard_live.clear()
    return prev


def guarded_empty(shape, dtype, device, zero=False, site=""):
    dtype = dtype or torch.get_default_dtype()
    shape = tuple(int(v) for v in shape)
    n = 1
    for v in shape:
        n *= v
    item = torch.empty((), dtype=dtype).element_size()
    nbytes = n * item
    pad = (-nbytes) % 16
    base = torch.empty(GUARD_BYTES + nbytes + pad + GUARD_BYTES, dtype=torch.uint8, device=device)
    base[:GUARD_BYTES] = _GUARD_BYTE
    base[GUARD_BYTES + nbytes:] = _GUARD_BYTE
    t = base[GUARD_BYTES:GUARD_BYTES + nbytes].view(dtype).view(shape)
    if zero:
        t.zero_()
    elif GUARD >= 2 and n:
        if dtype.is_floating_point:
            t.fill_(float("nan"))
        elif dtype == torch.bool:
            t.fill_(True)
        elif item == 1:
            t.fill_(0x7F)
        else:
            t.fill_(0x7F7F7F7F if item >= 4 else 0x7F7F)
    # (the registry holds the STORAGE, not a weak reference to ``base``: the Python object of a view's base dies with this frame although
    # the memory lives on through the view; an entry is dropped once nothing but the registry references its storage)
    _guard_live.append((base.untyped_storage(), nbytes, f"{dtype} {list(shape)} allocated at {site}"))
    guard_stats["allocations"] += 1
    return t


class GuardedTorch:
    """Stands in for the ``torch`` module inside ops.py while guard mode is on: empty / empty_like / zeros hand out guarded buffers,
    everything else is torch's."""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        return getattr(self._real, name)

    @staticmethod
    def _site():
        import sys

        f = sys._getframe(2)
        return f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno}"

    @staticmethod
    def _shape(size):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            return tuple(size[0])
        return tuple(size)

    def empty(self, *size, dtype=None, device=None, **kw):
        if device is None or torch.device(device).type != "cuda":
            return self._real.empty(*size, dtype=dtype, device=device, **kw)
        return guarded_empty(self._shape(size), dtype, device, site=self._site())

    def zeros(self, *size, dtype=None, device=None, **kw):
        if device is None or torch.device(device).type != "cuda":
            return self._real.zeros(*size, dtype=dtype, device=device, **kw)
        return guarded_empty(self._shape(size), dtype, device, zero=True, site=self._site())

    def empty_like(self, t, dtype=None, **kw):
        if not t.is_cuda:
            return self._real.empty_like(t, dtype=dtype, **kw)
        return guarded_empty(t.shape, dtype or t.dtype, t.device, site=self._site())


def guard_check(where):
    """Drain the stream and compare every live canary; raises A3DError naming the call, the buffer and the side."""
    torch.cuda.synchronize()
    live = [e for e in _guard_live if torch._C._storage_Use_Count(e[0]._cdata) > 1]  # (1 = the registry's own reference: the buffer is gone)
    _guard_live[:] = live
    whole = lambda st: torch.empty(0, dtype=torch.uint8, device=st.device).set_(st)
    parts = []
    for st, nbytes, what in live:
        base = whole(st)
        parts.append(base[:GUARD_BYTES])
        parts.append(base[GUARD_BYTES + nbytes:])
    guard_stats["checks"] += 1
    guard_stats["buffers_checked"] += len(live)
    if not parts:
        return
    by_device = {}
    for q in parts:
        by_device.setdefault(q.device, []).append(q)
    ok = all(bool((torch.cat(qs) == _GUARD_BYTE).all()) for qs in by_device.values())
    if ok:
        return
    bad = []
    for st, nbytes, what in live:
        base = whole(st)
        head, tail = bool((base[:GUARD_BYTES] == _GUARD_BYTE).all()), bool((base[GUARD_BYTES + nbytes:] == _GUARD_BYTE).all())
        if not head:
            bad.append(f"{what}: bytes BEFORE the buffer overwritten ({int((base[:GUARD_BYTES] != _GUARD_BYTE).sum())} of {GUARD_BYTES})")
        if not tail:
            t = base[GUARD_BYTES + nbytes:]
            first = int((t != _GUARD_BYTE).nonzero()[0])
            bad.append(f"{what}: bytes AFTER the buffer overwritten ({int((t != _GUARD_BYTE).sum())}, first {first} bytes past its end)")
            t.fill_(_GUARD_BYTE)
        if not head:
            base[:GUARD_BYTES] = _GUARD_BYTE
    raise A3DError(f"A3D_GUARD: buffer overrun detected after {where}: " + "; ".join(bad))


_HELD = []  # float32 / contiguous copies made for ONE call: kept alive until that call has been enqueued (see f32h)
_SYNC_EVERY_CALL = os.environ.get("A3D_SYNC_CALLS", "0") == "1"
_TRACE_FILE = open(os.environ["A3D_TRACE_CALLS"], "w") if os.environ.get("A3D_TRACE_CALLS") else None  # debugging aid: every entry point's name, flushed BEFORE the call (a device fault that aborts the process leaves the culprit as the last line; use with A3D_SYNC_CALLS=1)


def call(name: str, *args, tag: str = ""):
    """Invoke an int-returning entry point and raise on a non-zero status (``tag`` only labels KernelTimer records)."""
    timer = KernelTimer.active
    if _TRACE_FILE is not None:
        _TRACE_FILE.write(f"{name}{tag} {args}\n")
        _TRACE_FILE.flush()
    if timer is not None:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    rc = getattr(lib(), name)(*args)
    del _HELD[:]
    if GUARD and rc == 0:
        guard_check(name + tag)
    if _SYNC_EVERY_CALL:  # debugging aid (A3D_SYNC_CALLS=1): a device fault is reported at the entry point that caused it
        try:
            torch.cuda.synchronize()
        except Exception as e:
            raise A3DError(f"{name}{tag}: device fault surfaced after this call: {e}") from e
    if timer is not None:
        b.record()
        timer.records.setdefault(name + tag, []).append((a, b))
    if rc != 0:
        raise A3DError(f"{name} failed ({rc}): {lib().a3d_last_error().decode()}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- deferred device-side checks.  A data-dependent failure the reference reports from Python (estimate_bones: "no vertex in a leg
# quadrant", where the reference drops into pdb, skinning.py:183) must not cost the path a host synchronisation of its own -- and must
# not be a torch._assert_async either: on ROCm a failed device assert traps the wave and the PROCESS dies with "HSA_STATUS_ERROR_EXCEPTION:
# an HSAIL operation resulted in a hardware exception", no message, no Python traceback (tools/assert_async_probe.py; round 4's one
# unexplained abort of a 500-step Fauna run).  Instead the condition stays on the device as a flag and rides in the NEXT read-back the
# path performs anyway (the DMTet counts, the covered-pixel sums): one transfer, and a Python exception with the original message.
_deferred = []  # (0-dim bool tensor on the device: True = fine, message)


def defer_check(ok, message):
    _deferred.append((ok.detach().reshape(()), message))
    if len(_deferred) > 256:  # (nobody read anything back for a long time: look now)
        poll_deferred()


def _raise_deferred(items, flags):
    bad = [msg for (_, msg), f in zip(items, flags) if not f]
    if bad:
        raise A3DError("device-side check failed (reported at the next host read-back of the path): " + "; ".join(sorted(set(bad))))


def poll_deferred():
    """Read every pending flag now (one small transfer per device that has any).  Training loops call this once per iteration, before
    the optimiser step (pipeline.SyntheticScene.step does): a step whose skeleton was estimated from an empty leg
    quadrant must not update the weights, and a step without any read-back of its own (evaluation, a mask-only render) must not keep
    the failure pending.  Also registered with atexit."""
    if _deferred:
        items = list(_deferred)
        del _deferred[:]
        by_device = {}
        for it in items:
            by_device.setdefault(it[0].device, []).append(it)
        for group in by_device.values():
            _raise_deferred(group, torch.stack([ok.to(torch.bool) for ok, _ in group]).cpu().tolist())


def _poll_at_exit():
    try:
        poll_deferred()
    except A3DError as e:  # (cannot raise usefully at interpreter exit: say it)
        import sys

        print(f"3danimals_amd: {e}", file=sys.stderr)
    except Exception:
        pass


import atexit  # noqa: E402

atexit.register(_poll_at_exit)


def read_back(t):
    """t.cpu() for a read-back the path performs anyway; pending deferred checks on the same device travel in the same transfer."""
    items = [it for it in _deferred if it[0].device == t.device]
    if not items:
        return t.cpu()
    _deferred[:] = [it for it in _deferred if it[0].device != t.device]
    flat = torch.cat([t.reshape(-1), torch.stack([ok.to(t.dtype) for ok, _ in items])]).cpu()
    _raise_deferred(items, [bool(v) for v in flat[t.numel():].tolist()])
    return flat[: t.numel()].reshape(t.shape)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def require_device(*tensors, what: str = "op"):
    """Every tensor on the GPU, and on torch's CURRENT device: the kernels are enqueued on the current device's current stream."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise A3DError(f"{what}: tensor on {t.device}; the HIP hot path only runs on a ROCm device (no CPU fallback)")
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise A3DError(f"{what}: tensor on {t.device} while the current device is cuda:{cur}; wrap the call in torch.cuda.device(...)")


def fp32_region(fn):
    """The reference calls this path inside torch.autocast when mixed_precision is set (Trainer.py:208-218, AnimalModel.py:382-444) and
    forces .float() where it enters nvdiffrast (render.py:265,292).  Functions of the path that do their arithmetic in torch (camera /
    bone algebra: a handful of small matmuls) are wrapped with this: under autocast they run with autocast off on float32 copies of
    their half-precision tensor arguments, so that the path returns the same float32 results with and without mixed precision --
    a bf16 clip-space matmul would move vertices by 1e-2 of a pixel.  No effect (one flag test) outside autocast."""
    import functools

    def cast(x):
        if torch.is_tensor(x):
            return x.float() if x.is_floating_point() and x.dtype in (torch.float16, torch.bfloat16) else x
        if isinstance(x, (list, tuple)):
            vals = [cast(v) for v in x]
            return type(x)(*vals) if hasattr(x, "_fields") else type(x)(vals)  # (a namedtuple's constructor takes its fields one by one)
        if isinstance(x, dict):
            return {k: cast(v) for k, v in x.items()}
        return x

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        if not torch.is_autocast_enabled():
            return fn(*args, **kwargs)
        with torch.autocast("cuda", enabled=False):
            return fn(*cast(args), **cast(kwargs))

    return wrapped


def f32c(t):
    """float32, contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def f32h(t):
    """f32c for an argument that is only passed on as a raw pointer (``ptr(f32h(g))`` inside a ``call(...)``): a copy made here stays
    referenced until that call has been enqueued.  (Without that the temporary dies as soon as ptr() has returned its address, and the
    caching allocator may hand the same block to the NEXT temporary of the same argument list -- two pointers, one buffer.)"""
    u = f32c(t)
    if u is not t:
        _HELD.append(u)
    return u
