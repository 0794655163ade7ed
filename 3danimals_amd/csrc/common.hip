// Error string + version for the flat C ABI (include/a3d.h).
#include <stdarg.h>
#include <stdlib.h>

#include "a3d_common.h"

static thread_local char g_err[512] = "";

void a3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* a3d_last_error(void) { return g_err; }
extern "C" int a3d_version(void) { return 404; /* 0.4.x: round-4 ABI (404, round 6: a3d_dmtet_emit_opts.surf_pts / surf_bucket -- the surface rows of the SDF re-evaluation written by the emit launch --, the scan of a3d_dmtet_count inside its count launch; 403, round 6: the glue-free fused render -- a3d_shade_params / a3d_shade_bwd_rows, a3d_ca_shade.params, the appended fields of a3d_ca_buffer (bg_channels, g_stride, g_channels, vals_rows), a3d_gb_aux + g_tex of the G-buffer calls, a3d_xfm_points_fwd / _bwd, pixel strides of a3d_recon_losses_*, a3d_dmtet_surface_points; 402, round 5: a3d_bw_probe_fill / a3d_bw_probe_read -- the box fingerprint of bench.py -- and a3d_rast_opts.defer_resolve + a3d_rast_resolve / a3d_rast_resolve_gbuffer_fwd; 400: a3d_dmtet_count_ordered -- the culled count pass for grids in any numbering; 401: a3d_dmtet_emit_sparse, a3d_mask_aa_*, the optional groups of a3d_rast_fwd / a3d_dmtet_emit / a3d_composite_aa_fwd in structs); 0.3.x: round-3 ABI (skin_pose, scan-free covered-pixel list and DMTet, topology inside the DMTet emit; 301: culled DMTet count; 302: normals ride in the rasteriser launch; 303: the silhouette analysis rides in the compositor launch; 304: the DMTet emit writes the vertex -> face lists itself; 305: ... and covers only the blocks the count pass listed; 306: skin_pose_bwd without ticket; 307: link derivatives from the forward; 308: speculative DMTet emit) */ }

// experiment knob: only the experiment / profile builds of the library (build.py --exp / --profile) read the environment; the product
// library answers 0, so no switch of a measurement can change what it computes
int a3d_exp(void) {
#if defined(A3D_EXPERIMENT) || defined(A3D_PROFILE)
    const char* e = getenv("A3D_EXP");
    return e ? atoi(e) : 0;
#else
    return 0;
#endif
}

// ---- box fingerprint (bench.py: every line carries what THIS box's memory system gives a frame-sized streaming pass, so that a slow box
// and a regression can be told apart).  One float4 per thread, a work-group per 256 of them -- the compositor's shape.
typedef float a3d_v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void bw_fill_kernel(a3d_v4f* __restrict__ p, long long n4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const a3d_v4f v = {1.f, 2.f, 3.f, 4.f};
    if (i < n4) __builtin_nontemporal_store(v, p + i);
}

__global__ __launch_bounds__(256) void bw_read_kernel(const a3d_v4f* __restrict__ p, long long n4, float* __restrict__ sink) {
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const a3d_v4f v = __builtin_nontemporal_load(p + i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    if (s == 12345.678f) sink[0] = s;  // (never true for the fill's pattern: keeps the loads alive)
}

extern "C" int a3d_bw_probe_fill(float* dst, int64_t n_floats, a3d_stream_t stream) {
    A3D_CHECK_ARG(dst && n_floats > 0 && n_floats % 4 == 0 && ((uintptr_t)dst & 15) == 0);
    const long long n4 = n_floats / 4;
    hipLaunchKernelGGL(bw_fill_kernel, dim3((unsigned)a3d_div_up(n4, 256)), dim3(256), 0, (hipStream_t)stream, (a3d_v4f*)dst, n4);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_bw_probe_read(const float* src, int64_t n_floats, float* sink, a3d_stream_t stream) {
    A3D_CHECK_ARG(src && sink && n_floats > 0 && n_floats % 4 == 0 && ((uintptr_t)src & 15) == 0);
    const long long n4 = n_floats / 4;
    hipLaunchKernelGGL(bw_read_kernel, dim3(8192), dim3(256), 0, (hipStream_t)stream, (const a3d_v4f*)src, n4, sink);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}
