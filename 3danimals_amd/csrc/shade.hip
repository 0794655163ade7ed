// Per-point shading arithmetic on gfx950 -- the part of shade() between the field look-ups and the buffers
// (/root/reference/model/render/render.py:71-93): shading normal (renderutils/ops.py:194-227 -> bsdf.py:28-51 with the (0,0,1)
// perturbation), camera-space normal (render.py:73-74), directional light (light.py:186-190).
//
// One thread per covered pixel; everything lives in registers.  The reference runs ~30 elementwise torch kernels forward and
// ~70 backward over dense [B,H,W,3] frames for this; here it is one launch each way over the covered-pixel list, and the backward
// recomputes the forward instead of saving intermediates.
//   in : gb[P,12] (world position | face normal | smooth normal | canonical position, from a3d_gbuffer_fwd),
//        par[B,ncol] rows of the per-image quantities, read through the point -> image index img[P] (or, without img, one row per
//        point [P,ncol]): w2c rotation (9, row-major) | view position (3) | light (5: direction 3, ambient, diffuse) -- ncol 12
//        (no light) or 17,  kd[P,3] (row stride kd_stride floats)
//   out: nrm[P,3] shading normal, shading[P] = amb + diff*max(L.n_cam, 0), shaded[P,3] = shading*kd   (last two only with a light)
#include "a3d_common.h"
#include "shade_common.h"

namespace {

// ``img`` (point -> image, non-decreasing) given: par holds one row per IMAGE and every point reads its image's row (1 KB table, cache
// resident) instead of a [P,ncol] copy of it; null: one row per point.
__global__ __launch_bounds__(256) void sh_fwd_kernel(const float* __restrict__ gb, const float* __restrict__ par, int ncol,
                                                     const long long* __restrict__ img, const float* __restrict__ kd, int kd_stride,
                                                     long long P, int two_sided, float* __restrict__ nrm, float* __restrict__ shading,
                                                     float* __restrict__ shaded, float* __restrict__ clear, int n_clear) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // the backward's per-image row gradient (accumulated there with atomics): one memset less.  Grid-stride: B*ncol may exceed the
    // launch's thread count when only a few pixels are covered
    for (long long i = p; i < n_clear; i += (long long)gridDim.x * blockDim.x) clear[i] = 0.f;
    if (p >= P) return;
    const ShFwd f = sh_forward(gb + 12 * p, par + (long long)ncol * (img ? img[p] : p), ncol, two_sided);
    st3(nrm + 3 * p, f.N);
    if (ncol >= 17) {
        shading[p] = f.shading;
        st3(shaded + 3 * p, ld3(kd + (long long)kd_stride * p) * f.shading);
    }
}

__device__ __forceinline__ float sh_row16_sum(float r) {  // sum over the 16-lane DPP row, in every lane of the row
    r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x141, 0xF, 0xF, true));  // row_half_mirror
    r += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r), 0x140, 0xF, 0xF, true));  // row_mirror
    return r;
}

// With ``img`` the gradient of the per-image rows is reduced here: the points are sorted by image, so a work-group almost always sits
// inside one image -- 16-lane DPP sums, the 16 rows meet in LDS, ONE line-coalesced set of atomics per work-group onto g_par[B,ncol]
// (the [P,ncol] gradient and its segment-sum pass no longer exist).  A work-group that straddles images repeats the reduction once
// per image with the other images' points masked out.  (Letting those work-groups add per point -- 256 x 17 atomics onto the same
// 68 bytes -- cost 120 us for the 15 work-groups concerned: same-address device atomics serialise at ~30 ns each.)
__global__ __launch_bounds__(256) void sh_bwd_kernel(const float* __restrict__ g_nrm, const float* __restrict__ g_shading,
                                                     const float* __restrict__ g_shaded, const float* __restrict__ gb,
                                                     const float* __restrict__ par, int ncol, const long long* __restrict__ img,
                                                     const float* __restrict__ kd, int kd_stride, long long P, int two_sided,
                                                     float* __restrict__ g_gb, float* __restrict__ g_par, float* __restrict__ g_kd) {
    __shared__ float s_red[16][17];
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel id 0 = sh_bwd_kernel)
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < P;
    // the images this work-group's points belong to: img[first] .. img[last] (one, except at the ~B image boundaries of the list) -- read
    // up front: after the per-point work they would be one more dependent round trip
    const long long first = (long long)blockIdx.x * blockDim.x, last = min(first + (long long)blockDim.x, P) - 1;
    const long long b_first = img ? img[first] : 0, b_last = img ? img[last] : 0;
    float gp[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) gp[k] = 0.f;
    long long row = 0;
    if (live) {
        row = img ? img[p] : p;
        const float* pr = par + (long long)ncol * row;
        const ShFwd f = sh_forward(gb + 12 * p, pr, ncol, two_sided);
        V3 gN = g_nrm ? ld3(g_nrm + 3 * p) : V3{0.f, 0.f, 0.f};
        if (ncol >= 17) {
            const V3 k = ld3(kd + (long long)kd_stride * p);
            const V3 gs = g_shaded ? ld3(g_shaded + 3 * p) : V3{0.f, 0.f, 0.f};
            const float g_sh = (g_shading ? g_shading[p] : 0.f) + dot(gs, k);
            st3(g_kd + 3 * p, gs * f.shading);
            const float g_l = f.l >= 0.f ? g_sh * f.diff : 0.f;  // clamp(min=0) passes the gradient at l == 0
            const V3 g_L = f.cam * g_l;
            gp[12] = g_L.x; gp[13] = g_L.y; gp[14] = g_L.z;
            gp[15] = g_sh;
            gp[16] = g_sh * fmaxf(f.l, 0.f);
            const V3 g_cam = f.L * g_l;
            const float inv = 1.f / f.lenq;
            const V3 g_q = f.qq >= SH_EPS_SAFE ? (g_cam - f.cam * dot(f.cam, g_cam)) * inv : g_cam * inv;
            gp[0] = g_q.x * f.N.x; gp[1] = g_q.x * f.N.y; gp[2] = g_q.x * f.N.z;
            gp[3] = g_q.y * f.N.x; gp[4] = g_q.y * f.N.y; gp[5] = g_q.y * f.N.z;
            gp[6] = g_q.z * f.N.x; gp[7] = g_q.z * f.N.y; gp[8] = g_q.z * f.N.z;
            gN = gN + V3{pr[0] * g_q.x + pr[3] * g_q.y + pr[6] * g_q.z, pr[1] * g_q.x + pr[4] * g_q.y + pr[7] * g_q.z,
                         pr[2] * g_q.x + pr[5] * g_q.y + pr[8] * g_q.z};
        }
        // lerp(g, ns, t)
        const V3 g_g = gN * (1.f - f.t);
        V3 g_ns = gN * f.t;
        const float g_t = dot(gN, f.ns - f.g);
        const float g_c = (f.t_raw >= 0.f && f.t_raw <= 1.f) ? g_t / SH_NORMAL_THRESHOLD : 0.f;
        V3 g_v = f.ns * g_c;
        g_ns = g_ns + f.v * g_c;
        const V3 g_n2 = g_ns * f.sigma;
        const V3 g_n1 = normalize_b(g_n2, f.n2, f.len2);
        const V3 g_a = normalize_b(g_n1, f.n1, f.len1);
        const V3 g_d = normalize_b(g_v, f.v, f.lenv);  // d = view - pos
        float4* go = reinterpret_cast<float4*>(g_gb + 12 * p);
        const V3 gg = g_g * f.sigma;
        go[0] = make_float4(-g_d.x, -g_d.y, -g_d.z, gg.x);
        go[1] = make_float4(gg.y, gg.z, g_a.x, g_a.y);
        go[2] = make_float4(g_a.z, 0.f, 0.f, 0.f);
        gp[9] = g_d.x; gp[10] = g_d.y; gp[11] = g_d.z;
    }
    A3D_STAMP(0, 1);
    if (!img) {
        if (live) {
            float* gpr = g_par + (long long)ncol * p;
#pragma unroll
            for (int k = 0; k < 17; ++k)
                if (k < ncol) gpr[k] = gp[k];
        }
        return;
    }
    const long long b_lo = min(b_first, b_last), b_hi = max(b_first, b_last);
    const int lane = threadIdx.x & 63, r16 = threadIdx.x >> 4;
    for (long long bi = b_lo; bi <= b_hi; ++bi) {  // uniform bounds
        const bool mine = live && row == bi;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const float r = sh_row16_sum(mine ? gp[k] : 0.f);
            if ((lane & 15) == 0) s_red[r16][k] = r;
        }
        __syncthreads();
        if ((int)threadIdx.x < ncol) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += s_red[r][threadIdx.x];
            if (t != 0.f) atomicAdd(g_par + bi * ncol + threadIdx.x, t);
        }
        __syncthreads();
    }
    A3D_STAMP(0, 5);
}

}  // namespace

extern "C" int a3d_shade_fwd(const float* gb, const float* par, int ncol, const int64_t* img_or_null, const float* kd, int kd_stride, int64_t P,
                             int two_sided, float* nrm, float* shading, float* shaded, float* g_par_to_clear_or_null, int B,
                             a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && (ncol == 12 || ncol == 17) && (!g_par_to_clear_or_null || B > 0));
    if (P == 0) {
        if (g_par_to_clear_or_null) A3D_HIP(hipMemsetAsync(g_par_to_clear_or_null, 0, sizeof(float) * (size_t)B * ncol, (hipStream_t)stream));
        return A3D_OK;
    }
    A3D_CHECK_ARG(gb && par && nrm);
    A3D_CHECK_ARG(ncol == 12 || (kd && kd_stride >= 3 && shading && shaded));
    hipLaunchKernelGGL(sh_fwd_kernel, dim3(a3d_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, gb, par, ncol, (const long long*)img_or_null, kd,
                       kd_stride, (long long)P, two_sided, nrm, shading, shaded, g_par_to_clear_or_null, g_par_to_clear_or_null ? B * ncol : 0);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_shade_bwd(const float* g_nrm, const float* g_shading, const float* g_shaded, const float* gb, const float* par, int ncol,
                             const int64_t* img_or_null, int B, const float* kd, int kd_stride, int64_t P, int two_sided, float* g_gb, float* g_par,
                             float* g_kd, int g_par_is_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && (ncol == 12 || ncol == 17) && g_par && (!img_or_null || B > 0));
    hipStream_t s = (hipStream_t)stream;
    if (img_or_null && !g_par_is_clear) A3D_HIP(hipMemsetAsync(g_par, 0, sizeof(float) * (size_t)B * ncol, s));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(gb && par && g_gb);
    A3D_CHECK_ARG(ncol == 12 || (kd && kd_stride >= 3 && g_kd));
    hipLaunchKernelGGL(sh_bwd_kernel, dim3(a3d_div_up(P, 256)), dim3(256), 0, s, g_nrm, g_shading, g_shaded, gb, par, ncol,
                       (const long long*)img_or_null, kd, kd_stride, (long long)P, two_sided, g_gb, g_par, g_kd);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(shade)
