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

// where a point's per-image rows are: ``img`` (point -> image, non-decreasing; or, with img_div = H*W, the covered-pixel list itself: flat
// pixel index / pixels per image) given: one row per IMAGE and every point reads its image's rows (a 1 KB table, cache resident) instead
// of a [P,ncol] copy of them; null: one row per point.
__device__ __forceinline__ long long sh_image_of(const long long* __restrict__ img, unsigned img_div, long long p) {
    if (!img) return p;
    return img_div > 1u ? (long long)((unsigned)img[p] / img_div) : img[p];
}

__global__ __launch_bounds__(256) void sh_fwd_kernel(const float* __restrict__ gb, const ShPar par, const long long* __restrict__ img,
                                                     unsigned img_div, const float* __restrict__ kd, int kd_stride, long long P, int two_sided,
                                                     float* __restrict__ nrm, float* __restrict__ shading, float* __restrict__ shaded,
                                                     float* __restrict__ clear, int n_clear) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // the backward's per-image row gradient (accumulated there with atomics): one memset less.  Grid-stride: B*ncol may exceed the
    // launch's thread count when only a few pixels are covered
    for (long long i = p; i < n_clear; i += (long long)gridDim.x * blockDim.x) clear[i] = 0.f;
    if (p >= P) return;
    const ShFwd f = sh_forward(gb + 12 * p, par.row(sh_image_of(img, img_div, p)), two_sided);
    st3(nrm + 3 * p, f.N);
    if (par.light) {
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

// where the gradient of the per-image rows goes: the layout of ShPar, writable
struct ShParGrad {
    float *rot, *view, *light;
    long long rot_img, view_img, light_img;
    int rs;
    __device__ __forceinline__ float* at(long long b, int k) const {  // component k of 17: rotation 0..8 | view 9..11 | light 12..16
        if (k < 9) return rot + rot_img * b + rs * (k / 3) + (k % 3);
        if (k < 12) return view + view_img * b + (k - 9);
        return light + light_img * b + (k - 12);
    }
};

// With ``img`` the gradient of the per-image rows is reduced here: the points are sorted by image, so a work-group almost always sits
// inside one image -- 16-lane DPP sums, the 16 rows meet in LDS, ONE line-coalesced set of atomics per work-group onto the image's rows
// (the [P,ncol] gradient and its segment-sum pass no longer exist).  A work-group that straddles images repeats the reduction once
// per image with the other images' points masked out.  (Letting those work-groups add per point -- 256 x 17 atomics onto the same
// 68 bytes -- cost 120 us for the 15 work-groups concerned: same-address device atomics serialise at ~30 ns each.)
// g_kd rows: kd_cols floats each (3: the colour gradient alone; 9: the gradient of the texture field's whole output row, whose other
// columns the shading never reads: written as zeros), kd_rows >= P of them (rows past P: zeros -- the padding rows of the field's input).
__global__ __launch_bounds__(256) void sh_bwd_kernel(const float* __restrict__ g_nrm, const float* __restrict__ g_shading,
                                                     const float* __restrict__ g_shaded, const float* __restrict__ gb, const ShPar par,
                                                     const long long* __restrict__ img, unsigned img_div, const float* __restrict__ kd,
                                                     int kd_stride, long long P, int two_sided, float* __restrict__ g_gb, const ShParGrad g_par,
                                                     float* __restrict__ g_kd, int kd_cols, long long kd_rows) {
    __shared__ float s_red[16][17];
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel id 0 = sh_bwd_kernel)
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < P;
    const bool lit = par.light != nullptr;
    const int ncol = lit ? 17 : 12;
    if (!live && g_kd && p < kd_rows)
        for (int c = 0; c < kd_cols; ++c) g_kd[p * kd_cols + c] = 0.f;
    const long long first = (long long)blockIdx.x * blockDim.x;
    if (first >= P) return;  // (uniform: a work-group of padding rows only)
    // the images this work-group's points belong to: first .. last (one, except at the ~B image boundaries of the list) -- read
    // up front: after the per-point work they would be one more dependent round trip
    const long long last = min(first + (long long)blockDim.x, P) - 1;
    const long long b_first = img ? sh_image_of(img, img_div, first) : 0, b_last = img ? sh_image_of(img, img_div, last) : 0;
    float gp[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) gp[k] = 0.f;
    long long row = 0;
    if (live) {
        row = sh_image_of(img, img_div, p);
        const ShRow pr = par.row(row);
        const ShFwd f = sh_forward(gb + 12 * p, pr, two_sided);
        V3 gN = g_nrm ? ld3(g_nrm + 3 * p) : V3{0.f, 0.f, 0.f};
        if (lit) {
            const V3 k = ld3(kd + (long long)kd_stride * p);
            const V3 gs = g_shaded ? ld3(g_shaded + 3 * p) : V3{0.f, 0.f, 0.f};
            const float g_sh = (g_shading ? g_shading[p] : 0.f) + dot(gs, k);
            st3(g_kd + (long long)kd_cols * p, gs * f.shading);
            for (int c = 3; c < kd_cols; ++c) g_kd[p * kd_cols + c] = 0.f;
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
            const float *r0 = pr.rot, *r1 = pr.rot + pr.rs, *r2 = pr.rot + 2 * pr.rs;
            gN = gN + V3{r0[0] * g_q.x + r1[0] * g_q.y + r2[0] * g_q.z, r0[1] * g_q.x + r1[1] * g_q.y + r2[1] * g_q.z,
                         r0[2] * g_q.x + r1[2] * g_q.y + r2[2] * g_q.z};
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
#pragma unroll
            for (int k = 0; k < 17; ++k)
                if (k < ncol) *g_par.at(p, k) = gp[k];
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
            if (t != 0.f) atomicAdd(g_par.at(bi, (int)threadIdx.x), t);
        }
        __syncthreads();
    }
    A3D_STAMP(0, 5);
}

int sh_check_params(const a3d_shade_params* q, const char* who) {
    if (!q || q->size < sizeof(a3d_shade_params) || !q->rot || !q->view || (q->rot_row_stride != 3 && q->rot_row_stride != 4) ||
        q->rot_image_stride < 0 || q->view_image_stride < 0 || q->light_image_stride < 0) {
        a3d_set_error("%s: invalid a3d_shade_params (size, rot / view pointers, rot_row_stride 3 or 4, strides >= 0)", who);
        return A3D_EINVAL;
    }
    return A3D_OK;
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
    hipLaunchKernelGGL(sh_fwd_kernel, dim3(a3d_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, gb, sh_par_table(par, ncol),
                       (const long long*)img_or_null, 1u, kd, kd_stride, (long long)P, two_sided, nrm, shading, shaded, g_par_to_clear_or_null,
                       g_par_to_clear_or_null ? B * ncol : 0);
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
    const ShParGrad g = {g_par, g_par + 9, g_par + 12, ncol, ncol, ncol, 3};
    hipLaunchKernelGGL(sh_bwd_kernel, dim3(a3d_div_up(P, 256)), dim3(256), 0, s, g_nrm, g_shading, g_shaded, gb, sh_par_table(par, ncol),
                       (const long long*)img_or_null, 1u, kd, kd_stride, (long long)P, two_sided, g_gb, g, g_kd, 3, (long long)P);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// The shading adjoint of a fused render (round 6): what a3d_shade_bwd computes, for a caller that keeps the camera matrix, the view
// position and the light parameters where they are (a3d_shade_params: no [B,17] table to assemble, no gradient table to take apart),
// finds a point's image from the covered-pixel list itself (pix[p] / pixels_per_image) and wants the colour gradient as rows of the
// texture field's OUTPUT gradient (g_tex[tex_rows, tex_cols]: columns 0..2 = d/d kd, the rest and the rows past P zero).
extern "C" int a3d_shade_bwd_rows(const float* g_shaded, const float* gb, const a3d_shade_params* par, const a3d_shade_params* g_par,
                                  const int64_t* pix, int64_t pixels_per_image, const float* kd, int kd_stride, int64_t P, int two_sided,
                                  float* g_gb, float* g_tex, int tex_cols, int64_t tex_rows, a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && tex_rows >= P && tex_cols >= 3 && pixels_per_image > 0 && pixels_per_image < 0x7fffffffll);
    if (int rc = sh_check_params(par, __func__)) return rc;
    if (int rc = sh_check_params(g_par, __func__)) return rc;
    A3D_CHECK_ARG(par->light && g_par->light && g_par->rot_row_stride == par->rot_row_stride);
    A3D_CHECK_ARG(g_tex && (P == 0 || (g_shaded && gb && pix && kd && kd_stride >= 3 && g_gb)));
    if (tex_rows == 0) return A3D_OK;
    const ShParGrad g = {const_cast<float*>(g_par->rot), const_cast<float*>(g_par->view), const_cast<float*>(g_par->light),
                         g_par->rot_image_stride, g_par->view_image_stride, g_par->light_image_stride, g_par->rot_row_stride};
    hipLaunchKernelGGL(sh_bwd_kernel, dim3(a3d_div_up(tex_rows, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (const float*)nullptr,
                       g_shaded, gb, sh_par_of(par), (const long long*)pix, (unsigned)pixels_per_image, kd, kd_stride, (long long)P, two_sided, g_gb,
                       g, g_tex, tex_cols, (long long)tex_rows);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(shade)
