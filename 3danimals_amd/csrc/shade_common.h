// The forward arithmetic of the per-point shading (csrc/shade.hip), shared with the compositor (csrc/antialias.hip), which can compute the
// shaded colour of a covered pixel on the fly instead of reading it from a [P,3] array that a launch of its own wrote (a3d_ca_shade).
// Both translation units are compiled with -ffp-contract=off: same operations, same bits.
#pragma once
#include "a3d_common.h"

#ifdef __HIPCC__
namespace {

constexpr float SH_NORMAL_THRESHOLD = 0.1f;  // bsdf.py:13
constexpr float SH_EPS_NORMALIZE = 1e-12f;   // torch.nn.functional.normalize
constexpr float SH_EPS_SAFE = 1e-20f;        // render/util.py:28-32

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// y = x / max(|x|, eps)
__device__ __forceinline__ V3 normalize_f(V3 x, float* len_out) {
    const float len = fmaxf(sqrtf(dot(x, x)), SH_EPS_NORMALIZE);
    *len_out = len;
    return {x.x / len, x.y / len, x.z / len};
}
__device__ __forceinline__ V3 normalize_b(V3 gy, V3 y, float len) {
    // clamp_min passes no gradient to the norm when it is below eps
    return len > SH_EPS_NORMALIZE ? (gy - y * dot(y, gy)) * (1.f / len) : gy * (1.f / len);
}

// The per-image quantities of the shading as the kernels see them: the 3x3 camera rotation (row stride rs: 3 = packed, 4 = the top-left
// of a 4x4 world-to-camera matrix read in place), the view position, the light (direction 3 | ambient | diffuse; null: no light).
struct ShRow { const float* rot; int rs; const float* view; const float* light; };

// ... and where they live: one pointer and one image stride each (0 = one row shared by every image).  The [B,ncol] table of
// a3d_shade_fwd (rotation 9 | view 3 | light 5) is the special case rot = par, view = par + 9, light = par + 12, all strides ncol.
struct ShPar {
    const float* rot; const float* view; const float* light;
    long long rot_img, view_img, light_img;
    int rs;
    __host__ __device__ ShRow row(long long b) const { return {rot + rot_img * b, rs, view + view_img * b, light ? light + light_img * b : nullptr}; }
};
inline ShPar sh_par_table(const float* par, int ncol) {
    ShPar p;
    p.rot = par; p.view = par ? par + 9 : nullptr; p.light = (par && ncol >= 17) ? par + 12 : nullptr;
    p.rot_img = p.view_img = p.light_img = ncol; p.rs = 3;
    return p;
}
inline ShPar sh_par_of(const a3d_shade_params* q) {
    ShPar p;
    p.rot = q->rot; p.view = q->view; p.light = q->light;
    p.rot_img = q->rot_image_stride; p.view_img = q->view_image_stride; p.light_img = q->light_image_stride; p.rs = q->rot_row_stride;
    return p;
}

struct ShFwd {
    V3 n1, n2, v, ns, g, N, q, cam, L;
    float len1, len2, lenv, sigma, t_raw, t, qq, lenq, l, amb, diff, shading;
};

__device__ __forceinline__ ShFwd sh_forward(const float* __restrict__ gbp, const ShRow pr, int two_sided) {
    ShFwd f;
    // (the 48-byte G-buffer row as three aligned 16-byte loads; the canonical-position columns 9..11 are not needed here)
    const float4 g0 = reinterpret_cast<const float4*>(gbp)[0], g1 = reinterpret_cast<const float4*>(gbp)[1],
                 g2 = reinterpret_cast<const float4*>(gbp)[2];
    const V3 pos = {g0.x, g0.y, g0.z}, geo = {g0.w, g1.x, g1.y}, a = {g1.z, g1.w, g2.x}, view = ld3(pr.view);
    f.n1 = normalize_f(a, &f.len1);
    f.n2 = normalize_f(f.n1, &f.len2);
    f.v = normalize_f(view - pos, &f.lenv);
    f.sigma = (two_sided && !(dot(geo, f.v) > 0.f)) ? -1.f : 1.f;
    f.ns = f.n2 * f.sigma;
    f.g = geo * f.sigma;
    f.t_raw = dot(f.v, f.ns) / SH_NORMAL_THRESHOLD;
    f.t = fminf(fmaxf(f.t_raw, 0.f), 1.f);
    const V3 d = f.ns - f.g;  // torch.lerp: start + w*(end-start) below 0.5, end - (end-start)*(1-w) above
    f.N = f.t < 0.5f ? f.g + d * f.t : f.ns - d * (1.f - f.t);
    f.shading = 0.f;
    if (pr.light) {
        const float *r0 = pr.rot, *r1 = pr.rot + pr.rs, *r2 = pr.rot + 2 * pr.rs;
        f.q = {r0[0] * f.N.x + r0[1] * f.N.y + r0[2] * f.N.z, r1[0] * f.N.x + r1[1] * f.N.y + r1[2] * f.N.z,
               r2[0] * f.N.x + r2[1] * f.N.y + r2[2] * f.N.z};
        f.qq = dot(f.q, f.q);
        f.lenq = sqrtf(fmaxf(f.qq, SH_EPS_SAFE));
        f.cam = {f.q.x / f.lenq, f.q.y / f.lenq, f.q.z / f.lenq};
        f.L = ld3(pr.light);
        f.amb = pr.light[3];
        f.diff = pr.light[4];
        f.l = dot(f.L, f.cam);
        f.shading = f.amb + f.diff * fmaxf(f.l, 0.f);
    }
    return f;
}

}  // namespace
#endif
