// The G-buffer row of one covered pixel (csrc/gbuffer.hip), shared with the rasteriser's fused resolve launch (csrc/raster.hip):
// every product-sum that must round alike in both translation units is written with explicit fused multiply-adds, so the rows do not
// depend on the contraction mode either file is compiled with.
#pragma once
#include "a3d_common.h"

#ifdef __HIPCC__
__device__ __forceinline__ void gb_load3(const float* __restrict__ p, float& x, float& y, float& z) { x = p[0]; y = p[1]; z = p[2]; }

// u*a + v*b + w*c with the fused multiply-adds written out: the two kernels that inline gb_row must round identically, whatever
// contraction the compiler would pick in either context
__device__ __forceinline__ float gb_mix(float u, float a, float v, float b, float w, float c) { return __builtin_fmaf(u, a, __builtin_fmaf(v, b, w * c)); }

// what the fields take from the G-buffer (a3d_gb_aux): dense rows of the canonical position + the point -> image index, padded
struct GbAux {
    float* tex_out;
    long long* img_out;
    long long rows, pad_to;
};
__host__ __device__ inline GbAux gb_aux_of(const a3d_gb_aux* a) {
    GbAux g = {nullptr, nullptr, 0, 0};
    if (a) { g.tex_out = a->tex_out; g.img_out = (long long*)a->img_out; g.rows = a->rows; g.pad_to = a->pad_to; }
    return g;
}
// the padding rows behind a list of P points: zeros / the last image (whole work-group; call from ONE work-group of the launch)
__device__ __forceinline__ void gb_fill_padding(const GbAux& a, long long P, int last_image) {
    if (a.pad_to <= 0 || (!a.tex_out && !a.img_out)) return;
    long long end = (P + a.pad_to - 1) / a.pad_to * a.pad_to;
    if (end > a.rows) end = a.rows;
    if (a.tex_out)
        for (long long j = 3 * P + threadIdx.x; j < 3 * end; j += blockDim.x) a.tex_out[j] = 0.f;
    if (a.img_out)
        for (long long j = P + threadIdx.x; j < end; j += blockDim.x) a.img_out[j] = last_image;
}

// the G-buffer row of one covered pixel: texel r of flat pixel i -> out row p (+ the optional extra attribute)
__device__ __forceinline__ void gb_row(const float4 r, long long i, long long p, const int* __restrict__ tri, const float* __restrict__ v_pos,
                                       const float* __restrict__ v_nrm, const float* __restrict__ prior, int prior_batch, int V, int F,
                                       long long hw, float* __restrict__ out, const float* __restrict__ extra, int E,
                                       float* __restrict__ extra_out, float* __restrict__ tex_out = nullptr,
                                       long long* __restrict__ img_out = nullptr) {
    const int f = (int)r.w - 1;
    const long long b = i / hw;
    if (img_out) img_out[p] = b;  // point -> image: the index of the fields' per-image feature rows
    float4* o4 = reinterpret_cast<float4*>(out + p * 12);  // rows are 48 bytes: three aligned 16-byte stores
    float o[12];
    if (f < 0 || f >= F) {
        o4[0] = o4[1] = o4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (extra)
            for (int c = 0; c < E; ++c) extra_out[p * E + c] = 0.f;
        if (tex_out) { tex_out[3 * p] = 0.f; tex_out[3 * p + 1] = 0.f; tex_out[3 * p + 2] = 0.f; }
        return;
    }
    const int i0 = tri[3 * f], i1 = tri[3 * f + 1], i2 = tri[3 * f + 2];
    const float u = r.x, v = r.y, w = 1.f - u - v;
    const float* vp = v_pos + b * V * 3;
    float ax, ay, az, bx, by, bz, cx, cy, cz;
    gb_load3(vp + 3ll * i0, ax, ay, az);
    gb_load3(vp + 3ll * i1, bx, by, bz);
    gb_load3(vp + 3ll * i2, cx, cy, cz);
    o[0] = gb_mix(u, ax, v, bx, w, cx);
    o[1] = gb_mix(u, ay, v, by, w, cy);
    o[2] = gb_mix(u, az, v, bz, w, cz);
    // geometric normal: safe_normalize(cross(p1 - p0, p2 - p0))   (render.py:185-188, util.py:28-32)
    const float e1x = bx - ax, e1y = by - ay, e1z = bz - az, e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    const float nx = __builtin_fmaf(e1y, e2z, -(e1z * e2y)), ny = __builtin_fmaf(e1z, e2x, -(e1x * e2z)), nz = __builtin_fmaf(e1x, e2y, -(e1y * e2x));
    const float inv = 1.f / sqrtf(fmaxf(__builtin_fmaf(nx, nx, __builtin_fmaf(ny, ny, nz * nz)), 1e-20f));
    o[3] = nx * inv; o[4] = ny * inv; o[5] = nz * inv;
    const float* vn = v_nrm + b * V * 3;
    gb_load3(vn + 3ll * i0, ax, ay, az);
    gb_load3(vn + 3ll * i1, bx, by, bz);
    gb_load3(vn + 3ll * i2, cx, cy, cz);
    o[6] = gb_mix(u, ax, v, bx, w, cx);
    o[7] = gb_mix(u, ay, v, by, w, cy);
    o[8] = gb_mix(u, az, v, bz, w, cz);
    const float* pr = prior + (prior_batch == 1 ? 0ll : b * V * 3);
    gb_load3(pr + 3ll * i0, ax, ay, az);
    gb_load3(pr + 3ll * i1, bx, by, bz);
    gb_load3(pr + 3ll * i2, cx, cy, cz);
    o[9] = gb_mix(u, ax, v, bx, w, cx);
    o[10] = gb_mix(u, ay, v, by, w, cy);
    o[11] = gb_mix(u, az, v, bz, w, cz);
    o4[0] = make_float4(o[0], o[1], o[2], o[3]);
    o4[1] = make_float4(o[4], o[5], o[6], o[7]);
    o4[2] = make_float4(o[8], o[9], o[10], o[11]);
    // the canonical position once more as a row of its own: the [P,3] input of the texture / feature fields (render.py:53-57,209), dense,
    // so that their input gradient comes back dense too (no column slice of the 12-wide row and no padded gradient of that slice)
    if (tex_out) { tex_out[3 * p] = o[9]; tex_out[3 * p + 1] = o[10]; tex_out[3 * p + 2] = o[11]; }
    if (extra) {  // one more per-vertex attribute (the sequence models' 2-D motion, render.py:281-288), E <= 3 channels
        const float* eb = extra + b * V * E;
        for (int c = 0; c < E; ++c) extra_out[p * E + c] = gb_mix(u, eb[(long long)i0 * E + c], v, eb[(long long)i1 * E + c], w, eb[(long long)i2 * E + c]);
    }
}

#endif
