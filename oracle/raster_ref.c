/* Oracle: scalar CPU rasteriser.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * PARITY UNPINNED: the reference delegates this arithmetic to nvdiffrast
 * (dr.DepthPeeler(...).rasterize_next_layer(), /root/reference/model/render/render.py:292-294;
 * dr.rasterize, render.py:351), which is not vendored, not pinned and not installed, and the
 * reference has no test that pins its output.  This file restates the published operator
 * semantics: output texel = (u, v, z/w, triangle_id+1), empty = 0,
 * perspective-correct barycentrics from homogeneous 2-D edge functions evaluated at the pixel
 * centre, nearest z/w wins, no back-face culling, fragments outside -1<=z/w<=1 are clipped.
 *
 * The per-fragment function below is the SPECIFICATION the HIP kernel
 * (3danimals_amd/csrc/raster.hip) must reproduce operation by operation: every product and sum
 * is individually rounded (compile with -ffp-contract=off) except where fmaf is written out.
 * That makes shared edges watertight: the edge function of edge (j,k) seen from the neighbouring
 * triangle is the exact negation, and ties (value == 0) go to exactly one side by the sign of
 * the edge-function coefficients.  Depth ties go to the lower triangle id.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC raster_ref.c -o _build/libraster_ref.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float u, v, zw; int hit; } frag_t;

/* one triangle, one pixel centre (fx,fy in NDC) */
static frag_t frag(const float* p0, const float* p1, const float* p2, float fx, float fy) {
    frag_t r = {0.f, 0.f, 0.f, 0};
    float q0x = fmaf(-fx, p0[3], p0[0]), q0y = fmaf(-fy, p0[3], p0[1]);
    float q1x = fmaf(-fx, p1[3], p1[0]), q1y = fmaf(-fy, p1[3], p1[1]);
    float q2x = fmaf(-fx, p2[3], p2[0]), q2y = fmaf(-fy, p2[3], p2[1]);
    float a0 = q1x * q2y - q1y * q2x;
    float a1 = q2x * q0y - q2y * q0x;
    float a2 = q0x * q1y - q0y * q1x;
    float s = (a0 + a1) + a2;
    if (!(s != 0.f) || isnan(s)) return r;
    float sg = s > 0.f ? 1.f : -1.f;
    const float* P[3] = {p0, p1, p2};
    float a[3] = {a0, a1, a2};
    for (int i = 0; i < 3; ++i) {
        float e = a[i] * sg;
        if (e > 0.f) continue;
        if (e < 0.f || isnan(e)) return r;
        /* e == 0: pixel centre exactly on the edge line; owner decided by the line's normal */
        const float* pj = P[(i + 1) % 3];
        const float* pk = P[(i + 2) % 3];
        float A = (pj[1] * pk[3] - pj[3] * pk[1]) * sg;
        float B = (pj[3] * pk[0] - pj[0] * pk[3]) * sg;
        if (A > 0.f || (A == 0.f && B > 0.f)) continue;
        return r;
    }
    float zn = (p0[2] * a0 + p1[2] * a1) + p2[2] * a2;
    float wn = (p0[3] * a0 + p1[3] * a1) + p2[3] * a2;
    if (!(wn * sg > 0.f)) return r; /* behind the eye */
    float zw = zn / wn;
    if (!(zw >= -1.f && zw <= 1.f)) return r; /* near / far clip */
    float iw = 1.f / s;
    float u = a0 * iw, v = a1 * iw;
    r.u = fminf(fmaxf(u, 0.f), 1.f);
    r.v = fminf(fmaxf(v, 0.f), 1.f);
    r.zw = zw;
    r.hit = 1;
    return r;
}

/* pos [B,V,4] (or [1,V,4] with pos_batch==1), tri [F,3], out rast [B,H,W,4].
 * prev (NULL or [B,H,W,4]) = the previous depth layer (DepthPeeler.rasterize_next_layer, layer n > 0): a fragment is eligible only
 * where the previous layer is non-empty and only if it lies strictly behind it in (z/w, triangle id) order. */
int a3d_ref_rasterize_peel(const float* pos, int pos_batch, const int32_t* tri, int B, int V, int F, int H, int W,
                           const float* prev, float* rast) {
    const float xs = 2.f / (float)W, xo = 1.f / (float)W - 1.f;
    const float ys = 2.f / (float)H, yo = 1.f / (float)H - 1.f;
    size_t npix = (size_t)H * W;
    float* best_z = (float*)malloc(npix * sizeof(float));
    int32_t* best_t = (int32_t*)malloc(npix * sizeof(int32_t));
    if (!best_z || !best_t) return -1;
    for (int b = 0; b < B; ++b) {
        const float* pb = pos + (size_t)(pos_batch == 1 ? 0 : b) * V * 4;
        for (size_t i = 0; i < npix; ++i) { best_z[i] = 0.f; best_t[i] = -1; }
        for (int t = 0; t < F; ++t) {
            int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) continue;
            const float *p0 = pb + 4 * (size_t)i0, *p1 = pb + 4 * (size_t)i1, *p2 = pb + 4 * (size_t)i2;
            int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
            if (p0[3] > 0.f && p1[3] > 0.f && p2[3] > 0.f) {
                /* conservative pixel bounding box: pixel centres px+0.5 inside [min,max] widened by 1/32 px */
                float sx[3] = {p0[0] / p0[3], p1[0] / p1[3], p2[0] / p2[3]};
                float sy[3] = {p0[1] / p0[3], p1[1] / p1[3], p2[1] / p2[3]};
                float mnx = fminf(sx[0], fminf(sx[1], sx[2])), mxx = fmaxf(sx[0], fmaxf(sx[1], sx[2]));
                float mny = fminf(sy[0], fminf(sy[1], sy[2])), mxy = fmaxf(sy[0], fmaxf(sy[1], sy[2]));
                float fx0 = ceilf((mnx + 1.f) * 0.5f * W - 0.53125f), fx1 = floorf((mxx + 1.f) * 0.5f * W - 0.46875f);
                float fy0 = ceilf((mny + 1.f) * 0.5f * H - 0.53125f), fy1 = floorf((mxy + 1.f) * 0.5f * H - 0.46875f);
                if (!(fx1 >= 0.f) || !(fy1 >= 0.f) || !(fx0 <= (float)(W - 1)) || !(fy0 <= (float)(H - 1))) continue;
                x0 = fx0 < 0.f ? 0 : (int)fx0;
                y0 = fy0 < 0.f ? 0 : (int)fy0;
                x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
                y1 = fy1 > (float)(H - 1) ? H - 1 : (int)fy1;
            } else if (p0[3] <= 0.f && p1[3] <= 0.f && p2[3] <= 0.f) {
                continue;
            }
            for (int py = y0; py <= y1; ++py) {
                float fy = fmaf(ys, (float)py, yo);
                for (int px = x0; px <= x1; ++px) {
                    float fx = fmaf(xs, (float)px, xo);
                    frag_t f = frag(p0, p1, p2, fx, fy);
                    if (!f.hit) continue;
                    size_t pi = (size_t)py * W + px;
                    if (prev) {
                        const float* pr = prev + ((size_t)b * npix + pi) * 4;
                        if (!(pr[3] > 0.f)) continue;
                        if (f.zw < pr[2] || (f.zw == pr[2] && t <= (int)pr[3] - 1)) continue;
                    }
                    if (best_t[pi] < 0 || f.zw < best_z[pi]) { best_z[pi] = f.zw; best_t[pi] = t; }
                }
            }
        }
        float* out = rast + (size_t)b * npix * 4;
        for (int py = 0; py < H; ++py) {
            float fy = fmaf(ys, (float)py, yo);
            for (int px = 0; px < W; ++px) {
                size_t pi = (size_t)py * W + px;
                float* o = out + pi * 4;
                int t = best_t[pi];
                if (t < 0) { o[0] = o[1] = o[2] = o[3] = 0.f; continue; }
                float fx = fmaf(xs, (float)px, xo);
                frag_t f = frag(pb + 4 * (size_t)tri[3 * t], pb + 4 * (size_t)tri[3 * t + 1],
                                pb + 4 * (size_t)tri[3 * t + 2], fx, fy);
                o[0] = f.u; o[1] = f.v; o[2] = f.zw; o[3] = (float)(t + 1);
            }
        }
    }
    free(best_z);
    free(best_t);
    return 0;
}

int a3d_ref_rasterize(const float* pos, int pos_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* rast) {
    return a3d_ref_rasterize_peel(pos, pos_batch, tri, B, V, F, H, W, 0, rast);
}
