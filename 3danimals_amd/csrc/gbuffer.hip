// Fused G-buffer over the covered-pixel list on gfx950.
//
// The reference builds its G-buffer with five dr.interpolate calls over the full image (world position, geometric
// normal through a [[f,f,f]] index buffer, smooth normal, tangent, canonical position; model/render/render.py:182-209)
// plus ~6 torch ops for the face normals (render.py:185-188), and in backward runs five interpolate-backward kernels
// and the rasteriser's.  Shading only ever needs the covered pixels (everything else is overwritten by lerp(bg,.,0),
// render.py:261-262), so here ONE kernel walks the compact list of covered pixels:
//   fwd : texel (u,v,id) -> 3 vertex ids -> gathers of v_pos / v_nrm / canonical v_pos -> 12 floats per point
//         [ world position | normalised face normal | smooth normal | canonical position ]
//   bwd : one thread per covered pixel: the adjoint of the pixel onto the three corners of its triangle -- the four attribute
//         adjoints (bary weighted), the face-normal adjoint (normalise + cross product) and, folded in, the rasteriser's
//         backward: d/du, d/dv of all attributes pushed straight through u = a0/(a0+a1+a2), v = a1/(...) onto clip-space x, y, w.
//         No [B,H,W,4] gradient image is ever materialised.  The 36 scatter-adds per pixel go through the work-group's LDS:
//         list entries on one triangle 1 and 8 lanes apart are merged through DPP, every (pixel, corner) row is STAGED with plain stores
//         and linked into the list of its vertex (hash claim + one integer exchange), then 16-lane groups walk the lists with
//         lane = component and add each vertex's sums to ONE 64-byte gradient row [B*V, 16] -- a single line request.
//         What the measurements behind this say (bench workload, B=16, 2e5 covered pixels; DESIGN.md has the full table):
//         LDS atomics cost ~3.5 clocks per active lane whatever the kind (36 ds_add_f32 per pixel: 42 us); device float atomics
//         are priced per 64-byte line request, not per lane (140k rows x 12 adds: 86 us as single-lane instructions, 8 us as one
//         16-lane instruction per row); same-address device atomics serialise at ~30 ns.  Earlier designs, all slower: LDS hash
//         table with float atomics + per-vertex flush of 12 single-lane atomics 62-72 us; per-triangle pixel slots filled by the
//         forward + gather per (vertex, face) 38 (fwd) + 55 + 59; atomic-free face pass over each triangle's pixel box + vertex
//         gather 83 + 9; screen-aligned 16x16 regions + vertex pass 131 + 6; table replicas 70 / 85; no LDS aggregation at all
//         (every staged entry straight to its row: 410k contended row atomics) 60.  Shipped: 32 us with the 1-lane merge, 21 us with the
//         8-lane merge on top.
// The shared canonical mesh accumulates per image ([B,V,3]) and is reduced by the caller.
// HBM traffic per covered pixel: fwd 8 (index) + 16 (texel) + 48 (out) B; bwd 8 + 16 + 48 B in; vertex data lives in L2.
#include "a3d_common.h"
#include "cover_common.h"
#include "gbuffer_common.h"

__global__ __launch_bounds__(256) void gb_fwd_kernel(const float4* __restrict__ rast, const int* __restrict__ tri, const long long* __restrict__ pix,
                                                     long long P, const float* __restrict__ v_pos, const float* __restrict__ v_nrm,
                                                     const float* __restrict__ prior, int prior_batch, int V, int F, long long hw,
                                                     float* __restrict__ out, const float* __restrict__ extra, int E,
                                                     float* __restrict__ extra_out, float4* __restrict__ zero_rows, long long n_zero4,
                                                     const GbAux aux, int last_image) {
    float* __restrict__ tex_out = aux.tex_out;
    long long* __restrict__ img_out = aux.img_out;
    if (blockIdx.x == 0) gb_fill_padding(aux, P, last_image);
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // the backward's gradient rows cleared here, while this launch is waiting for its gathers anyway (saves the backward its memset)
    for (long long z = p; z < n_zero4; z += (long long)gridDim.x * blockDim.x) zero_rows[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p >= P) return;
    const long long i = pix[p];
    gb_row(rast[i], i, p, tri, v_pos, v_nrm, prior, prior_batch, V, F, hw, out, extra, E, extra_out, tex_out, img_out);
}

// The covered-pixel list AND its G-buffer rows in one launch (a3d_cover_emit + a3d_gbuffer_fwd): thread = position k of the tile-ordered
// pixel space; the texel is read ONCE (its id decides coverage, its barycentrics feed the row), the list position comes from the block
// counts / group sums the rasteriser's resolve left (cover_common.h), covered pixels write their list entry, the pixel -> entry map and
// their row.  One launch and one pass over the id channel less; the rows' gathers hide behind the 80 % of threads that only write -1.
__global__ __launch_bounds__(256) void gb_cover_fwd_kernel(const float4* __restrict__ rast, long long n, int H, int W,
                                                           const int* __restrict__ block_count, const int* __restrict__ group_sum,
                                                           long long* __restrict__ pix, int* __restrict__ inv, const int* __restrict__ tri,
                                                           const float* __restrict__ v_pos, const float* __restrict__ v_nrm,
                                                           const float* __restrict__ prior, int prior_batch, int V, int F,
                                                           float* __restrict__ out, const float* __restrict__ extra, int E,
                                                           float* __restrict__ extra_out, float4* __restrict__ zero_rows, long long n_zero4,
                                                           const GbAux aux, long long P, int last_image) {
    float* __restrict__ tex_out = aux.tex_out;
    long long* __restrict__ img_out = aux.img_out;
    if (blockIdx.x == 0) gb_fill_padding(aux, P, last_image);
    __shared__ int wave_n[4];
    __shared__ int s_off;
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel ids of this file: 0 = gb_cover_fwd_kernel, 1 = gb_bwd_kernel)
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    for (long long z = k; z < n_zero4; z += (long long)gridDim.x * 256) zero_rows[z] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long flat = k < n ? cv_flat(k, H, W, 8) : 0;
    // Three work-groups in four cover background only, and the block counts say so before a texel is read: such a work-group writes its
    // 256 map entries and leaves -- no 4 KB of texels, no list offset, no barrier (the launch is bound by seats x lifetime, and these
    // held theirs for a texel round trip + the offset's loads to write -1).
    if (block_count[blockIdx.x] == 0) {  // (uniform)
        if (inv && k < n) inv[flat] = -1;
        A3D_STAMP(0, 5);
        return;
    }
    const float4 r = k < n ? rast[flat] : make_float4(0.f, 0.f, 0.f, 0.f);  // (issued before the offset's loads: the latencies overlap)
    const int wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int off = cv_block_offset_wave0(block_count, group_sum, (int)blockIdx.x);
        if (threadIdx.x == 0) s_off = off;
    }
    const bool on = r.w > 0.f;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) wave_n[wave] = __popcll(m);
    __syncthreads();
    A3D_STAMP(0, 1);
    if (!on) {
        if (inv && k < n) inv[flat] = -1;
        A3D_STAMP(0, 5);
        return;
    }
    int o = s_off + a3d_wave_prefix(m);
    for (int w = 0; w < wave; ++w) o += wave_n[w];
    pix[o] = flat;
    if (inv) inv[flat] = o;
    gb_row(r, flat, o, tri, v_pos, v_nrm, prior, prior_batch, V, F, (long long)H * W, out, extra, E, extra_out, tex_out, img_out);
    A3D_STAMP(0, 5);
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// Vertex data of one triangle of one image, loaded once and reused by every pixel of the triangle.
struct GbTri {
    float pos[3][3], nrm[3][3], pri[3][3];
    // adjoint helpers of the face normal that do not depend on the pixel
    float e1[3], e2[3], fn[3], inv_len;
    bool degenerate;
};

__device__ __forceinline__ void gb_load_tri(GbTri& t, const float* __restrict__ v_pos, const float* __restrict__ v_nrm,
                                            const float* __restrict__ prior, long long vb3, long long pb3, int i0, int i1, int i2) {
    const int idx[3] = {i0, i1, i2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gb_load3(v_pos + vb3 + 3ll * idx[k], t.pos[k][0], t.pos[k][1], t.pos[k][2]);
        gb_load3(v_nrm + vb3 + 3ll * idx[k], t.nrm[k][0], t.nrm[k][1], t.nrm[k][2]);
        gb_load3(prior + pb3 + 3ll * idx[k], t.pri[k][0], t.pri[k][1], t.pri[k][2]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { t.e1[c] = t.pos[1][c] - t.pos[0][c]; t.e2[c] = t.pos[2][c] - t.pos[0][c]; }
    const float nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1], ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2], nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    const float d = nx * nx + ny * ny + nz * nz;
    t.degenerate = !(d > 1e-20f);
    t.inv_len = t.degenerate ? 1e10f : 1.f / sqrtf(d);
    t.fn[0] = nx * t.inv_len; t.fn[1] = ny * t.inv_len; t.fn[2] = nz * t.inv_len;
}

// Adjoint of one covered pixel onto the three corners of its triangle, ADDED to acc[corner][0..11]: [0..2] d/d v_pos, [3..5]
// d/d v_nrm, [6..8] d/d canonical position, [9..11] d/d clip x, y, w.  (u, v) = barycentrics of the texel, g = the 12 incoming
// gradients of the G-buffer row, p0..p2 = clip-space vertices.
template <int NC>
__device__ __forceinline__ void gb_pixel_adjoint(const GbTri& t, const float4 p0, const float4 p1, const float4 p2, float u, float v,
                                                 const float g[12], int px, int py, int H, int W, bool want_clip, float acc[3][NC],
                                                 const float ex[3][3], const float ge[3]) {
    const float w = 1.f - u - v;
    float gu = 0.f, gv = 0.f;
    if (NC > 12) {  // the extra attribute: acc[.][12..14] d/d extra, and its share of d/du, d/dv
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gu += ge[c] * (ex[0][c] - ex[2][c]);
            gv += ge[c] * (ex[1][c] - ex[2][c]);
            acc[0][12 + c] += u * ge[c]; acc[1][12 + c] += v * ge[c]; acc[2][12 + c] += w * ge[c];
        }
    }
    // world position
    const float gx = g[0], gy = g[1], gz = g[2];
    gu += gx * (t.pos[0][0] - t.pos[2][0]) + gy * (t.pos[0][1] - t.pos[2][1]) + gz * (t.pos[0][2] - t.pos[2][2]);
    gv += gx * (t.pos[1][0] - t.pos[2][0]) + gy * (t.pos[1][1] - t.pos[2][1]) + gz * (t.pos[1][2] - t.pos[2][2]);
    // face normal n = e1 x e2, normalised: q = adjoint of the un-normalised normal; g_e1 = e2 x q, g_e2 = q x e1
    const float hx = g[3], hy = g[4], hz = g[5];
    float qx, qy, qz;
    if (!t.degenerate) {
        const float dot = t.fn[0] * hx + t.fn[1] * hy + t.fn[2] * hz;
        qx = (hx - t.fn[0] * dot) * t.inv_len; qy = (hy - t.fn[1] * dot) * t.inv_len; qz = (hz - t.fn[2] * dot) * t.inv_len;
    } else {
        qx = hx * 1e10f; qy = hy * 1e10f; qz = hz * 1e10f;
    }
    const float g1x = t.e2[1] * qz - t.e2[2] * qy, g1y = t.e2[2] * qx - t.e2[0] * qz, g1z = t.e2[0] * qy - t.e2[1] * qx;
    const float g2x = qy * t.e1[2] - qz * t.e1[1], g2y = qz * t.e1[0] - qx * t.e1[2], g2z = qx * t.e1[1] - qy * t.e1[0];
    acc[0][0] += u * gx - (g1x + g2x); acc[0][1] += u * gy - (g1y + g2y); acc[0][2] += u * gz - (g1z + g2z);
    acc[1][0] += v * gx + g1x; acc[1][1] += v * gy + g1y; acc[1][2] += v * gz + g1z;
    acc[2][0] += w * gx + g2x; acc[2][1] += w * gy + g2y; acc[2][2] += w * gz + g2z;
    // smooth normal
    const float mx = g[6], my = g[7], mz = g[8];
    gu += mx * (t.nrm[0][0] - t.nrm[2][0]) + my * (t.nrm[0][1] - t.nrm[2][1]) + mz * (t.nrm[0][2] - t.nrm[2][2]);
    gv += mx * (t.nrm[1][0] - t.nrm[2][0]) + my * (t.nrm[1][1] - t.nrm[2][1]) + mz * (t.nrm[1][2] - t.nrm[2][2]);
    acc[0][3] += u * mx; acc[0][4] += u * my; acc[0][5] += u * mz;
    acc[1][3] += v * mx; acc[1][4] += v * my; acc[1][5] += v * mz;
    acc[2][3] += w * mx; acc[2][4] += w * my; acc[2][5] += w * mz;
    // canonical position
    const float cx = g[9], cy = g[10], cz = g[11];
    gu += cx * (t.pri[0][0] - t.pri[2][0]) + cy * (t.pri[0][1] - t.pri[2][1]) + cz * (t.pri[0][2] - t.pri[2][2]);
    gv += cx * (t.pri[1][0] - t.pri[2][0]) + cy * (t.pri[1][1] - t.pri[2][1]) + cz * (t.pri[1][2] - t.pri[2][2]);
    acc[0][6] += u * cx; acc[0][7] += u * cy; acc[0][8] += u * cz;
    acc[1][6] += v * cx; acc[1][7] += v * cy; acc[1][8] += v * cz;
    acc[2][6] += w * cx; acc[2][7] += w * cy; acc[2][8] += w * cz;
    if (want_clip && (gu != 0.f || gv != 0.f)) {  // rasteriser backward (same algebra as rs_bwd_kernel)
        const float fx = ((float)px + 0.5f) * (2.f / (float)W) - 1.f;
        const float fy = ((float)py + 0.5f) * (2.f / (float)H) - 1.f;
        const float q0x = p0.x - fx * p0.w, q0y = p0.y - fy * p0.w;
        const float q1x = p1.x - fx * p1.w, q1y = p1.y - fy * p1.w;
        const float q2x = p2.x - fx * p2.w, q2y = p2.y - fy * p2.w;
        const float a0 = q1x * q2y - q1y * q2x, a1 = q2x * q0y - q2y * q0x, a2 = q0x * q1y - q0y * q1x;
        const float s = a0 + a1 + a2;
        if (s != 0.f) {
            const float is = 1.f / s;
            const float uu = a0 * is, vv = a1 * is;
            const float tt = gu * uu + gv * vv;
            const float ga0 = (gu - tt) * is, ga1 = (gv - tt) * is, ga2 = -tt * is;
            const float o0x = -ga1 * q2y + ga2 * q1y, o0y = ga1 * q2x - ga2 * q1x;
            const float o1x = ga0 * q2y - ga2 * q0y, o1y = -ga0 * q2x + ga2 * q0x;
            const float o2x = -ga0 * q1y + ga1 * q0y, o2y = ga0 * q1x - ga1 * q0x;
            acc[0][9] += o0x; acc[0][10] += o0y; acc[0][11] += -fx * o0x - fy * o0y;
            acc[1][9] += o1x; acc[1][10] += o1y; acc[1][11] += -fx * o1x - fy * o1y;
            acc[2][9] += o2x; acc[2][10] += o2y; acc[2][11] += -fx * o2x - fy * o2y;
        }
    }
}

// Two sizes of the work-group tables, chosen by the entry point from the pixels-per-triangle ratio of the call:
//   <512, 640>  triangles of a few pixels (256 pixels reference ~150-200 distinct vertices, ~512 entries after the pair merge): 39 KB
//   <1024, 768> sub-pixel triangles (every pixel its own three vertices: up to 768 entries and as many vertices): 52 KB
// Whatever still does not fit goes to the gradient rows directly (twelve single-lane atomics: slow, so it has to stay rare -- with
// the small tables the R = 128 grid, 0.3 pixels per triangle, ran at 74 us).
#define GB_PROBES 16
#define GB_ROW 16       // floats per gradient row = one 64-byte line: [0..2] v_pos, [3..5] v_nrm, [6..8] canonical, [9..11] extra, [12] [13] [15] clip x y w

// find-or-claim the table slot of a vertex row
template <int GB_SLOTS>
__device__ __forceinline__ int gb_slot(int* s_key, int key) {
    unsigned h = ((unsigned)key * 2654435761u) >> (GB_SLOTS == 512 ? 23 : 22);  // top 9 / 10 bits
#pragma unroll 1
    for (int t = 0; t < GB_PROBES; ++t) {
        const int old = atomicCAS(&s_key[h], -1, key);
        if (old == -1 || old == key) return (int)h;
        h = (h + 1) & (GB_SLOTS - 1);
    }
    return -1;  // table crowded: the caller falls back to global atomics
}

// component -> column of the gradient row: 0..8 in place, 9..11 (clip x, y, w) -> 12, 13, 15, 12..14 (extra attribute) -> 9..11
__device__ __forceinline__ int gb_col(int k) { return k < 9 ? k : (k < 12 ? (k == 11 ? 15 : k + 3) : k - 3); }
__device__ __forceinline__ bool gb_comp_on(int k, int NC, bool want_prior, bool want_clip) {
    return k < NC && (k < 6 || k >= 9 || want_prior) && (k < 9 || k >= 12 || want_clip);
}

template <int NC>
__device__ __forceinline__ void gb_row_direct(float* g_rows, long long row, const float c[NC], bool want_prior, bool want_clip) {
    float* r = g_rows + GB_ROW * row;
#pragma unroll
    for (int k = 0; k < NC; ++k)
        if (gb_comp_on(k, NC, want_prior, want_clip)) atomicAdd(r + gb_col(k), c[k]);
}

// lane ^ 1 through the DPP quad permute (no LDS traffic)
__device__ __forceinline__ float gb_xor1(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}
__device__ __forceinline__ int gb_xor1(int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); }

template <int GB_SLOTS, int GB_ENTRIES, int NC>  // NC = components per (pixel, corner): 12, or 15 with the extra attribute
__global__ __launch_bounds__(256) void gb_bwd_kernel(const float* __restrict__ g_out, const float4* __restrict__ rast, const int* __restrict__ tri,
                                                     const long long* __restrict__ pix, long long P, const float* __restrict__ v_pos,
                                                     const float* __restrict__ v_nrm, const float* __restrict__ prior, int prior_batch,
                                                     const float4* __restrict__ clip, int V, int F, int H, int W, float* __restrict__ g_rows,
                                                     int want_prior, const float* __restrict__ extra, int E,
                                                     const float* __restrict__ g_extra, const float* __restrict__ g_tex) {
    constexpr int ST = NC == 12 ? 13 : 17;  // floats per staged entry: NC sums + the previous entry of the same slot, odd stride
    __shared__ int s_key[GB_SLOTS];    // vertex row (b*V + v) of a slot, -1 = free
    __shared__ int s_head[GB_SLOTS];   // last staged entry of the slot's list
    __shared__ int s_used[GB_SLOTS];   // claimed slots, in claim order
    __shared__ float s_stage[GB_ENTRIES * ST];
    __shared__ int s_n[2];             // staged entries, used slots
    A3D_STAMP(1, 0);
    for (int i = threadIdx.x; i < GB_SLOTS; i += blockDim.x) { s_key[i] = -1; s_head[i] = -1; }
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    const bool want_clip = clip != nullptr;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = p < P ? pix[p] : 0;
    const float4 r = p < P ? rast[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int f = (int)r.w - 1;
    const bool live = p < P && f >= 0 && f < F;
    float acc[3][NC];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[c][k] = 0.f;
    int i0 = 0, i1 = 0, i2 = 0;
    int b = 0;
    if (live) {
        const unsigned hw = (unsigned)H * (unsigned)W;
        b = (int)((unsigned)i / hw);
        const unsigned rem = (unsigned)i - (unsigned)b * hw;
        const int py = (int)(rem / (unsigned)W), px = (int)(rem - (unsigned)py * (unsigned)W);
        i0 = tri[3 * f]; i1 = tri[3 * f + 1]; i2 = tri[3 * f + 2];
        const long long vb3 = (long long)b * V * 3, pb3 = prior_batch == 1 ? 0ll : vb3;
        GbTri t;
        gb_load_tri(t, v_pos, v_nrm, prior, vb3, pb3, i0, i1, i2);
        float4 p0 = make_float4(0.f, 0.f, 0.f, 1.f), p1 = p0, p2 = p0;
        if (want_clip) { const float4* cb = clip + (long long)b * V; p0 = cb[i0]; p1 = cb[i1]; p2 = cb[i2]; }
        const float4* gp = reinterpret_cast<const float4*>(g_out + p * 12);
        const float4 ga = gp[0], gb4 = gp[1], gc = gp[2];
        float g[12] = {ga.x, ga.y, ga.z, ga.w, gb4.x, gb4.y, gb4.z, gb4.w, gc.x, gc.y, gc.z, gc.w};
        if (g_tex) {  // the canonical position's gradient arrives as rows of its own (the fields' input gradient): columns 9..11 of g_out are not read
            g[9] = g_tex[3 * p]; g[10] = g_tex[3 * p + 1]; g[11] = g_tex[3 * p + 2];
        }
        float ex[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, ge[3] = {0.f, 0.f, 0.f};
        if (NC > 12) {
            const float* eb = extra + (long long)b * V * E;
            for (int c = 0; c < E; ++c) {
                ex[0][c] = eb[(long long)i0 * E + c]; ex[1][c] = eb[(long long)i1 * E + c]; ex[2][c] = eb[(long long)i2 * E + c];
                ge[c] = g_extra[p * E + c];
            }
        }
        gb_pixel_adjoint<NC>(t, p0, p1, p2, r.x, r.y, g, px, py, H, W, want_clip, acc, ex, ge);
    }
    A3D_STAMP(1, 1);
    // neighbouring list entries on the same triangle of the same image: the even lane takes the odd lane's sums, a third fewer entries
    const int tkey = live ? b * F + f : -1 - (int)threadIdx.x;  // (B*F < 2^31 is checked by the entry point)
    const int tkey1 = gb_xor1(tkey);
    const bool same = live && tkey1 == tkey;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const float o = gb_xor1(acc[c][k]);
            if (same) acc[c][k] += o;
        }
    bool active = live && !(same && (threadIdx.x & 1));
    // a second round of the same merge with the lane 8 away (DPP row_ror:8) -- the pixel below in a full 8x8 tile, and triangles
    // of a few pixels are usually two rows tall: 34 -> 21 us, every entry saved is an LDS claim, an exchange and a list hop less.
    // (Any two live entries on one triangle may merge, adjacency only makes it likely.  Further rounds -- 2 away, mirrored lanes,
    // 16 / 32 away -- cost more exchanges than they save: 21.1 - 23.1 us.)
    {
        const int mykey = active ? tkey : -1 - (int)threadIdx.x;
        const int other = __builtin_amdgcn_mov_dpp(mykey, 0x128, 0xF, 0xF, true);  // (outside the &&: every lane must take part)
        const bool same_r = active && other == mykey;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float o = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[c][k]), 0x128, 0xF, 0xF, true));
                if (same_r) acc[c][k] += o;
            }
        active = active && !(same_r && (threadIdx.x & 8));
    }
    // stage: each (pixel, corner) row goes to LDS with plain stores and is linked into the list of its vertex (one integer exchange);
    // entries are handed out per wave (one counter update per wave, not per lane)
    A3D_STAMP(1, 2);
    const unsigned long long amask = __ballot(active);
    int wbase = 0;
    if (a3d_lane_id() == 0 && amask) wbase = atomicAdd(&s_n[0], 3 * __popcll(amask));
    wbase = __builtin_amdgcn_readfirstlane(wbase);
    if (active) {
        const int rowb = b * V;
        const int idx[3] = {i0, i1, i2};
        const int base = wbase + 3 * a3d_wave_prefix(amask);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int key = rowb + idx[c];
            const int e = base + c;
            const int slot = e < GB_ENTRIES ? gb_slot<GB_SLOTS>(s_key, key) : -1;
            if (slot >= 0) {
                float* dst = s_stage + e * ST;
#pragma unroll
                for (int k = 0; k < NC; ++k) dst[k] = acc[c][k];
                dst[ST - 1] = __int_as_float(atomicExch(&s_head[slot], e));
            } else {
                gb_row_direct<NC>(g_rows, key, acc[c], want_prior != 0, want_clip);
            }
        }
    }
    __syncthreads();
    A3D_STAMP(1, 3);
    // the claimed slots, compacted (ballot + one counter update per wave)
    for (int sidx = threadIdx.x; sidx < GB_SLOTS; sidx += blockDim.x) {
        const bool used = s_key[sidx] >= 0;
        const unsigned long long um = __ballot(used);
        int ub = 0;
        if (a3d_lane_id() == 0 && um) ub = atomicAdd(&s_n[1], __popcll(um));
        ub = __builtin_amdgcn_readfirstlane(ub);
        if (used) s_used[ub + a3d_wave_prefix(um)] = sidx;
    }
    __syncthreads();
    // reduce + flush: 16 lanes per vertex, lane = component, so the twelve atomics of a vertex are ONE 64-byte line request
    // (line-coalesced device atomics are ~10x cheaper than the same number of scattered ones, see the header); two vertices in
    // flight per group so that the list walks (one LDS round trip per entry) overlap
    A3D_STAMP(1, 4);
    const int n_used = s_n[1];
    const int k = threadIdx.x & 15, kk = k < NC ? k : ST - 1;
    const bool lane_on = gb_comp_on(k, NC, want_prior != 0, want_clip);
    for (int j = threadIdx.x >> 4; j < n_used; j += 32) {
        const int slot_a = s_used[j], slot_b = j + 16 < n_used ? s_used[j + 16] : -1;
        int ea = s_head[slot_a], eb = slot_b >= 0 ? s_head[slot_b] : -1;
        float sum_a = 0.f, sum_b = 0.f;
        while (ea >= 0 || eb >= 0) {
            if (ea >= 0) { const float* src = s_stage + ea * ST; const float v = src[kk]; ea = __float_as_int(src[ST - 1]); if (k < NC) sum_a += v; }
            if (eb >= 0) { const float* src = s_stage + eb * ST; const float v = src[kk]; eb = __float_as_int(src[ST - 1]); if (k < NC) sum_b += v; }
        }
        if (lane_on) {
            atomicAdd(g_rows + (long long)GB_ROW * s_key[slot_a] + gb_col(k), sum_a);
            if (slot_b >= 0) atomicAdd(g_rows + (long long)GB_ROW * s_key[slot_b] + gb_col(k), sum_b);
        }
    }
    A3D_STAMP(1, 5);
}

extern "C" int a3d_gbuffer_fwd(const float* rast, const int32_t* tri, const int64_t* pix, int64_t P, const float* v_pos, const float* v_nrm,
                               const float* prior, int prior_batch, int B, int V, int F, int H, int W, float* out, const float* extra_or_null,
                               int E, float* extra_out_or_null, float* g_rows_to_clear_or_null, const a3d_gb_aux* aux_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0);
    A3D_CHECK_ARG(!aux_or_null || (aux_or_null->size >= sizeof(a3d_gb_aux) && aux_or_null->rows >= P && aux_or_null->pad_to >= 0));
    A3D_CHECK_ARG(prior_batch == 1 || prior_batch == B);
    A3D_CHECK_ARG(!extra_or_null || (E >= 1 && E <= 3 && (extra_out_or_null || P == 0)));  // (an empty list has no output storage)
    A3D_CHECK_ARG(!g_rows_to_clear_or_null || ((uintptr_t)g_rows_to_clear_or_null & 63) == 0);
    const long long n_zero4 = g_rows_to_clear_or_null ? (long long)B * V * (GB_ROW / 4) : 0;
    if (P == 0) {
        if (n_zero4) A3D_HIP(hipMemsetAsync(g_rows_to_clear_or_null, 0, sizeof(float4) * (size_t)n_zero4, (hipStream_t)stream));
        return A3D_OK;
    }
    A3D_CHECK_ARG(rast && tri && pix && v_pos && v_nrm && prior && out);
    hipLaunchKernelGGL(gb_fwd_kernel, dim3(a3d_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)rast, tri, (const long long*)pix,
                       (long long)P, v_pos, v_nrm, prior, prior_batch, V, F, (long long)H * W, out, extra_or_null, E, extra_out_or_null,
                       (float4*)g_rows_to_clear_or_null, n_zero4, gb_aux_of(aux_or_null), B - 1);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_cover_gbuffer_fwd(const float* rast, const int32_t* tri, int B, int V, int F, int H, int W, const void* cover_scratch,
                                     int64_t P, int64_t* pix, int32_t* inv_or_null, const float* v_pos, const float* v_nrm, const float* prior,
                                     int prior_batch, float* out, const float* extra_or_null, int E, float* extra_out_or_null,
                                     float* g_rows_to_clear_or_null, const a3d_gb_aux* aux_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(!aux_or_null || (aux_or_null->size >= sizeof(a3d_gb_aux) && aux_or_null->rows >= P && aux_or_null->pad_to >= 0));
    A3D_CHECK_ARG(rast && cover_scratch && P >= 0 && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0 && (long long)B * H * W < 0x7fffffffll);
    A3D_CHECK_ARG(H % 8 == 0 && W % 8 == 0);  // the tile-ordered list (a3d_cover_count / a3d_rast_fwd's resolve with tile = 8)
    A3D_CHECK_ARG(prior_batch == 1 || prior_batch == B);
    A3D_CHECK_ARG(!extra_or_null || (E >= 1 && E <= 3 && (extra_out_or_null || P == 0)));
    A3D_CHECK_ARG(!g_rows_to_clear_or_null || ((uintptr_t)g_rows_to_clear_or_null & 63) == 0);
    A3D_CHECK_ARG(P == 0 || (tri && pix && v_pos && v_nrm && prior && out));
    A3D_CHECK_ARG(P > 0 || inv_or_null || g_rows_to_clear_or_null);  // (something to do)
    const long long n = (long long)B * H * W;
    const int nb = a3d_div_up(n, 256);
    const long long n_zero4 = g_rows_to_clear_or_null ? (long long)B * V * (GB_ROW / 4) : 0;
    hipLaunchKernelGGL(gb_cover_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float4*)rast, n, H, W, (const int*)cover_scratch,
                       (const int*)cover_scratch + nb, (long long*)pix, inv_or_null, tri, v_pos, v_nrm, prior, prior_batch, V, F, out,
                       extra_or_null, E, extra_out_or_null, (float4*)g_rows_to_clear_or_null, n_zero4, gb_aux_of(aux_or_null), (long long)P, B - 1);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

template <int NC>
static void gb_launch_bwd(bool big, hipStream_t s, const float* g_out, const float* rast, const int32_t* tri, const int64_t* pix, int64_t P,
                          const float* v_pos, const float* v_nrm, const float* prior, int prior_batch, const float* clip, int V, int F, int H, int W,
                          float* g_rows, int want_prior, const float* extra, int E, const float* g_extra, const float* g_tex) {
    const dim3 grid(a3d_div_up(P, 256)), block(256);
    if (!big)
        hipLaunchKernelGGL((gb_bwd_kernel<512, 640, NC>), grid, block, 0, s, g_out, (const float4*)rast, tri, (const long long*)pix, (long long)P, v_pos,
                           v_nrm, prior, prior_batch, (const float4*)clip, V, F, H, W, g_rows, want_prior, extra, E, g_extra, g_tex);
    else
        hipLaunchKernelGGL((gb_bwd_kernel<1024, 768, NC>), grid, block, 0, s, g_out, (const float4*)rast, tri, (const long long*)pix, (long long)P, v_pos,
                           v_nrm, prior, prior_batch, (const float4*)clip, V, F, H, W, g_rows, want_prior, extra, E, g_extra, g_tex);
}

extern "C" int a3d_gbuffer_bwd(const float* g_out, const float* rast, const int32_t* tri, const int64_t* pix, int64_t P, const float* v_pos,
                               const float* v_nrm, const float* prior, int prior_batch, const float* clip_or_null, int B, int V, int F, int H, int W,
                               float* g_rows, int g_rows_are_clear, int want_prior, const float* extra_or_null, int E,
                               const float* g_extra_out_or_null, const float* g_tex_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(P >= 0 && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0 && (long long)B * H * W < 0x7fffffffll);
    A3D_CHECK_ARG((long long)B * V < 0x7fffffffll && (long long)B * (F + 1) < 0x7fffffffll);
    A3D_CHECK_ARG(prior_batch == 1 || prior_batch == B);
    A3D_CHECK_ARG(g_rows && ((uintptr_t)g_rows & 63) == 0);
    A3D_CHECK_ARG(!extra_or_null || (E >= 1 && E <= 3 && (g_extra_out_or_null || P == 0)));
    hipStream_t s = (hipStream_t)stream;
    if (!g_rows_are_clear) A3D_HIP(hipMemsetAsync(g_rows, 0, sizeof(float) * GB_ROW * (size_t)B * V, s));
    if (P == 0) return A3D_OK;
    A3D_CHECK_ARG(g_out && rast && tri && pix && v_pos && v_nrm && prior);
    // covered pixels per triangle of the call (all triangles, visible or not): below ~0.6 most pixels own their three vertices
    const bool big = (double)P < 0.6 * (double)B * (double)F;
    if (extra_or_null)
        gb_launch_bwd<15>(big, s, g_out, rast, tri, pix, P, v_pos, v_nrm, prior, prior_batch, clip_or_null, V, F, H, W, g_rows, want_prior,
                          extra_or_null, E, g_extra_out_or_null, g_tex_or_null);
    else
        gb_launch_bwd<12>(big, s, g_out, rast, tri, pix, P, v_pos, v_nrm, prior, prior_batch, clip_or_null, V, F, H, W, g_rows, want_prior, nullptr, 0,
                          nullptr, g_tex_or_null);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// d/d canonical position of a SHARED canonical mesh: the sum over the images of columns 6..8 of the gradient rows (the backward
// accumulates them per image: sixteen images' atomics on one row would serialise) -- a strided torch reduction of 7 us otherwise
__global__ __launch_bounds__(256) void gb_prior_sum_kernel(const float* __restrict__ rows, int B, int V, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * V) return;
    const int v = i / 3, c = i - 3 * v;
    const float* p = rows + (long long)v * GB_ROW + 6 + c;
    float s = 0.f;
    int b = 0;
    for (; b + 4 <= B; b += 4) {  // (four rows in flight; ascending image order: the order torch's sum over dim 0 is NOT bound to, the result is a float sum either way)
        const float a0 = p[(long long)b * V * GB_ROW], a1 = p[(long long)(b + 1) * V * GB_ROW], a2 = p[(long long)(b + 2) * V * GB_ROW],
                    a3 = p[(long long)(b + 3) * V * GB_ROW];
        s += (a0 + a1) + (a2 + a3);
    }
    for (; b < B; ++b) s += p[(long long)b * V * GB_ROW];
    out[i] = s;
}

extern "C" int a3d_gbuffer_prior_grad(const float* g_rows, int B, int V, float* g_prior, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_rows && g_prior && B > 0 && V > 0 && (long long)3 * V < 0x7fffffffll);
    hipLaunchKernelGGL(gb_prior_sum_kernel, dim3(a3d_div_up(3ll * V, 256)), dim3(256), 0, (hipStream_t)stream, g_rows, B, V, g_prior);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(gbuffer)
