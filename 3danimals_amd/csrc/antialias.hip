// Analytic silhouette antialiasing on gfx950 -- replaces dr.antialias(color, rast, pos, tri)
// (model/render/render.py:264-267; nvdiffrast, third party).  Specification: oracle/raster_ref.py::antialias.
//
//   topology : edge -> adjacent-triangle table built with a 64-bit-key open-addressing hash (atomicCAS claim,
//              atomicMin on a (face*4+corner) code per traversal direction => deterministic), flattened to
//              opp[F,3] so the per-pixel pass does ONE 12-byte read instead of three hash probes.
//              Once per mesh topology (= once per DMTet call), shared by every image and colour buffer.
//   analyze  : pixel-space vertex positions once per (image, vertex); then one thread per (pixel, direction) inspects the
//              right / lower neighbour; id discontinuities are analysed (3+3 8-byte gathers, no divisions) and the rare
//              true silhouette crossings are appended to a work list of AA_SHARDS segments
//              with ONE atomicAdd per wave (ballot + mbcnt) on the segment's own counter.  Once per (rast, clip): the reference repeats this
//              for every colour buffer it antialiases (render.py:311-315).
//   fwd/bwd  : out = color (+) blends over the work list; backward adds colour gradients and sends
//              d(alpha)/d(clip) to the two vertices of the crossing edge.
// HBM traffic: analyze reads 16 B/pixel; fwd/bwd copy 4C B/pixel in and out; the work list is a few thousand
// 16-byte records per image.  Compiled with -ffp-contract=off so sign tests agree with the oracle.
#include "a3d_common.h"
#include "shade_common.h"
#include "topo_common.h"

#define AA_SHARDS 256  // segments of the crossing work list (the consumers' work-group size: one thread per segment scans the fill counts)

struct AaRec {
    int pix0;     // flat index (b*H + y)*W + x of the pair's first pixel
    int tri;      // triangle that owns the crossing edge
    float alpha;  // signed blend weight
    int flags;    // bit0 d (0: right neighbour, 1: lower), bits1-2 edge, bit3 triangle belongs to 2nd pixel, bit4 dc clamped
};

__global__ __launch_bounds__(256) void aa_hash_insert_kernel(const int* __restrict__ tri, int F, unsigned mask,
                                                             unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 3 * F) aa_insert_edge(tri, idx, mask, keys, vals);
}

__global__ __launch_bounds__(256) void aa_hash_lookup_kernel(const int* __restrict__ tri, int F, unsigned mask,
                                                             const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                             int* __restrict__ opp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 3 * F) opp[idx] = aa_lookup_edge(tri, idx, mask, keys, vals);
}

__device__ __forceinline__ bool aa_same_sign(float a, float b) { return ((__float_as_uint(a) ^ __float_as_uint(b)) >> 31) == 0u; }

struct AaEdge {
    float xa, ya, xb, yb;  // pair direction is the first coordinate
    int va, vb;            // vertex indices of the edge
};

// Shared by analysis and backward: pixel-space vertices of triangle t relative to pixel (px,py).
__device__ __forceinline__ void aa_project(const float4 p, float fx, float fy, float xh, float yh, float& X, float& Y) {
    X = p.x / p.w * xh - fx;
    Y = p.y / p.w * yh - fy;
}

// pixel-space position of every vertex relative to the image centre: the two divisions of aa_project, once per (image, vertex)
// instead of 12 times per pixel pair.  (p.x / p.w * xh) - fx is evaluated unfused (this TU is built with -ffp-contract=off), so
// subtracting fx from the stored product gives the same bits as aa_project.
__global__ __launch_bounds__(256) void aa_screen_kernel(const float4* __restrict__ clip, long long n, float xh, float yh,
                                                        float2* __restrict__ screen, int* __restrict__ count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < AA_SHARDS) count[i] = 0;  // the analysis launch that follows appends from zero (one launch less than a memset)
    if (i >= n) return;
    const float4 p = clip[i];
    screen[i] = make_float2(p.x / p.w * xh, p.y / p.w * yh);
}

// what the analysis reads and writes (kernel argument of its stand-alone launch and of the compositor launch it can ride in)
struct AaAnalyzeJob {
    const float4* rast;
    const float2* screen;
    const int *tri, *opp, *off, *adj;
    AaRec* work;
    int* count;
    int clip_batch, V, F, H, W, capacity, B, stride;  // stride: layout of off / adj (topo_common.h: vf_list)
};

// One pixel pair (pixel i = image b, row y, column x; its right (d = 0) or lower (d = 1) neighbour, which the caller has established
// to exist): true + the record if the pair straddles a silhouette edge.
__device__ __forceinline__ bool aa_pair_record(const AaAnalyzeJob& a, long long i, int b, int x, int y, int d, AaRec& rec) {
    const float4* __restrict__ rast = a.rast;
    const float2* __restrict__ screen = a.screen;
    const int* __restrict__ tri = a.tri;
    const int* __restrict__ opp = a.opp;
    const int* __restrict__ off = a.off;
    const int* __restrict__ adj = a.adj;
    const int clip_batch = a.clip_batch, V = a.V, F = a.F, H = a.H, W = a.W;
    bool emit = false;
    {
        {
            // (only depth and id of the two texels: the second half of each 16-byte texel, 8 bytes instead of 16 per read)
            const float2 r0 = reinterpret_cast<const float2*>(rast + i)[1];
            const float2 r1 = reinterpret_cast<const float2*>(rast + i + (d == 0 ? 1 : W))[1];
            const int id0 = (int)r0.y - 1, id1 = (int)r1.y - 1;
            if (id0 != id1) {
                int t = id0 >= 0 ? id0 : id1;
                if (id0 >= 0 && id1 >= 0) t = (r0.x < r1.x) ? id0 : id1;
                const bool use1 = (t == id1);
                if (t >= 0 && t < F) {
                    const float xh = 0.5f * W, yh = 0.5f * H;
                    const float2* sb = screen + (clip_batch == 1 ? 0ll : (long long)b * V);
                    const int px = x + (use1 ? 1 - d : 0), py = y + (use1 ? d : 0);
                    const float ds = use1 ? -1.f : 1.f;
                    const int v0 = tri[3 * t], v1 = tri[3 * t + 1], v2 = tri[3 * t + 2];
                    const float fx = (float)px + 0.5f - xh, fy = (float)py + 0.5f - yh;
                    const float2 a0 = sb[v0], a1 = sb[v1], a2 = sb[v2];
                    float x0 = a0.x - fx, y0 = a0.y - fy, x1 = a1.x - fx, y1 = a1.y - fy, x2 = a2.x - fx, y2 = a2.y - fy;
                    const float bb = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
                    // the geometric tests first (which edge the pair's line leaves the triangle through, its slope, the crossing distance);
                    // the silhouette test -- the only part that needs the neighbouring triangle -- only for the pairs that pass them, and
                    // only for that ONE edge.  Same operations on the same operands as testing all three edges up front: same records.
                    float px0 = x0, py0 = y0, px1 = x1, py1 = y1, px2 = x2, py2 = y2;
                    if (d == 1) {  // pair direction becomes the first coordinate
                        px0 = y0; py0 = x0; px1 = y1; py1 = x1; px2 = y2; py2 = x2;
                    }
                    // edge k joins vertices (k+1, k+2)
                    const float exa[3] = {px1, px2, px0}, eya[3] = {py1, py2, py0}, exb[3] = {px2, px0, px1}, eyb[3] = {py2, py0, py1};
                    float best = -INFINITY, bdx = 0.f, bdy = 1.f;
                    int di = 0;
                    bool bstr = false;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float dxe = exb[k] - exa[k], dye = eyb[k] - eya[k];
                        const bool str = !aa_same_sign(eya[k], eyb[k]);
                        const float ratio = str ? (ds * (exa[k] * dye - eya[k] * dxe)) / dye : -INFINITY;
                        const bool better = (k == 0) ? true : (ratio > best);
                        if (better) { best = ratio; di = k; bdx = dxe; bdy = dye; bstr = str; }
                    }
                    const float dc = best;
                    if (bstr && (fabsf(bdy) >= fabsf(bdx)) && (dc > -0.0625f) && (dc < 1.0625f)) {
                        // silhouette: the vertex opposite to edge di in the adjacent triangle lies on the same side as the own third vertex
                        // (no neighbour: the own vertex, i.e. always a silhouette)
                        int o = opp ? opp[3 * t + di] : aa_opposite_from_lists(tri, off, adj, a.stride, F, t, di);
                        o = o >= 0 ? o : (di == 0 ? v0 : (di == 1 ? v1 : v2));
                        const float2 bo = sb[o];
                        const float ox = bo.x - fx, oy = bo.y - fy;
                        const float w = di == 0 ? (x1 - ox) * (y2 - oy) - (x2 - ox) * (y1 - oy)
                                      : (di == 1 ? (x2 - ox) * (y0 - oy) - (x0 - ox) * (y2 - oy) : (x0 - ox) * (y1 - oy) - (x1 - ox) * (y0 - oy));
                        if (aa_same_sign(w, bb)) {
                            const float dcc = fminf(fmaxf(dc, 0.f), 1.f);
                            rec.pix0 = (int)i;
                            rec.tri = t;
                            rec.alpha = ds * (0.5f - dcc);
                            rec.flags = d | (di << 1) | (use1 ? 8 : 0) | ((dc < 0.f || dc > 1.f) ? 16 : 0);
                            emit = true;
                        }
                    }
                }
            }
        }
    }
    return emit;
}

// wave-aggregated append into segment ``shard`` of the work list: a single append counter would serialise every wave of the launch on
// one address (~12 ns per returning atomic: 31 of the analysis' 46 us on the bench workload, round 1)
__device__ __forceinline__ void aa_append(const AaAnalyzeJob& a, bool emit, const AaRec& rec, int shard) {
    const unsigned long long m = __ballot(emit);
    if (m) {
        const int seg_cap = a.capacity / AA_SHARDS;
        int basei = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (a3d_lane_id() == leader) basei = atomicAdd(a.count + shard, __popcll(m));
        basei = __shfl(basei, leader);
        if (emit) {
            const int slot = basei + a3d_wave_prefix(m);
            if (slot < seg_cap) a.work[(long long)shard * seg_cap + slot] = rec;
        }
    }
}

// One thread per pixel, BOTH its pairs (right neighbour d = 0, lower neighbour d = 1): work-group (bx, b) of a (ceil(H W / 256), B) grid.
// Work-group L of that grid appends to segment L % AA_SHARDS; a segment holds 512 records for each of its work-groups.
// Two phases: every thread compares the triangle id of its pixel with the two neighbours' (three 4-byte reads); the pairs that differ
// (~15 % of them: with triangles the size of a pixel an id discontinuity is the rule inside the object) are POOLED through LDS and
// worked off by the first ceil(n / 64) waves with full lanes -- the long path (triangle, three screen positions, the opposite vertex:
// five dependent gathers, ~200 instructions) used to run in every wave that held a single such pair, at ~15 % of its lanes.
// (Until round 4 a work-group took ONE direction of its 256 pixels: twice the work-groups, each texel's id read by both.  Wherever the
// analysis rides -- the compositor's first launch -- the launch is bound by the work-groups it has to seat, tools/kernel_phases.py:
// 8192 analysis work-groups of 3.7 us were half of its residency.)
__device__ __forceinline__ void aa_analyze_body(const AaAnalyzeJob& a, unsigned bx, int b) {
    __shared__ unsigned s_cand[512];  // (y << 16 | x) | d << 31
    __shared__ int s_ncand;
    const unsigned hw = (unsigned)a.H * (unsigned)a.W;
    const unsigned rem = bx * 256u + threadIdx.x;
    if (threadIdx.x == 0) s_ncand = 0;
    __syncthreads();
    bool cand0 = false, cand1 = false;
    unsigned xy = 0;
    if (rem < hw) {
        const unsigned y = rem / (unsigned)a.W, x = rem - y * (unsigned)a.W;  // (a 64-bit division costs ~150 instructions)
        const long long i = (long long)b * hw + rem;
        const bool has0 = (int)x + 1 < a.W, has1 = (int)y + 1 < a.H;
        // (unconditional reads at clamped positions, the selects after them)
        const float w0 = reinterpret_cast<const float*>(a.rast + i)[3];
        const float wr = reinterpret_cast<const float*>(a.rast + (has0 ? i + 1 : i))[3];
        const float wd = reinterpret_cast<const float*>(a.rast + (has1 ? i + a.W : i))[3];
        cand0 = has0 && (int)w0 != (int)wr;
        cand1 = has1 && (int)w0 != (int)wd;
        xy = (y << 16) | x;
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const bool cand = d ? cand1 : cand0;
        const unsigned long long m = __ballot(cand);
        int base = 0;
        if (m) {
            const int leader = __ffsll((long long)m) - 1;
            if (a3d_lane_id() == leader) base = atomicAdd(&s_ncand, __popcll(m));
            base = __shfl(base, leader);
        }
        if (cand) s_cand[base + a3d_wave_prefix(m)] = xy | ((unsigned)d << 31);
    }
    __syncthreads();
    const int n = s_ncand;
    const unsigned lin = bx + ((hw + 255u) / 256u) * (unsigned)b;
    for (int c0 = 0; c0 < n; c0 += 256) {
        if (c0 + (int)(threadIdx.x & ~63u) >= n) break;  // (whole waves without work leave; the ballots of aa_append see whole waves)
        AaRec rec;
        bool emit = false;
        if (c0 + (int)threadIdx.x < n) {
            const unsigned c = s_cand[c0 + threadIdx.x];
            const int d = (int)(c >> 31), y = (int)((c >> 16) & 0x7FFFu), x = (int)(c & 0xFFFFu);
            emit = aa_pair_record(a, (long long)b * hw + (unsigned)y * (unsigned)a.W + (unsigned)x, b, x, y, d, rec);
        }
        aa_append(a, emit, rec, (int)(lin & (AA_SHARDS - 1)));
    }
}

__global__ __launch_bounds__(256) void aa_analyze_kernel(AaAnalyzeJob a) { aa_analyze_body(a, blockIdx.x, (int)blockIdx.y); }

// consumers: exclusive prefix of the segment fills into LDS (call with all threads of the block), then record r -> its slot
__device__ __forceinline__ int aa_segment_offsets(const int* __restrict__ count, int capacity, int* s_off) {
    const int seg_cap = capacity / AA_SHARDS;
    __shared__ int s_wsum[AA_SHARDS / 64];
    {   // (blockDim.x == 256 == AA_SHARDS in every consumer)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int c = (int)threadIdx.x < AA_SHARDS ? min(count[threadIdx.x], seg_cap) : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        if ((int)threadIdx.x < AA_SHARDS) s_off[threadIdx.x + 1] = before + incl;
        if (threadIdx.x == 0) s_off[0] = 0;
    }
    __syncthreads();
    return s_off[AA_SHARDS];
}

__device__ __forceinline__ long long aa_record_slot(int r, const int* s_off, int capacity) {
    int lo = 0, hi = AA_SHARDS;  // largest seg with s_off[seg] <= r
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int mid = (lo + hi) >> 1;
        if (s_off[mid] <= r) lo = mid; else hi = mid;
    }
    return (long long)lo * (capacity / AA_SHARDS) + (r - s_off[lo]);
}

__global__ __launch_bounds__(256) void aa_fwd_kernel(const float* __restrict__ color, int C, const AaRec* __restrict__ work,
                                                     const int* __restrict__ count, int capacity, int W, float* __restrict__ out) {
    __shared__ int s_off[AA_SHARDS + 1];
    const int n = aa_segment_offsets(count, capacity, s_off);
    const long long total = (long long)n * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / C), c = (int)(idx - (long long)r * C);
        const AaRec rec = work[aa_record_slot(r, s_off, capacity)];
        const long long p0 = rec.pix0, p1 = p0 + ((rec.flags & 1) ? W : 1);
        const long long dst = rec.alpha > 0.f ? p0 : p1;
        atomicAdd(out + dst * C + c, rec.alpha * (color[p1 * C + c] - color[p0 * C + c]));
    }
}

// d(loss)/d(alpha) of one crossing record pushed onto the two vertices of the crossing edge (clip x, y, w)
__device__ __forceinline__ void aa_edge_adjoint(const AaRec& rec, unsigned p0, int d, int di, bool use1, float dd, const float4* __restrict__ clip,
                                                int clip_batch, const int* __restrict__ tri, int V, int H, int W, float xh, float yh,
                                                float* __restrict__ g_clip) {
    // alpha = ds*0.5 - xint,  xint = (Xa*Yb - Ya*Xb)/(Yb - Ya) in the (pair-direction, other) frame
    const int b = (int)(p0 / ((unsigned)H * (unsigned)W));
    const int rem = (int)(p0 - (unsigned)b * ((unsigned)H * (unsigned)W));
    const int y = rem / W, x = rem - y * W;
    const int px = x + (use1 ? 1 - d : 0), py = y + (use1 ? d : 0);
    const float fx = (float)px + 0.5f - xh, fy = (float)py + 0.5f - yh;
    const long long vb = clip_batch == 1 ? 0ll : (long long)b * V;
    const int ia = tri[3 * rec.tri + (di + 1) % 3], ib = tri[3 * rec.tri + (di + 2) % 3];
    const float4 pa = clip[vb + ia], pbv = clip[vb + ib];
    float sxa, sya, sxb, syb;
    aa_project(pa, fx, fy, xh, yh, sxa, sya);
    aa_project(pbv, fx, fy, xh, yh, sxb, syb);
    const float Xa = d ? sya : sxa, Ya = d ? sxa : sya, Xb = d ? syb : sxb, Yb = d ? sxb : syb;
    const float D = Yb - Ya, iD = 1.f / D, iD2 = iD * iD;
    const float gx = -dd;  // dL/dxint
    const float gXa = gx * Yb * iD, gXb = -gx * Ya * iD;
    const float gYa = gx * Yb * (Xa - Xb) * iD2, gYb = gx * Ya * (Xb - Xa) * iD2;
    // back to screen x / y
    const float gsxa = d ? gYa : gXa, gsya = d ? gXa : gYa, gsxb = d ? gYb : gXb, gsyb = d ? gXb : gYb;
    float* oa = g_clip + (vb + ia) * 4;
    float* ob = g_clip + (vb + ib) * 4;
    const float iwa = 1.f / pa.w, iwb = 1.f / pbv.w;
    atomicAdd(oa, gsxa * xh * iwa);
    atomicAdd(oa + 1, gsya * yh * iwa);
    atomicAdd(oa + 3, -(gsxa * pa.x * xh + gsya * pa.y * yh) * iwa * iwa);
    atomicAdd(ob, gsxb * xh * iwb);
    atomicAdd(ob + 1, gsyb * yh * iwb);
    atomicAdd(ob + 3, -(gsxb * pbv.x * xh + gsyb * pbv.y * yh) * iwb * iwb);
}

// 32 lanes per crossing record: lane l takes channels l, l + 32, ... (colour gradients of both pixels, two atomics per channel) and
// the lanes' partial d(loss)/d(alpha) meet in lane 0, which pushes it onto the two vertices of the crossing edge.  (One thread per
// record walked the channels serially: 21 us for the 17-channel buffer.)
__global__ __launch_bounds__(256) void aa_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ color, int C,
                                                     const AaRec* __restrict__ work, const int* __restrict__ count, int capacity,
                                                     const float4* __restrict__ clip, int clip_batch, const int* __restrict__ tri, int V,
                                                     int H, int W, float* __restrict__ g_color, float* __restrict__ g_clip) {
    __shared__ int s_off[AA_SHARDS + 1];
    const int n = aa_segment_offsets(count, capacity, s_off);
    const float xh = 0.5f * W, yh = 0.5f * H;
    const int sub = threadIdx.x & 31, groups = (gridDim.x * blockDim.x) >> 5;
    for (int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; r < n; r += groups) {
        const AaRec rec = work[aa_record_slot(r, s_off, capacity)];
        const int d = rec.flags & 1, di = (rec.flags >> 1) & 3;
        const bool use1 = rec.flags & 8, clamped = rec.flags & 16;
        const long long p0 = rec.pix0, p1 = p0 + (d ? W : 1);
        const long long dst = rec.alpha > 0.f ? p0 : p1;
        float dd = 0.f;
        for (int c = sub; c < C; c += 32) {
            const float gd = g_out[dst * C + c];
            if (gd != 0.f) {
                atomicAdd(g_color + p1 * C + c, rec.alpha * gd);
                atomicAdd(g_color + p0 * C + c, -rec.alpha * gd);
                dd += gd * (color[p1 * C + c] - color[p0 * C + c]);
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);  // stays inside the 32-lane half
        if (sub != 0 || clamped || dd == 0.f) continue;
        aa_edge_adjoint(rec, (unsigned)p0, d, di, use1, dd, clip, clip_batch, tri, V, H, W, xh, yh, g_clip);
    }
}

// g_color = g_out (streamed, 16 bytes per lane) and g_clip = 0 in ONE launch (was a device-to-device memcpy + a memset)
__global__ __launch_bounds__(256) void aa_copy_zero_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n,
                                                           float* __restrict__ zero, long long nz) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n4 = n >> 2;
    for (long long i = t; i < n4; i += stride) reinterpret_cast<float4*>(dst)[i] = a3d_load_stream4(src + 4 * i);
    for (long long i = (n4 << 2) + t; i < n; i += stride) dst[i] = src[i];
    for (long long i = t; i < nz; i += stride) zero[i] = 0.f;
}

extern "C" int a3d_aa_shards(void) { return AA_SHARDS; }

// records the work list must hold for a [B,H,W] frame: AA_SHARDS segments of 512 records (256 pixels x 2 directions) per analysis
// work-group mapped to them
extern "C" int a3d_aa_capacity(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return AA_SHARDS;
    const long long groups = (long long)B * a3d_div_up((long long)H * W, 256);
    return (int)(a3d_div_up(groups, AA_SHARDS) * 512 * AA_SHARDS);
}

// ---- compositing fused with the antialiasing -------------------------------------------------------------------------------------------
// render_mesh composites the shaded points over the background (coverage is 0 or 1, so lerp(bg, [rgb, 1], alpha), render.py:261-262, is
// a select) and antialiases the result (render.py:311-312).  As separate steps that is a background fill, a scatter of the points, the
// antialiasing's copy of the whole image and its blends -- and in backward the same passes again.  Here the image is written ONCE from
// its sources (point rows through the pixel -> point map `inv`, background elsewhere), the blends read the same sources (dr.antialias
// blends input colours, never already blended ones), and the backward never materialises a dense colour gradient: the incoming
// gradient is gathered at the covered pixels and the blend adjoints are added to the point rows directly.
struct CaSrc {
    const float* vals;  // [P, C]; null with ``rast``: every covered pixel has the value 1 in all channels (a3d_mask_aa_*)
    const int* inv;     // [B*H*W] point of a pixel, -1 = uncovered; null: coverage is read from ``rast``
    const float4* rast; // [B*H*W] raster texels (covered <=> id channel > 0), only without ``inv``
    // vals == null with sh_gb (C == 3): the value of point q is kd[q] * shading(q), computed on the spot from the G-buffer row, the image's
    // camera / light row and kd -- a3d_shade_fwd's arithmetic (shade_common.h) without its launch and without the [P,3] round trip
    const float* sh_gb;   // [P,12]
    ShPar sh_par;         // the images' camera / light rows (a [B,17] table, or the caller's own tensors: a3d_shade_params)
    const float* sh_kd;   // [P,3], row stride sh_kd_stride
    int sh_kd_stride, sh_two_sided;
    float* sh_out;        // compose launch only: [P,3] the colour it computed, kept for the blend launch and the backward (or null)
    const float* bg;    // [bg_batch, H, W, bgC] or null (zeros); channels bgC .. C read as 0
    int bg_shared;      // bg_batch == 1
    int bgC;
    int C;
    unsigned hw;
};

__device__ __forceinline__ int ca_point(const CaSrc& s, unsigned p) {  // >= 0: covered (the point row, 0 without a list); -1: uncovered
    return s.inv ? s.inv[p] : (s.rast[p].w > 0.f ? 0 : -1);
}

__device__ __forceinline__ float3 ca_shaded(const CaSrc& s, int q, unsigned p) {  // (sh_gb given) the shaded colour of point q = pixel p
    const ShFwd f = sh_forward(s.sh_gb + 12ll * q, s.sh_par.row(p / s.hw), s.sh_two_sided);
    const float* k = s.sh_kd + (long long)s.sh_kd_stride * q;
    return make_float3(k[0] * f.shading, k[1] * f.shading, k[2] * f.shading);
}

__device__ __forceinline__ float ca_pre(const CaSrc& s, unsigned p, int c) {
    const int q = ca_point(s, p);
    if (q >= 0 && c < s.C && !s.vals && s.sh_gb) {
        const float3 v = ca_shaded(s, q, p);
        return c == 0 ? v.x : (c == 1 ? v.y : v.z);
    }
    if (q >= 0) return (c < s.C && s.vals) ? s.vals[(long long)q * s.C + c] : 1.f;
    if (!s.bg || c >= s.bgC) return 0.f;
    const unsigned r = s.bg_shared ? p % s.hw : p;
    return s.bg[(long long)r * s.bgC + c];
}

// One or two buffers per launch (blockIdx.y picks the job): the render path composites and antialiases the colour image and the
// feature image of a step against the same pixel list and the same crossing records, and four launches each way instead of eight is
// the larger part of what these latency-bound kernels cost.
struct CaJob {
    CaSrc s;
    float* clear;        // forward: n_clear floats zeroed by the compose launch (a3d_ca_shade: the shading backward's per-image rows)
    int n_clear;
    float* out;          // forward: [B,H,W,oC], the first oC <= C+1 channels of the composited image
    int oC;
    const float* g_out;  // backward: [B,H,W,gS] of which the first gC channels are the gradient of the image's (the rest: zero)
    int gS, gC;
    int gHW;             // backward, a3d_mask_aa_bwd only: > 0 = the gradient is channels-FIRST, [B,gS,H,W] with gHW = H W (0: channels-last)
    float* g_vals;       // backward: [vals_rows >= P, C], rows past P zero
    long long vals_rows;
};

// 256 pixels per work-group: the pixel -> source map goes through LDS once, then the work-group writes the pixels' C+1 floats as one
// contiguous run.  (j / C1 for j < 256*C1 as a float multiply: exact in that range, and an integer division per element -- ~40
// instructions, 150 in 64 bits -- made this pass instruction bound: 39 us for the 17-channel image.)  C + 1 == 4: one 16-byte texel
// per thread.  Stores are non-temporal: the image is far larger than the L2s (17-channel image: 20.3 -> 17.2 us).
// Extra work-groups (blockIdx.x >= nb_compose): the silhouette analysis of this frame (aa_analyze_body) when it has not run yet -- the
// blend launch that follows is its first consumer, and this pass, which only moves pixels, leaves the gather path idle: as a launch
// of its own the analysis is 15 us of kernel plus a launch gap, here it adds ~5.
// (8 waves per SIMD asked for: the launch is bound by the work-groups it can seat -- see the loop below -- and the 67 registers the riding
// analysis and the on-the-fly shading need leave 7; at 64 a few values of those two paths spill to scratch (72 bytes per lane) and the call
// is 3 us shorter, 46.0-47.4 -> 43.0-43.8.  The same request on ca_bwd_kernel, 83 registers: 6 waves 24.9 against 23.1 us, 8 waves 28.5.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void ca_compose_kernel(CaJob ja, CaJob jb, unsigned n_pix, unsigned nb_compose, AaAnalyzeJob an, int extra_first) {
    __shared__ int s_src[256];  // >= 0: point row; -1: zero; <= -2: background texel -(v + 2)
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel ids of this file: 0 = ca_compose_kernel, 1 = ca_blend_kernel, 2 = ca_gather_kernel, 3 = ca_bwd_kernel)
    // (extra_first: the analysis work-groups -- gather chains -- are dispatched BEFORE the pixel movers of their row, not as the launch's tail)
    unsigned bx = blockIdx.x;
    if (extra_first) {
        const unsigned n_extra = gridDim.x - nb_compose;
        bx = bx < n_extra ? nb_compose + bx : bx - n_extra;
    }
    if (bx >= nb_compose) {
        const unsigned nbx = ((unsigned)an.H * (unsigned)an.W + 255u) / 256u, total = nbx * (unsigned)an.B;
        const unsigned j = blockIdx.y * (gridDim.x - nb_compose) + (bx - nb_compose);  // flat work-group of the analysis
        if (j >= total) return;
        const unsigned b = j / nbx;
        aa_analyze_body(an, j - b * nbx, (int)b);
        A3D_STAMP(0, 4);  // (analysis work-groups end at slot 4, the movers of the first buffer at slot 5, of the second at slot 3)
        return;
    }
    const CaJob& job = blockIdx.y ? jb : ja;
    const CaSrc& s = job.s;
    float* __restrict__ out = job.out;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const unsigned base = bx * 256u, p = base + threadIdx.x;
    for (unsigned z = p; z < (unsigned)job.n_clear; z += nb_compose * 256u) job.clear[z] = 0.f;
    if (s.C == 3 && job.oC == 4 && (((uintptr_t)out | (s.bgC == 4 ? (uintptr_t)s.bg : 0)) & 15) == 0) {
        if (p >= n_pix) return;
        const int q = ca_point(s, p);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q >= 0) {
            if (s.vals) { const float* r = s.vals + 3ll * q; v = make_float4(r[0], r[1], r[2], 1.f); }
            else if (s.sh_gb) {
                const float3 c3 = ca_shaded(s, q, p);
                v = make_float4(c3.x, c3.y, c3.z, 1.f);
                if (s.sh_out) { float* so = s.sh_out + 3ll * q; so[0] = c3.x; so[1] = c3.y; so[2] = c3.z; }
            }
            else v = make_float4(1.f, 1.f, 1.f, 1.f);
        }
        else if (s.bg) {
            const unsigned r = s.bg_shared ? p % s.hw : p;
            if (s.bgC == 4) v = reinterpret_cast<const float4*>(s.bg)[r];
            else {  // the reference's 3-channel background as it is: alpha 0 (render.py:254-256 appends it per call)
                const float* g = s.bg + (long long)r * s.bgC;
                v = make_float4(s.bgC > 0 ? g[0] : 0.f, s.bgC > 1 ? g[1] : 0.f, s.bgC > 2 ? g[2] : 0.f, 0.f);
            }
        }
        v4f nt; nt.x = v.x; nt.y = v.y; nt.z = v.z; nt.w = v.w;
        __builtin_nontemporal_store(nt, reinterpret_cast<v4f*>(out) + p);
        A3D_STAMP(0, 5);
        return;
    }
    if (s.C == 16 && job.oC == 16 && s.vals && !s.bg && s.inv && ((((uintptr_t)out | (uintptr_t)s.vals) & 15) == 0)) {
        // (round 6) the 16-channel feature image without its alpha channel: 64-byte pixels and 64-byte value rows, both 16-byte aligned --
        // every thread moves four float4 (pixel = index / 4, channel group = index % 4): one 16-byte load and one 16-byte store per
        // 16 bytes of image, no pixel -> source map through LDS, no per-float index arithmetic (the general path below: four scalar
        // loads and a float division per 16 bytes stored)
        const unsigned i0 = base * 4u + threadIdx.x, n4 = n_pix * 4u;
        int q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const unsigned i = i0 + 256u * k; q[k] = i < n4 ? s.inv[i >> 2] : -1; }
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned i = i0 + 256u * k;
            v[k] = q[k] >= 0 ? reinterpret_cast<const float4*>(s.vals)[(long long)q[k] * 4 + (i & 3u)] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned i = i0 + 256u * k;
            if (i < n4) { v4f nt; nt.x = v[k].x; nt.y = v[k].y; nt.z = v[k].z; nt.w = v[k].w; __builtin_nontemporal_store(nt, reinterpret_cast<v4f*>(out) + i); }
        }
        if (blockIdx.y) A3D_STAMP(0, 3);
        else A3D_STAMP(0, 5);
        return;
    }
    const int C1 = job.oC;  // (floats per pixel of the image as it is stored)
    if (p < n_pix) {
        const int q = ca_point(s, p);
        s_src[threadIdx.x] = q >= 0 ? q : (s.bg ? -2 - (int)(s.bg_shared ? p % s.hw : p) : -1);
    }
    __syncthreads();
    A3D_STAMP(0, 1);
    const int nloc = (int)min(256u, n_pix - base) * C1;
    const float rc = 1.f / (float)C1;
    float* o = out + (long long)base * C1;
    // 16 bytes per lane (the run starts on a 16-byte boundary: 256 * C1 floats per work-group); a ragged tail goes float by float.
    // (Round 4, tools/kernel_phases.py: the launch holds ~1500 work-groups at any time -- 8192 of the analysis at 3.7 us each, 4096 of
    // the 18-channel image at 4.7, 4096 of the 4-channel one at 2.1 -- and ends when the last has been through; it is bound by what the
    // memory system returns to that many waiters, not by any work-group's own chain: while its movers store they store at ~6.5 TB/s
    // between them (a frame-sized fill alone reaches 6.2: tools/bw_probe), but a work-group also holds its seat for the 1.4 us its
    // pixel -> source map takes to arrive, and the 4-channel image's work-groups hold theirs for 2.4 us to write 4 KB -- 3.9 TB/s over
    // the launch.  Measured and dropped: a zero-fill path for the three work-groups in four that hold background only (same 3.6 us of
    // stores: they wait on the same write path), the pixel / channel split stepped instead of recomputed per float;
    // the loads of three / five of the stores below issued first, unconditionally at clamped indices -- loop 3.6 -> 4.2 us, launch 38.9
    // -> 40.8 (five: 94 registers, two waves per SIMD fewer, 54); 2 / 4 / 8 tiles per work-group -- call 50.1 -> 52.0 / 57.7 / 84.4.)
    const int n4 = (((uintptr_t)out & 15) == 0) ? (nloc & ~3) : 0;
    for (int j0 = 4 * threadIdx.x; j0 < n4; j0 += 1024) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            const int pl = (int)(((float)j + 0.5f) * rc), c = j - pl * C1;
            const int src = s_src[pl];
            v[k] = 0.f;
            if (src >= 0) v[k] = (c < s.C && s.vals) ? s.vals[(long long)src * s.C + c] : 1.f;
            else if (src <= -2) v[k] = c < s.bgC ? s.bg[(long long)(-2 - src) * s.bgC + c] : 0.f;
        }
        v4f q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3];
        __builtin_nontemporal_store(q, reinterpret_cast<v4f*>(o + j0));
    }
    for (int j = n4 + threadIdx.x; j < nloc; j += 256) {
        const int pl = (int)(((float)j + 0.5f) * rc), c = j - pl * C1;
        const int src = s_src[pl];
        float v = 0.f;
        if (src >= 0) v = (c < s.C && s.vals) ? s.vals[(long long)src * s.C + c] : 1.f;
        else if (src <= -2) v = c < s.bgC ? s.bg[(long long)(-2 - src) * s.bgC + c] : 0.f;
        o[j] = v;
    }
    if (blockIdx.y) A3D_STAMP(0, 3);
    else A3D_STAMP(0, 5);
}

__global__ __launch_bounds__(256) void ca_blend_kernel(CaJob ja, CaJob jb, const AaRec* __restrict__ work, const int* __restrict__ count,
                                                       int capacity, int W) {
    __shared__ int s_off[AA_SHARDS + 1];
    A3D_STAMP(1, 0);
    const CaJob& job = blockIdx.y ? jb : ja;
    const CaSrc& s = job.s;
    float* __restrict__ out = job.out;
    const int n = aa_segment_offsets(count, capacity, s_off);
    A3D_STAMP(1, 1);
    const unsigned C1 = (unsigned)job.oC;
    const unsigned total = (unsigned)n * C1;  // (n <= capacity records, C1 <= 4096)
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned ru = idx / C1;
        const int r = (int)ru, c = (int)(idx - ru * C1);
        const AaRec rec = work[aa_record_slot(r, s_off, capacity)];
        const unsigned p0 = (unsigned)rec.pix0, p1 = p0 + ((rec.flags & 1) ? (unsigned)W : 1u);
        const unsigned dst = rec.alpha > 0.f ? p0 : p1;
        const float d = ca_pre(s, p1, c) - ca_pre(s, p0, c);
        if (d != 0.f) atomicAdd(out + (long long)dst * C1 + c, rec.alpha * d);
    }
    A3D_STAMP(1, 5);
}

// g_vals[p, :C] = g_out[pix[p], :C] (the select's adjoint), 256 points per work-group (same index arithmetic as the compositor); the
// first job's work-groups also zero g_clip
__global__ __launch_bounds__(256) void ca_gather_kernel(CaJob ja, CaJob jb, const long long* __restrict__ pix, long long P,
                                                        float* __restrict__ zero, long long nz) {
    __shared__ unsigned s_pix[256];
    A3D_STAMP(2, 0);
    const CaJob& job = blockIdx.y ? jb : ja;
    const int C = job.s.C;
    const float* __restrict__ g_out = job.g_out;
    const long long base = (long long)blockIdx.x * 256, p = base + threadIdx.x;
    if (p < P) s_pix[threadIdx.x] = (unsigned)pix[p];
    if (blockIdx.y == 0)
        for (long long i = p; i < nz; i += (long long)gridDim.x * 256) zero[i] = 0.f;
    __syncthreads();
    A3D_STAMP(2, 1);
    {   // rows past P (the padding rows of a field's point list): zeros
        const long long z0 = max(base, P) * C, z1 = min(base + 256, job.vals_rows) * C;
        for (long long j = z0 + threadIdx.x; j < z1; j += 256) job.g_vals[j] = 0.f;
    }
    if (base >= P) return;
    // (round 6, measured and dropped: the same four-float4-per-thread form for the 16-channel gather -- 11.8 -> 20.3 us in the step; the
    // element form below keeps 16 lanes on one pixel's 64 bytes and four such rows in flight per lane)
    const int nloc = (int)min(256ll, P - base) * C, gS = job.gS, gC = job.gC;
    const float rc = 1.f / (float)C;
    float* o = job.g_vals + base * C;
    // four elements per thread in flight: written as a plain loop every load waited for the store before it (the compiler cannot rule
    // out that g_vals aliases g_out), i.e. C dependent round trips per work-group
    for (int j0 = threadIdx.x; j0 < nloc; j0 += 4 * 256) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + 256 * k, jc = j < nloc ? j : j0;
            const int pl = (int)(((float)jc + 0.5f) * rc), c = jc - pl * C;
            v[k] = c < gC ? g_out[(long long)s_pix[pl] * gS + c] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (j0 + 256 * k < nloc) o[j0 + 256 * k] = v[k];
    }
    A3D_STAMP(2, 5);
}

// aa_bwd_kernel on the composited image: colours come from the sources, colour adjoints go to the point rows
__global__ __launch_bounds__(256) void ca_bwd_kernel(CaJob ja, CaJob jb, const AaRec* __restrict__ work, const int* __restrict__ count,
                                                     int capacity, const float4* __restrict__ clip, int clip_batch, const int* __restrict__ tri,
                                                     int V, int H, int W, float* __restrict__ g_clip) {
    __shared__ int s_off[AA_SHARDS + 1];
    A3D_STAMP(3, 0);
    const CaJob& job = blockIdx.y ? jb : ja;
    const CaSrc& s = job.s;
    const float* __restrict__ g_out = job.g_out;
    float* __restrict__ g_vals = job.g_vals;
    const int n = aa_segment_offsets(count, capacity, s_off);
    A3D_STAMP(3, 1);
    const float xh = 0.5f * W, yh = 0.5f * H;
    const int C = s.C, C1 = min(s.C + 1, job.gC), gS = job.gS;  // (channels without a gradient contribute nothing)
    const int sub = threadIdx.x & 31, groups = (gridDim.x * blockDim.x) >> 5;
    for (int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; r < n; r += groups) {
        const AaRec rec = work[aa_record_slot(r, s_off, capacity)];
        const int d = rec.flags & 1, di = (rec.flags >> 1) & 3;
        const bool use1 = rec.flags & 8, clamped = rec.flags & 16;
        const unsigned p0 = (unsigned)rec.pix0, p1 = p0 + (d ? (unsigned)W : 1u);
        const unsigned dst = rec.alpha > 0.f ? p0 : p1;
        const int q0 = ca_point(s, p0), q1 = ca_point(s, p1);
        float dd = 0.f;
        for (int c = sub; c < C1; c += 32) {
            // (channels-first: pixel dst = image b, pixel q  ->  (b gS + c) gHW + q = dst + b (gS - 1) gHW + c gHW)
            const float gd = job.gHW ? g_out[(long long)dst + ((long long)(dst / (unsigned)job.gHW) * (gS - 1) + c) * job.gHW] : g_out[(long long)dst * gS + c];
            if (gd != 0.f) {
                if (c < C && g_vals) {  // (a constant colour has no adjoint)
                    if (q1 >= 0) atomicAdd(g_vals + (long long)q1 * C + c, rec.alpha * gd);
                    if (q0 >= 0) atomicAdd(g_vals + (long long)q0 * C + c, -rec.alpha * gd);
                }
                dd += gd * (ca_pre(s, p1, c) - ca_pre(s, p0, c));
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dd += __shfl_xor(dd, o, 64);  // stays inside the 32-lane half
        if (sub != 0 || clamped || dd == 0.f) continue;
        aa_edge_adjoint(rec, p0, d, di, use1, dd, clip, clip_batch, tri, V, H, W, xh, yh, g_clip);
    }
    A3D_STAMP(3, 5);
}

extern "C" size_t a3d_aa_hash_bytes(int F) { return (size_t)aa_slots(F < 1 ? 1 : F) * (8 + 4 * AA_VALS); }

extern "C" int a3d_aa_topology(const int32_t* tri, int F, int V, void* hash, int32_t* opp, a3d_stream_t stream) {
    A3D_CHECK_ARG(F >= 0 && V > 0);
    if (F == 0) return A3D_OK;
    A3D_CHECK_ARG(tri && hash && opp);
    hipStream_t s = (hipStream_t)stream;
    const unsigned n = aa_slots(F);
    unsigned long long* keys = (unsigned long long*)hash;
    int* vals = (int*)(keys + n);
    A3D_HIP(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * n, s));
    A3D_HIP(hipMemsetAsync(vals, 0x7F, sizeof(int) * AA_VALS * (size_t)n, s));
    hipLaunchKernelGGL(aa_hash_insert_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, s, tri, F, n - 1, keys, vals);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(aa_hash_lookup_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, s, tri, F, n - 1, keys, vals, opp);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// the whole table from the lists (every corner looked up the way a3d_aa_analyze does it for the pairs it needs): same table, bit for bit,
// as a3d_aa_topology's hash -- checks the list walk against the hash and the oracle, and serves callers that have the lists anyway
__global__ __launch_bounds__(256) void aa_opp_from_lists_kernel(int stride, const int* __restrict__ tri, int F, const int* __restrict__ off,
                                                                const int* __restrict__ adj, int* __restrict__ opp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 3 * F) opp[idx] = aa_opposite_from_lists(tri, off, adj, stride, F, idx / 3, idx - 3 * (idx / 3));
}

extern "C" int a3d_aa_topology_from_lists(const int32_t* tri, int F, const int32_t* off, const int32_t* adj, int32_t* opp, int lists_stride,
                                          a3d_stream_t stream) {
    A3D_CHECK_ARG(F >= 0 && (long long)3 * F < 0x7fffffffll && lists_stride >= 0);
    if (F == 0) return A3D_OK;
    A3D_CHECK_ARG(tri && off && adj && opp);
    hipLaunchKernelGGL(aa_opp_from_lists_kernel, dim3(a3d_div_up(3ll * F, 256)), dim3(256), 0, (hipStream_t)stream, lists_stride, tri, F, off, adj, opp);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_aa_analyze(const float* rast, const float* clip, int clip_batch, const int32_t* tri, const int32_t* opp_or_null, int B, int V,
                              int F, int H, int W, float* screen, void* work, int capacity, int32_t* count, int prepared,
                              const int32_t* off_or_null, const int32_t* adj_or_null, int lists_stride, a3d_stream_t stream) {
    const int32_t* opp = opp_or_null;
    A3D_CHECK_ARG(lists_stride >= 0);
    A3D_CHECK_ARG(rast && clip && screen && work && count && B > 0 && V > 0 && F >= 0 && H > 0 && W > 0 && capacity > 0);
    A3D_CHECK_ARG(clip_batch == 1 || clip_batch == B);
    A3D_CHECK_ARG((long long)B * H * W < 0x7FFFFFFFll && H < 32768 && W < 65536);  // (a candidate pair is packed as y << 16 | x | d << 31)
    hipStream_t s = (hipStream_t)stream;
    A3D_CHECK_ARG(capacity >= a3d_aa_capacity(B, H, W));
    if (F == 0) {
        A3D_HIP(hipMemsetAsync(count, 0, sizeof(int) * AA_SHARDS, s));
        return A3D_OK;
    }
    A3D_CHECK_ARG(tri && (opp || (off_or_null && adj_or_null)));  // the opposite-vertex table, or the vertex -> face lists to find them in
    const long long nvert = (long long)clip_batch * V;
    A3D_CHECK_ARG(B <= 65535);
    if (!prepared) {  // (prepared: a3d_rast_fwd of the same clip filled `screen` and zeroed `count` in its own launch)
        hipLaunchKernelGGL(aa_screen_kernel, dim3(a3d_div_up(nvert > AA_SHARDS ? nvert : AA_SHARDS, 256)), dim3(256), 0, s, (const float4*)clip, nvert,
                           0.5f * W, 0.5f * H, (float2*)screen, count);
        A3D_LAUNCH_CHECK();
    }
    AaAnalyzeJob an;
    an.rast = (const float4*)rast; an.screen = (const float2*)screen; an.tri = tri; an.opp = opp; an.off = off_or_null; an.adj = adj_or_null;
    an.work = (AaRec*)work; an.count = count; an.clip_batch = clip_batch; an.V = V; an.F = F; an.H = H; an.W = W; an.capacity = capacity; an.B = B; an.stride = lists_stride;
    hipLaunchKernelGGL(aa_analyze_kernel, dim3(a3d_div_up((long long)H * W, 256), B), dim3(256), 0, s, an);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_aa_fwd(const float* color, int C, const void* work, const int32_t* count, int capacity, int B, int H, int W, float* out,
                          a3d_stream_t stream) {
    A3D_CHECK_ARG(color && work && count && out && C > 0 && B > 0 && H > 0 && W > 0 && capacity > 0);
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemcpyAsync(out, color, sizeof(float) * (size_t)B * H * W * C, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(aa_fwd_kernel, dim3(512), dim3(256), 0, s, color, C, (const AaRec*)work, count, capacity, W, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_aa_bwd(const float* g_out, const float* color, int C, const void* work, const int32_t* count, int capacity,
                          const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F, int H, int W, float* g_color,
                          float* g_clip, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && color && work && count && clip && g_color && g_clip && C > 0 && B > 0 && V > 0 && H > 0 && W > 0);
    A3D_CHECK_ARG(clip_batch == 1 || clip_batch == B);
    hipStream_t s = (hipStream_t)stream;
    {
        const long long n = (long long)B * H * W * C, nz = 4ll * clip_batch * V;
        int blocks = a3d_div_up(n >> 2, 256 * 4);
        blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
        hipLaunchKernelGGL(aa_copy_zero_kernel, dim3(blocks), dim3(256), 0, s, g_out, g_color, n, g_clip, nz);
        A3D_LAUNCH_CHECK();
    }
    if (F == 0) return A3D_OK;
    A3D_CHECK_ARG(tri);
    hipLaunchKernelGGL(aa_bwd_kernel, dim3(1024), dim3(256), 0, s, g_out, color, C, (const AaRec*)work, count, capacity,
                       (const float4*)clip, clip_batch, tri, V, H, W, g_color, g_clip);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

static CaJob ca_job(const float* vals, int C, const int32_t* inv, const float* bg, int bg_batch, int H, int W, float* out, const float* g_out,
                    float* g_vals, const a3d_ca_buffer* ext = nullptr, long long P = 0) {
    CaJob j;
    j.s.vals = vals; j.s.inv = inv; j.s.rast = nullptr; j.s.sh_gb = nullptr; j.s.sh_par = ShPar{}; j.s.sh_kd = nullptr; j.s.sh_kd_stride = 0; j.s.sh_out = nullptr;
    j.s.sh_two_sided = 0; j.s.bg = bg; j.s.bg_shared = bg_batch == 1; j.s.C = C; j.s.hw = (unsigned)H * (unsigned)W;
    j.s.bgC = (ext && ext->bg_channels > 0) ? ext->bg_channels : C + 1;
    j.out = out; j.g_out = g_out; j.g_vals = g_vals; j.clear = nullptr; j.n_clear = 0; j.gHW = 0;
    j.oC = (ext && ext->out_channels > 0) ? ext->out_channels : C + 1;
    j.gS = (ext && ext->g_stride > 0) ? ext->g_stride : C + 1;
    j.gC = (ext && ext->g_channels > 0) ? ext->g_channels : C + 1;
    j.vals_rows = (ext && ext->vals_rows > P) ? ext->vals_rows : P;
    return j;
}
static bool ca_ext_ok(const a3d_ca_buffer* b) {
    return b->bg_channels >= 0 && b->bg_channels <= b->C + 1 && b->g_channels >= 0 && b->g_channels <= b->C + 1 && b->g_stride >= 0 &&
           (b->g_stride == 0 || b->g_stride >= (b->g_channels > 0 ? b->g_channels : b->C + 1)) && b->vals_rows >= 0 && b->out_channels >= 0 &&
           b->out_channels <= b->C + 1;
}

// the deferred shading of a compositor call's first buffer (a3d_ca_shade): the job's values come from sh_forward instead of vals
static int ca_shade(const a3d_ca_shade* sh, int C, const float* vals, CaJob* j, bool forward) {
    if (!sh) return A3D_OK;
    A3D_CHECK_ARG(sh->size >= sizeof(a3d_ca_shade));
    A3D_CHECK_ARG(!vals && C == 3 && sh->gb && (sh->par || sh->params) && sh->kd && sh->kd_stride >= 3 && sh->n_clear >= 0 && (sh->n_clear == 0 || sh->clear));
    if (sh->params) {
        const a3d_shade_params* q = sh->params;
        A3D_CHECK_ARG(q->size >= sizeof(a3d_shade_params) && q->rot && q->view && q->light && (q->rot_row_stride == 3 || q->rot_row_stride == 4));
        A3D_CHECK_ARG(q->rot_image_stride >= 0 && q->view_image_stride >= 0 && q->light_image_stride >= 0);
    }
    // (the colour computed on the spot exists only on the compose kernel's 16-byte path: a forward call whose image or 4-channel background
    // is not 16-byte aligned would take the general path, which has no shading source -- refused instead of composited wrongly)
    A3D_CHECK_ARG(!forward || (j->oC == 4 && (((uintptr_t)j->out | (j->s.bgC == 4 ? (uintptr_t)j->s.bg : 0)) & 15) == 0));
    j->s.sh_gb = sh->gb; j->s.sh_par = sh->params ? sh_par_of(sh->params) : sh_par_table(sh->par, 17); j->s.sh_kd = sh->kd;
    j->s.sh_kd_stride = sh->kd_stride; j->s.sh_two_sided = sh->two_sided;
    j->s.sh_out = forward ? sh->shaded_out : nullptr;
    if (forward) { j->clear = sh->clear; j->n_clear = sh->n_clear; }
    return A3D_OK;
}

// the riding analysis of a compositor call: the job for the extra work-groups of its first launch, or nothing
static int ca_ride(const a3d_aa_ride* r, const float* rast_override, void* work, int32_t* count, int capacity, int B, int H, int W, AaAnalyzeJob* an,
                   unsigned* nb_an) {
    *an = AaAnalyzeJob{};
    *nb_an = 0;
    if (!r) return A3D_OK;
    A3D_CHECK_ARG(r->size >= sizeof(a3d_aa_ride));
    const float* rast = rast_override ? rast_override : r->rast;
    A3D_CHECK_ARG(rast && r->screen && r->tri && r->V > 0 && r->F > 0 && (r->clip_batch == 1 || r->clip_batch == B));
    A3D_CHECK_ARG(r->opp || (r->off && r->adj));
    A3D_CHECK_ARG(capacity >= a3d_aa_capacity(B, H, W) && B <= 65535 && r->lists_stride >= 0 && H < 32768 && W < 65536);
    an->rast = (const float4*)rast; an->screen = (const float2*)r->screen; an->tri = r->tri; an->opp = r->opp;
    an->off = r->off; an->adj = r->adj; an->work = (AaRec*)work; an->count = count;
    an->clip_batch = r->clip_batch; an->V = r->V; an->F = r->F; an->H = H; an->W = W; an->capacity = capacity; an->B = B; an->stride = r->lists_stride;
    *nb_an = (unsigned)a3d_div_up((long long)H * W, 256) * (unsigned)B;
    return A3D_OK;
}

extern "C" int a3d_composite_aa_fwd(const a3d_ca_buffer* first, const a3d_ca_buffer* second_or_null, const int32_t* inv, void* work,
                                    int32_t* count, int capacity, int B, int H, int W, const a3d_aa_ride* analyze_or_null,
                                    const a3d_ca_shade* shade_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(first && first->size >= sizeof(a3d_ca_buffer) && (!second_or_null || second_or_null->size >= sizeof(a3d_ca_buffer)));
    const float *vals = first->vals, *bg_or_null = first->bg, *vals2_or_null = second_or_null ? second_or_null->vals : nullptr;
    const float* bg2_or_null = second_or_null ? second_or_null->bg : nullptr;
    const int C = first->C, bg_batch = first->bg_batch, C2 = second_or_null ? second_or_null->C : 0, bg2_batch = second_or_null ? second_or_null->bg_batch : 0;
    float *out = first->out, *out2_or_null = second_or_null ? second_or_null->out : nullptr;
    A3D_CHECK_ARG(!second_or_null || out2_or_null);
    A3D_CHECK_ARG(inv && work && count && out && C > 0 && C + 1 <= 4096 && B > 0 && H > 0 && W > 0 && capacity > 0);
    AaAnalyzeJob an;
    unsigned nb_an;
    if (int rc = ca_ride(analyze_or_null, nullptr, work, count, capacity, B, H, W, &an, &nb_an)) return rc;
    A3D_CHECK_ARG((long long)B * H * W < 0x7FFFFFFFll && (!bg_or_null || bg_batch == 1 || bg_batch == B));
    const bool two = out2_or_null != nullptr;
    A3D_CHECK_ARG(!two || (C2 > 0 && C2 + 1 <= 4096 && (!bg2_or_null || bg2_batch == 1 || bg2_batch == B)));
    hipStream_t s = (hipStream_t)stream;
    A3D_CHECK_ARG(ca_ext_ok(first) && (!second_or_null || ca_ext_ok(second_or_null)));
    CaJob ja = ca_job(vals, C, inv, bg_or_null, bg_batch, H, W, out, nullptr, nullptr, first);
    if (int rc = ca_shade(shade_or_null, C, vals, &ja, true)) return rc;
    const CaJob jb = two ? ca_job(vals2_or_null, C2, inv, bg2_or_null, bg2_batch, H, W, out2_or_null, nullptr, nullptr, second_or_null) : ja;
    const unsigned n_pix = (unsigned)B * ja.s.hw;
    const unsigned nb_compose = (unsigned)a3d_div_up(n_pix, 256), rows = two ? 2u : 1u;
    hipLaunchKernelGGL(ca_compose_kernel, dim3(nb_compose + (nb_an + rows - 1) / rows, rows), dim3(256), 0, s, ja, jb, n_pix, nb_compose, an, a3d_exp() == 43 ? 0 : 1);
    A3D_LAUNCH_CHECK();
    if (ja.s.sh_gb && ja.s.sh_out) {  // the colours the compose launch kept: the blends read them as plain value rows
        ja.s.vals = ja.s.sh_out;
        ja.s.sh_gb = nullptr;
    }
    hipLaunchKernelGGL(ca_blend_kernel, dim3(512, two ? 2 : 1), dim3(256), 0, s, ja, two ? jb : ja, (const AaRec*)work, count, capacity, W);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_composite_aa_bwd(const a3d_ca_buffer* first, const a3d_ca_buffer* second_or_null, const int64_t* pix, int64_t P,
                                    const int32_t* inv, const void* work, const int32_t* count, int capacity, const float* clip, int clip_batch,
                                    const int32_t* tri, int B, int V, int F, int H, int W, float* g_clip, const a3d_ca_shade* shade_or_null,
                                    a3d_stream_t stream) {
    A3D_CHECK_ARG(first && first->size >= sizeof(a3d_ca_buffer) && (!second_or_null || second_or_null->size >= sizeof(a3d_ca_buffer)));
    const float *g_out = first->g_out, *vals = first->vals, *bg_or_null = first->bg;
    const int C = first->C, bg_batch = first->bg_batch;
    float* g_vals = first->g_vals;
    const float *g_out2_or_null = second_or_null ? second_or_null->g_out : nullptr, *vals2 = second_or_null ? second_or_null->vals : nullptr;
    const float* bg2_or_null = second_or_null ? second_or_null->bg : nullptr;
    const int C2 = second_or_null ? second_or_null->C : 0, bg2_batch = second_or_null ? second_or_null->bg_batch : 0;
    float* g_vals2 = second_or_null ? second_or_null->g_vals : nullptr;
    A3D_CHECK_ARG(!second_or_null || g_out2_or_null);
    A3D_CHECK_ARG(g_out && inv && work && count && clip && g_clip && C > 0 && C <= 4096 && B > 0 && V > 0 && H > 0 && W > 0 && P >= 0 && capacity > 0);
    A3D_CHECK_ARG((long long)B * H * W < 0x7FFFFFFFll && (clip_batch == 1 || clip_batch == B) && (!bg_or_null || bg_batch == 1 || bg_batch == B));
    A3D_CHECK_ARG(P == 0 || ((vals || shade_or_null) && pix && g_vals));
    const bool two = g_out2_or_null != nullptr;
    A3D_CHECK_ARG(!two || (C2 > 0 && C2 <= 4096 && (!bg2_or_null || bg2_batch == 1 || bg2_batch == B) && (P == 0 || (vals2 && g_vals2))));
    hipStream_t s = (hipStream_t)stream;
    A3D_CHECK_ARG(ca_ext_ok(first) && (!second_or_null || ca_ext_ok(second_or_null)));
    CaJob ja = ca_job(vals, C, inv, bg_or_null, bg_batch, H, W, nullptr, g_out, g_vals, first, (long long)P);
    if (int rc = ca_shade(shade_or_null, C, vals, &ja, false)) return rc;
    const CaJob jb = two ? ca_job(vals2, C2, inv, bg2_or_null, bg2_batch, H, W, nullptr, g_out2_or_null, g_vals2, second_or_null, (long long)P) : ja;
    {
        const long long nz = 4ll * clip_batch * V;
        long long blocks = a3d_div_up(ja.vals_rows > jb.vals_rows ? ja.vals_rows : jb.vals_rows, 256);
        if (blocks < 64) blocks = 64;  // enough work-groups to zero g_clip when the point list is short
        hipLaunchKernelGGL(ca_gather_kernel, dim3((unsigned)blocks, two ? 2 : 1), dim3(256), 0, s, ja, jb, (const long long*)pix, (long long)P, g_clip, nz);
        A3D_LAUNCH_CHECK();
    }
    if (F == 0) return A3D_OK;
    A3D_CHECK_ARG(tri);
    hipLaunchKernelGGL(ca_bwd_kernel, dim3(1024, two ? 2 : 1), dim3(256), 0, s, ja, jb, (const AaRec*)work, count, capacity, (const float4*)clip,
                       clip_batch, tri, V, H, W, g_clip);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

// ---- the texture-less, light-less render: every covered pixel is (1, .., 1), the only thing a caller differentiates is the silhouette
// (Fauna's random-view mask for the discriminator, /root/reference/model/models/Fauna.py:111-173: render_mesh(material = None, lgt = None,
// render_modes = ['shaded']) and only the alpha channel leaves).  As a general render that is a covered-pixel list + G-buffer launch, a
// shading launch, a host read-back of the list's length and a compositor call over [P,3] rows of ones; here the compositor takes its
// coverage straight from the raster texels: no list, no rows, no read-back -- two launches forward (the silhouette analysis riding in
// the first as usual), one backward.
extern "C" int a3d_mask_aa_fwd(const float* rast, int C, const float* bg_or_null, int bg_batch, float* out, void* work, int32_t* count,
                               int capacity, int B, int H, int W, const a3d_aa_ride* analyze_or_null, a3d_stream_t stream) {
    A3D_CHECK_ARG(rast && work && count && out && C > 0 && C + 1 <= 4096 && B > 0 && H > 0 && W > 0 && capacity > 0);
    A3D_CHECK_ARG((long long)B * H * W < 0x7FFFFFFFll && (!bg_or_null || bg_batch == 1 || bg_batch == B));
    AaAnalyzeJob an;
    unsigned nb_an;
    if (int rc = ca_ride(analyze_or_null, rast, work, count, capacity, B, H, W, &an, &nb_an)) return rc;
    hipStream_t s = (hipStream_t)stream;
    CaJob ja = ca_job(nullptr, C, nullptr, bg_or_null, bg_batch, H, W, out, nullptr, nullptr);
    ja.s.rast = (const float4*)rast;
    const unsigned n_pix = (unsigned)B * ja.s.hw, nb_compose = (unsigned)a3d_div_up(n_pix, 256);
    hipLaunchKernelGGL(ca_compose_kernel, dim3(nb_compose + nb_an, 1), dim3(256), 0, s, ja, ja, n_pix, nb_compose, an, a3d_exp() == 43 ? 0 : 1);
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(ca_blend_kernel, dim3(512, 1), dim3(256), 0, s, ja, ja, (const AaRec*)work, count, capacity, W);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_mask_aa_bwd(const float* g_out, const float* rast, int C, const float* bg_or_null, int bg_batch, const void* work,
                               const int32_t* count, int capacity, const float* clip, int clip_batch, const int32_t* tri, int B, int V, int F,
                               int H, int W, float* g_clip, int g_channels_first, a3d_stream_t stream) {
    A3D_CHECK_ARG(g_out && rast && work && count && clip && g_clip && C > 0 && C <= 4096 && B > 0 && V > 0 && H > 0 && W > 0 && capacity > 0);
    A3D_CHECK_ARG((long long)B * H * W < 0x7FFFFFFFll && (clip_batch == 1 || clip_batch == B) && (!bg_or_null || bg_batch == 1 || bg_batch == B));
    hipStream_t s = (hipStream_t)stream;
    A3D_HIP(hipMemsetAsync(g_clip, 0, sizeof(float) * 4 * (size_t)clip_batch * V, s));
    if (F == 0) return A3D_OK;
    A3D_CHECK_ARG(tri);
    CaJob ja = ca_job(nullptr, C, nullptr, bg_or_null, bg_batch, H, W, nullptr, g_out, nullptr);
    ja.s.rast = (const float4*)rast;
    ja.gHW = g_channels_first ? H * W : 0;
    hipLaunchKernelGGL(ca_bwd_kernel, dim3(1024, 1), dim3(256), 0, s, ja, ja, (const AaRec*)work, count, capacity, (const float4*)clip, clip_batch,
                       tri, V, H, W, g_clip);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(antialias)
