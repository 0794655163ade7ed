// DMTet marching tetrahedra on gfx950 -- sort-free formulation.
//
// The reference (model/geometry/dmtet.py:104-155) gathers the six edges of every surface-crossing tet,
// sorts them globally inside torch.unique(dim=0) and numbers the surface vertices by rank.  Because the
// grid's unique edge list is static and already lexicographically sorted (dmtet.py:283-288), that rank is
// an exclusive prefix sum of the "sign crossing" flag over the static list.  So:
//   count : wave-ballot popcounts of the crossing flag per 1024-edge block and of the 1-/2-triangle case
//           per 1024-tet block, the ballots themselves kept as bit planes (1 bit/edge, 4 bits/tet), then one
//           work-group scan per block-sum array (-> V, n1, n2).  With the grid's static word groups (round 3) the pass is
//           CULLED: a sign-plane pre-pass, then a word of 64 index rows is read only if the <= 8 sixteen-vertex groups that
//           cover its vertices do not all lie on one side of the surface -- 96-98 % of the index stream is never read.
//   emit  : ONE launch over the bit planes: a crossing edge's vertex id is block prefix + word prefix + popcount below its bit;
//           crossing edges place their vertex, surface tets read their tet2edge row and write int64 faces -- and (round 3) the int32
//           triangle list of the render kernels and the mesh's vertex -> face lists, every vertex owning a fixed number of slots
//           (the grid bounds the valence), so that no topology launch follows.
//           (Until round 2 the emit re-gathered every SDF value and re-read both index arrays in two launches
//           chained through an edge -> vertex table: 23 us of kernel time against 8 now.)
// Integer/byte work: streaming form 8 B/edge + 16 B/tet of reads plus ~1e7 L2-resident 4-byte SDF gathers in the count pass (TA
// line rate bound), culled form 32 B per 64 rows of group ids + the rows of the few words that are read; ~1 MB of bit planes in the
// emit pass.  This TU is compiled with -ffp-contract=off: the vertex placement
// must round exactly like the reference's separate torch kernels (mul, mul, add).
#include "a3d_common.h"

#define DM_THREADS 256
#define DM_SLABS 4
#define DM_BLOCK_ITEMS (DM_THREADS * DM_SLABS)

// reference dmtet.py:26-45
__constant__ signed char c_tri_table[16][6] = {
    {-1, -1, -1, -1, -1, -1}, {1, 0, 2, -1, -1, -1}, {4, 0, 3, -1, -1, -1}, {1, 4, 2, 1, 3, 4},
    {3, 1, 5, -1, -1, -1},    {2, 3, 0, 2, 5, 3},    {1, 4, 0, 1, 5, 4},    {4, 2, 5, -1, -1, -1},
    {4, 5, 2, -1, -1, -1},    {4, 1, 0, 4, 5, 1},    {3, 2, 0, 3, 5, 2},    {1, 3, 5, -1, -1, -1},
    {4, 1, 2, 4, 3, 1},       {3, 0, 4, -1, -1, -1}, {2, 0, 1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}};
// number of triangles per case packed 2 bits each: [0,1,1,2,1,2,2,1,1,2,2,1,2,1,1,0]
#define DM_NTRI(c) ((0x16696994u >> (2 * (c))) & 3u)

static_assert(((0x16696994u >> 0) & 3) == 0 && ((0x16696994u >> 6) & 3) == 2 && ((0x16696994u >> 30) & 3) == 0, "ntri pack");

// "inside" = sdf > 0 (dmtet.py:107; 0 counts as outside) -- read from the SDF itself, or (BITS) from a 1-bit-per-vertex plane a small
// pre-pass left (dm_sign_kernel): on large grids the count pass is bound by its ~1e8 four-byte gathers (one cache line per 32 vertices);
// from the bit plane the same wave touches a handful of lines (one per 1024 vertices)
template <bool BITS>
__device__ __forceinline__ bool dm_inside(const void* __restrict__ src, int v) {
    if (BITS) return (reinterpret_cast<const unsigned*>(src)[v >> 5] >> (v & 31)) & 1u;
    return reinterpret_cast<const float*>(src)[v] > 0.f;
}

template <bool BITS>
__device__ __forceinline__ int dm_tet_case(const void* __restrict__ src, int4 t) {
    return (dm_inside<BITS>(src, t.x) ? 1 : 0) | (dm_inside<BITS>(src, t.y) ? 2 : 0) | (dm_inside<BITS>(src, t.z) ? 4 : 0) |
           (dm_inside<BITS>(src, t.w) ? 8 : 0);
}

template <bool BITS>
__device__ __forceinline__ bool dm_edge_cross(const void* __restrict__ src, int2 e) {
    return dm_inside<BITS>(src, e.x) != dm_inside<BITS>(src, e.y);  // exactly one endpoint inside (dmtet.py:118)
}

// every wave leaves DM_SIGN_WORDS words: lane l reads the vertices base + 64 j + l, all loads in flight, one ballot per word
#define DM_SIGN_WORDS 4
#define DM_TICKET_SLOT 8  // list_len[8]: work-groups of the culled count launch that have finished (dm_scan_tail)
__global__ __launch_bounds__(256) void dm_sign_kernel(const float* __restrict__ sdf, int Nv, unsigned long long* __restrict__ bits,
                                                      int* __restrict__ list_len) {
    A3D_STAMP(0, 0);  // (A3D_STAMP kernel ids of this file: 0 = dm_sign_kernel, 1 = dm_count_cull_kernel, 3 = dm_emit_kernel, 4 = dm_bwd_kernel)
    const int lane = threadIdx.x & 63;
    // (the two append counters of the count launch and, 32 bytes into the first one's line, the arrival ticket of its folded scan)
    if (list_len && blockIdx.x == 0 && threadIdx.x < 3) list_len[threadIdx.x == 2 ? DM_TICKET_SLOT : 16 * threadIdx.x] = 0;
    const long long w0 = ((long long)blockIdx.x * (256 / 64) + (threadIdx.x >> 6)) * DM_SIGN_WORDS;  // first word of this wave
    float x[DM_SIGN_WORDS];
#pragma unroll
    for (int j = 0; j < DM_SIGN_WORDS; ++j) {
        const long long v = (w0 + j) * 64 + lane;
        x[j] = sdf[v < Nv ? v : 0];
    }
#pragma unroll
    for (int j = 0; j < DM_SIGN_WORDS; ++j) {
        const unsigned long long m = __ballot((w0 + j) * 64 + lane < Nv && x[j] > 0.f);
        if (lane == 0 && (w0 + j) * 64 < Nv) bits[w0 + j] = m;
    }
    A3D_STAMP(0, 5);
}

// ------------------------------------------------------------------------------------------------ count
// Besides the block sums the pass leaves what it found as bit planes, so that the emit pass never gathers an SDF value or reads an
// index row for a tet or edge that is not on the surface (~1 % are): edge_bits[word] = crossing flags of 64 consecutive edges,
// tet_bits[word*4 + j] = bit j of the marching-tets case of 64 consecutive tets.
//
// Flags of the grid vertices at the ends of crossing edges (the only ones the surface's gradient reaches), one bit per vertex.  Straight
// into the plane every crossing edge is two device atomics, and a word of the plane -- 32 neighbouring vertices, each on ~7 crossing
// edges -- takes dozens of them: same-address atomics serialise at the memory side, and in the culled pass, where the few blocks
// that hold crossings ARE the critical path, draining them was a third of the kernel (R = 64: 8.4 us without the plane, 12.6 with).
// A work-group therefore ORs into a window of the plane kept in LDS (edges are sorted by their first vertex: a block's vertices start
// at its first edge's and span about one grid layer) and flushes the non-zero words once per block; what falls outside the window
// (irregular grids) goes to the plane directly.
#define DM_VWIN 1024  // words: 32k vertices
__device__ __forceinline__ void dm_flag_vertex(unsigned* __restrict__ vbits, unsigned* s_vwin, int vbase, int v) {
    const unsigned w = (unsigned)((v >> 5) - vbase);
    if (s_vwin && w < (unsigned)DM_VWIN) atomicOr(s_vwin + w, 1u << (v & 31));
    else atomicOr(vbits + (v >> 5), 1u << (v & 31));
}

// One 1024-item block (16 words; wave w owns the words k*4 + w).  ``skip`` (wave-uniform): bit k set = word k of this wave is known to
// hold no crossing (dm_count_cull_kernel below) -- its index rows are not loaded, its bits are written as zeros.
// a block sum: a plain store, or (publish: the scan rides in this launch, dm_scan_tail) a device-scope atomic, performed at the memory
// side like the plane's ORs and the list appends -- the scanning work-group sits on another XCD, whose L2 this XCD's stores do not reach
// inside a kernel.  (A release fence per work-group instead -- plain stores + __threadfence() -- writes the XCD's L2 back 841 times:
// the count call went from 20 to 88 us.)
__device__ __forceinline__ void dm_put(int* p, int v, bool publish) {
    if (publish) __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool BITS>
__device__ __forceinline__ void dm_count_block(const void* __restrict__ sdf, const int2* __restrict__ edges, const int4* __restrict__ tets,
                                               int Ne, int Nt, bool is_edge, int blk, unsigned skip, int* __restrict__ blk_e,
                                               int* __restrict__ blk_t1, int* __restrict__ blk_t2, unsigned long long* __restrict__ edge_bits,
                                               unsigned long long* __restrict__ tet_bits, int* __restrict__ wlocal,
                                               unsigned* __restrict__ vbits, int (*s_cnt)[DM_THREADS / A3D_WAVE], int* s_pc,
                                               int* __restrict__ list_len = nullptr, int* __restrict__ list = nullptr,
                                               unsigned* s_vwin = nullptr, bool publish = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c0 = 0, c1 = 0;
    const long long base = (long long)blk * DM_BLOCK_ITEMS;
    const int vbase = (is_edge && vbits && s_vwin) ? (edges[base < Ne ? base : 0].x >> 5) : 0;  // (uniform; in flight with the rows below)
    if (!BITS) {
        // SDF values gathered directly (grids below DM_SIGN_PLANE_MIN_NV vertices): row by row -- the pass is bound by the ~1e7 4-byte
        // gathers (TA line rate), and more of them in flight per lane only made it slower (17.7 -> 22 us at R = 64)
        if (is_edge) {
#pragma unroll
            for (int k = 0; k < DM_SLABS; ++k) {
                long long i = base + k * DM_THREADS + tid;
                const int2 e = i < Ne ? edges[i] : make_int2(0, 0);
                const bool f = i < Ne && dm_edge_cross<false>(sdf, e);
                if (f && vbits) {
                    dm_flag_vertex(vbits, s_vwin, vbase, e.x);
                    dm_flag_vertex(vbits, s_vwin, vbase, e.y);
                }
                const unsigned long long m = __ballot(f);
                if (lane == 0) {
                    edge_bits[(base >> 6) + k * (DM_THREADS / A3D_WAVE) + wave] = m;
                    s_pc[k * (DM_THREADS / A3D_WAVE) + wave] = __popcll(m);
                }
                c0 += __popcll(m);  // wave-uniform
            }
        } else {
#pragma unroll
            for (int k = 0; k < DM_SLABS; ++k) {
                long long i = base + k * DM_THREADS + tid;
                const int cs = i < Nt ? dm_tet_case<false>(sdf, tets[i]) : 0;
                const unsigned long long w0 = __ballot(cs & 1), w1 = __ballot(cs & 2), w2 = __ballot(cs & 4), w3 = __ballot(cs & 8);
                if (lane < 4) tet_bits[((base >> 6) + k * (DM_THREADS / A3D_WAVE) + wave) * 4 + lane] = lane == 0 ? w0 : (lane == 1 ? w1 : (lane == 2 ? w2 : w3));
                const unsigned n = DM_NTRI(cs);
                c0 += __popcll(__ballot(n == 1u));
                c1 += __popcll(__ballot(n == 2u));
            }
        }
    } else {
    // Signs from the bit plane (large grids): the gathers hit a handful of cache lines per wave, what is left is streaming 8 B/edge +
    // 16 B/tet -- and with one index row in flight per wave (load -> gathers -> ballot -> next row) that ran at 3 TB/s: too few bytes in
    // flight for the HBM latency.  All DM_SLABS rows of a thread are loaded up front, then all their sign lookups (R = 128: 98 -> 86 us)
    if (is_edge) {
        int2 e[DM_SLABS];
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            const long long i = base + k * DM_THREADS + tid;
            e[k] = make_int2(0, 0);
            if (!((skip >> k) & 1u) && i < Ne) e[k] = edges[i];
        }
        bool f[DM_SLABS];
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k)
            f[k] = !((skip >> k) & 1u) && (base + k * DM_THREADS + tid) < Ne && dm_edge_cross<BITS>(sdf, e[k]);
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            if (f[k] && vbits) {
                dm_flag_vertex(vbits, s_vwin, vbase, e[k].x);
                dm_flag_vertex(vbits, s_vwin, vbase, e[k].y);
            }
            const unsigned long long m = __ballot(f[k]);
            if (lane == 0) {
                edge_bits[(base >> 6) + k * (DM_THREADS / A3D_WAVE) + wave] = m;
                s_pc[k * (DM_THREADS / A3D_WAVE) + wave] = __popcll(m);
            }
            c0 += __popcll(m);  // wave-uniform
        }
    } else {
        int4 t[DM_SLABS];
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            const long long i = base + k * DM_THREADS + tid;
            t[k] = make_int4(0, 0, 0, 0);
            if (!((skip >> k) & 1u) && i < Nt) t[k] = tets[i];
        }
        int cs[DM_SLABS];
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k)
            cs[k] = (!((skip >> k) & 1u) && (base + k * DM_THREADS + tid) < Nt) ? dm_tet_case<BITS>(sdf, t[k]) : 0;
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            const unsigned long long w0 = __ballot(cs[k] & 1), w1 = __ballot(cs[k] & 2), w2 = __ballot(cs[k] & 4), w3 = __ballot(cs[k] & 8);
            if (lane < 4) tet_bits[((base >> 6) + k * (DM_THREADS / A3D_WAVE) + wave) * 4 + lane] = lane == 0 ? w0 : (lane == 1 ? w1 : (lane == 2 ? w2 : w3));
            const unsigned n = DM_NTRI(cs[k]);
            c0 += __popcll(__ballot(n == 1u));
            c1 += __popcll(__ballot(n == 2u));
        }
    }
    }
    if (lane == 0) { s_cnt[0][wave] = c0; s_cnt[1][wave] = c1; }
    __syncthreads();
    if (is_edge && vbits && s_vwin) {  // the window's non-zero words to the plane; left zeroed for the work-group's next block
        for (int i = tid; i < DM_VWIN; i += DM_THREADS) {
            const unsigned m = s_vwin[i];
            if (m) { atomicOr(vbits + vbase + i, m); s_vwin[i] = 0u; }
        }
    }
    if (tid == 0) {
        int a = 0, b = 0;
        for (int w = 0; w < DM_THREADS / A3D_WAVE; ++w) { a += s_cnt[0][w]; b += s_cnt[1][w]; }
        if (is_edge) dm_put(blk_e + blk, a, publish);
        else { dm_put(blk_t1 + blk, a, publish); dm_put(blk_t2 + blk, b, publish); }
        // the blocks that hold something, in any order: the emit launch then has one work-group set per listed block instead of one
        // per block of the grid (R = 128: ~1.4k of 27k)
        if (list && (a | b) != 0) list[atomicAdd(list_len, 1)] = blk;
    }
    if (is_edge && tid < DM_BLOCK_ITEMS / 64) {  // crossings of this block before word tid
        int before = 0;
        for (int j = 0; j < tid; ++j) before += s_pc[j];
        wlocal[(long long)blk * (DM_BLOCK_ITEMS / 64) + tid] = before;
    }
}

template <bool BITS>
__global__ __launch_bounds__(DM_THREADS) void dm_count_kernel(const void* __restrict__ sdf, const int2* __restrict__ edges,
                                                                const int4* __restrict__ tets, int Ne, int Nt, int nbe,
                                                                int* __restrict__ blk_e, int* __restrict__ blk_t1,
                                                                int* __restrict__ blk_t2, unsigned long long* __restrict__ edge_bits,
                                                                unsigned long long* __restrict__ tet_bits, int* __restrict__ wlocal,
                                                                unsigned* __restrict__ vbits) {
    __shared__ int s_cnt[2][DM_THREADS / A3D_WAVE];
    __shared__ int s_pc[DM_BLOCK_ITEMS / 64];  // crossings per 64-edge word of this block
    const bool is_edge = (int)blockIdx.x < nbe;
    dm_count_block<BITS>(sdf, edges, tets, Ne, Nt, is_edge, is_edge ? (int)blockIdx.x : (int)blockIdx.x - nbe, 0u, blk_e, blk_t1, blk_t2,
                         edge_bits, tet_bits, wlocal, vbits, s_cnt, s_pc);
}

// ---- the scan inside the count launch (round 6, VERDICT r5 item 3: built, measured, kept OFF -- see a3d_dmtet_count).  dm_scan_kernel
// below is four tiny scans in a launch of their own: 6.3 us of kernel for ~5k integers, plus the gap in front of it, on the critical
// path between the count and the emit.  Here every work-group of the culled count launch publishes its block sums as device-scope
// atomics (dm_put) and takes a ticket; the one that draws the LAST ticket acquires and scans: wave w owns
// array w (edge blocks, one-triangle tets, two-triangle tets, 1024-vertex chunks of the surface-vertex plane) as rows of 64 consecutive
// sums -- every row loaded up front (coalesced), scanned with shuffles, chained through a wave-uniform carry: no LDS, no barrier, for
// up to DM_TAIL_ROWS * 64 sums per array.  Larger grids keep the separate launch (its 1024 threads and 96 KB of LDS are what 15k sums need).
// The chunk popcounts of the vertex plane are taken by all four waves first (8 lanes per 128-byte chunk) into LDS.
#define DM_TAIL_ROWS 32
#define DM_TAIL_MAX (DM_TAIL_ROWS * 64)

// in-place exclusive scan of arr[0, n) (global, or LDS for the chunk counts -> written to `dst`) by ONE wave; returns the total.  (The rows'
// inclusive scans: a3d_wave_incl_scan, six DPP additions.  The shuffle form -- six ds_bpermute round trips per row of sums, chained row
// after row behind wave-uniform branches -- made the folded scan 17 us slower than the launch it replaces.)
template <bool COHERENT>
__device__ __forceinline__ int dm_wave_scan_rows(const int* src, int* dst, int n) {
    const int lane = threadIdx.x & 63;
    int v[DM_TAIL_ROWS];
#pragma unroll
    for (int j = 0; j < DM_TAIL_ROWS; ++j) {
        const int i = j * 64 + lane;
        v[j] = i < n ? (COHERENT ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i]) : 0;
    }
    int carry = 0;
#pragma unroll
    for (int j = 0; j < DM_TAIL_ROWS; ++j) {  // (every row, no branch: the rows' chains interleave; rows past n hold zeros)
        const int incl = a3d_wave_incl_scan(v[j]);
        const int i = j * 64 + lane;
        if (i < n) dst[i] = carry + incl - v[j];
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    return carry;
}

__device__ __forceinline__ void dm_scan_tail(int* __restrict__ blk_e, int* __restrict__ blk_t1, int* __restrict__ blk_t2, int nbe, int nbt,
                                             int* __restrict__ counts, const unsigned* __restrict__ vbits, int* __restrict__ vchunk, int nvc,
                                             const int* __restrict__ list_len, int* s_chunk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (vbits) {  // chunk popcounts -> LDS (nvc <= DM_VWIN, checked by the host)
        const uint4* plane = reinterpret_cast<const uint4*>(vbits);
        const int n16 = nvc * 8;
        for (int i0 = tid; i0 < n16; i0 += 8 * DM_THREADS) {
            uint4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = plane[i0 + DM_THREADS * k < n16 ? i0 + DM_THREADS * k : i0];  // (first touch after the acquire)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int c = __popc(x[k].x) + __popc(x[k].y) + __popc(x[k].z) + __popc(x[k].w);
                c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
                if ((lane & 7) == 0 && i0 + DM_THREADS * k < n16) s_chunk[(i0 + DM_THREADS * k) >> 3] = c;
            }
        }
        __syncthreads();
    }
    if (wave == 3) {
        const int total = vbits ? dm_wave_scan_rows<false>(s_chunk, vchunk, nvc) : 0;
        if (lane == 0) counts[3] = total;
    } else {
        int* arr = wave == 0 ? blk_e : (wave == 1 ? blk_t1 : blk_t2);
        const int total = dm_wave_scan_rows<true>(arr, arr, wave == 0 ? nbe : nbt);
        if (lane == 0) {
            counts[wave] = total;
            // how many edge / tet blocks the count launch listed as non-empty
            if (wave == 1) counts[4] = __hip_atomic_load(list_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (wave == 2) counts[5] = __hip_atomic_load(list_len + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The same pass with a cull in front.  ~99 % of the words hold no crossing, and which vertices a word's 64 rows touch is a property of
// the grid: ``groups`` [words x 8] (built once per grid, model/geometry/dmtet.py: TetGridTopology.word_groups) lists the <= 8 aligned
// 16-vertex groups that cover them -- on a grid numbered along its rows (the Kuhn grids; any grid whose generator walks space) 64
// consecutive tets touch ~48 vertices in 4 short runs.  A word all of whose groups read 0x0000 or all 0xffff in the sign plane has all
// its vertices on one side: no row of it crosses, its 1 KB of index rows is never read.  What is read instead is 32 B of group ids
// and 8 two-byte sign fields (cache resident).  Words with more than 8 groups carry 0xffffffff (always processed); a grid where most
// do (scrambled numbering) is run through dm_count_kernel instead.  G consecutive blocks per work-group: their 4 G words per wave are
// culled with ALL group ids in flight at once, then ALL sign fields -- two round trips per G blocks instead of per block.
#define DM_CULL_SLOTS 8
#define DM_CULL_BLOCKS 4

template <int G, int SLOTS>
__global__ __launch_bounds__(DM_THREADS) void dm_count_cull_kernel(const unsigned* __restrict__ sign, const int2* __restrict__ edges,
                                                                   const int4* __restrict__ tets, int Ne, int Nt, int nbe, int nbt, int nge,
                                                                   const unsigned* __restrict__ edge_groups,
                                                                   const unsigned* __restrict__ tet_groups, int* __restrict__ blk_e,
                                                                   int* __restrict__ blk_t1, int* __restrict__ blk_t2,
                                                                   unsigned long long* __restrict__ edge_bits,
                                                                   unsigned long long* __restrict__ tet_bits, int* __restrict__ wlocal,
                                                                   unsigned* __restrict__ vbits, int* __restrict__ list_len,
                                                                   int* __restrict__ elist, int* __restrict__ tlist, int fold_scan,
                                                                   int* __restrict__ counts, int* __restrict__ vchunk, int nvc,
                                                                   int* __restrict__ clear, int n_clear) {
    __shared__ int s_cnt[2][DM_THREADS / A3D_WAVE];
    __shared__ int s_pc[DM_BLOCK_ITEMS / 64];
    __shared__ unsigned s_nib[G][DM_THREADS / A3D_WAVE];
    __shared__ unsigned s_vwin[DM_VWIN];
    __shared__ int s_last;
    A3D_STAMP(1, 0);
    // (folded scan only) the valence counters of the mesh the emit launch is about to build, zeroed here as dm_scan_kernel does
    if (fold_scan)
        for (int z = blockIdx.x * DM_THREADS + threadIdx.x; z < n_clear; z += gridDim.x * DM_THREADS) clear[z] = 0;
    // SLOTS = 8 (grids numbered along their rows: the Kuhn grids) or 16 (round 4: spatially coherent files whose words touch more groups
    // -- a BCC lattice in its generator's order: 10-13 -- at 64 instead of 32 bytes of table per word)
    constexpr int WPB = DM_BLOCK_ITEMS / 64, WPR = 64 / SLOTS, ROUNDS = (G * DM_SLABS * SLOTS + 63) / 64;
    static_assert((SLOTS == 8 || SLOTS == 16) && DM_SLABS == 4 && DM_THREADS == 256, "SLOTS bits of a ballot per word, one nibble of skip bits per block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool is_edge = (int)blockIdx.x < nge;
    // block g of this work-group = first + g * step: blocks that hold crossings come in runs (the surface), and a work-group that owned
    // G neighbours would process a whole run serially while the others write zeros
    const int nblk = is_edge ? nbe : nbt, step = is_edge ? nge : (int)gridDim.x - nge;
    const int first = is_edge ? (int)blockIdx.x : (int)blockIdx.x - nge;
    const unsigned* __restrict__ groups = is_edge ? edge_groups : tet_groups;
    // entry = (word slot q = g * DM_SLABS + k, group j): lane -> (q = lane / SLOTS + WPR r, j = lane % SLOTS)
    unsigned gid[ROUNDS], val[ROUNDS];
    bool in[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int q = lane / SLOTS + WPR * r, g = q / DM_SLABS, k = q - g * DM_SLABS;
        in[r] = q < G * DM_SLABS && first + g * step < nblk;
        // (unconditional loads at a clamped index, then the select: a load under a condition is a branch with its own wait)
        gid[r] = groups[((long long)(in[r] ? first + g * step : first) * WPB + k * (DM_THREADS / A3D_WAVE) + wave) * SLOTS + (lane % SLOTS)];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned field = reinterpret_cast<const unsigned short*>(sign)[gid[r] != 0xFFFFFFFFu ? gid[r] : 0u];
        val[r] = gid[r] != 0xFFFFFFFFu ? field : 1u;  // 1 = mixed
    }
    unsigned skip = 0;  // bit q: word slot q holds no crossing
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned long long z = __ballot(in[r] && val[r] == 0u), o = __ballot(in[r] && val[r] == 0xFFFFu);
        constexpr unsigned long long M = SLOTS == 16 ? 0xFFFFull : 0xFFull;
#pragma unroll
        for (int b = 0; b < WPR; ++b)
            if (((z >> (SLOTS * b)) & M) == M || ((o >> (SLOTS * b)) & M) == M) skip |= 1u << (WPR * r + b);
    }
    A3D_STAMP(1, 1);
    if (lane < G) s_nib[lane][wave] = (skip >> (DM_SLABS * lane)) & 15u;
    if (vbits && is_edge)
        for (int i = tid; i < DM_VWIN; i += DM_THREADS) s_vwin[i] = 0u;
    __syncthreads();
    A3D_STAMP(1, 2);
    bool lds_used = false;
    for (int g = 0; g < G; ++g) {
        const int blk = first + g * step;
        if (blk >= nblk) break;
        if ((s_nib[g][0] & s_nib[g][1] & s_nib[g][2] & s_nib[g][3]) == 15u) {
            // no word of this block holds a crossing (most blocks): its outputs are zeros, written by the first threads -- no loads, no
            // ballots, no barrier
            if (is_edge) {
                if (tid < WPB) { edge_bits[(long long)blk * WPB + tid] = 0ull; wlocal[(long long)blk * WPB + tid] = 0; }
                if (tid == 0) dm_put(blk_e + blk, 0, fold_scan);
            } else {
                if (tid < 4 * WPB) tet_bits[(long long)blk * 4 * WPB + tid] = 0ull;
                if (tid == 0) { dm_put(blk_t1 + blk, 0, fold_scan); dm_put(blk_t2 + blk, 0, fold_scan); }
            }
            continue;
        }
        if (lds_used) __syncthreads();  // s_cnt / s_pc of the previous processed block consumed
        lds_used = true;
        dm_count_block<true>(sign, edges, tets, Ne, Nt, is_edge, blk, (skip >> (DM_SLABS * g)) & 15u, blk_e, blk_t1, blk_t2, edge_bits, tet_bits,
                             wlocal, vbits, s_cnt, s_pc, list_len + (is_edge ? 0 : 16), is_edge ? elist : tlist, s_vwin, fold_scan != 0);
    }
    A3D_STAMP(1, 5);
    if (fold_scan) {
        // everything the scan reads -- block sums, plane words, list lengths -- left this work-group as device-scope atomics: once every
        // thread's have been acknowledged (vmcnt) the work-group draws its ticket; no fence, no cache write-back
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) s_last = __hip_atomic_fetch_add(list_len + DM_TICKET_SLOT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
        __syncthreads();
        A3D_STAMP(1, 6);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (this XCD's L2 forgets what it may hold of those lines)
            dm_scan_tail(blk_e, blk_t1, blk_t2, nbe, nbt, counts, vbits, vchunk, nvc, list_len, reinterpret_cast<int*>(s_vwin));
            A3D_STAMP(1, 7);
        }
    }
}

// ------------------------------------------------------------------------------------------------ count, grid in any numbering
// A grid file written by an external mesher (the reference's data/tets/{128,256}_tets.npz, Quartet) numbers its vertices and orders its
// rows without regard to space: 64 consecutive rows of `edges` / `tets` then touch ~70 / ~250 vertices anywhere in the grid, two words
// in three hold a crossing, and the cull above has nothing to skip -- while the OUTPUT order (vertex = rank of its crossing edge in
// the sorted list, faces in tet order) is defined by exactly that numbering.  So the signs are evaluated in an order of OUR choosing
// and only the results are carried back:
//   * static per grid (a3d_dmtet_order; model/geometry/dmtet.py: TetGridTopology.spatial_order): the vertices ranked along a Morton
//     curve through their positions, the edge rows rewritten in ranks and sorted by (lower rank, higher rank), the tet rows rewritten
//     in ranks (corner order kept: the case index depends on it) and sorted by their lowest rank, each row with the index of the
//     row it came from, and the word-group tables of THOSE lists (16 slots: 92 % of the words of a scrambled BCC lattice skip);
//   * dm_sign_order_kernel: one sign bit per RANK (sdf gathered through vertex_of_rank: Nv four-byte gathers), and the planes and
//     block sums of the original order zeroed by the same launch;
//   * dm_count_order_kernel: the cull over the ranked rows, wave by wave, no LDS, no barrier; a crossing edge / a surface tet (~1 % of
//     the rows of the ~8 % of the words that are read at all) ORs its bit(s) into the plane of the ORIGINAL order at the row it came
//     from and adds one to that row's block sum -- fire-and-forget atomics, about 2 V + 3 T of them;
//   * the scan launch, as before, plus work-groups that leave wlocal (crossings of the same block before a word) from the finished
//     edge plane, which the streaming pass gets from its ballots.
// From there on (a3d_dmtet_emit) nothing differs: same planes (a tet with all four corners inside reads as case 0 instead of 15; both
// mean "no triangle"), same prefixes, same output bits.
__global__ __launch_bounds__(256) void dm_sign_order_kernel(const float* __restrict__ sdf, const int* __restrict__ vertex_of_rank, int Nv,
                                                            unsigned long long* __restrict__ bits, uint4* __restrict__ clear,
                                                            long long n_clear16) {
    const int lane = threadIdx.x & 63;
    const long long w0 = ((long long)blockIdx.x * (256 / 64) + (threadIdx.x >> 6)) * DM_SIGN_WORDS;  // first word of this wave
    int v[DM_SIGN_WORDS];
#pragma unroll
    for (int j = 0; j < DM_SIGN_WORDS; ++j) {
        const long long r = (w0 + j) * 64 + lane;
        v[j] = vertex_of_rank[r < Nv ? r : 0];
    }
    for (long long z = (long long)blockIdx.x * 256 + threadIdx.x; z < n_clear16; z += (long long)gridDim.x * 256) clear[z] = make_uint4(0u, 0u, 0u, 0u);
    float x[DM_SIGN_WORDS];
#pragma unroll
    for (int j = 0; j < DM_SIGN_WORDS; ++j) x[j] = sdf[v[j]];
#pragma unroll
    for (int j = 0; j < DM_SIGN_WORDS; ++j) {
        const unsigned long long m = __ballot((w0 + j) * 64 + lane < Nv && x[j] > 0.f);
        if (lane == 0 && (w0 + j) * 64 < Nv) bits[w0 + j] = m;
    }
}

#define DM_ORDER_WORDS 16  // words per wave, strided over the list: the words that hold crossings come in runs (the surface)

template <int SLOTS>
__global__ __launch_bounds__(256) void dm_count_order_kernel(const unsigned* __restrict__ sign, const int2* __restrict__ edges_r,
                                                             const int* __restrict__ edge_of_row, const int4* __restrict__ tets_r,
                                                             const int* __restrict__ tet_of_row, const int* __restrict__ vertex_of_rank,
                                                             int Ne, int Nt, int nwe, int nwt, int waves_e, int waves_t,
                                                             const unsigned* __restrict__ edge_groups, const unsigned* __restrict__ tet_groups,
                                                             int* __restrict__ blk_e, int* __restrict__ blk_t1, int* __restrict__ blk_t2,
                                                             unsigned long long* __restrict__ edge_bits,
                                                             unsigned long long* __restrict__ tet_bits, unsigned* __restrict__ vbits) {
    static_assert(SLOTS == 8 || SLOTS == 16, "group ids per word");
    static_assert(DM_BLOCK_ITEMS == 1024, "block of a row = row >> 10");
    constexpr int WPR = 64 / SLOTS, ROUNDS = DM_ORDER_WORDS / WPR;  // words per round of 64 lanes
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave_g >= waves_e + waves_t) return;
    const bool is_edge = wave_g < waves_e;
    const int wv = is_edge ? wave_g : wave_g - waves_e, nwaves = is_edge ? waves_e : waves_t, nw = is_edge ? nwe : nwt;
    const unsigned* __restrict__ groups = is_edge ? edge_groups : tet_groups;
    // word q of this wave = q * nwaves + wv; entry = (word, slot): lane -> (q = lane / SLOTS + WPR r, slot = lane % SLOTS); all group
    // ids in flight, then all sign fields (unconditional loads at a clamped index, select afterwards)
    unsigned gid[ROUNDS], val[ROUNDS];
    bool in[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const long long w = (long long)(lane / SLOTS + WPR * r) * nwaves + wv;
        in[r] = w < nw;
        gid[r] = groups[(in[r] ? w : (long long)wv) * SLOTS + (lane % SLOTS)];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned field = reinterpret_cast<const unsigned short*>(sign)[gid[r] != 0xFFFFFFFFu ? gid[r] : 0u];
        val[r] = gid[r] != 0xFFFFFFFFu ? field : 1u;  // 1 = mixed
    }
    unsigned todo = 0;  // bit q: word q exists and may hold a crossing
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const unsigned long long z = __ballot(in[r] && val[r] == 0u), o = __ballot(in[r] && val[r] == 0xFFFFu), e = __ballot(in[r]);
        constexpr unsigned long long M = SLOTS == 16 ? 0xFFFFull : 0xFFull;
#pragma unroll
        for (int b = 0; b < WPR; ++b) {
            const bool all_out = ((z >> (SLOTS * b)) & M) == M, all_in = ((o >> (SLOTS * b)) & M) == M, exists = ((e >> (SLOTS * b)) & 1ull) != 0;
            if (exists && !all_out && !all_in) todo |= 1u << (WPR * r + b);
        }
    }
    while (todo) {  // (wave-uniform)
        const int q = __ffs(todo) - 1;
        todo &= todo - 1u;
        const long long i = ((long long)q * nwaves + wv) * 64 + lane;
        if (is_edge) {
            const bool ok = i < Ne;
            const int2 e = edges_r[ok ? i : 0];
            const int orig = edge_of_row[ok ? i : 0];
            if (ok && dm_edge_cross<true>(sign, e)) {
                atomicOr(edge_bits + (orig >> 6), 1ull << (orig & 63));
                atomicAdd(blk_e + (orig >> 10), 1);
                if (vbits) {  // the grid vertices at the ends of crossing edges, in the numbering of the file
                    const int a = vertex_of_rank[e.x], b = vertex_of_rank[e.y];
                    atomicOr(vbits + (a >> 5), 1u << (a & 31));
                    atomicOr(vbits + (b >> 5), 1u << (b & 31));
                }
            }
        } else {
            const bool ok = i < Nt;
            const int4 t = tets_r[ok ? i : 0];
            const int orig = tet_of_row[ok ? i : 0];
            const int cs = ok ? dm_tet_case<true>(sign, t) : 0;
            const unsigned n = DM_NTRI(cs);
            if (n) {
                unsigned long long* w = tet_bits + 4ll * (orig >> 6);
                const unsigned long long bit = 1ull << (orig & 63);
                if (cs & 1) atomicOr(w + 0, bit);
                if (cs & 2) atomicOr(w + 1, bit);
                if (cs & 4) atomicOr(w + 2, bit);
                if (cs & 8) atomicOr(w + 3, bit);
                atomicAdd((n == 1u ? blk_t1 : blk_t2) + (orig >> 10), 1);
            }
        }
    }
}

// three work-groups, one per block-sum array: in-place exclusive scan, totals to counts[0..2].  With the count pass's wlocal (crossings
// of the same block before a 64-edge word) a surface vertex id is blk_e[e >> 10] + wlocal[e >> 6] + popcount(edge_bits[e >> 6] below
// bit e & 63): three small loads, no edge -> vertex table.  (Extending this scan to words here, 29k of them in one work-group, cost 22 us.)
#define DM_SCAN_LDS 24576  // block sums a scan work-group stages through LDS (96 KB of the CU's 160)
__global__ __launch_bounds__(1024) void dm_scan_kernel(int* __restrict__ blk_e, int* __restrict__ blk_t1, int* __restrict__ blk_t2,
                                                       int nbe, int nbt, int* __restrict__ counts, const unsigned* __restrict__ vbits,
                                                       int* __restrict__ vchunk, int nvc, int* __restrict__ clear, int n_clear,
                                                       const int* __restrict__ list_len, const unsigned long long* __restrict__ edge_bits,
                                                       int* __restrict__ wlocal, int n_scans, const unsigned long long* __restrict__ tet_bits,
                                                       int* __restrict__ tlocal) {
    __shared__ int s_wave[16];
    __shared__ int s_arr[DM_SCAN_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int which = blockIdx.x;
    // the valence counters of the mesh this extraction is about to emit (a3d_dmtet_emit: topo_count), zeroed here: no memset launch
    for (int z = blockIdx.x * 1024 + tid; z < n_clear; z += gridDim.x * 1024) clear[z] = 0;
    if (which >= n_scans) {
        // (ordered count pass only) wlocal from the finished edge plane: thread = word, the 16 words of a block = 16 neighbouring lanes
        static_assert(DM_BLOCK_ITEMS / 64 == 16, "a block's words are one 16-lane row");
        // and tlocal (one- / two-triangle tets of the same block before a word, 16 + 16 bits) from the tet planes
        const int wg_e = (nbe * 16 + 1023) / 1024;
        const bool is_edge = which - n_scans < wg_e;
        const long long w = (long long)(which - n_scans - (is_edge ? 0 : wg_e)) * 1024 + tid;
        int c = 0;
        if (is_edge) {
            if (w < (long long)nbe * 16) c = __popcll(edge_bits[w]);
        } else if (w < (long long)nbt * 16) {
            const ulonglong2 pa = reinterpret_cast<const ulonglong2*>(tet_bits)[2 * w], pb = reinterpret_cast<const ulonglong2*>(tet_bits)[2 * w + 1];
            const unsigned long long odd = pa.x ^ pa.y ^ pb.x ^ pb.y;
            c = __popcll(odd) | (__popcll(~odd & (pa.x | pa.y | pb.x | pb.y) & ~(pa.x & pa.y & pb.x & pb.y)) << 16);
        }
        int incl = c;  // (both halves at once: neither sum reaches 2^16)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int up = __shfl_up(incl, o, 16);
            if ((lane & 15) >= o) incl += up;
        }
        if (is_edge) {
            if (w < (long long)nbe * 16) wlocal[w] = incl - c;
        } else if (w < (long long)nbt * 16) tlocal[w] = incl - c;
        return;
    }
    if (which == 3) {  // surface-adjacent grid vertices: bits per 1024-vertex chunk (32 words), exclusive prefix over the chunks -> counts[3]
        if (nvc <= DM_SCAN_LDS) {
            // the plane is read with consecutive lanes on consecutive 16 bytes (eight lanes = one chunk, their popcounts met through
            // shuffles) and the chunk counts scanned in LDS.  One thread per chunk run -- lanes 384 bytes apart at R = 128 -- made every
            // load instruction 64 separate lines, and this single work-group the slowest of the launch: 12 of the scan's 13 us there
            const uint4* plane = reinterpret_cast<const uint4*>(vbits);
            const int n16 = nvc * 8;
            for (int i0 = tid; i0 < n16; i0 += 8 * 1024) {
                uint4 x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = plane[i0 + 1024 * k < n16 ? i0 + 1024 * k : i0];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int c = __popc(x[k].x) + __popc(x[k].y) + __popc(x[k].z) + __popc(x[k].w);
                    c += __shfl_xor(c, 1, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 4, 64);
                    if ((lane & 7) == 0 && i0 + 1024 * k < n16) s_arr[(i0 + 1024 * k) >> 3] = c;
                }
            }
            __syncthreads();
            const int per = ((nvc + 1023) / 1024) | 1;
            const int lo = min(tid * per, nvc), hi = min(lo + per, nvc);
            int mine = 0;
            for (int i = lo; i < hi; ++i) mine += s_arr[i];
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                int up = __shfl_up(incl, o, 64);
                if (lane >= o) incl += up;
            }
            if (lane == 63) s_wave[wave] = incl;
            __syncthreads();
            int run = incl - mine;
            for (int w = 0; w < wave; ++w) run += s_wave[w];
            for (int i = lo; i < hi; ++i) { const int v = s_arr[i]; s_arr[i] = run; run += v; }
            __syncthreads();
            for (int i = tid; i < nvc; i += 1024) vchunk[i] = s_arr[i];
            if (tid == 1023) counts[3] = run;
            return;
        }
        const int per = (nvc + 1023) / 1024;
        const int lo = min(tid * per, nvc), hi = min(lo + per, nvc);
        for (int c = lo; c < hi; ++c) {
            const uint4* w = reinterpret_cast<const uint4*>(vbits + 32ll * c);
            int n = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { const uint4 x = w[q]; n += __popc(x.x) + __popc(x.y) + __popc(x.z) + __popc(x.w); }
            vchunk[c] = n;
        }
        const int mine = a3d_run_sum(vchunk, lo, hi);  // (this thread's own stores)
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int w = 0; w < wave; ++w) run += s_wave[w];
        run = a3d_run_scan<false>(vchunk, vchunk, lo, hi, run);
        if (tid == 1023) counts[3] = run;
        return;
    }
    int* arr = which == 0 ? blk_e : (which == 1 ? blk_t1 : blk_t2);
    const int n = which == 0 ? nbe : nbt;
    // every thread owns a contiguous run of ceil(n / 1024) block sums: one pass, one barrier (a loop over 1024-element slabs with
    // three barriers each cost 19 us at n = 1.5e4, the R = 128 grid).  Up to DM_SCAN_LDS sums the array goes through LDS: read and
    // written back with coalesced accesses, eight in flight per thread -- the runs themselves, read from memory, are 60-byte strides
    // between lanes and, at 15 elements per run, eight dependent round trips (13.8 us of the R = 128 count call)
    const bool staged = n <= DM_SCAN_LDS;
    if (staged) {
        for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
            int v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = arr[i0 + 1024 * k < n ? i0 + 1024 * k : i0];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + 1024 * k < n) s_arr[i0 + 1024 * k] = v[k];
        }
        __syncthreads();
    }
    const int per = staged ? (((n + 1023) / 1024) | 1) : (n + 1023) / 1024;  // (odd: the runs of consecutive lanes start in different banks)
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int mine = 0;
    if (staged) for (int i = lo; i < hi; ++i) mine += s_arr[i];
    else mine = a3d_run_sum(arr, lo, hi);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    if (staged) {
        for (int i = lo; i < hi; ++i) { const int v = s_arr[i]; s_arr[i] = run; run += v; }
        __syncthreads();
        for (int i0 = tid; i0 < n; i0 += 8 * 1024) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i0 + 1024 * k < n) arr[i0 + 1024 * k] = s_arr[i0 + 1024 * k];
        }
    } else {
        run = a3d_run_scan<false>(arr, arr, lo, hi, run);
    }
    if (tid == 1023) {
        counts[which] = run;
        if (which == 0 && !vbits) counts[3] = 0;
        // how many edge / tet blocks the count launch listed as non-empty (-1: no lists, a3d_dmtet_emit covers every block)
        if (which == 1) counts[4] = list_len ? list_len[0] : -1;
        if (which == 2) counts[5] = list_len ? list_len[16] : -1;
    }
}

// sorted list of the flagged grid vertices: chunk c = vertices [1024 c, 1024 c + 1024); entry = chunk prefix + bits below.  Runs as
// extra work-groups of the emit launch; every work-group clears the 32 words it consumed, which leaves the bit plane armed for the
// next count (vertex_scratch_is_clean: no memset).
// (round 6) ``pts`` (optional): the grid positions of the listed vertices, row for row -- pts[i] = pos[idx[i]] -- and zero rows behind up to
// the next multiple of ``bucket`` (n_true = the length of the list): the block the SDF network is re-evaluated on
// (DMTetGeometry._get_mesh_surface_backward), which was a gather launch of its own (a3d_dmtet_gather_rows) right behind this one.
__device__ __forceinline__ void dm_surface_vertices_chunk(int c, unsigned* __restrict__ vbits, const int* __restrict__ vchunk, int Nv,
                                                          long long* __restrict__ idx, int* s_pre, const float* __restrict__ pos = nullptr,
                                                          float* __restrict__ pts = nullptr, int bucket = 0, int n_true = 0) {
    if (pts && c == 0 && bucket > 0) {  // the padding rows
        const int rows = (n_true + bucket - 1) / bucket * bucket;
        for (int z = threadIdx.x; z < 3 * (rows - n_true); z += blockDim.x) pts[3ll * n_true + z] = 0.f;
    }
    unsigned mine = 0;
    if (threadIdx.x < 64) {  // (first wave) exclusive prefix of the 32 word popcounts through shuffles
        mine = threadIdx.x < 32 ? vbits[32ll * c + threadIdx.x] : 0u;
        const int n = __popc(mine);
        int incl = n;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += up;
        }
        if (threadIdx.x < 32) s_pre[threadIdx.x] = incl - n;
    }
    __syncthreads();
    const int base = vchunk[c];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int local = k * 256 + threadIdx.x, v = c * 1024 + local;
        const unsigned word = vbits[32ll * c + (local >> 5)];
        if (v < Nv && ((word >> (local & 31)) & 1u)) {
            const long long slot = base + s_pre[local >> 5] + __popc(word & ((1u << (local & 31)) - 1u));
            idx[slot] = v;
            if (pts) {
                const float x = pos[3ll * v], y = pos[3ll * v + 1], z = pos[3ll * v + 2];
                pts[3 * slot] = x; pts[3 * slot + 1] = y; pts[3 * slot + 2] = z;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 32 && mine) vbits[32ll * c + threadIdx.x] = 0u;
}

// ------------------------------------------------------------------------------------------------ emit
// One launch: edge work-groups place the surface vertices, tet work-groups write the faces.  Both read the bit planes of the count
// pass (8 B per 64 edges, 32 B per 64 tets); only crossing edges read their index pair / SDF / positions and only surface tets read
// their tet2edge row.  A vertex id is block prefix + word prefix + a popcount, so the tets do not wait for the edges (no edge -> vertex table).
__device__ __forceinline__ int dm_vertex_of_edge(int eid, const unsigned long long* __restrict__ edge_bits, const int* __restrict__ wlocal,
                                                 const int* __restrict__ blk_e) {
    const unsigned long long word = edge_bits[eid >> 6];
    return blk_e[eid / DM_BLOCK_ITEMS] + wlocal[eid >> 6] + __popcll(word & ((1ull << (eid & 63)) - 1ull));
}

// the surface vertex of crossing edge i (reference dmtet.py:124-131: w = flip([s_a, -s_b]) / (s_a + (-s_b)); v = p_a*w_a + p_b*w_b)
__device__ __forceinline__ void dm_place_vertex(long long i, int vid, const float* __restrict__ pos, const float* __restrict__ sdf,
                                                const int2* __restrict__ edges, float* __restrict__ verts, int* __restrict__ vert_edge) {
    const int2 e = edges[i];
    const float sa = sdf[e.x], sb = sdf[e.y];
    const float nsb = -sb;
    const float den = sa + nsb;
    const float wa = nsb / den, wb = sa / den;
    const float* pa = pos + 3ll * e.x;
    const float* pb = pos + 3ll * e.y;
    float* o = verts + 3ll * vid;
    o[0] = pa[0] * wa + pb[0] * wb;
    o[1] = pa[1] * wa + pb[1] * wb;
    o[2] = pa[2] * wa + pb[2] * wb;
    vert_edge[vid] = (int)i;
}

// the n (1 or 2) triangles of surface tet t (case cs), faces slot .. slot + n - 1
__device__ __forceinline__ void dm_write_faces(long long t, int cs, unsigned n, long long slot, const int* __restrict__ tet2edge,
                                               const unsigned long long* __restrict__ edge_bits, const int* __restrict__ wlocal,
                                               const int* __restrict__ blk_e, long long* __restrict__ faces, long long* __restrict__ uv_idx,
                                               int* __restrict__ tri32, int* __restrict__ topo_cnt, int* __restrict__ topo_adj,
                                               int topo_stride, int F) {
    const int* te = tet2edge + 6ll * t;
    int ev[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) ev[j] = te[j];
    const signed char* row = c_tri_table[cs];
    for (unsigned q = 0; q < n; ++q) {
        long long* fo = faces + 3 * (slot + q);
        long long* uo = uv_idx + 3 * (slot + q);
        int ids[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int slot_e = row[3 * q + j];
            int eid = ev[0];
#pragma unroll
            for (int s = 1; s < 6; ++s) eid = (slot_e == s) ? ev[s] : eid;  // register select, no scratch
            ids[j] = dm_vertex_of_edge(eid, edge_bits, wlocal, blk_e);
            fo[j] = (long long)ids[j];
        }
        if (tri32) {
            // the first half of the mesh topology (topology.hip: a3d_mesh_topology_finalize does the rest in ONE launch): the int32
            // triangle list the render kernels read and the valence counts of the vertex -> face lists, from the three ids this
            // thread holds anyway (fire-and-forget adds; this used to be a launch of its own over the finished list)
            const int fi = (int)(slot + q);
            tri32[3 * fi] = ids[0]; tri32[3 * fi + 1] = ids[1]; tri32[3 * fi + 2] = ids[2];
            if (topo_adj) {
                // ... or the WHOLE of it: the valence of a surface vertex is bounded by the grid (at most two triangles from each
                // tet around its edge), so every vertex owns topo_stride slots and the returned count is the entry's place in its
                // list -- no scan, no second launch.  Three returning atomics in flight per face thread (+2 us on this launch
                // against the 11 us entry point it makes unnecessary); the lists are unordered, their readers sort the keys.
                const int s0 = atomicAdd(topo_cnt + ids[0], 1), s1 = atomicAdd(topo_cnt + ids[1], 1), s2 = atomicAdd(topo_cnt + ids[2], 1);
                if (s0 < topo_stride) topo_adj[(long long)ids[0] * topo_stride + s0] = fi;
                if (s1 < topo_stride) topo_adj[(long long)ids[1] * topo_stride + s1] = F + fi;
                if (s2 < topo_stride) topo_adj[(long long)ids[2] * topo_stride + s2] = 2 * F + fi;
            } else {
                atomicAdd(topo_cnt + ids[0], 1); atomicAdd(topo_cnt + ids[1], 1); atomicAdd(topo_cnt + ids[2], 1);
            }
        }
        // reference dmtet.py:91-96 with face_gidx = 2t + q
        uo[0] = 4ll * t;
        uo[1] = 4ll * t + q + 1;
        uo[2] = 4ll * t + q + 2;
    }
}

__global__ __launch_bounds__(DM_THREADS) void dm_emit_kernel(const float* __restrict__ pos, const float* __restrict__ sdf,
                                                             const int2* __restrict__ edges, const int* __restrict__ tet2edge, int Ne, int Nt,
                                                             int nbe, const int* __restrict__ blk_e, const int* __restrict__ blk_t1,
                                                             const int* __restrict__ blk_t2, const unsigned long long* __restrict__ edge_bits,
                                                             const unsigned long long* __restrict__ tet_bits, const int* __restrict__ wlocal,
                                                             int n1, float* __restrict__ verts, int* __restrict__ vert_edge,
                                                             long long* __restrict__ faces, long long* __restrict__ uv_idx, int nbt,
                                                             unsigned* __restrict__ vbits, const int* __restrict__ vchunk, int Nv,
                                                             long long* __restrict__ surf_idx, float* __restrict__ clear, int n_clear,
                                                             int* __restrict__ tri32, int* __restrict__ topo_cnt, int wg_per_block,
                                                             int* __restrict__ topo_adj, int topo_stride, int F,
                                                             const int* __restrict__ elist, const int* __restrict__ tlist, int n_eblocks,
                                                             int n_tblocks, int nvc, const int* __restrict__ dev_counts, int cap_V,
                                                             int cap_F, int cap_surf, float* __restrict__ surf_pts, int surf_bucket) {
    __shared__ int s_pre[32];
    A3D_STAMP(3, 0);  // (only work-groups that reach a stage stamp it: most leave at one of the early exits)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // SPECULATIVE launch (dev_counts = the counts of a3d_dmtet_count, still on the device: the host has not read them yet and sized
    // buffers and grid by a guess): the sizes come from there; if anything does not fit, NO work-group touches anything and the host,
    // which sees the same numbers a moment later, launches again with exact sizes.  n_eblocks / n_tblocks then are the capacities the
    // grid was sized for.
    int live_e = n_eblocks, live_t = n_tblocks;
    if (dev_counts) {
        const int dV = dev_counts[0], d1 = dev_counts[1], d2 = dev_counts[2], dS = dev_counts[3], dE = dev_counts[4], dT = dev_counts[5];
        if (dV > cap_V || d1 + 2 * d2 > cap_F || (vbits && dS > cap_surf) || dE < 0 || dE > n_eblocks || dT > n_tblocks) return;
        n1 = d1;
        F = d1 + 2 * d2;
        live_e = dE;
        live_t = (d1 + d2) > 0 ? dT : 0;
        cap_surf = dS;  // (from here on: the true length of the surface-vertex list)
    }
    // the backward's dense SDF gradient (scattered into with atomics) cleared here: one memset less on the backward path
    for (int z = blockIdx.x * blockDim.x + tid; z < n_clear; z += gridDim.x * blockDim.x) clear[z] = 0.f;
    A3D_STAMP(3, 1);
    // wg_per_block = 4: one SLAB (256 items) per work-group -- a slab is a chain of three dependent gathers (bit planes -> index row ->
    // vertex ids), and four of them in series per work-group made the launch pure latency at the bench size (16 us for ~1 MB; four
    // times the work-groups overlap them: 12.8 us).  wg_per_block = 1: all four slabs in one work-group, for grids whose block count
    // alone fills the machine many times over (R = 128: 27k blocks; 107k work-groups cost 62 us against 41 us)
    // n_eblocks / n_tblocks work-group sets: every block of the grid (elist / tlist null), or only the blocks the culled count pass
    // listed as non-empty (any order: a block's outputs go to places its prefixes name)
    if ((int)blockIdx.x >= wg_per_block * (n_eblocks + n_tblocks)) {
        const int c = (int)blockIdx.x - wg_per_block * (n_eblocks + n_tblocks);
        if (c < nvc) dm_surface_vertices_chunk(c, vbits, vchunk, Nv, surf_idx, s_pre, pos, surf_pts, surf_bucket, cap_surf);  // (past the chunks: work-groups that only clear)
        return;
    }
    constexpr int WPS = DM_THREADS / A3D_WAVE;  // words per slab
    const int set = blockIdx.x / wg_per_block;
    const bool is_edge = set < n_eblocks;
    if (is_edge ? set >= live_e : set - n_eblocks >= live_t) return;  // (speculative launch: sets past the listed blocks)
    const int blk = is_edge ? (elist ? elist[set] : set) : nbe + (tlist ? tlist[set - n_eblocks] : set - n_eblocks);
    const int slab0 = (blockIdx.x % wg_per_block) * (DM_SLABS / wg_per_block), slab1 = slab0 + DM_SLABS / wg_per_block;
    if (is_edge) {
        const long long base = (long long)blk * DM_BLOCK_ITEMS;
        // this wave's (up to four) words at once: ~99 % of the blocks hold no crossing edge, and finding that out must not take a
        // dependent load per slab
        unsigned long long words[DM_SLABS];
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) words[k] = (k >= slab0 && k < slab1) ? edge_bits[(base >> 6) + k * WPS + wave] : 0ull;
        if (!(words[0] | words[1] | words[2] | words[3])) return;
#pragma unroll
        for (int k = 0; k < DM_SLABS; ++k) {
            if (k < slab0 || k >= slab1) continue;
            const long long wi = (base >> 6) + k * WPS + wave;
            const unsigned long long word = words[k];
            if (!((word >> lane) & 1ull)) continue;
            const long long i = base + k * DM_THREADS + tid;
            dm_place_vertex(i, blk_e[blk] + wlocal[wi] + a3d_wave_prefix(word), pos, sdf, edges, verts, vert_edge);
        }
        return;
    }
    const int bt = blk - nbe;
    const long long base = (long long)bt * DM_BLOCK_ITEMS;
    // per-chunk counts of the 16 chunks of this block (lane j < 16 holds chunk j's): n == 1 <=> odd number of inside corners
    int c1 = 0, c2 = 0;
    unsigned long long pw0 = 0, pw1 = 0, pw2 = 0, pw3 = 0;  // lane j < 16: the four bit planes of chunk j
    if (lane < DM_BLOCK_ITEMS / 64) {
        const unsigned long long* w = tet_bits + ((base >> 6) + lane) * 4;
        pw0 = w[0]; pw1 = w[1]; pw2 = w[2]; pw3 = w[3];
        const unsigned long long odd = pw0 ^ pw1 ^ pw2 ^ pw3;
        c1 = __popcll(odd);
        c2 = __popcll(~odd & (pw0 | pw1 | pw2 | pw3) & ~(pw0 & pw1 & pw2 & pw3));
    }
    if (!__ballot((c1 | c2) != 0)) return;  // no surface tet in the whole block (the usual case): one round trip, not one per slab
    const int blk1 = blk_t1[bt], blk2 = blk_t2[bt];
    for (int k = slab0; k < slab1; ++k) {
        const int chunk = k * WPS + wave;
        // (the chunk's planes come from the lane that loaded them for the counts)
        const unsigned long long w0 = __shfl(pw0, chunk, 64), w1 = __shfl(pw1, chunk, 64), w2 = __shfl(pw2, chunk, 64), w3 = __shfl(pw3, chunk, 64);
        const unsigned long long odd = w0 ^ w1 ^ w2 ^ w3;
        const unsigned long long m1 = odd, m2 = ~odd & (w0 | w1 | w2 | w3) & ~(w0 & w1 & w2 & w3);
        if (!(m1 | m2)) continue;  // wave-uniform: no surface tet among these 64
        int run1 = blk1, run2 = blk2;
        for (int j = 0; j < chunk; ++j) { run1 += __shfl(c1, j, 64); run2 += __shfl(c2, j, 64); }
        const int cs = (int)((w0 >> lane) & 1ull) | ((int)((w1 >> lane) & 1ull) << 1) | ((int)((w2 >> lane) & 1ull) << 2) |
                       ((int)((w3 >> lane) & 1ull) << 3);
        const unsigned n = DM_NTRI(cs);
        if (n == 0u) continue;
        const long long t = base + k * DM_THREADS + tid;
        const long long slot = (n == 1u) ? (long long)(run1 + a3d_wave_prefix(m1)) : (long long)n1 + 2ll * (run2 + a3d_wave_prefix(m2));
        dm_write_faces(t, cs, n, slot, tet2edge, edge_bits, wlocal, blk_e, faces, uv_idx, tri32, topo_cnt, topo_adj, topo_stride, F);
    }
    A3D_STAMP(3, 5);
}

// The emit launch for SPARSE planes (the ordered count pass: the ~1 % surface items of a grid in a random numbering are spread evenly
// over the planes -- at the "256" class 1.5 crossing edges per 1024-edge block -- so every block holds something, a list of non-empty
// blocks saves nothing, and a work-group per block is 27k work-groups that each run a chain of five dependent gathers for one or two
// items: 62 us).  Thread = WORD of a plane: the word loads of a wave are one coalesced read, a work-group whose 256 words are all empty
// is done, the others pool their set bits (see below); the in-block prefixes come from the scan launch (wlocal; tlocal = the same
// for the one- / two-triangle tets, packed 16 + 16 bits).  The grid does not depend on the counts, so the speculative form (sizes
// from dev_counts) needs no block capacities.
__global__ __launch_bounds__(DM_THREADS) void dm_emit_words_kernel(const float* __restrict__ pos, const float* __restrict__ sdf,
                                                                   const int2* __restrict__ edges, const int* __restrict__ tet2edge, int nwe,
                                                                   int nwt, const int* __restrict__ blk_e, const int* __restrict__ blk_t1,
                                                                   const int* __restrict__ blk_t2, const unsigned long long* __restrict__ edge_bits,
                                                                   const unsigned long long* __restrict__ tet_bits, const int* __restrict__ wlocal,
                                                                   const int* __restrict__ tlocal, int n1, float* __restrict__ verts,
                                                                   int* __restrict__ vert_edge, long long* __restrict__ faces,
                                                                   long long* __restrict__ uv_idx, unsigned* __restrict__ vbits,
                                                                   const int* __restrict__ vchunk, int Nv, long long* __restrict__ surf_idx,
                                                                   float* __restrict__ clear, int n_clear, int* __restrict__ tri32,
                                                                   int* __restrict__ topo_cnt, int* __restrict__ topo_adj, int topo_stride, int F,
                                                                   int wg_e, int wg_t, int nvc, const int* __restrict__ dev_counts, int cap_V,
                                                                   int cap_F, int cap_surf, float* __restrict__ surf_pts, int surf_bucket) {
    __shared__ int s_pre[32];
    __shared__ unsigned short s_items[DM_THREADS * 64];  // (word of this work-group) << 6 | bit, 32 KB
    __shared__ int s_wave[DM_THREADS / A3D_WAVE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool faces_live = F > 0;
    if (dev_counts) {
        const int dV = dev_counts[0], d1 = dev_counts[1], d2 = dev_counts[2], dS = dev_counts[3];
        if (dV > cap_V || d1 + 2 * d2 > cap_F || (vbits && dS > cap_surf)) return;
        n1 = d1;
        F = d1 + 2 * d2;
        faces_live = F > 0;
        cap_surf = dS;  // (from here on: the true length of the surface-vertex list)
    }
    for (int z = blockIdx.x * blockDim.x + tid; z < n_clear; z += gridDim.x * blockDim.x) clear[z] = 0.f;
    const int b = blockIdx.x;
    if (b >= wg_e + wg_t) {
        if (b - wg_e - wg_t < nvc) dm_surface_vertices_chunk(b - wg_e - wg_t, vbits, vchunk, Nv, surf_idx, s_pre, pos, surf_pts, surf_bucket, cap_surf);
        return;
    }
    // The set bits of the work-group's 256 words are POOLED: every thread lists its word's bits in LDS (pure ALU + LDS stores), then
    // the items are dealt out evenly over the 256 lanes -- a word with forty crossings (a plane in a spatial row order: the surface
    // sits in a few dense words) costs forty LDS stores instead of forty serial chains of dependent gathers in one lane (125 us on
    // the BCC lattice in its generator's numbering); on sparse planes the pool is a few dozen items and one round.
    const bool is_edge = b < wg_e;
    const int w = (is_edge ? b : b - wg_e) * DM_THREADS + tid;
    unsigned long long mine = 0ull;  // the bits of this thread's word that are items
    if (is_edge) {
        if (w < nwe) mine = edge_bits[w];
    } else if (w < nwt && faces_live) {
        const ulonglong2 pa = reinterpret_cast<const ulonglong2*>(tet_bits)[2ll * w], pb = reinterpret_cast<const ulonglong2*>(tet_bits)[2ll * w + 1];
        mine = (pa.x | pa.y | pb.x | pb.y) & ~(pa.x & pa.y & pb.x & pb.y);  // surface tets: neither case 0 nor case 15
    }
    const int c = __popcll(mine);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = incl - c, total = 0;
#pragma unroll
    for (int k = 0; k < DM_THREADS / A3D_WAVE; ++k) {
        if (k < wave) off += s_wave[k];
        total += s_wave[k];
    }
    if (total == 0) return;  // (uniform)
    while (mine) {
        s_items[off++] = (unsigned short)((tid << 6) | (__ffsll((long long)mine) - 1));
        mine &= mine - 1ull;
    }
    __syncthreads();
    const int w_base = (is_edge ? b : b - wg_e) * DM_THREADS;
    for (int j = tid; j < total; j += DM_THREADS) {
        const int it = s_items[j], wj = w_base + (it >> 6), bit = it & 63;
        if (is_edge) {
            const unsigned long long word = edge_bits[wj];
            dm_place_vertex(64ll * wj + bit, blk_e[wj >> 4] + wlocal[wj] + __popcll(word & ((1ull << bit) - 1ull)), pos, sdf, edges, verts, vert_edge);
        } else {
            const ulonglong2 pa = reinterpret_cast<const ulonglong2*>(tet_bits)[2ll * wj], pb = reinterpret_cast<const ulonglong2*>(tet_bits)[2ll * wj + 1];
            const unsigned long long w0 = pa.x, w1 = pa.y, w2 = pb.x, w3 = pb.y;
            const unsigned long long odd = w0 ^ w1 ^ w2 ^ w3;
            const unsigned long long m1 = odd, m2 = ~odd & (w0 | w1 | w2 | w3) & ~(w0 & w1 & w2 & w3);
            const unsigned long long below = (1ull << bit) - 1ull;
            const int cs = (int)((w0 >> bit) & 1ull) | ((int)((w1 >> bit) & 1ull) << 1) | ((int)((w2 >> bit) & 1ull) << 2) | ((int)((w3 >> bit) & 1ull) << 3);
            const unsigned n = DM_NTRI(cs);
            const int tl = tlocal[wj];
            const long long slot = (n == 1u) ? (long long)(blk_t1[wj >> 4] + (tl & 0xFFFF) + __popcll(m1 & below))
                                             : (long long)n1 + 2ll * (blk_t2[wj >> 4] + (tl >> 16) + __popcll(m2 & below));
            dm_write_faces(64ll * wj + bit, cs, n, slot, tet2edge, edge_bits, wlocal, blk_e, faces, uv_idx, tri32, topo_cnt, topo_adj, topo_stride, F);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
__global__ __launch_bounds__(256) void dm_bwd_kernel(const float* __restrict__ g_verts, const float* __restrict__ pos,
                                                     const float* __restrict__ sdf, const int2* __restrict__ edges,
                                                     const int* __restrict__ vert_edge, int V, float* __restrict__ g_pos,
                                                     float* __restrict__ g_sdf) {
    A3D_STAMP(4, 0);
    int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int2 e = edges[vert_edge[v]];
    float sa = sdf[e.x], sb = sdf[e.y];
    float den = sa - sb, inv = 1.f / den, inv2 = inv * inv;
    float gx = g_verts[3ll * v], gy = g_verts[3ll * v + 1], gz = g_verts[3ll * v + 2];
    const float* pa = pos + 3ll * e.x;
    const float* pb = pos + 3ll * e.y;
    float g_wa = gx * pa[0] + gy * pa[1] + gz * pa[2];
    float g_wb = gx * pb[0] + gy * pb[1] + gz * pb[2];
    // w_a = -s_b/den, w_b = s_a/den
    atomicAdd(g_sdf + e.x, (g_wa - g_wb) * sb * inv2);
    atomicAdd(g_sdf + e.y, (g_wb - g_wa) * sa * inv2);
    if (g_pos) {
        float wa = -sb * inv, wb = sa * inv;
        atomicAdd(g_pos + 3ll * e.x + 0, gx * wa); atomicAdd(g_pos + 3ll * e.x + 1, gy * wa); atomicAdd(g_pos + 3ll * e.x + 2, gz * wa);
        atomicAdd(g_pos + 3ll * e.y + 0, gx * wb); atomicAdd(g_pos + 3ll * e.y + 1, gy * wb); atomicAdd(g_pos + 3ll * e.y + 2, gz * wb);
    }
    A3D_STAMP(4, 5);
}

// ------------------------------------------------------------------------------------------------ C ABI
struct DmScratch {
    int *be, *b1, *b2, *wlocal, *tlocal;
    unsigned long long *edge_bits, *tet_bits;
    int nbe, nbt;
    size_t sign_off;
    int *list_len, *elist, *tlist;  // non-empty blocks of the culled count pass: two append counters (64 bytes apart), the two lists
};

#define DM_SIGN_PLANE_MIN_NV (1 << 20)  // below this the pre-pass launch costs what the cheaper gathers save (R = 64: 2.7e5 vertices)

static size_t dm_split_scratch(void* scratch, int Ne, int Nt, DmScratch* d) {
    d->nbe = a3d_div_up(Ne, DM_BLOCK_ITEMS);
    d->nbt = a3d_div_up(Nt, DM_BLOCK_ITEMS);
    const size_t nwe = (size_t)d->nbe * (DM_BLOCK_ITEMS / 64), nwt = (size_t)d->nbt * (DM_BLOCK_ITEMS / 64);
    unsigned long long* q = (unsigned long long*)scratch;  // the 8-byte arrays first
    d->edge_bits = q;
    d->tet_bits = q + nwe;
    int* p = (int*)(q + nwe + 4 * nwt);
    d->wlocal = p;
    d->be = p + nwe;
    d->b1 = d->be + d->nbe;
    d->b2 = d->b1 + d->nbt;
    d->tlocal = d->b2 + d->nbt;  // (ordered pass only) one- / two-triangle tets of the same block before a word, 16 + 16 bits
    d->sign_off = (sizeof(unsigned long long) * (nwe + 4 * nwt) + sizeof(int) * (nwe + d->nbe + 2 * (size_t)d->nbt + nwt + 4) + 63) & ~(size_t)63;
    const size_t list_off = (d->sign_off + ((size_t)Ne / 4 + 64) + 63) & ~(size_t)63;  // (after a sign plane for up to 2 Ne grid vertices)
    d->list_len = (int*)((char*)scratch + list_off);
    d->elist = d->list_len + 32;
    d->tlist = d->elist + d->nbe;
    return list_off + sizeof(int) * (32 + (size_t)d->nbe + d->nbt);
}

extern "C" size_t a3d_dmtet_scratch_bytes(int Ne, int Nt) {
    DmScratch d;
    return dm_split_scratch(nullptr, Ne < 1 ? 1 : Ne, Nt < 1 ? 1 : Nt, &d);
}

// surface-adjacent vertex list (optional): 1 bit per grid vertex in 1024-vertex chunks + one int per chunk
extern "C" size_t a3d_dmtet_vertex_scratch_bytes(int Nv) {
    const size_t nvc = (size_t)a3d_div_up(Nv < 1 ? 1 : Nv, 1024);
    return nvc * 128 + nvc * sizeof(int);
}

extern "C" int a3d_dmtet_word_group_slots(void) { return DM_CULL_SLOTS; }
extern "C" int a3d_dmtet_word_group_bits(void) { return 4; }  // 16 vertices per group: a two-byte field of the sign plane
extern "C" int a3d_dmtet_block_items(void) { return DM_BLOCK_ITEMS; }

extern "C" int a3d_dmtet_count(const float* sdf, const int32_t* edges, const int32_t* tets, int Ne, int Nt, void* scratch,
                               int32_t* counts, void* vertex_scratch_or_null, int vertex_scratch_is_clean, int Nv,
                               const uint32_t* edge_groups_or_null, const uint32_t* tet_groups_or_null, int group_slots,
                               int32_t* words_to_clear_or_null, int n_words_to_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(n_words_to_clear >= 0 && (n_words_to_clear == 0 || words_to_clear_or_null));
    A3D_CHECK_ARG(!edge_groups_or_null || group_slots == 8 || group_slots == 16);
    A3D_CHECK_ARG(sdf && edges && tets && scratch && counts && ((uintptr_t)scratch & 7) == 0);
    A3D_CHECK_ARG((edge_groups_or_null == nullptr) == (tet_groups_or_null == nullptr));
    A3D_CHECK_ARG(!edge_groups_or_null || Nv > 0);
    A3D_CHECK_ARG(Ne > 0 && Nt > 0);
    A3D_CHECK_ARG(!vertex_scratch_or_null || (Nv > 0 && ((uintptr_t)vertex_scratch_or_null & 15) == 0));
    A3D_CHECK_ARG(Nv >= 0);  // (Nv = 0: unknown -- the sign-plane path for large grids is then not taken)
    DmScratch d;
    dm_split_scratch(scratch, Ne, Nt, &d);
    hipStream_t s = (hipStream_t)stream;
    unsigned* vbits = (unsigned*)vertex_scratch_or_null;
    const int nvc = vbits ? a3d_div_up(Nv, 1024) : 0;
    int* vchunk = vbits ? (int*)(vbits + 32ll * nvc) : nullptr;
    if (vbits && !vertex_scratch_is_clean) A3D_HIP(hipMemsetAsync(vbits, 0, 128 * (size_t)nvc, s));
    const int* list_len = nullptr;
    if (edge_groups_or_null && (long long)Nv <= 2ll * Ne) {  // (the caller decides: ops.DMTET_CULL_MIN_VERTS)
        unsigned long long* sign = (unsigned long long*)((char*)scratch + d.sign_off);
        hipLaunchKernelGGL(dm_sign_kernel, dim3(a3d_div_up(Nv, 256 * DM_SIGN_WORDS)), dim3(256), 0, s, sdf, Nv, sign, d.list_len);
        A3D_LAUNCH_CHECK();
        list_len = d.list_len;
        // (G = 1, 2, 4, 8 blocks per work-group measured within 1 us of each other once the blocks of a work-group are strided)
        const int nge = a3d_div_up(d.nbe, DM_CULL_BLOCKS), ngt = a3d_div_up(d.nbt, DM_CULL_BLOCKS);
        // (experiment builds only, A3D_EXP=44) the scan rides in the count launch -- dm_scan_tail.  MEASURED AND DROPPED in round 6: the
        // call went from 20.5 to 28.7 us at the bench size (in-kernel stamps: every work-group spends a median of 4.4 us between its last
        // store and its ticket -- acknowledgements of its stores + one returning device atomic --, and the last one another 6.7 us on
        // the scans: an acquire, two dependent round trips to memory, the stores), against 6.2 us of kernel + the gap for the launch it
        // replaces; with release fences instead of atomics 88 us (841 L2 write-backs)
        const int fold = d.nbe <= DM_TAIL_MAX && d.nbt <= DM_TAIL_MAX && nvc <= DM_VWIN && a3d_exp() == 44;
#define DM_CULL_LAUNCH(SLOTS)                                                                                                                  \
    hipLaunchKernelGGL((dm_count_cull_kernel<DM_CULL_BLOCKS, SLOTS>), dim3(nge + ngt), dim3(DM_THREADS), 0, s, (const unsigned*)sign,           \
                       (const int2*)edges, (const int4*)tets, Ne, Nt, d.nbe, d.nbt, nge, edge_groups_or_null, tet_groups_or_null, d.be, d.b1,    \
                       d.b2, d.edge_bits, d.tet_bits, d.wlocal, vbits, d.list_len, d.elist, d.tlist, fold, counts, vchunk, nvc,                  \
                       words_to_clear_or_null, words_to_clear_or_null ? n_words_to_clear : 0)
        if (group_slots == 16) DM_CULL_LAUNCH(16);
        else DM_CULL_LAUNCH(8);
#undef DM_CULL_LAUNCH
        if (fold) {
            A3D_LAUNCH_CHECK();
            return A3D_OK;
        }
    } else if (Nv >= DM_SIGN_PLANE_MIN_NV && (long long)Nv <= 2ll * Ne) {
        unsigned long long* sign = (unsigned long long*)((char*)scratch + d.sign_off);
        hipLaunchKernelGGL(dm_sign_kernel, dim3(a3d_div_up(Nv, 256 * DM_SIGN_WORDS)), dim3(256), 0, s, sdf, Nv, sign, (int*)nullptr);
        A3D_LAUNCH_CHECK();
        hipLaunchKernelGGL(dm_count_kernel<true>, dim3(d.nbe + d.nbt), dim3(DM_THREADS), 0, s, (const void*)sign, (const int2*)edges,
                           (const int4*)tets, Ne, Nt, d.nbe, d.be, d.b1, d.b2, d.edge_bits, d.tet_bits, d.wlocal, vbits);
    } else {
        hipLaunchKernelGGL(dm_count_kernel<false>, dim3(d.nbe + d.nbt), dim3(DM_THREADS), 0, s, (const void*)sdf, (const int2*)edges,
                           (const int4*)tets, Ne, Nt, d.nbe, d.be, d.b1, d.b2, d.edge_bits, d.tet_bits, d.wlocal, vbits);
    }
    A3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(dm_scan_kernel, dim3(vbits ? 4 : 3), dim3(1024), 0, s, d.be, d.b1, d.b2, d.nbe, d.nbt, counts, vbits, vchunk, nvc,
                       words_to_clear_or_null, words_to_clear_or_null ? n_words_to_clear : 0, list_len, (const unsigned long long*)nullptr,
                       (int*)nullptr, vbits ? 4 : 3, (const unsigned long long*)nullptr, (int*)nullptr);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_dmtet_count_ordered(const float* sdf, int Nv, int Ne, int Nt, const a3d_dmtet_order* order, void* scratch,
                                       int32_t* counts, void* vertex_scratch_or_null, int vertex_scratch_is_clean,
                                       int32_t* words_to_clear_or_null, int n_words_to_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(order && order->size >= sizeof(a3d_dmtet_order));  // (a newer caller may append fields; an older, shorter struct is refused)
    A3D_CHECK_ARG(order->group_slots == 8 || order->group_slots == 16);
    A3D_CHECK_ARG(order->vertex_of_rank && order->edges_ranked && order->edge_of_row && order->tets_ranked && order->tet_of_row &&
                  order->edge_groups && order->tet_groups);
    A3D_CHECK_ARG(n_words_to_clear >= 0 && (n_words_to_clear == 0 || words_to_clear_or_null));
    A3D_CHECK_ARG(sdf && scratch && counts && ((uintptr_t)scratch & 15) == 0);
    A3D_CHECK_ARG(Ne > 0 && Nt > 0 && Nv > 0 && (long long)Nv <= 2ll * Ne);
    A3D_CHECK_ARG(!vertex_scratch_or_null || ((uintptr_t)vertex_scratch_or_null & 15) == 0);
    DmScratch d;
    dm_split_scratch(scratch, Ne, Nt, &d);
    hipStream_t s = (hipStream_t)stream;
    unsigned* vbits = (unsigned*)vertex_scratch_or_null;
    const int nvc = vbits ? a3d_div_up(Nv, 1024) : 0;
    int* vchunk = vbits ? (int*)(vbits + 32ll * nvc) : nullptr;
    if (vbits && !vertex_scratch_is_clean) A3D_HIP(hipMemsetAsync(vbits, 0, 128 * (size_t)nvc, s));
    unsigned long long* sign = (unsigned long long*)((char*)scratch + d.sign_off);
    // planes, wlocal and block sums are one stretch of the scratch that ends where the sign plane begins (a multiple of 64 bytes)
    hipLaunchKernelGGL(dm_sign_order_kernel, dim3(a3d_div_up(Nv, 256 * DM_SIGN_WORDS)), dim3(256), 0, s, sdf, order->vertex_of_rank, Nv, sign,
                       (uint4*)scratch, (long long)(d.sign_off / 16));
    A3D_LAUNCH_CHECK();
    const int nwe = d.nbe * (DM_BLOCK_ITEMS / 64), nwt = d.nbt * (DM_BLOCK_ITEMS / 64);  // rows of the group tables (padded to whole blocks)
    const int waves_e = a3d_div_up(nwe, DM_ORDER_WORDS), waves_t = a3d_div_up(nwt, DM_ORDER_WORDS);
    const dim3 grid(a3d_div_up(waves_e + waves_t, 4));
#define DM_ORDER_LAUNCH(SLOTS)                                                                                                              \
    hipLaunchKernelGGL(dm_count_order_kernel<SLOTS>, grid, dim3(256), 0, s, (const unsigned*)sign, (const int2*)order->edges_ranked,        \
                       order->edge_of_row, (const int4*)order->tets_ranked, order->tet_of_row, order->vertex_of_rank, Ne, Nt, nwe, nwt,     \
                       waves_e, waves_t, order->edge_groups, order->tet_groups, d.be, d.b1, d.b2, d.edge_bits, d.tet_bits, vbits)
    if (order->group_slots == 16) DM_ORDER_LAUNCH(16);
    else DM_ORDER_LAUNCH(8);
#undef DM_ORDER_LAUNCH
    A3D_LAUNCH_CHECK();
    const int n_scans = vbits ? 4 : 3;
    hipLaunchKernelGGL(dm_scan_kernel, dim3(n_scans + a3d_div_up(nwe, 1024) + a3d_div_up(nwt, 1024)), dim3(1024), 0, s, d.be, d.b1, d.b2, d.nbe,
                       d.nbt, counts, vbits, vchunk, nvc, words_to_clear_or_null, words_to_clear_or_null ? n_words_to_clear : 0, (const int*)nullptr,
                       (const unsigned long long*)d.edge_bits, d.wlocal, n_scans, (const unsigned long long*)d.tet_bits, d.tlocal);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_dmtet_emit(const float* pos, const float* sdf, const int32_t* edges, const int32_t* tet2edge, int Ne, int Nt,
                              const void* scratch, int V, int n1, int n2, float* verts, int32_t* vert_edge, int64_t* faces, int64_t* uv_idx,
                              const a3d_dmtet_emit_opts* opts_or_null, a3d_stream_t stream) {
    a3d_dmtet_emit_opts o = {};
    if (opts_or_null) {
        A3D_CHECK_ARG(opts_or_null->size >= sizeof(a3d_dmtet_emit_opts));
        o = *opts_or_null;
    }
    void* vertex_scratch_or_null = o.vertex_scratch;
    const int Nv = o.Nv, n_surf = o.n_surf, topo_stride = o.topo_stride;
    int64_t* surf_idx_or_null = o.surf_idx;
    float* g_sdf_to_clear_or_null = o.g_sdf_to_clear;
    int32_t *tri32_or_null = o.tri32, *topo_count_or_null = o.topo_count, *topo_adj_or_null = o.topo_adj;
    const int n_edge_blocks_listed = o.use_block_lists ? o.n_edge_blocks_listed : -1, n_tet_blocks_listed = o.use_block_lists ? o.n_tet_blocks_listed : -1;
    const int32_t* device_counts_or_null = o.device_counts;
    A3D_CHECK_ARG(pos && sdf && edges && tet2edge && scratch);
    const bool spec = device_counts_or_null != nullptr;  // V, n1 (= F), n_surf, n_*_blocks_listed are CAPACITIES; n2 is ignored
    A3D_CHECK_ARG(!spec || (n_edge_blocks_listed >= 0 && n_tet_blocks_listed >= 0 && V > 0));
    if (spec) n2 = 0;
    A3D_CHECK_ARG(Ne > 0 && Nt > 0 && V >= 0 && n1 >= 0 && n2 >= 0);
    A3D_CHECK_ARG(V == 0 || (verts && vert_edge));
    A3D_CHECK_ARG((n1 + n2) == 0 || (faces && uv_idx));
    A3D_CHECK_ARG(!vertex_scratch_or_null || (Nv > 0 && n_surf >= 0 && (n_surf == 0 || surf_idx_or_null)));
    A3D_CHECK_ARG(!o.surf_pts || (surf_idx_or_null && o.surf_bucket > 0 && (!spec || n_surf % o.surf_bucket == 0)));
    A3D_CHECK_ARG(!g_sdf_to_clear_or_null || Nv > 0);
    A3D_CHECK_ARG((tri32_or_null == nullptr) == (topo_count_or_null == nullptr));
    A3D_CHECK_ARG(!topo_adj_or_null || (tri32_or_null && topo_stride > 0 && (long long)V * topo_stride < 0x7fffffffll));
    if (V == 0) {  // no crossing edge, hence no surface tet and no flagged vertex
        if (g_sdf_to_clear_or_null) A3D_HIP(hipMemsetAsync(g_sdf_to_clear_or_null, 0, sizeof(float) * (size_t)Nv, (hipStream_t)stream));
        return A3D_OK;
    }
    DmScratch d;
    dm_split_scratch((void*)scratch, Ne, Nt, &d);
    unsigned* vbits = (unsigned*)vertex_scratch_or_null;
    const int nvc = vbits ? a3d_div_up(Nv, 1024) : 0;
    const int cap_F = n1 + 2 * n2;
    const int nbt = (n1 + n2) > 0 ? d.nbt : 0;
    // counts[4], counts[5] of a3d_dmtet_count: the non-empty blocks its culled pass listed in the scratch (-1: none listed)
    const bool listed = n_edge_blocks_listed >= 0 && n_tet_blocks_listed >= 0;
    A3D_CHECK_ARG(!listed || (n_edge_blocks_listed <= d.nbe && n_tet_blocks_listed <= d.nbt));
    const int ne = listed ? n_edge_blocks_listed : d.nbe, nt = listed ? (nbt ? n_tet_blocks_listed : 0) : nbt;
    const int wgpb = (ne + nt) <= 8192 ? DM_SLABS : 1;
    // (a short block list must not leave the clear of the dense SDF gradient to a handful of work-groups: <= 16 floats per thread)
    const int wg_work = wgpb * (ne + nt) + nvc, wg_clear = g_sdf_to_clear_or_null ? a3d_div_up(Nv, DM_THREADS * 16) : 0;
    hipLaunchKernelGGL(dm_emit_kernel, dim3(wg_work > wg_clear ? wg_work : wg_clear), dim3(DM_THREADS), 0, (hipStream_t)stream, pos, sdf, (const int2*)edges, tet2edge,
                       Ne, Nt, d.nbe, d.be, d.b1, d.b2, d.edge_bits, d.tet_bits, d.wlocal, n1, verts, vert_edge, (long long*)faces,
                       (long long*)uv_idx, nbt, vbits, vbits ? (const int*)(vbits + 32ll * nvc) : nullptr, Nv, (long long*)surf_idx_or_null,
                       g_sdf_to_clear_or_null, g_sdf_to_clear_or_null ? Nv : 0, tri32_or_null, topo_count_or_null, wgpb, topo_adj_or_null,
                       topo_stride, n1 + 2 * n2, listed ? (const int*)d.elist : nullptr, listed ? (const int*)d.tlist : nullptr, ne, nt, nvc,
                       device_counts_or_null, V, cap_F, n_surf, o.surf_pts, o.surf_bucket);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_dmtet_emit_sparse(const float* pos, const float* sdf, const int32_t* edges, const int32_t* tet2edge, int Ne, int Nt,
                                     const void* scratch, int V, int n1, int n2, float* verts, int32_t* vert_edge, int64_t* faces,
                                     int64_t* uv_idx, const a3d_dmtet_emit_opts* opts_or_null, a3d_stream_t stream) {
    a3d_dmtet_emit_opts o = {};
    if (opts_or_null) {
        A3D_CHECK_ARG(opts_or_null->size >= sizeof(a3d_dmtet_emit_opts));
        o = *opts_or_null;
    }
    A3D_CHECK_ARG(pos && sdf && edges && tet2edge && scratch);
    const bool spec = o.device_counts != nullptr;  // V, n1 (read as F) and n_surf then are CAPACITIES; n2 is ignored
    if (spec) n2 = 0;
    A3D_CHECK_ARG(!spec || V > 0);
    A3D_CHECK_ARG(Ne > 0 && Nt > 0 && V >= 0 && n1 >= 0 && n2 >= 0);
    A3D_CHECK_ARG(V == 0 || (verts && vert_edge));
    A3D_CHECK_ARG((n1 + n2) == 0 || (faces && uv_idx));
    A3D_CHECK_ARG(!o.vertex_scratch || (o.Nv > 0 && o.n_surf >= 0 && (o.n_surf == 0 || o.surf_idx)));
    A3D_CHECK_ARG(!o.surf_pts || (o.surf_idx && o.surf_bucket > 0 && (!spec || o.n_surf % o.surf_bucket == 0)));
    A3D_CHECK_ARG(!o.g_sdf_to_clear || o.Nv > 0);
    A3D_CHECK_ARG((o.tri32 == nullptr) == (o.topo_count == nullptr));
    A3D_CHECK_ARG(!o.topo_adj || (o.tri32 && o.topo_stride > 0 && (long long)V * o.topo_stride < 0x7fffffffll));
    if (V == 0) {  // no crossing edge, hence no surface tet and no flagged vertex
        if (o.g_sdf_to_clear) A3D_HIP(hipMemsetAsync(o.g_sdf_to_clear, 0, sizeof(float) * (size_t)o.Nv, (hipStream_t)stream));
        return A3D_OK;
    }
    DmScratch d;
    dm_split_scratch((void*)scratch, Ne, Nt, &d);
    unsigned* vbits = (unsigned*)o.vertex_scratch;
    const int nvc = vbits ? a3d_div_up(o.Nv, 1024) : 0;
    const int nwe = d.nbe * (DM_BLOCK_ITEMS / 64), nwt = d.nbt * (DM_BLOCK_ITEMS / 64);
    const int wg_e = a3d_div_up(nwe, DM_THREADS), wg_t = a3d_div_up(nwt, DM_THREADS);
    const int wg_work = wg_e + wg_t + nvc, wg_clear = o.g_sdf_to_clear ? a3d_div_up(o.Nv, DM_THREADS * 16) : 0;
    hipLaunchKernelGGL(dm_emit_words_kernel, dim3(wg_work > wg_clear ? wg_work : wg_clear), dim3(DM_THREADS), 0, (hipStream_t)stream, pos, sdf,
                       (const int2*)edges, tet2edge, nwe, nwt, d.be, d.b1, d.b2, d.edge_bits, d.tet_bits, d.wlocal, d.tlocal, n1, verts, vert_edge,
                       (long long*)faces, (long long*)uv_idx, vbits, vbits ? (const int*)(vbits + 32ll * nvc) : nullptr, o.Nv,
                       (long long*)o.surf_idx, o.g_sdf_to_clear, o.g_sdf_to_clear ? o.Nv : 0, o.tri32, o.topo_count, o.topo_adj, o.topo_stride,
                       n1 + 2 * n2, wg_e, wg_t, nvc, o.device_counts, V, n1 + 2 * n2, o.n_surf, o.surf_pts, o.surf_bucket);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

extern "C" int a3d_dmtet_bwd(const float* g_verts, const float* pos, const float* sdf, const int32_t* edges, const int32_t* vert_edge,
                             int V, int Nv, float* g_pos_or_null, float* g_sdf, int g_sdf_is_clear, a3d_stream_t stream) {
    A3D_CHECK_ARG(pos && sdf && edges && g_sdf && Nv > 0 && V >= 0);
    hipStream_t s = (hipStream_t)stream;
    if (!g_sdf_is_clear) A3D_HIP(hipMemsetAsync(g_sdf, 0, sizeof(float) * (size_t)Nv, s));
    if (g_pos_or_null) A3D_HIP(hipMemsetAsync(g_pos_or_null, 0, sizeof(float) * 3 * (size_t)Nv, s));
    if (V > 0) {
        A3D_CHECK_ARG(g_verts && vert_edge);
        hipLaunchKernelGGL(dm_bwd_kernel, dim3(a3d_div_up(V, 256)), dim3(256), 0, s, g_verts, pos, sdf, (const int2*)edges, vert_edge, V,
                           g_pos_or_null, g_sdf);
        A3D_LAUNCH_CHECK();
    }
    return A3D_OK;
}

// out[i, :] = src[idx[i], :] for i < n, zeros for n <= i < rows: the surface-adjacent grid vertices' rows gathered into the bucket-padded
// list the SDF field is re-evaluated on (DMTetGeometry._get_mesh_surface_backward: pos[idx] + pad forward, g_sdf[idx] + pad backward --
// an index kernel, a pad copy and the slice's zero-fill + copy each way as torch ops)
__global__ __launch_bounds__(256) void dm_gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx, long long n,
                                                             long long rows, int C, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= rows * C) return;
    const long long i = t / C;
    const int c = (int)(t - i * C);
    out[t] = i < n ? src[idx[i] * C + c] : 0.f;
}

extern "C" int a3d_dmtet_gather_rows(const float* src, const int64_t* idx, int64_t n, int64_t rows, int C, float* out, a3d_stream_t stream) {
    A3D_CHECK_ARG(n >= 0 && rows >= n && C >= 1 && C <= 16);
    if (rows == 0) return A3D_OK;
    A3D_CHECK_ARG(out && (n == 0 || (src && idx)));
    hipLaunchKernelGGL(dm_gather_rows_kernel, dim3(a3d_div_up(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, src, (const long long*)idx,
                       (long long)n, (long long)rows, C, out);
    A3D_LAUNCH_CHECK();
    return A3D_OK;
}

A3D_PROFILE_TU(dmtet)
