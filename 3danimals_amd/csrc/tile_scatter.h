// Per-tile reduction of a pixel -> vertex scatter in LDS (round 6: the modular backward kernels behind the nvdiffrast stand-in,
// a3d_rast_bwd / a3d_interp_bwd).  A work-group owns a 16 x 16 pixel tile; its pixels' (vertex, row) contributions meet in an LDS
// table keyed by the vertex -- neighbouring pixels share their triangles' vertices: ~50 distinct vertices for the 768 contributions of a
// covered tile at the bench mesh -- and leave as ONE row of adjacent global atomics per vertex and tile instead of one per pixel.
// The table only NAMES the vertices (slot = find-or-claim by compare-and-swap); the contributions are staged as entries with plain stores,
// linked into their slot's list with one integer exchange (the scheme of the fused path's G-buffer backward, gbuffer.hip) and summed by
// the lanes that walk a list.  Measured on a3d_rast_bwd (B = 16, 256 x 256; kernel us): float atomics into a [slot][column] table 29.7,
// of which the adds 14 -- an LDS float atomic costs ~10 cycles PER LANE whether or not lanes meet at an address; lanes that own a slot
// scanning the whole entry list 110 (a few hundred dependent LDS reads per work-group).  A contribution that finds no slot within
// TS_PROBES steps goes to global memory directly: correctness never depends on the table's size.
#pragma once
#include "a3d_common.h"

#define TS_SLOTS 512  // (power of two; 768 contributions at most per tile)
#define TS_PROBES 8
#define TS_ENTRIES 768  // (three per pixel at most)
#define TS_TILE 16

// Pixel of a thread: a wave owns an 8 x 8 block of the tile (lane bits 0..2 = x, 3..5 = y), so that lane ^ 1, 2, 4 are horizontal and
// lane ^ 8, 16, 32 vertical neighbours at doubling distances.
__device__ __forceinline__ void ts_pixel(int tile_x0, int tile_y0, int& px, int& py) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    px = tile_x0 + 8 * (wave & 1) + (lane & 7);
    py = tile_y0 + 8 * (wave >> 1) + (lane >> 3);
}

// Quadtree merge inside the wave BEFORE the table: pixels on the same triangle (key; < 0 = nothing to contribute) carry contributions
// to the same three vertices, and a triangle of the bench mesh covers ~35 pixels.  Round m: the lane whose bit m is clear absorbs its
// partner lane ^ m when both hold the same key; the partner retires.  Same-address LDS float atomics serialise (measured: the 9 adds
// per pixel of a3d_rast_bwd were 24 of its 37 us with every pixel going to the table itself), so every merged pair is three contended
// adds less.  ROUNDS: 6 = down to one entry per aligned run of the block, fewer for wide rows (each round moves N values).
template <int M>
__device__ __forceinline__ int ts_xor(int x) {  // lane ^ M: register moves (DPP) where the pattern exists, ds_bpermute otherwise
    if (M == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
    if (M == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
    if (M == 8) return __builtin_amdgcn_mov_dpp(x, 0x128 /* row_ror:8 */, 0xF, 0xF, true);
    return __shfl_xor(x, M, 64);
}
template <int N, int M>
__device__ __forceinline__ void ts_merge_round(int& key, float (&v)[N]) {
    const int lane = threadIdx.x & 63;
    const int other = ts_xor<M>(key);
    const bool same = key >= 0 && other == key;
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float o = __int_as_float(ts_xor<M>(__float_as_int(v[n])));
        if (same && !(lane & M)) v[n] += o;
    }
    if (same && (lane & M)) key = -1;
}
template <int N, int ROUNDS>
__device__ __forceinline__ void ts_merge(int& key, float (&v)[N]) {
    ts_merge_round<N, 1>(key, v);
    if (ROUNDS > 1) ts_merge_round<N, 8>(key, v);
    if (ROUNDS > 2) ts_merge_round<N, 2>(key, v);
    if (ROUNDS > 3) ts_merge_round<N, 16>(key, v);
    if (ROUNDS > 4) ts_merge_round<N, 4>(key, v);
    if (ROUNDS > 5) ts_merge_round<N, 32>(key, v);
}

struct TileScatter {
    int* key;     // [TS_SLOTS] vertex row (global: image base included), -1 = free
    int* head;    // [TS_SLOTS] last staged entry of the slot's list, -1 = none
    int* used;    // [TS_SLOTS] claimed slots, compacted by flush()
    int* n;       // [4] staged entries, claimed slots (+ padding)
    int* e_next;  // [TS_ENTRIES] the entry staged before this one in the same slot
    float* e_val; // [TS_ENTRIES][C]
    int C;

    __device__ __forceinline__ void init(void* lds, int C_) {
        key = (int*)lds;
        head = key + TS_SLOTS;
        used = head + TS_SLOTS;
        n = used + TS_SLOTS;
        e_next = n + 4;  // (e_val stays 16-byte aligned)
        e_val = (float*)(e_next + TS_ENTRIES);
        C = C_;
        for (int i = threadIdx.x; i < TS_SLOTS; i += blockDim.x) { key[i] = -1; head[i] = -1; }
        if (threadIdx.x < 2) n[threadIdx.x] = 0;
    }
    // the slot of a vertex row, or -1 when the probe sequence is taken by others
    __device__ __forceinline__ int slot(int row) const {
        unsigned h = ((unsigned)row * 0x9E3779B1u) >> (32 - 9);
#pragma unroll 1
        for (int p = 0; p < TS_PROBES; ++p, h = (h + 1) & (TS_SLOTS - 1)) {
            const int was = atomicCAS(&key[h], -1, row);
            if (was == -1 || was == row) return (int)h;
        }
        return -1;
    }
    // every lane of the wave calls this (after the table's barrier): ``rows`` entries for each active lane, handed out per wave (one
    // counter update per wave); returns the lane's first entry
    __device__ __forceinline__ int entries(bool active, int rows) const {
        const unsigned long long m = __ballot(active);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && m) base = atomicAdd(&n[0], rows * __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        return base + rows * __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
    }
    // entry e (its C values already stored in e_val) joins the list of its slot: one integer exchange
    __device__ __forceinline__ void link(int e, int s) const { e_next[e] = atomicExch(&head[s], e); }
    // after a barrier: G lanes per claimed slot (lane = column) walk the slot's list and add the row's sum to g[row * stride + c] as
    // adjacent atomics; two lists in flight per group so that the hops (one LDS round trip each) overlap.  ``skip`` = a column never
    // written.  Contains a barrier.
    template <int G>
    __device__ __forceinline__ void flush(float* __restrict__ g, int stride, int skip) const {
        // the claimed slots, compacted (ballot + one counter update per wave): ~50 of the 512 at the bench mesh -- walking all of them
        // was 6 of a work-group's 11 us
        for (int s = threadIdx.x; s < TS_SLOTS; s += blockDim.x) {
            const bool u = head[s] >= 0;
            const unsigned long long um = __ballot(u);
            int ub = 0;
            if ((threadIdx.x & 63) == 0 && um) ub = atomicAdd(&n[1], __popcll(um));
            ub = __builtin_amdgcn_readfirstlane(ub);
            if (u) used[ub + __popcll(um & ((1ull << (threadIdx.x & 63)) - 1ull))] = s;
        }
        __syncthreads();
        const int n_used = n[1], c = threadIdx.x & (G - 1), groups = blockDim.x / G;
        const bool on = c < C && c != skip;
        for (int j = threadIdx.x / G; j < n_used; j += 2 * groups) {
            const int sa = used[j], sb = j + groups < n_used ? used[j + groups] : -1;
            int ea = head[sa], eb = sb >= 0 ? head[sb] : -1;
            float sum_a = 0.f, sum_b = 0.f;
            while (ea >= 0 || eb >= 0) {
                if (ea >= 0) { const float v = c < C ? e_val[ea * C + c] : 0.f; ea = e_next[ea]; sum_a += v; }
                if (eb >= 0) { const float v = c < C ? e_val[eb * C + c] : 0.f; eb = e_next[eb]; sum_b += v; }
            }
            if (on && sum_a != 0.f) atomicAdd(g + (long long)key[sa] * stride + c, sum_a);
            if (on && sb >= 0 && sum_b != 0.f) atomicAdd(g + (long long)key[sb] * stride + c, sum_b);
        }
    }
    static size_t lds_bytes(int C) { return sizeof(int) * (3 * TS_SLOTS + 4 + TS_ENTRIES) + sizeof(float) * TS_ENTRIES * (size_t)C; }
};
