// Mesh-topology helpers shared by normals.hip (vertex -> face CSR), antialias.hip (edge -> opposite vertex table) and
// topology.hip (both in one pass over the triangle list).  Device code only; every kernel here lives in an anonymous namespace
// so that each translation unit gets its own copy.
#pragma once
#include "a3d_common.h"

#define AA_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define AA_NONE 0x7F7F7F7F
#define AA_VALS 4  // ints per hash slot: lowest code per traversal direction [0..1], minus the highest code per direction [2..3]

static inline unsigned aa_slots(int F) {
    unsigned n = 64;
    while (n < (unsigned)(6 * (long long)F)) n <<= 1;  // load factor <= 1/2
    return n;
}

#ifdef __HIPCC__
__device__ __forceinline__ unsigned aa_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}

// corner i of face f owns the edge opposite to it, (a, b) = (tri[f][(i+1)%3], tri[f][(i+2)%3]); the table keeps, per undirected edge and
// per traversal direction, the smallest AND the largest code f*4+i that claimed it (atomicMin => deterministic).  The neighbour of a
// face across an edge: the lowest code that traverses it the OTHER way (a consistently wound mesh); else, of the codes that traverse
// it the SAME way, the lowest that is not the face itself, else the highest that is not (round 4: with the lowest alone, of two
// faces that traverse a shared edge in the same direction -- a mesh wound inconsistently, e.g. marching tets over a grid file whose
// tets are not uniformly oriented -- only one found the other, and the other took its interior edges for silhouettes; nvdiffrast's
// hash stores the first two opposite vertices of an undirected edge whatever the winding)
__device__ __forceinline__ void aa_insert_edge_ab(int a, int b, int code, unsigned mask, unsigned long long* __restrict__ keys,
                                                  int* __restrict__ vals) {
    if (a == b) return;
    const int d = a < b ? 0 : 1;
    const unsigned long long key = a < b ? (((unsigned long long)(unsigned)a << 32) | (unsigned)b)
                                         : (((unsigned long long)(unsigned)b << 32) | (unsigned)a);
    unsigned h = aa_hash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long old = atomicCAS(&keys[h], AA_EMPTY_KEY, key);
        if (old == AA_EMPTY_KEY || old == key) {
            atomicMin(&vals[AA_VALS * h + d], code);
            atomicMin(&vals[AA_VALS * h + 2 + d], -code);
            return;
        }
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ void aa_insert_edge(const int* __restrict__ tri, int idx, unsigned mask, unsigned long long* __restrict__ keys,
                                               int* __restrict__ vals) {
    const int f = idx / 3, i = idx - 3 * f;
    aa_insert_edge_ab(tri[3 * f + (i + 1) % 3], tri[3 * f + (i + 2) % 3], f * 4 + i, mask, keys, vals);
}

// the rule above: opposite = lowest code of the other direction, same_lo / same_hi = lowest / highest code (-1: none) of the own direction
__device__ __forceinline__ int aa_pick_neighbour(int own, int opposite, int same_lo, int same_hi) {
    if (opposite != AA_NONE) return opposite;
    if (same_lo != AA_NONE && same_lo != own) return same_lo;
    if (same_hi >= 0 && same_hi != own) return same_hi;
    return AA_NONE;
}

// vertex opposite to corner idx's edge in the adjacent triangle (-1: boundary edge)
__device__ __forceinline__ int aa_lookup_edge(const int* __restrict__ tri, int idx, unsigned mask, const unsigned long long* __restrict__ keys,
                                              const int* __restrict__ vals) {
    const int f = idx / 3, i = idx - 3 * f;
    const int a = tri[3 * f + (i + 1) % 3], b = tri[3 * f + (i + 2) % 3];
    if (a == b) return -1;
    const int d = a < b ? 0 : 1;
    const unsigned long long key = a < b ? (((unsigned long long)(unsigned)a << 32) | (unsigned)b)
                                         : (((unsigned long long)(unsigned)b << 32) | (unsigned)a);
    unsigned h = aa_hash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long k = keys[h];
        if (k == key) {
            const int nhi = vals[AA_VALS * h + 2 + d];
            const int other = aa_pick_neighbour(f * 4 + i, vals[AA_VALS * h + (1 - d)], vals[AA_VALS * h + d], nhi == AA_NONE ? -1 : -nhi);
            return other != AA_NONE ? tri[3 * (other >> 2) + (other & 3)] : -1;
        }
        if (k == AA_EMPTY_KEY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// The same answer WITHOUT a hash, from the vertex -> (corner, face) lists (any storage order): every face that holds both end points of
// the edge is found in a's list; per traversal direction the lowest and the highest code f'*4+i' are kept, exactly what the hash's atomicMins leave.
// ~valence x (1 + 3) L2-resident loads: for callers that need a few thousand lookups (the silhouette analysis asks only for pixel
// pairs that passed every geometric test), not a table for all 3F corners.
__device__ __forceinline__ void aa_consider_face(int key, int u0, int u1, int u2, int F, int a, int b, int slot[4]) {
    const int c = key >= 2 * F ? 2 : (key >= F ? 1 : 0);  // a sits at corner c of face g
    const int g = key - c * F;
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
        const int ub = cb == 0 ? u0 : (cb == 1 ? u1 : u2);
        if (cb == c || ub != b) continue;
        const int io = 3 - c - cb;  // the corner opposite to the edge in g; g traverses the edge from corner io+1 to corner io+2
        const int ea = (io + 1) % 3 == c ? a : b, eb = (io + 1) % 3 == c ? b : a;
        const int dd = ea < eb ? 0 : 1;
        const int code = g * 4 + io;
        slot[dd] = code < slot[dd] ? code : slot[dd];
        slot[2 + dd] = code > slot[2 + dd] ? code : slot[2 + dd];
    }
}

// The lists come in two layouts.  CSR (stride 0): list v = adj[off[v] .. off[v + 1]) -- a3d_normals_adjacency, a3d_mesh_topology[_finalize].
// Fixed stride (stride > 0): list v = adj[v * stride ..], off[v] = its length -- written by the DMTet emit launch itself, whose grid bounds
// the valence (a crossing edge is shared by at most `tets per edge` tets, each with at most two triangles at it): no scan, no launch.
__device__ __forceinline__ void vf_list(const int* __restrict__ off, int stride, int v, int& lo, int& n) {
    if (stride > 0) { lo = v * stride; n = min(off[v], stride); }
    else { lo = off[v]; n = off[v + 1] - lo; }
}

__device__ __forceinline__ int aa_opposite_from_lists(const int* __restrict__ tri, const int* __restrict__ off, const int* __restrict__ adj,
                                                      int stride, int F, int f, int i) {
    const int a = tri[3 * f + (i + 1) % 3], b = tri[3 * f + (i + 2) % 3];
    if (a == b) return -1;
    const int d = a < b ? 0 : 1;
    int slot[4] = {AA_NONE, AA_NONE, -1, -1};  // lowest code per direction, highest code per direction
    int lo, n;
    vf_list(off, stride, a, lo, n);
    // up to eight entries at once: all keys in flight, then all index rows in flight (two round trips instead of two per entry)
    int keys[8], rows[8][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) keys[k] = adj[lo + (k < n ? k : 0)];  // (n >= 1: the face f itself is in a's list)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = keys[k] >= 2 * F ? 2 : (keys[k] >= F ? 1 : 0);
        const int g = keys[k] - c * F;
        rows[k][0] = tri[3 * g]; rows[k][1] = tri[3 * g + 1]; rows[k][2] = tri[3 * g + 2];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < n) aa_consider_face(keys[k], rows[k][0], rows[k][1], rows[k][2], F, a, b, slot);
    for (int e = 8; e < n; ++e) {  // valence above eight: the rest one by one
        const int key = adj[lo + e];
        const int c = key >= 2 * F ? 2 : (key >= F ? 1 : 0);
        const int g = key - c * F;
        aa_consider_face(key, tri[3 * g], tri[3 * g + 1], tri[3 * g + 2], F, a, b, slot);
    }
    const int other = aa_pick_neighbour(f * 4 + i, slot[1 - d], slot[d], slot[2 + d]);
    return other != AA_NONE ? tri[3 * (other >> 2) + (other & 3)] : -1;
}

// Vertex -> (corner, face) lists may be stored in any order (the fused DMTet path fills them with atomics and does not sort); the
// consumers visit a list in ascending key order all the same -- corner-major, the order of the reference's three scatter_add_ passes
// (mesh.py:291-293) -- so their float sums do not depend on the fill order: the smallest key above ``last``.  Lists are short
// (valence ~6): up to 8 keys are held in registers (``regs``, loaded once by nr_load_keys), longer lists are re-read from memory.
__device__ __forceinline__ void nr_load_keys(const int* __restrict__ adj, int lo, int n, int regs[8]) {
    // eight UNCONDITIONAL loads (clamped index) in flight, then the selects: a load under `k < n ? .. : ..` becomes a branch with its
    // own wait, i.e. eight dependent round trips (measured: the normals kernels went from 9 to 13-15 us)
    int raw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = adj[lo + (k < n ? k : 0)];  // (callers guarantee n >= 1)
#pragma unroll
    for (int k = 0; k < 8; ++k) regs[k] = k < n ? raw[k] : 0x7fffffff;
}
// smallest key above ``last`` in a list read from memory (lists longer than the register slots)
__device__ __forceinline__ int nr_next_key_mem(const int* __restrict__ adj, int lo, int n, int last) {
    int best = 0x7fffffff;
    for (int k = 0; k < n; ++k) {
        const int a = adj[lo + k];
        best = (a > last && a < best) ? a : best;
    }
    return best;
}

namespace {

constexpr int NR_SCAN_THREADS = 1024;

// single work-group exclusive scan of cnt[V] -> off[V+1]; cnt is reset to 0 (it becomes the fill cursor).  Every thread owns a
// contiguous run of ceil(V / 1024) elements, so the whole array is one pass with two barriers (the first version looped over 1024-
// element slabs with three barriers each: 9 us at V = 5.7k).
__global__ __launch_bounds__(NR_SCAN_THREADS) void nr_adj_scan_kernel(int* __restrict__ cnt, int V, int* __restrict__ off) {
    __shared__ int wave_tot[NR_SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (V + NR_SCAN_THREADS - 1) / NR_SCAN_THREADS;
    const int lo = min((int)threadIdx.x * per, V), hi = min(lo + per, V);
    const int mine = a3d_run_sum(cnt, lo, hi);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    run = a3d_run_scan<true>(cnt, off, lo, hi, run);
    if (threadIdx.x == NR_SCAN_THREADS - 1) off[V] = run;
}

// per-vertex lists into ascending key order, which makes the summation order independent of the fill atomics.  Lists are short
// (valence ~6): up to 8 entries are sorted in registers with a 19-comparator network, longer ones by insertion in place.
__device__ __forceinline__ void nr_adj_sort_vertex(const int* __restrict__ off, int v, int* __restrict__ adj) {
    const int lo = off[v], hi = off[v + 1], n = hi - lo;
    if (n <= 1) return;
    if (n <= 8) {
        int a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = k < n ? adj[lo + k] : 0x7fffffff;
#define NR_CX(i, j) { const int x = min(a[i], a[j]), y = max(a[i], a[j]); a[i] = x; a[j] = y; }
        NR_CX(0, 1) NR_CX(2, 3) NR_CX(4, 5) NR_CX(6, 7)
        NR_CX(0, 2) NR_CX(1, 3) NR_CX(4, 6) NR_CX(5, 7)
        NR_CX(1, 2) NR_CX(5, 6) NR_CX(0, 4) NR_CX(3, 7)
        NR_CX(1, 5) NR_CX(2, 6)
        NR_CX(1, 4) NR_CX(3, 6)
        NR_CX(2, 4) NR_CX(3, 5)
        NR_CX(3, 4)
#undef NR_CX
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < n) adj[lo + k] = a[k];
        return;
    }
    for (int i = lo + 1; i < hi; ++i) {
        const int key = adj[i];
        int j = i - 1;
        while (j >= lo && adj[j] > key) { adj[j + 1] = adj[j]; --j; }
        adj[j + 1] = key;
    }
}

__global__ __launch_bounds__(256) void nr_adj_sort_kernel(const int* __restrict__ off, int V, int* __restrict__ adj) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) nr_adj_sort_vertex(off, v, adj);
}

}  // namespace
#endif
