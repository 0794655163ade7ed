// Mesh-topology helpers shared by normals.hip (vertex -> face CSR), antialias.hip (edge -> opposite vertex table) and
// topology.hip (both in one pass over the triangle list).  Device code only; every kernel here lives in an anonymous namespace
// so that each translation unit gets its own copy.
#pragma once
#include "a3d_common.h"

#define AA_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define AA_NONE 0x7F7F7F7F

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

// corner idx = 3f + i owns the edge opposite to it, (tri[f][(i+1)%3], tri[f][(i+2)%3]); the table keeps, per undirected edge and
// per traversal direction, the smallest code f*4+i that claimed it (atomicMin => deterministic)
__device__ __forceinline__ void aa_insert_edge(const int* __restrict__ tri, int idx, unsigned mask, unsigned long long* __restrict__ keys,
                                               int* __restrict__ vals) {
    const int f = idx / 3, i = idx - 3 * f;
    const int a = tri[3 * f + (i + 1) % 3], b = tri[3 * f + (i + 2) % 3];
    if (a == b) return;
    const int d = a < b ? 0 : 1;
    const unsigned long long key = a < b ? (((unsigned long long)(unsigned)a << 32) | (unsigned)b)
                                         : (((unsigned long long)(unsigned)b << 32) | (unsigned)a);
    unsigned h = aa_hash(key) & mask;
    for (unsigned probe = 0; probe <= mask; ++probe) {
        const unsigned long long old = atomicCAS(&keys[h], AA_EMPTY_KEY, key);
        if (old == AA_EMPTY_KEY || old == key) {
            atomicMin(&vals[2 * h + d], f * 4 + i);
            return;
        }
        h = (h + 1) & mask;
    }
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
            int other = vals[2 * h + (1 - d)];
            if (other == AA_NONE) {
                const int same = vals[2 * h + d];
                if (same != AA_NONE && same != f * 4 + i) other = same;
            }
            return other != AA_NONE ? tri[3 * (other >> 2) + (other & 3)] : -1;
        }
        if (k == AA_EMPTY_KEY) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

namespace {

constexpr int NR_SCAN_THREADS = 1024;

// single work-group exclusive scan of cnt[V] -> off[V+1]; cnt is reset to 0 (it becomes the fill cursor)
__global__ __launch_bounds__(NR_SCAN_THREADS) void nr_adj_scan_kernel(int* __restrict__ cnt, int V, int* __restrict__ off) {
    __shared__ int wave_tot[NR_SCAN_THREADS / 64];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < V; base += NR_SCAN_THREADS) {
        const int i = base + threadIdx.x;
        const int c = i < V ? cnt[i] : 0;
        int incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (i < V) { off[i] = before + incl - c; cnt[i] = 0; }
        __syncthreads();
        if (threadIdx.x == NR_SCAN_THREADS - 1) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[V] = carry_s;
}

// lists are short (valence ~6): insertion sort in place makes the summation order independent of the fill atomics
__global__ __launch_bounds__(256) void nr_adj_sort_kernel(const int* __restrict__ off, int V, int* __restrict__ adj) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int lo = off[v], hi = off[v + 1];
    for (int i = lo + 1; i < hi; ++i) {
        const int key = adj[i];
        int j = i - 1;
        while (j >= lo && adj[j] > key) { adj[j + 1] = adj[j]; --j; }
        adj[j + 1] = key;
    }
}

}  // namespace
#endif
