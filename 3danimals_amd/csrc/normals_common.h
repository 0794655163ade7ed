// Vertex-normal helpers shared by normals.hip (the stand-alone kernels) and raster.hip (the same forward pass as extra work-groups of
// the rasteriser's triangle launch).  Device code only; both translation units are compiled with -ffp-contract=off, so the two
// instances of nr_fwd_vertex round identically.
#pragma once
#include "a3d_common.h"
#include "topo_common.h"

struct NrFace { int i0, i1, i2, c; };

__device__ __forceinline__ NrFace nr_decode(int key, int F, const int* __restrict__ tri) {
    NrFace r;
    r.c = key >= 2 * F ? 2 : (key >= F ? 1 : 0);
    const int f = key - r.c * F;
    r.i0 = tri[3 * f]; r.i1 = tri[3 * f + 1]; r.i2 = tri[3 * f + 2];
    return r;
}

// The walk over a vertex's list is a chain of dependent gathers (list entry -> index row -> three positions): done entry by entry it
// costs three round trips per incident face and the kernel is pure latency (9 us for 1e5 vertices, the same for 6e3).  Lists are short
// (valence ~6), so a thread takes up to NR_SLOTS entries at once: all keys in flight, then all index rows, then all positions -- three
// round trips per VERTEX -- and sorts the keys in registers in between (19 compare-exchanges), so the sums run in ascending key order
// (the reference's scatter_add_ order, mesh.py:291-293) whatever order the list is stored in.  Longer lists take the entry-by-entry
// loop for the rest (keys picked in ascending order from memory).
#define NR_SLOTS 8

__device__ __forceinline__ void nr_sort8(int a[8]) {
#define NR_CX(i, j) { const int x = min(a[i], a[j]), y = max(a[i], a[j]); a[i] = x; a[j] = y; }
    NR_CX(0, 1) NR_CX(2, 3) NR_CX(4, 5) NR_CX(6, 7)
    NR_CX(0, 2) NR_CX(1, 3) NR_CX(4, 6) NR_CX(5, 7)
    NR_CX(1, 2) NR_CX(5, 6) NR_CX(0, 4) NR_CX(3, 7)
    NR_CX(1, 5) NR_CX(2, 6)
    NR_CX(1, 4) NR_CX(3, 6)
    NR_CX(2, 4) NR_CX(3, 5)
    NR_CX(3, 4)
#undef NR_CX
}

// the NR_SLOTS smallest keys of the list, ascending (0x7fffffff = none); for lists of up to NR_SLOTS entries: the whole list
__device__ __forceinline__ void nr_first_keys(const int* __restrict__ adj, int lo, int n, int keys[NR_SLOTS]) {
    nr_load_keys(adj, lo, n, keys);  // eight unconditional loads in flight
    if (n > NR_SLOTS) {  // rare: keep the eight smallest of the whole list
        for (int e = NR_SLOTS; e < n; ++e) {
            const int k = adj[lo + e];
            int imax = 0, vmax = keys[0];
#pragma unroll
            for (int q = 1; q < NR_SLOTS; ++q)
                if (keys[q] > vmax) { vmax = keys[q]; imax = q; }
            if (k < vmax) {
#pragma unroll
                for (int q = 0; q < NR_SLOTS; ++q)
                    if (q == imax) keys[q] = k;
            }
        }
    }
    nr_sort8(keys);
}

// One (image, vertex) of the forward pass: the un-normalised sum of the incident faces' cross products in ascending key order -> acc,
// its safe-normalised value -> nrm.  ``vp`` = this image's vertex array, ``o`` = 3 * (image * V + vertex).
// BATCH = index rows / positions in flight at once (8: three round trips per vertex, 107 registers; 4: five round trips, for hosts
// that cannot afford the registers -- the rasteriser's triangle kernel runs at 72).  Same operations in the same order either way.
template <int BATCH>
__device__ __forceinline__ void nr_fwd_vertex(const float* __restrict__ vp, const int* __restrict__ tri, const int* __restrict__ off,
                                              const int* __restrict__ adj, int stride, int F, int vi, float* __restrict__ acc,
                                              float* __restrict__ nrm, long long o) {
    static_assert(NR_SLOTS % BATCH == 0, "whole batches");
    float x = 0.f, y = 0.f, z = 0.f;
    int lo, cnt;
    vf_list(off, stride, vi, lo, cnt);
    if (cnt > 0) {  // (an isolated vertex -- or F == 0 -- touches neither adj nor tri)
    int keys[NR_SLOTS];
    nr_first_keys(adj, lo, cnt, keys);
#pragma unroll
    for (int k0 = 0; k0 < NR_SLOTS; k0 += BATCH) {
        if (k0 > 0 && k0 >= cnt) break;
        NrFace t[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) t[k] = nr_decode(k0 + k < cnt ? keys[k0 + k] : keys[0], F, tri);  // (the first face again for unused slots: no branch)
        float p[BATCH][9];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const float* p0 = vp + 3ll * t[k].i0;
            const float* p1 = vp + 3ll * t[k].i1;
            const float* p2 = vp + 3ll * t[k].i2;
            p[k][0] = p0[0]; p[k][1] = p0[1]; p[k][2] = p0[2];
            p[k][3] = p1[0]; p[k][4] = p1[1]; p[k][5] = p1[2];
            p[k][6] = p2[0]; p[k][7] = p2[1]; p[k][8] = p2[2];
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            if (k0 + k < cnt) {
                const float ax = p[k][3] - p[k][0], ay = p[k][4] - p[k][1], az = p[k][5] - p[k][2];
                const float bx = p[k][6] - p[k][0], by = p[k][7] - p[k][1], bz = p[k][8] - p[k][2];
                x += ay * bz - az * by; y += az * bx - ax * bz; z += ax * by - ay * bx;
            }
        }
    }
    int last = keys[NR_SLOTS - 1];
    for (int e = NR_SLOTS; e < cnt; ++e) {  // valence above NR_SLOTS: the remaining entries one by one, in ascending key order
        last = nr_next_key_mem(adj, lo, cnt, last);
        const NrFace tt = nr_decode(last, F, tri);
        const float* p0 = vp + 3ll * tt.i0;
        const float* p1 = vp + 3ll * tt.i1;
        const float* p2 = vp + 3ll * tt.i2;
        const float ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
        const float bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
        x += ay * bz - az * by; y += az * bx - ax * bz; z += ax * by - ay * bx;
    }
    }
    acc[o] = x; acc[o + 1] = y; acc[o + 2] = z;
    float d = x * x + y * y + z * z;
    if (!(d > 1e-20f)) { x = 0.f; y = 0.f; z = 1.f; d = 1.f; }
    const float len = sqrtf(fmaxf(d, 1e-20f));  // x / sqrt(clamp(dot, 1e-20)) as render/util.py:28-32 writes it
    nrm[o] = x / len; nrm[o + 1] = y / len; nrm[o + 2] = z / len;
}
