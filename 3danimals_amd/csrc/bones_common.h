// Kinematic-chain helpers shared by bones.hip (stand-alone a3d_bone_transforms_*) and skin.hip (a3d_skin_pose_*, where every
// work-group of the skinning launch composes its image's chains itself).  Device code only.
#pragma once
#include "a3d_common.h"

#define BN_MAXK 64
#define BN_MAXD 8

struct A34 {  // row-major 3x4 affine (last row 0 0 0 1 implied)
    float m[12];
};

__device__ __forceinline__ A34 bn_identity() {
    A34 a;
#pragma unroll
    for (int i = 0; i < 12; ++i) a.m[i] = 0.f;
    a.m[0] = a.m[5] = a.m[10] = 1.f;
    return a;
}

__device__ __forceinline__ A34 bn_mul(const A34& a, const A34& b) {  // a . b
    A34 c;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int col = 0; col < 4; ++col) {
            float s = a.m[4 * r] * b.m[col] + a.m[4 * r + 1] * b.m[4 + col] + a.m[4 * r + 2] * b.m[8 + col];
            if (col == 3) s += a.m[4 * r + 3];
            c.m[4 * r + col] = s;
        }
    }
    return c;
}

__device__ __forceinline__ void bn_normalize(float& x, float& y, float& z) {  // torch.nn.functional.normalize, eps 1e-12
    const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    x /= n; y /= n; z /= n;
}

// rest frame R (row-major 3x3, columns right | up | forward) of a bone a -> b   (skinning.py:251-270)
__device__ __forceinline__ void bn_rest(const float* __restrict__ bone, float R[9], float t[3]) {
    float fx = bone[3] - bone[0], fy = bone[4] - bone[1], fz = bone[5] - bone[2];
    bn_normalize(fx, fy, fz);
    // up = normalize(forward x (1,0,0)) = normalize(0, fz, -fy)
    float ux = 0.f, uy = fz, uz = -fy;
    bn_normalize(ux, uy, uz);
    // right = up x forward
    const float rx = uy * fz - uz * fy, ry = uz * fx - ux * fz, rz = ux * fy - uy * fx;
    bn_normalize(ux, uy, uz);
    R[0] = rx; R[1] = ux; R[2] = fx;
    R[3] = ry; R[4] = uy; R[5] = fy;
    R[6] = rz; R[7] = uz; R[8] = fz;
    t[0] = bone[0]; t[1] = bone[1]; t[2] = bone[2];
}

// sine / cosine of the three Euler angles of a link (sx, cx, sy, cy, sz, cz)
struct BnTrig {
    float sx, cx, sy, cy, sz, cz;
};

__device__ __forceinline__ BnTrig bn_trig(const float* __restrict__ ang) {
    BnTrig t;
    sincosf(ang[0], &t.sx, &t.cx);
    sincosf(ang[1], &t.sy, &t.cy);
    sincosf(ang[2], &t.sz, &t.cz);
    return t;
}

__device__ __forceinline__ void bn_euler_from(const BnTrig& g, float Rot[9]) {  // Rx(x) Ry(y) Rz(z)
    const float cx = g.cx, sx = g.sx, cy = g.cy, sy = g.sy, cz = g.cz, sz = g.sz;
    Rot[0] = cy * cz;                 Rot[1] = -cy * sz;                Rot[2] = sy;
    Rot[3] = sx * sy * cz + cx * sz;  Rot[4] = -sx * sy * sz + cx * cz; Rot[5] = -sx * cy;
    Rot[6] = -cx * sy * cz + sx * sz; Rot[7] = cx * sy * sz + sx * cz;  Rot[8] = cx * cy;
}

__device__ __forceinline__ void bn_euler(const float* __restrict__ ang, float Rot[9]) { bn_euler_from(bn_trig(ang), Rot); }

__device__ __forceinline__ void bn_mat3(const float A[9], const float B[9], float C[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// the link of a bone from its rest frame (bn_rest) and the sines / cosines of its angles: [R Rot R^T | t - R Rot R^T t]
__device__ __forceinline__ A34 bn_link_from(const float R[9], const float t[3], const BnTrig& g) {
    float Rot[9], T1[9], Lr[9];
    bn_euler_from(g, Rot);
    bn_mat3(R, Rot, T1);
    // Lr = T1 . R^T
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Lr[3 * r + c] = T1[3 * r] * R[3 * c] + T1[3 * r + 1] * R[3 * c + 1] + T1[3 * r + 2] * R[3 * c + 2];
    A34 L;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        L.m[4 * r] = Lr[3 * r]; L.m[4 * r + 1] = Lr[3 * r + 1]; L.m[4 * r + 2] = Lr[3 * r + 2];
        L.m[4 * r + 3] = t[r] - (Lr[3 * r] * t[0] + Lr[3 * r + 1] * t[1] + Lr[3 * r + 2] * t[2]);
    }
    return L;
}

__device__ __forceinline__ A34 bn_link(const float* __restrict__ bone, const float* __restrict__ ang) {
    float R[9], t[3];
    bn_rest(bone, R, t);
    return bn_link_from(R, t, bn_trig(ang));
}

__device__ __forceinline__ void bn_store(float* __restrict__ dst, const A34& a) {
#pragma unroll
    for (int q = 0; q < 12; ++q) dst[q] = a.m[q];
}
// (16-byte aligned destination: three 16-byte stores instead of twelve scalar ones -- a scattered store costs its lane a line request
// whatever its width)
__device__ __forceinline__ void bn_store16(float* __restrict__ dst, const A34& a) {
    float4* d = reinterpret_cast<float4*>(dst);
    d[0] = make_float4(a.m[0], a.m[1], a.m[2], a.m[3]);
    d[1] = make_float4(a.m[4], a.m[5], a.m[6], a.m[7]);
    d[2] = make_float4(a.m[8], a.m[9], a.m[10], a.m[11]);
}
__device__ __forceinline__ A34 bn_load(const float* __restrict__ src) {
    A34 a;
#pragma unroll
    for (int q = 0; q < 12; ++q) a.m[q] = src[q];
    return a;
}


// adjoint of ONE chain link: g (= dM of the bone, 12 floats), the prefix product P and the suffix product S of the link's position in the
// bone's chain (row-major 3x4), the link's bone and angles -> the three angle gradients
__device__ __forceinline__ void bn_link_adjoint(const float* g, const float* P, const float* S, const float* __restrict__ bone,
                                        const float* __restrict__ ang, float& gx, float& gy, float& gz) {
    // M = P L S  =>  dL = P_r^T dM S_full^T  (dM's implied last row is zero; S_full = [S; 0 0 0 1])
    float T1[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) T1[4 * r + c] = P[r] * g[c] + P[4 + r] * g[4 + c] + P[8 + r] * g[8 + c];
    float GL[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            GL[4 * r + c] = T1[4 * r] * S[4 * c] + T1[4 * r + 1] * S[4 * c + 1] + T1[4 * r + 2] * S[4 * c + 2] + T1[4 * r + 3] * S[4 * c + 3];
        GL[4 * r + 3] = T1[4 * r + 3];
    }
    // link: Lr = R Rot R^T, Lt = t - Lr t   =>  g_Lr_total = g_Lr - g_Lt (x) t ;  g_Rot = R^T g_Lr_total R
    float R[9], t[3];
    bn_rest(bone, R, t);
    float GLr[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) GLr[3 * r + c] = GL[4 * r + c] - GL[4 * r + 3] * t[c];
    float T2[9], GRot[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T2[3 * r + c] = R[r] * GLr[c] + R[3 + r] * GLr[3 + c] + R[6 + r] * GLr[6 + c];  // R^T . GLr
    bn_mat3(T2, R, GRot);
    // d Rot / d angles for Rot = Rx Ry Rz
    const float x = ang[0], y = ang[1], z = ang[2];
    const float cx = cosf(x), sx = sinf(x), cy = cosf(y), sy = sinf(y), cz = cosf(z), sz = sinf(z);
    const float dX[9] = {0.f, 0.f, 0.f,
                         cx * sy * cz - sx * sz, -cx * sy * sz - sx * cz, -cx * cy,
                         sx * sy * cz + cx * sz, -sx * sy * sz + cx * cz, -sx * cy};
    const float dY[9] = {-sy * cz, sy * sz, cy,
                         sx * cy * cz, -sx * cy * sz, sx * sy,
                         -cx * cy * cz, cx * cy * sz, -cx * sy};
    const float dZ[9] = {-cy * sz, -cy * cz, 0.f,
                         -sx * sy * sz + cx * cz, -sx * sy * cz - cx * sz, 0.f,
                         cx * sy * sz + sx * cz, cx * sy * cz - sx * sz, 0.f};
#pragma unroll
    for (int q = 0; q < 9; ++q) { gx += GRot[q] * dX[q]; gy += GRot[q] * dY[q]; gz += GRot[q] * dZ[q]; }
        }

// d L / d angle_c (c = x, y, z) of one link, as three row-major 3x4 blocks: L = [R Rot R^T | t - R Rot R^T t]  =>
// dL = [R dRot R^T | -(R dRot R^T) t].  With them the adjoint of a link is <P^T g S^T, dL_c>: 36 multiply-adds, no sin / cos, no
// normalisation -- computed once per link by the forward (a3d_skin_pose_fwd) instead of once per (bone, chain position) pair and
// work-group by the backward.
// (one angle c of the three: 12 floats to out)
__device__ __forceinline__ void bn_link_derivative_from(const float R[9], const float t[3], const BnTrig& g, int c, float* __restrict__ out) {
    const float cx = g.cx, sx = g.sx, cy = g.cy, sy = g.sy, cz = g.cz, sz = g.sz;
    float dR[9];
    if (c == 0) {
        dR[0] = 0.f; dR[1] = 0.f; dR[2] = 0.f;
        dR[3] = cx * sy * cz - sx * sz; dR[4] = -cx * sy * sz - sx * cz; dR[5] = -cx * cy;
        dR[6] = sx * sy * cz + cx * sz; dR[7] = -sx * sy * sz + cx * cz; dR[8] = -sx * cy;
    } else if (c == 1) {
        dR[0] = -sy * cz; dR[1] = sy * sz; dR[2] = cy;
        dR[3] = sx * cy * cz; dR[4] = -sx * cy * sz; dR[5] = sx * sy;
        dR[6] = -cx * cy * cz; dR[7] = cx * cy * sz; dR[8] = -cx * sy;
    } else {
        dR[0] = -cy * sz; dR[1] = -cy * cz; dR[2] = 0.f;
        dR[3] = -sx * sy * sz + cx * cz; dR[4] = -sx * sy * cz - cx * sz; dR[5] = 0.f;
        dR[6] = cx * sy * sz + sx * cz; dR[7] = cx * sy * cz - sx * sz; dR[8] = 0.f;
    }
    float T1[9];
    bn_mat3(R, dR, T1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float lr[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) lr[q] = T1[3 * r] * R[3 * q] + T1[3 * r + 1] * R[3 * q + 1] + T1[3 * r + 2] * R[3 * q + 2];  // (T1 . R^T)
        out[4 * r] = lr[0]; out[4 * r + 1] = lr[1]; out[4 * r + 2] = lr[2];
        out[4 * r + 3] = -(lr[0] * t[0] + lr[1] * t[1] + lr[2] * t[2]);
    }
}

__device__ __forceinline__ void bn_link_derivatives(const float* __restrict__ bone, const float* __restrict__ ang, float* __restrict__ out) {
    float R[9], t[3];
    bn_rest(bone, R, t);
    const BnTrig g = bn_trig(ang);
#pragma unroll
    for (int c = 0; c < 3; ++c) bn_link_derivative_from(R, t, g, c, out + 12 * c);
}

// the same adjoint as bn_link_adjoint from the precomputed derivatives dL[3][12] of the link
__device__ __forceinline__ void bn_link_adjoint_dl(const float* g, const float* P, const float* S, const float* __restrict__ dL, float& gx,
                                                   float& gy, float& gz) {
    float T1[12];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) T1[4 * r + c] = P[r] * g[c] + P[4 + r] * g[4 + c] + P[8 + r] * g[8 + c];
    float GL[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            GL[4 * r + c] = T1[4 * r] * S[4 * c] + T1[4 * r + 1] * S[4 * c + 1] + T1[4 * r + 2] * S[4 * c + 2] + T1[4 * r + 3] * S[4 * c + 3];
        GL[4 * r + 3] = T1[4 * r + 3];
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) { gx += GL[q] * dL[q]; gy += GL[q] * dL[12 + q]; gz += GL[q] * dL[24 + q]; }
}

// Adjoint of the chain composition for ONE instance, run by a whole work-group (any size): g_M [K,12] -> g_angles [K,3].
// phase 1 links -> LDS; phase 2 (thread = bone, two threads per bone) prefix products P_j = L_0..L_{j-1} and suffix products
// S_j = L_{j+1}..L_{D-1} of its chain -> LDS; phase 3 (thread = (bone, chain position)) the adjoint of one link pushed through the
// conjugation and the Euler factors -> 3 floats in LDS; phase 4 (thread = (link, component)) sums the contributions of every bone whose
// chain holds the link, in bone order: no atomics, no memset, bit-reproducible.  ``s_dyn``: bn_bwd_lds(K, D) bytes of LDS.
// COHERENT: g_M was accumulated by device-scope atomics of OTHER work-groups of the same launch (a3d_skin_pose_bwd's last-work-group
// tail): read it with agent-scope loads.  ``g_extra`` (or null) is added to g_M (a gradient that reached the transforms directly).
template <bool COHERENT>
__device__ __forceinline__ void bn_chain_adjoint(const float* g_M, const float* __restrict__ g_extra, const float* __restrict__ bb,
                                                 const float* __restrict__ aa, const int* __restrict__ chain, int K, int D,
                                                 float* __restrict__ g_angles, float* s_dyn) {
    float (*s_L)[13] = (float (*)[13])s_dyn;                    // [K][13]
    float (*s_P)[13] = (float (*)[13])(s_dyn + 13 * K);         // [K*D][13]
    float (*s_S)[13] = (float (*)[13])(s_dyn + 13 * K * (1 + D));  // [K*D][13]
    float (*s_c)[3] = (float (*)[3])(s_dyn + 13 * K * (1 + 2 * D));  // [K*D][3]
    int* s_chain = (int*)(s_dyn + 13 * K * (1 + 2 * D) + 3 * K * D);     // [K*D]
    int* s_pos = s_chain + K * D;                                        // [K*K]: chain slot of link i in bone k's chain, or -1
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) s_chain[w] = chain[w];
    for (int w = threadIdx.x; w < K * K; w += blockDim.x) s_pos[w] = -1;
    for (int i = threadIdx.x; i < K; i += blockDim.x) bn_store(s_L[i], bn_link(bb + 6 * i, aa + 3 * i));
    __syncthreads();
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) {  // a link occurs at most once in a chain: no write conflicts
        const int i = s_chain[w];
        if (i >= 0) s_pos[(w / D) * K + i] = w;
    }
    for (int w = threadIdx.x; w < 2 * K; w += blockDim.x) {
        const int k = w >> 1;
        if ((w & 1) == 0) {
            A34 run = bn_identity();
            for (int j = 0; j < D; ++j) {
                bn_store(s_P[k * D + j], run);
                const int i = s_chain[k * D + j];
                if (i >= 0) run = bn_mul(run, bn_load(s_L[i]));
            }
        } else {
            A34 run = bn_identity();
            for (int j = D - 1; j >= 0; --j) {
                bn_store(s_S[k * D + j], run);
                const int i = s_chain[k * D + j];
                if (i >= 0) run = bn_mul(bn_load(s_L[i]), run);
            }
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) {
        const int k = w / D;
        const int i = s_chain[w];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (i >= 0) {
            float g[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                g[q] = COHERENT ? __hip_atomic_load(g_M + 12 * k + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : g_M[12 * k + q];
                if (g_extra) g[q] += g_extra[12 * k + q];
            }
            bn_link_adjoint(g, s_P[w], s_S[w], bb + 6 * i, aa + 3 * i, gx, gy, gz);
        }
        s_c[w][0] = gx; s_c[w][1] = gy; s_c[w][2] = gz;
    }
    __syncthreads();
    // phase 4: thread = (link, component); the contributions of the bones whose chain holds the link, in bone order
    for (int t = threadIdx.x; t < 3 * K; t += blockDim.x) {
        const int i = t / 3, comp = t - 3 * i;
        float g = 0.f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const int w = s_pos[k * K + i];
            g += w >= 0 ? s_c[w][comp] : 0.f;
        }
        g_angles[3 * i + comp] = g;
    }
}

static inline size_t bn_bwd_lds(int K, int D) { return sizeof(float) * ((size_t)13 * K * (1 + 2 * D) + (size_t)4 * K * D + (size_t)K * K); }



// ---- the same adjoint with the prefix / suffix products PRECOMPUTED by the forward (a3d_skin_pose_fwd): what is left for the work-group
// that finishes an image last is one parallel phase over the K*D (bone, position) pairs and the per-link sums -- ~3 us instead of ~10.
// PS[K][D][2][12]: P_j and S_j of bone k's chain (identity for padded positions).
__device__ __forceinline__ void bn_chain_products(const float (*s_L)[13], const int* s_chain, int K, int D, float* __restrict__ PS) {
    // thread = (bone, direction).  The chain's link indices are read up front and padded positions multiply by the identity: with
    // ``i = s_chain[..]; if (i >= 0) run = run . L[i]`` inside the loop every step was two dependent LDS round trips behind a branch
    // (0.46 us per step -- this work-group was the last of its launch to finish)
    for (int w = threadIdx.x; w < 2 * K; w += blockDim.x) {
        const int k = w >> 1;
        const bool suffix = w & 1;
        int idx[BN_MAXD];
#pragma unroll
        for (int j = 0; j < BN_MAXD; ++j) idx[j] = j < D ? s_chain[k * D + j] : -1;
        A34 run = bn_identity();
#pragma unroll
        for (int jj = 0; jj < BN_MAXD; ++jj) {
            const int j = suffix ? D - 1 - jj : jj;  // (suffix: positions D-1 .. 0; idx[] is indexed with constants below)
            if (jj < D) {
                bn_store16(PS + ((long long)(k * D + j) * 2 + (suffix ? 1 : 0)) * 12, run);  // (PS: 16-byte aligned, a3d_skin_pose_fwd checks)
                int i = -1;
#pragma unroll
                for (int q = 0; q < BN_MAXD; ++q) i = q == j ? idx[q] : i;
                A34 L = bn_load(s_L[i >= 0 ? i : 0]);
                if (i < 0) L = bn_identity();
                run = suffix ? bn_mul(L, run) : bn_mul(run, L);
            }
        }
    }
}

// Floats per image of the buffer a3d_skin_pose_fwd leaves for the backward: PS[K][D][2][12] (bn_chain_products) + dL[K][3][12]
// (bn_link_derivatives).
__host__ __device__ static inline size_t bn_products_floats(int K, int D) { return (size_t)K * D * 24 + (size_t)K * 36; }

// s_mem: (4*K*D + K*K) words of LDS.  g_M is ONE work-group's share of the image's transform gradient, in LDS; the result is ADDED to
// g_angles (zero on entry): the adjoint is linear in g_M, so the sum over the work-groups of adjoint(share) is adjoint(sum) and no
// work-group has to wait for the others.  PS / dL: the precomputed products and link derivatives of this image.
__device__ __forceinline__ void bn_chain_adjoint_ps(const float* g_M, const float* __restrict__ PS, const float* __restrict__ dL,
                                                    const int* __restrict__ chain, int K, int D, float* __restrict__ g_angles, float* s_mem,
                                                    int chain_of_thread = -2) {
    // chain_of_thread (optional): chain[threadIdx.x], loaded by the caller long before (-2: not given) -- the rows of PS / dL below hang on
    // it, and a load issued here is a round trip of its own in front of theirs
    float (*s_c)[3] = (float (*)[3])s_mem;       // [K*D][3]
    int* s_chain = (int*)(s_mem + 3 * K * D);    // [K*D]
    int* s_pos = s_chain + K * D;                // [K*K]
    for (int w = threadIdx.x; w < K * K; w += blockDim.x) s_pos[w] = -1;
    __syncthreads();
    for (int w = threadIdx.x; w < K * D; w += blockDim.x) {
        const int k = w / D, i = (chain_of_thread != -2 && w == (int)threadIdx.x) ? chain_of_thread : chain[w];
        s_chain[w] = i;
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (i >= 0) {
            s_pos[k * K + i] = w;
            float g[12], P[12], S[12], d[36];
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                g[q] = g_M[12 * k + q];
                P[q] = PS[(long long)w * 24 + q];
                S[q] = PS[(long long)w * 24 + 12 + q];
            }
#pragma unroll
            for (int q = 0; q < 36; ++q) d[q] = dL[36 * i + q];
            bn_link_adjoint_dl(g, P, S, d, gx, gy, gz);
        }
        s_c[w][0] = gx; s_c[w][1] = gy; s_c[w][2] = gz;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * K; t += blockDim.x) {
        const int i = t / 3, comp = t - 3 * i;
        float g = 0.f;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
            const int w = s_pos[k * K + i];
            g += w >= 0 ? s_c[w][comp] : 0.f;
        }
        atomicAdd(g_angles + 3 * i + comp, g);
    }
}
