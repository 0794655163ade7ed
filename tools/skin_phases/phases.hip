// Where the microseconds of the two skinning launches go: a stand-alone harness (no Python, no torch) that compiles csrc/skin.hip with
// A3D_PROFILE (a3d_common.h: thread 0 of every work-group stamps the 100 MHz wall clock at the phase boundaries; tools/kernel_phases.py
// does the same for every kernel of the library inside the real step) and runs a3d_skin_pose_fwd /
// a3d_skin_pose_bwd at the bench size (B = 16, V = 5928, K = 20 bones, chains of <= 8 links) or the one given on the command line.
//
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA3D_PROFILE -I include -I 3danimals_amd/csrc tools/skin_phases/phases.hip \
//           3danimals_amd/csrc/common.hip -o gpurun_out/skin_phases && gpurun_out/skin_phases [B V K D]
//
// Prints, per launch: the event-timed duration without stamps (mean of 200), the spread of the work-groups' start times, and for
// every phase the median / maximum duration over the work-groups, plus the time from the first start to the last end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../3danimals_amd/csrc/skin.hip"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)(s >> 8) / 16777216.f;
}

template <typename T>
static T* upload(const std::vector<T>& h) {
    T* d;
    CK(hipMalloc(&d, h.size() * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static void report(const char* name, const std::vector<unsigned long long>& st, int n_wg, const char* const* phase, int n_stamps) {
    unsigned long long t0 = ~0ull, t1 = 0, s_max = 0;
    for (int w = 0; w < n_wg; ++w) {
        t0 = std::min(t0, st[8 * w]);
        s_max = std::max(s_max, st[8 * w]);
        for (int k = 0; k < n_stamps; ++k) t1 = std::max(t1, st[8 * w + k]);
    }
    printf("%s: %d work-groups, starts spread over %.2f us, first start -> last stamp %.2f us\n", name, n_wg, (s_max - t0) * 0.01, (t1 - t0) * 0.01);
    for (int k = 1; k < n_stamps; ++k) {
        std::vector<double> d;
        for (int w = 0; w < n_wg; ++w)
            if (st[8 * w + k] && st[8 * w + k - 1] && st[8 * w + k] >= st[8 * w + k - 1]) d.push_back((st[8 * w + k] - st[8 * w + k - 1]) * 0.01);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("    %-44s median %6.2f  p90 %6.2f  max %6.2f us   (%zu work-groups)\n", phase[k], d[d.size() / 2], d[d.size() * 9 / 10], d.back(), d.size());
    }
    {   // shader clocks per wall-clock microsecond over the work-groups' lifetimes (stamps 6, 7 = clock64() at stamps 0, 5)
        std::vector<double> f;
        for (int w = 0; w < n_wg; ++w)
            if (st[8 * w + 5] > st[8 * w] && st[8 * w + 7] > st[8 * w + 6]) f.push_back((double)(st[8 * w + 7] - st[8 * w + 6]) / ((st[8 * w + 5] - st[8 * w]) * 0.01));
        if (!f.empty()) {
            std::sort(f.begin(), f.end());
            printf("    shader clock while the work-groups ran: median %.0f MHz (min %.0f, max %.0f)\n", f[f.size() / 2], f.front(), f.back());
        }
    }
    std::vector<double> life;
    for (int w = 0; w < n_wg; ++w) {
        unsigned long long e = 0;
        for (int k = 0; k < n_stamps; ++k) e = std::max(e, st[8 * w + k]);
        life.push_back((e - st[8 * w]) * 0.01);
    }
    std::sort(life.begin(), life.end());
    printf("    %-44s median %6.2f  p90 %6.2f  max %6.2f us\n", "work-group lifetime (first -> last stamp)", life[life.size() / 2], life[life.size() * 9 / 10], life.back());
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16, V = argc > 2 ? atoi(argv[2]) : 5928, K = argc > 3 ? atoi(argv[3]) : 20, D = argc > 4 ? atoi(argv[4]) : 8;
    unsigned seed = 1;
    std::vector<float> v((size_t)B * V * 3), bones((size_t)B * K * 6), angles((size_t)B * K * 3), g((size_t)B * V * 3);
    for (auto& x : v) x = 2.f * frand(seed) - 1.f;
    for (auto& x : bones) x = 2.f * frand(seed) - 1.f;
    for (auto& x : angles) x = 0.6f * frand(seed) - 0.3f;
    for (auto& x : g) x = 2.f * frand(seed) - 1.f;
    std::vector<int> chain((size_t)K * D, -1);
    for (int k = 0; k < K; ++k) {  // bone k hangs below the <= D - 1 bones numbered before it: chains of 1 .. D links, root first
        const int len = std::min(k + 1, D);
        for (int j = 0; j < len; ++j) chain[(size_t)k * D + j] = k - len + 1 + j;
    }
    float *d_v = upload(v), *d_b = upload(bones), *d_a = upload(angles), *d_g = upload(g);
    int* d_c = upload(chain);
    float *d_out, *d_T, *d_PS, *d_gang, *d_gv;
    const size_t psf = a3d_skin_pose_products_floats(K, D);
    CK(hipMalloc(&d_out, sizeof(float) * B * V * 3));
    CK(hipMalloc(&d_gv, sizeof(float) * B * V * 3));
    CK(hipMalloc(&d_T, sizeof(float) * B * K * 12));
    CK(hipMalloc(&d_PS, sizeof(float) * B * psf));
    CK(hipMalloc(&d_gang, sizeof(float) * B * K * 3));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto fwd = [&]() {
        if (a3d_skin_pose_fwd(d_v, B, d_b, B, d_a, d_c, B, V, K, D, 0.05f, d_out, d_T, d_PS, d_gang, s)) { printf("fwd refused: %s\n", a3d_last_error()); exit(1); }
    };
    auto bwd = [&]() {
        if (a3d_skin_pose_bwd(d_g, d_v, B, d_b, B, d_T, d_PS, d_a, d_c, B, V, K, D, 0.05f, d_gv, nullptr, d_gang, 0, s)) { printf("bwd refused: %s\n", a3d_last_error()); exit(1); }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](auto fn, const char* name) {
        for (int i = 0; i < 20; ++i) fn();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 200; ++i) fn();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per call back to back (events over 200 calls; B = %d, V = %d, K = %d, D = %d)\n", name, ms * 5.f, B, V, K, D);
    };
    timeit(fwd, "a3d_skin_pose_fwd");
    timeit(bwd, "a3d_skin_pose_bwd (incl. its memset of g_angles)");
    const int max_wg = 1 << 16;
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, sizeof(unsigned long long) * 8 * max_wg));
    auto profile = [&](auto fn, const char* name, int n_wg, const char* const* phase, int n_stamps, int kid) {
        for (int rep = 0; rep < 3; ++rep) {  // (the last repetition is reported)
            CK(hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * 8 * max_wg, s));
            CK(hipStreamSynchronize(s));
            if (a3d_profile_set_skin(d_st, kid)) exit(1);
            fn();
            CK(hipStreamSynchronize(s));
        }
        if (a3d_profile_set_skin(nullptr, -1)) exit(1);
        std::vector<unsigned long long> st((size_t)8 * n_wg);
        CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        report(name, st, n_wg, phase, n_stamps);
    };
    {
        const int ngroups = (V + 63) / 64, groups = std::max(1, (int)(((long long)ngroups * B + 1023) / 1024));  // (a3d_skin_pose_fwd's rule)
        const int gx = (ngroups + groups - 1) / groups + 1;
        static const char* const ph[] = {"", "links + bones into LDS (-> barrier)", "chain products (thread 0's bone)", "barrier (all chains done)", "", "logits, softmax, blend, store"};
        // (stamp 4 is unused in the forward: 3 -> 5 is reported under slot 5 by copying 3 into 4 below)
        profile([&]() { fwd(); }, "sk_fwd_kernel<20, true>", gx * B, ph, 4, 0);
        // the vertex phase separately
        std::vector<unsigned long long> st((size_t)8 * gx * B);
        CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> d;
        for (int w = 0; w < gx * B; ++w)
            if (st[8 * w + 3] && st[8 * w + 5] >= st[8 * w + 3]) d.push_back((st[8 * w + 5] - st[8 * w + 3]) * 0.01);
        std::sort(d.begin(), d.end());
        if (!d.empty()) printf("    %-44s median %6.2f  p90 %6.2f  max %6.2f us   (%zu work-groups)\n", ph[5], d[d.size() / 2], d[d.size() * 9 / 10], d.back(), d.size());
        std::vector<double> pw;
        for (int b = 0; b < B; ++b) {
            const int w = b * gx + gx - 1;
            if (st[8 * w + 5] >= st[8 * w + 1]) pw.push_back((st[8 * w + 5] - st[8 * w + 1]) * 0.01);
        }
        std::sort(pw.begin(), pw.end());
        if (!pw.empty()) printf("    %-44s median %6.2f            max %6.2f us   (%zu work-groups)\n", "products work-group: prefix / suffix + d link", pw[pw.size() / 2], pw.back(), pw.size());
    }
    {
        const int chunks = (V + 255) / 256;
        int cpb = (int)(((long long)chunks * B + 767) / 768);
        if (cpb < 1) cpb = 1;
        const int gx = (chunks + cpb - 1) / cpb;
        static const char* const ph[] = {"", "stage + softmax weights, g_v (phase 1)", "g_T on the matrix pipe (phase 2)", "the waves' tiles into LDS (-> barrier)", "share of g_T into LDS (-> barrier)", "chain adjoint + atomics on g_angles"};
        profile([&]() { bwd(); }, "sk_bwd_kernel<5, true>", gx * B, ph, 6, 1);
    }
    return 0;
}
