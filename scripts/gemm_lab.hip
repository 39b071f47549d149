// Lab harness for the config-5 dense GEMM kernels: runs dense_scores_256w<NBUF, LAB> on random fp16 data at the bench shape,
// checks a sample of the scores against a plain per-element dot product computed on the device in f32, and times it
// with HIP events. Diagnostic only (scripts/gpu_gemm_lab.sh builds and runs it on the GPU box).
#define MSR_GEMM_LAB 1
#include "gemm_lab_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

using namespace msr;

#define CK(x)                                                                   \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

__global__ void ref_dots(const _Float16* Q, const _Float16* P, const uint32_t* qs, const uint32_t* ds, float* o, uint32_t n,
                         uint32_t H) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const _Float16* q = Q + (uint64_t)qs[i] * H;
    const _Float16* p = P + (uint64_t)ds[i] * H;
    float s = 0.f;
    for (uint32_t k = 0; k < H; ++k) s += (float)q[k] * (float)p[k];
    o[i] = s;
}

typedef void (*gemm_kernel)(const _Float16*, const _Float16*, uint32_t*, uint32_t, uint32_t, uint32_t, uint64_t, uint32_t, uint32_t,
                            uint32_t);
static int run_kernel(gemm_kernel kern, int lds_bytes, const _Float16* dq, const _Float16* dp, uint32_t* dout, uint32_t M, uint32_t N,
                      uint32_t H, uint64_t ld, int reps, hipStream_t st, float* ms_out);
template <int NBUF, int LAB = 0>
static int run(const _Float16* dq, const _Float16* dp, uint32_t* dout, uint32_t M, uint32_t N, uint32_t H, uint64_t ld,
               int reps, hipStream_t st, float* ms_out) {
    return run_kernel(dense_scores_256w<NBUF, LAB>, NBUF * kGwStage, dq, dp, dout, M, N, H, ld, reps, st, ms_out);
}
static int run_kernel(gemm_kernel kern, int lds_bytes, const _Float16* dq, const _Float16* dp, uint32_t* dout, uint32_t M, uint32_t N,
                      uint32_t H, uint64_t ld, int reps, hipStream_t st, float* ms_out) {
    const uint32_t qb_n = (M + 255) / 256, db_n = (N + 255) / 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                           lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid((qb_n * db_n + 7) / 8 * 8);
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, dq, dp, dout, M, N, H, ld, qb_n, db_n, 1u);
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, dq, dp, dout, M, N, H, ld, qb_n, db_n, 1u);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(ms_out, e0, e1));
    *ms_out /= reps;
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 25010, N = argc > 2 ? atoi(argv[2]) : 5000, H = argc > 3 ? atoi(argv[3]) : 4096;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int nbuf = argc > 5 ? atoi(argv[5]) : 4;
    const uint32_t Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256;
    const uint64_t ld = Np;
    std::vector<_Float16> hq((size_t)Mp * H, (_Float16)0.f), hp((size_t)Np * H, (_Float16)0.f);
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (size_t i = 0; i < (size_t)M * H; ++i) hq[i] = (_Float16)(nd(g) * 0.05f);
    for (size_t i = 0; i < (size_t)N * H; ++i) hp[i] = (_Float16)(nd(g) * 0.05f);
    _Float16 *dq, *dp;
    uint32_t* dout;
    CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMalloc(&dp, hp.size() * 2));
    CK(hipMalloc(&dout, (size_t)Mp * ld * 4));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xFF, (size_t)Mp * ld * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float ms = 0;
    const int lab = argc > 6 ? atoi(argv[6]) : 0;
    int rc = nbuf == 642 ? run_kernel(lab == 5 ? dense_scores_256k<5, 2> : dense_scores_256k<0, 2>, 2 * kGkStage, dq, dp, dout, M, N, H, ld, reps, st, &ms)
             : nbuf == 64 ? run_kernel(lab == 3 ? dense_scores_256k<3> : lab == 5 ? dense_scores_256k<5> : lab == 6 ? dense_scores_256k<6> : lab == 7 ? dense_scores_256k<7> : dense_scores_256k<0>, 2 * kGkStage, dq, dp, dout, M, N, H, ld, reps, st, &ms)
             : nbuf == 16 ? run_kernel(lab == 3 ? dense_scores_256r16<3> : dense_scores_256r16<0>, 2 * kGwStage, dq, dp, dout, M, N, H, ld, reps, st, &ms)
             : nbuf == 2 ? run_kernel(lab == 1 ? dense_scores_256r<1> : lab == 3 ? dense_scores_256r<3> : lab == 4 ? dense_scores_256r<4> : dense_scores_256r<0>, 2 * kGwStage, dq, dp, dout, M, N, H, ld, reps, st, &ms)
             : lab == 3 ? run<4, 3>(dq, dp, dout, M, N, H, ld, reps, st, &ms)
             : lab == 1 ? run<4, 1>(dq, dp, dout, M, N, H, ld, reps, st, &ms) : nbuf == 5 ? run<5>(dq, dp, dout, M, N, H, ld, reps, st, &ms)
                       : (nbuf == 3 ? run<3>(dq, dp, dout, M, N, H, ld, reps, st, &ms) : run<4>(dq, dp, dout, M, N, H, ld, reps, st, &ms));
    if (rc) return rc;
    printf("dense_scores_256w<%d, lab %d> %u x %u x %u: %.4f ms  %.1f TFLOP/s\n", nbuf, lab, M, N, H, ms, 2.0 * M * N * H / (ms * 1e-3) / 1e12);
    if (lab >= 5) {  // per-phase laps of the 64-deep kernel: cycles per step in half 0, wait + barrier, half 1
        std::vector<uint64_t> stamps(8 * 4096);
        CK(hipMemcpyFromSymbol(stamps.data(), HIP_SYMBOL(g_gemm_lab_stamps), stamps.size() * 8));
        const uint32_t nb = std::min<uint32_t>(4096, ((M + 255) / 256) * ((N + 255) / 256));
        std::vector<double> l0, l1, l2;
        for (uint32_t b = 0; b < nb; ++b)
            if (stamps[8 * b]) l0.push_back(stamps[8 * b] / (H / 64.0)), l1.push_back(stamps[8 * b + 1] / (H / 64.0)), l2.push_back(stamps[8 * b + 2] / (H / 64.0));
        std::sort(l0.begin(), l0.end()), std::sort(l1.begin(), l1.end()), std::sort(l2.begin(), l2.end());
        if (!l0.empty())
            printf("laps per 64-deep step (median cycles): before the barrier %.0f, wait + barrier %.0f, after %.0f (2048 in all = MFMA-bound)\n", l0[l0.size() / 2],
                   l1[l1.size() / 2], l2[l2.size() / 2]);
    }
    if (lab == 3 || lab == 4) {  // stamps of the last launch: K-loop cycles per 32-deep sub-step, in-kernel clock, block lives
        std::vector<uint64_t> stamps(8 * 4096);
        CK(hipMemcpyFromSymbol(stamps.data(), HIP_SYMBOL(g_gemm_lab_stamps), stamps.size() * 8));
        const uint32_t nb = std::min<uint32_t>(4096, ((M + 255) / 256) * ((N + 255) / 256));
        std::vector<double> cyc, clk, life, loop_us, pro, epi, drain;
        uint64_t first = ~0ull, last = 0;
        double busy = 0;
        for (uint32_t b = 0; b < nb; ++b)
            if (stamps[8 * b + 1]) {
                cyc.push_back((double)stamps[8 * b] / (H / 32)), clk.push_back((double)stamps[8 * b] / stamps[8 * b + 1] * 0.1);
                loop_us.push_back(stamps[8 * b + 1] * 0.01);
                if (stamps[8 * b + 3]) {
                    life.push_back((stamps[8 * b + 3] - stamps[8 * b + 2]) * 0.01);
                    busy += life.back();
                    pro.push_back((stamps[8 * b + 4] - stamps[8 * b + 2]) * 0.01), epi.push_back((stamps[8 * b + 6] - stamps[8 * b + 5]) * 0.01),
                        drain.push_back((stamps[8 * b + 3] - stamps[8 * b + 6]) * 0.01);
                    first = std::min(first, stamps[8 * b + 2]), last = std::max(last, stamps[8 * b + 3]);
                }
            }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        std::sort(loop_us.begin(), loop_us.end());
        if (!cyc.empty())
            printf("stamps: %zu blocks; cycles per sub-step p10/median/p90 %.0f / %.0f / %.0f (1024 = MFMA-bound); clock median %.3f GHz; "
                   "K loop median %.1f us\n",
                   cyc.size(), cyc[cyc.size() / 10], cyc[cyc.size() / 2], cyc[cyc.size() * 9 / 10], clk[clk.size() / 2],
                   loop_us[loop_us.size() / 2]);
        if (!life.empty()) {
            std::sort(life.begin(), life.end());
            std::sort(pro.begin(), pro.end()), std::sort(epi.begin(), epi.end()), std::sort(drain.begin(), drain.end());
            printf("medians: prologue %.1f us, epilogue until stores issued %.1f us, store drain %.1f us\n", pro[pro.size() / 2], epi[epi.size() / 2],
                   drain[drain.size() / 2]);
            printf("block lives: median %.1f us, p90 %.1f us; first start -> last end %.1f us; sum of lives / 256 CUs %.1f us\n",
                   life[life.size() / 2], life[life.size() * 9 / 10], (last - first) * 0.01, busy / 256);
            // how many blocks are alive at 20 instants of the launch
            printf("alive:");
            for (int t = 0; t < 20; ++t) {
                const uint64_t at = first + (last - first) * (2 * t + 1) / 40;
                int n = 0;
                for (uint32_t b = 0; b < nb; ++b) n += stamps[8 * b + 3] && stamps[8 * b + 2] <= at && at < stamps[8 * b + 3];
                printf(" %d", n);
            }
            printf("\n");
        }
    }
    // sample check: 4096 random (q, d) pairs + the corners
    const uint32_t ns = 4096;
    std::vector<uint32_t> qs(ns), ds(ns);
    for (uint32_t i = 0; i < ns; ++i) {
        qs[i] = g() % M;
        ds[i] = g() % N;
    }
    qs[0] = 0, ds[0] = 0, qs[1] = M - 1, ds[1] = N - 1, qs[2] = 0, ds[2] = N - 1, qs[3] = M - 1, ds[3] = 0;
    uint32_t *dqs, *dds;
    float* dref;
    CK(hipMalloc(&dqs, ns * 4));
    CK(hipMalloc(&dds, ns * 4));
    CK(hipMalloc(&dref, ns * 4));
    CK(hipMemcpy(dqs, qs.data(), ns * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dds, ds.data(), ns * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_dots, dim3((ns + 255) / 256), dim3(256), 0, st, dq, dp, dqs, dds, dref, ns, H);
    std::vector<float> ref(ns);
    CK(hipMemcpyAsync(ref.data(), dref, ns * 4, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    double worst = 0;
    int bad = 0;
    for (uint32_t i = 0; i < ns; ++i) {
        float got;
        CK(hipMemcpy(&got, dout + (uint64_t)qs[i] * ld + ds[i], 4, hipMemcpyDeviceToHost));
        const double d = fabs((double)got - ref[i]);
        if (d > worst) worst = d;
        if (!(d <= 2e-4)) ++bad;
    }
    // padding columns of a real row are 0, rows beyond M untouched (0xFFFFFFFF)
    uint32_t padv = 1, beyond = 0;
    if (Np > N) CK(hipMemcpy(&padv, dout + (uint64_t)0 * ld + N, 4, hipMemcpyDeviceToHost));
    else padv = 0;
    if (Mp > M) CK(hipMemcpy(&beyond, dout + (uint64_t)M * ld, 4, hipMemcpyDeviceToHost));
    else beyond = 0xFFFFFFFFu;
    printf("check: %d of %u samples off by > 2e-4 (worst %.3g); pad column %u (want 0); row beyond M %08x (want ffffffff)\n", bad,
           ns, worst, padv, beyond);
    return bad || padv || beyond != 0xFFFFFFFFu;
}
