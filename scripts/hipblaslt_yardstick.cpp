// Vendor yardstick for the config-5 dense GEMM: the same shape (scores[nq][n] = Q[nq][H] . P[n][H]^T, fp16 in, f32 out)
// through hipBLASLt's heuristic-picked kernels, timed with HIP events on random data.  Diagnostic only: nothing in
// libmsr links hipBLASLt.  Build: hipcc --offload-arch=gfx950 -O2 scripts/hipblaslt_yardstick.cpp -lhipblaslt -o ...
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        auto e_ = (x);                                                    \
        if ((int)e_ != 0) {                                               \
            fprintf(stderr, "%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main(int argc, char** argv) {
    const int64_t nq = argc > 1 ? atoll(argv[1]) : 25010, n = argc > 2 ? atoll(argv[2]) : 5000, H = argc > 3 ? atoll(argv[3]) : 4096;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const hipDataType out_t = (argc > 5 && atoi(argv[5]) == 16) ? HIP_R_16F : HIP_R_32F;
    std::vector<__half> hq((size_t)nq * H), hp((size_t)n * H);
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& x : hq) x = __float2half(nd(g) * 0.05f);
    for (auto& x : hp) x = __float2half(nd(g) * 0.05f);
    __half *dq, *dp;
    void* dd;
    const size_t out_b = out_t == HIP_R_32F ? 4 : 2;
    CK(hipMalloc(&dq, hq.size() * 2));
    CK(hipMalloc(&dp, hp.size() * 2));
    CK(hipMalloc(&dd, (size_t)nq * n * out_b));
    CK(hipMemcpy(dq, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
    hipblasLtHandle_t lt;
    CK(hipblasLtCreate(&lt));
    hipblasLtMatmulDesc_t md;
    CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t tA = HIPBLAS_OP_T, tB = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &tA, sizeof(tA)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &tB, sizeof(tB)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, H, n, H));    // P as col-major H x n, used transposed
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, H, nq, H));   // Q as col-major H x nq
    CK(hipblasLtMatrixLayoutCreate(&lc, out_t, n, nq, n));       // scores^T col-major n x nq == scores row-major
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    size_t ws = 256u << 20;
    void* dws;
    CK(hipMalloc(&dws, ws));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws)));
    hipblasLtMatmulHeuristicResult_t res[8];
    int found = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(lt, md, la, lb, lc, lc, pref, 8, res, &found));
    printf("shape nq=%lld n=%lld H=%lld out=%s: %d algos\n", (long long)nq, (long long)n, (long long)H,
           out_t == HIP_R_32F ? "f32" : "f16", found);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const float alpha = 1.f, beta = 0.f;
    double best = 0;
    for (int a = 0; a < found; ++a) {
        for (int w = 0; w < 3; ++w)
            CK(hipblasLtMatmul(lt, md, &alpha, dp, la, dq, lb, &beta, dd, lc, dd, lc, &res[a].algo, dws, ws, st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r)
            CK(hipblasLtMatmul(lt, md, &alpha, dp, la, dq, lb, &beta, dd, lc, dd, lc, &res[a].algo, dws, ws, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double tf = 2.0 * nq * n * H / (ms * 1e-3) / 1e12;
        printf("  algo %d: %.4f ms  %.1f TFLOP/s  (workspace %zu)\n", a, ms, tf, res[a].workspaceSize);
        if (tf > best) best = tf;
    }
    printf("best %.1f TFLOP/s\n", best);
    return 0;
}
