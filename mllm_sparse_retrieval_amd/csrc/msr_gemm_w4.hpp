// dense_scores_256k — the production 256 x 256 GEMM block of the hybrid path (Q . P^T, fp16 in, f32 accumulate) on FOUR
// waves, one per SIMD, each 128 x 128 = 8 x 8 tiles of v_mfma_f32_16x16x32_f16; see the kernel's own comment below and
// DESIGN.md section 5. The experimental kernels it grew out of (LDS-DMA fill, 32 x 32 x 16 MFMAs, 32-deep steps) are
// in scripts/gemm_lab_kernels.hpp, built only by the lab harness scripts/gemm_lab.hip.
#pragma once

#include <type_traits>

#include "msr_select.hpp"

namespace msr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1 (the index is a constant expression in the body,
// whatever the unroller would have made of a plain loop)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

typedef float float4v __attribute__((ext_vector_type(4)));

// LAB != 0 instantiations (scripts/gemm_lab.hip only, which defines MSR_GEMM_LAB) stamp the K loop of every block:
// shader cycles and 100 MHz ticks of the loop, the block's life (start, loop start, loop end, stores issued, stores done)
#ifdef MSR_GEMM_LAB
__device__ uint64_t g_gemm_lab_stamps[8 * 4096];
#else
extern __device__ uint64_t g_gemm_lab_stamps[];  // never referenced by the LAB == 0 instantiations the library builds
#endif

// XCD-aware patch-major block order shared by the 256 x 256 kernels: workgroup L runs on XCD L % 8 as that XCD's
// (L / 8)-th block; every XCD walks a contiguous eighth of the order (patches of 8 query blocks x 4 doc blocks)
__device__ __forceinline__ bool patch_major_block(uint32_t L, uint32_t qb_n, uint32_t db_n, uint32_t& qb, uint32_t& db) {
    const uint32_t n_blk = qb_n * db_n, per = (n_blk + 7u) / 8u;
    const uint32_t xcd = L & 7u, slot = L >> 3;
    const uint32_t t = xcd * per + slot;
    if (slot >= per || t >= n_blk) return false;
    const uint32_t pr = t / (8u * db_n), rem = t - pr * 8u * db_n;
    const uint32_t hp = min(8u, qb_n - 8u * pr);
    const uint32_t pc = rem / (hp * 4u), rem2 = rem - pc * hp * 4u;
    qb = pr * 8u + rem2 % hp;
    db = pc * 4u + rem2 / hp;
    return true;
}

// dense_scores_256k: a 256 x 256 block on four waves (2 x 2), each 128 x 128 = 8 x 8 tiles of v_mfma_f32_16x16x32_f16
// (fragment map: lane (c = l & 15, g = l >> 4) holds k = 8g .. 8g+7 of row c of A / of column c of B; C/D: column l & 15,
// rows 4 (l >> 4) + reg). The 256 accumulator registers live in the AGPR half of the unified file (launch bound 256
// threads, one workgroup per CU: 512 registers per lane).
//   * 16 x 16 x 32 rather than 32 x 32 x 16: the same FLOPs per cycle, but the chip holds a higher clock under it
//     (MI355X_MICROARCH.md, DVFS give-back (7); +7 % measured in this loop).
//   * K in steps of 64: every global load fetches FULL 128-byte lines (eight lanes per row; 32-deep steps fetch every
//     line twice, 64 B at a time: twice the TA / L2 requests). ONE staging set of 16 x 16 B per lane holds a step.
//   * Two 64-KiB LDS buffers hold a step each: rows of 128 B, slot swizzle slot ^= (row >> 1) & 7 — conflict-free for
//     the four 16-lane groups of ds_read_b128 under this fragment map (found by enumeration) and for the 8-lane row writes.
//   * One barrier per step, after MFMA 39 of 128:
//       MFMAs 0-63 run on the k-half-0 fragments, 64-127 on the k-half-1 fragments of step S;
//       MFMAs 0-31   : the 16 reads of fragments(S, k-half 1) from buffer S & 1, one per second MFMA
//       after MFMA 39: lgkmcnt(0) + barrier — every wave is done with buffer S & 1, step S + 1 is complete in the other
//       MFMAs 40-119 : the 16 staged stores of step S + 2 into buffer S & 1, one per fifth MFMA (a ds_write_b128 costs
//                      its wave ~20 cycles of issue whatever the spacing), each followed by the load that refills its
//                      registers with step S + 3
//       MFMAs 64-95  : the 16 reads of fragments(S + 1, k-half 0) from the other buffer
//   * The MFMAs are inline asm with the accumulator tied in place: left to the register allocator, 64 four-register
//     accumulators end in a storm of v_accvgpr copies and scratch spills. Inline asm is invisible to the
//     sched_group_barrier classes, so the issue order is pinned with sched_barrier(0) after every MFMA, and the
//     MFMA -> accvgpr-read hazard after the loop is covered by hand.
//   * The scores leave through LDS (four passes of 64 query rows x 1 KiB), so that every global store is a full 1-KiB
//     row piece instead of 16 lanes x 4 B of the C/D layout (17 us per block instead of 8).
constexpr int kGkStage = 2 * 256 * 128;  // bytes per 64-deep step buffer: A rows then B rows

constexpr int kGkBarrierAt = 39;   // the step's barrier follows this MFMA (the 16 fragment reads of buffer `cur` end at MFMA 31)
constexpr int kGkStoreEvery = 5;   // 16 staged stores, one per this many MFMAs, after the barrier (16 x 5 = 80 <= 88)

template <int LAB = 0, int ADEPTH = 1>  // ADEPTH 2: a second staging set for the A operand (loads two steps ahead): measured no gain
__global__ __launch_bounds__(256, 1) void dense_scores_256k(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                            uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                            uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 buffers of kGkStage
    uint64_t life0 = 0;
    if constexpr (LAB == 3) life0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const uint32_t c = lane & 15, g = lane >> 4;
    uint32_t qb, db;
    if (!patch_major_block(blockIdx.x, qb_n, db_n, qb, db)) return;
    const uint32_t q_blk = qb * 256, d_blk = db * 256;
    if (d_blk >= N) {  // padding docs: keys 0
        for (uint32_t i = tid; i < 256 * 64; i += 256) {
            const uint32_t q = q_blk + i / 64;
            if (q < M) reinterpret_cast<uint4*>(out + (uint64_t)q * ld + d_blk)[i % 64] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const uint32_t wr = wave >> 1, wm = wr * 128, wn = (wave & 1) * 128;
    float4v acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    // staging plan: chunk tid + 256 i (i = 0..7) of an operand's step = 16 B: row (tid >> 3) + 32 i, segment tid & 7
    // (eight lanes fetch one 128-byte line); LDS slot = segment ^ ((row >> 1) & 7), the same for every i
    const char* const qbase = reinterpret_cast<const char*>(Q + (uint64_t)q_blk * H);
    const char* const pbase = reinterpret_cast<const char*>(P + (uint64_t)d_blk * H);
    const uint32_t srow = tid >> 3, sseg = tid & 7;
    const uint32_t goff = srow * H * 2 + sseg * 16;
    const uint32_t loff = srow * 128 + (sseg ^ ((srow >> 1) & 7)) * 16;
    const uint64_t gstep = (uint64_t)32 * H * 2;  // 32 rows further
    const uint32_t KS = H / 64;
    u32x4 ra[8], rb[8], ra2[ADEPTH == 2 ? 8 : 1];
    auto gload_all = [&](uint32_t S) {
        const uint64_t k0 = (uint64_t)min(S, KS - 1) * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(qbase + k0 + i * gstep + goff);
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pbase + k0 + i * gstep + goff);  // (the loop's order)
    };
    auto lstore_all = [&](uint32_t stage) {
        uint8_t* const st = smem + stage * kGkStage;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<u32x4*>(st + loff + i * 32 * 128) = ra[i];
            *reinterpret_cast<u32x4*>(st + 256 * 128 + loff + i * 32 * 128) = rb[i];
        }
    };
    // fragment addresses: row (w? + 16 i + c) -> swizzle (c >> 1) & 7; k-half h reads slot (4 h + g) ^ swizzle
    const uint32_t fs0 = (g ^ ((c >> 1) & 7)) * 16, fs1 = fs0 ^ 64;
    const uint32_t fa = (wm + c) * 128, fb = 256 * 128 + (wn + c) * 128;
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };
    half8 xa[8], xb[8], ya[8], yb[8];
    // ---- prologue: steps 0 and 1 into the LDS buffers, step 2 into the staging registers
    {   // (steps 0 and 1 are requested together — one memory round trip, not two; the extra registers die here)
        u32x4 ta[8], tb[8];
        const uint64_t k1 = (uint64_t)min(1u, KS - 1) * 128;
        gload_all(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) ta[i] = *reinterpret_cast<const u32x4*>(qbase + k1 + i * gstep + goff);
#pragma unroll
        for (int i = 0; i < 8; ++i) tb[i] = *reinterpret_cast<const u32x4*>(pbase + k1 + i * gstep + goff);
        lstore_all(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<u32x4*>(smem + kGkStage + loff + i * 32 * 128) = ta[i];
            *reinterpret_cast<u32x4*>(smem + kGkStage + 256 * 128 + loff + i * 32 * 128) = tb[i];
        }
    }
    if constexpr (ADEPTH == 2) {  // (the loop's order of requests: A of step 2, A of step 3, B of step 2)
        const uint64_t k2 = (uint64_t)min(2u, KS - 1) * 128, k3 = (uint64_t)min(3u, KS - 1) * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(qbase + k2 + i * gstep + goff);
#pragma unroll
        for (int i = 0; i < 8; ++i) ra2[i] = *reinterpret_cast<const u32x4*>(qbase + k3 + i * gstep + goff);
#pragma unroll
        for (int i = 0; i < 8; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pbase + k2 + i * gstep + goff);
    } else {
        gload_all(2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xa[i] = *reinterpret_cast<const half8*>(smem + fa + i * 16 * 128 + fs0);
        xb[i] = *reinterpret_cast<const half8*>(smem + fb + i * 16 * 128 + fs0);
    }
    uint64_t t0 = 0, r0 = 0;
    if constexpr (LAB == 3) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t lap0 = 0, lap1 = 0, lap2 = 0;
    // one 64-deep step; raS holds the A operand of step S + 2 and is refilled with step S + 1 + a_ahead
    auto do_step = [&](uint32_t S, u32x4 (&raS)[8]) {
        constexpr uint32_t a_ahead = ADEPTH == 2 ? 4 : 3;
        uint64_t s0 = 0, s1 = 0, s2 = 0;
        if constexpr (LAB >= 5) s0 = __builtin_amdgcn_s_memtime();
        const uint8_t* const cur = smem + (S & 1) * kGkStage;
        const uint8_t* const oth = smem + ((S & 1) ^ 1) * kGkStage;
        uint8_t* const curw = smem + (S & 1) * kGkStage;
        const uint64_t k0 = (uint64_t)min(S + 3, KS - 1) * 128, k0a = (uint64_t)min(S + a_ahead, KS - 1) * 128;
        // 128 MFMAs: 0-63 on the k-half-0 fragments (xa, xb), 64-127 on the k-half-1 fragments (ya, yb)
        static_for<0, 128>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int i = (m & 63) >> 3, j = m & 7;
            if constexpr (m < 64) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(xa[i]), "v"(xb[j]));
            else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(ya[i]), "v"(yb[j]));
            if constexpr (m < 32 && (m & 1)) {  // the k-half-1 fragments of this step
                constexpr int q = m >> 1;
                if constexpr (q < 8) yb[q] = *reinterpret_cast<const half8*>(cur + fb + q * 16 * 128 + fs1);
                else ya[q - 8] = *reinterpret_cast<const half8*>(cur + fa + (q - 8) * 16 * 128 + fs1);
            }
            if constexpr (m == kGkBarrierAt) {  // every wave has read all it needs from buffer `cur`; step S + 1 is complete in `oth`
                if constexpr (LAB >= 5) s1 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
                if constexpr (LAB >= 5) s2 = __builtin_amdgcn_s_memtime();
            }
            if constexpr (m > kGkBarrierAt && (m - kGkBarrierAt - 1) % kGkStoreEvery == kGkStoreEvery / 2 &&
                          (m - kGkBarrierAt - 1) / kGkStoreEvery < 16) {
                // a staged store of step S + 2 every kGkStoreEvery-th MFMA (the LDS store path moves ~79 B/clk for the
                // whole CU: packed densely the stores stall their waves), each followed by the load that refills its
                // registers with step S + 3
                constexpr int q = (m - kGkBarrierAt - 1) / kGkStoreEvery;
                if constexpr (q < 8) {
                    if constexpr (LAB != 7) *reinterpret_cast<u32x4*>(curw + loff + q * 32 * 128) = raS[q];
                    if constexpr (LAB != 6) raS[q] = *reinterpret_cast<const u32x4*>(qbase + k0a + q * gstep + goff);
                } else {
                    if constexpr (LAB != 7) *reinterpret_cast<u32x4*>(curw + 256 * 128 + loff + (q - 8) * 32 * 128) = rb[q - 8];
                    if constexpr (LAB != 6) rb[q - 8] = *reinterpret_cast<const u32x4*>(pbase + k0 + (q - 8) * gstep + goff);
                }
            }
            if constexpr (m >= 64 && m < 96 && (m & 1)) {  // the k-half-0 fragments of the next step
                constexpr int q = (m - 64) >> 1;
                if constexpr (q < 8) xb[q] = *reinterpret_cast<const half8*>(oth + fb + q * 16 * 128 + fs0);
                else xa[q - 8] = *reinterpret_cast<const half8*>(oth + fa + (q - 8) * 16 * 128 + fs0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (LAB >= 5) lap0 += s1 - s0, lap1 += s2 - s1, lap2 += __builtin_amdgcn_s_memtime() - s2;
    };
    if constexpr (ADEPTH == 2) {
#pragma nounroll
        for (uint32_t S = 0; S + 1 < KS; S += 2) {
            do_step(S, ra);
            do_step(S + 1, ra2);
        }
        if (KS & 1) do_step(KS - 1, ra);
    } else {
#pragma nounroll
        for (uint32_t S = 0; S < KS; ++S) do_step(S, ra);
    }
    if constexpr (LAB >= 5)
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x] = lap0, g_gemm_lab_stamps[8 * blockIdx.x + 1] = lap1, g_gemm_lab_stamps[8 * blockIdx.x + 2] = lap2;
    if constexpr (LAB == 3) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x] = t1 - t0, g_gemm_lab_stamps[8 * blockIdx.x + 1] = r1 - r0,
            g_gemm_lab_stamps[8 * blockIdx.x + 4] = r0, g_gemm_lab_stamps[8 * blockIdx.x + 5] = r1;
    }
    // The compiler does not know the asm statements are MFMAs: their write -> accvgpr-read hazard is covered by hand.
    // The empty asms tie every accumulator to a point AFTER the nops (volatile asms keep their order), so that no read
    // of a result can be scheduled above them.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    // ---- epilogue through LDS: pass t carries the tile rows i = 2t, 2t + 1 of every wave:
    // 64 query rows (2 wave rows x 32) x 256 docs = 64 KiB, alternating between the two step buffers
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bar();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float* const stg = reinterpret_cast<float*>(smem + (t & 1) * kGkStage);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) stg[(wr * 32 + 16 * ii + 4 * g + e) * 256 + wn + 16 * j + c] = acc[2 * t + ii][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        u32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const u32x4*>(stg + (wave * 16 + k) * 256 + lane * 4);
        const uint32_t d = d_blk + lane * 4;
        const uint32_t keep0 = d + 0 < N ? ~0u : 0u, keep1 = d + 1 < N ? ~0u : 0u, keep2 = d + 2 < N ? ~0u : 0u,
                       keep3 = d + 3 < N ? ~0u : 0u;
        const uint32_t as_key = raw ? 0u : ~0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t row = wave * 16 + k;
            const uint32_t q = q_blk + (row >> 5) * 128 + 32 * t + (row & 31);
            u32x4 o = v[k];
#pragma unroll
            for (int x = 0; x < 4; ++x) o[x] ^= (((uint32_t)((int32_t)o[x] >> 31)) | 0x80000000u) & as_key;
            o[0] &= keep0, o[1] &= keep1, o[2] &= keep2, o[3] &= keep3;
            if (q < M) *reinterpret_cast<u32x4*>(out + (uint64_t)q * ld + d) = o;
        }
    }
    if constexpr (LAB == 3) {
        const uint64_t issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x + 2] = life0, g_gemm_lab_stamps[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime(),
            g_gemm_lab_stamps[8 * blockIdx.x + 6] = issued;
    }
}

}  // namespace msr
