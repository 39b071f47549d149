// Experimental GEMM kernels of the hybrid path, kept for scripts/gemm_lab.hip (nothing in libmsr.so includes this file):
// the steps between the 8-wave ping-pong kernel (csrc/msr_hybrid.hip: dense_scores_256p) and the production kernel
// (csrc/msr_gemm_w4.hpp: dense_scores_256k). What each one measured is in DESIGN.md section 5.
//
// dense_scores_256w: the 256 x 256 block of Q . P^T on FOUR waves, one per SIMD, each 128 x 128 = 4 x 4 MFMA tiles of
// 32 x 32 (v_mfma_f32_32x32x16_f16), software-pipelined inside the single wave instead of ping-pong between two.
//
// Why: per 32-deep K sub-step a 128 x 64 wave tile (dense_scores_256p) reads 12 fragments of 1 KiB for 16 MFMAs, a
// 128 x 128 tile 16 for 32 — the LDS read pipe goes from ~75 % to ~50 % of the MFMA time, there are half as many waves
// at every barrier, and the vendor library's best kernels for this very shape (hipBLASLt heuristics on 25 010 x 5 000 x
// 4 096: MT256x256x64, MIWT8_8 = 128 x 128 per wave, 256 threads; scripts/hipblaslt_yardstick.cpp: 1 200 TFLOP/s) use
// the same wave tile. The 256 accumulator registers live in the AGPR half of the unified file (launch bound 256
// threads, one workgroup per CU: 512 registers per lane).
//
// Pipeline (NBUF LDS buffers of 32 KiB, A rows then B rows, 64 B per row, filled by LDS-DMA as in dense_scores_256p,
// same source-side bank swizzle): at the top of sub-step p the wave holds fragments(p) in registers; it waits for its
// own DMA pieces of sub-step p + 1 (counted vmcnt: the NBUF - 2 newer sub-steps stay in flight) and for its fragment
// reads (lgkmcnt(0)), passes ONE barrier — now every wave's pieces of p + 1 have landed and buffer p % NBUF has been
// read by everyone — and then issues, interleaved by the scheduler hints below: 32 MFMAs on fragments(p), the 16
// fragment reads of p + 1 into the other register set, and the 8 DMAs of sub-step p + NBUF into buffer p % NBUF.
// The DMAs are issued unconditionally (past the end they re-fetch the last sub-step into a buffer nobody reads), so
// the vmcnt bookkeeping is one constant and the loop body is one basic block.
#pragma once

#include "../mllm_sparse_retrieval_amd/csrc/msr_gemm_w4.hpp"

namespace msr {

constexpr int kGwStage = 2 * 256 * 64;  // bytes per sub-step buffer

// Epilogue of the 2 x 2-wave kernels: the block's scores leave through LDS so that every global store is one full
// 1-KiB row piece (64 lanes x 16 B = a query's 256 docs) instead of 32 lanes x 4 B. Measured on the w kernel with the
// direct stores of the C/D layout (256 dword stores per wave, two 128-B pieces each): the K loops of a block's life
// take 75 % of the kernel time, the rest is mostly this tail (store-ISSUE-bound: ~4.5 B/cycle/CU).
// Four passes; pass i carries every wave's tiles acc[i][*]: 64 query rows (2 wave rows x 32) x 256 docs x 4 B = 64 KiB
// of LDS, two alternating regions when the kernel owns 128 KiB.
// C/D map of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
template <int LDS_BYTES>
__device__ __forceinline__ void store_block_via_lds(const float16v (&acc)[4][4], uint8_t* smem, uint32_t* __restrict__ out,
                                                    uint32_t M, uint32_t N, uint64_t ld, uint32_t q_blk, uint32_t d_blk,
                                                    uint32_t wave, uint32_t lane, uint32_t raw) {
    static_assert(LDS_BYTES >= 64 * 1024, "one 64-KiB staging region at least");
    constexpr bool kTwo = LDS_BYTES >= 128 * 1024;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t wr = wave >> 1, wn = (wave & 1) * 128;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");  // the K loop's buffers are dead for every wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* const stg = reinterpret_cast<float*>(smem + (kTwo ? (i & 1) * 64 * 1024 : 0));
        if (!kTwo && i) asm volatile("s_barrier" ::: "memory");  // the previous pass has been read out
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stg[(wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 256 + wn + 32 * j + r] = acc[i][j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        // wave w stores rows 16 w .. 16 w + 15 of the 64: row = (wave row) * 32 + (row in the 32 x 32 tile); all reads
        // first (native vector type: one ds_read_b128 each), then branch-free conversion and one store per row
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
            const uint32_t q = q_blk + (row >> 5) * 128 + 32 * i + (row & 31);
            u32x4 o = v[k];
            // f32_to_key without a branch: negative floats flip all bits, the others the sign bit
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] ^= (((uint32_t)((int32_t)o[c] >> 31)) | 0x80000000u) & as_key;
            o[0] &= keep0, o[1] &= keep1, o[2] &= keep2, o[3] &= keep3;
            if (q < M) *reinterpret_cast<u32x4*>(out + (uint64_t)q * ld + d) = o;
        }
    }
}

template <int NBUF, int LAB = 0>  // LAB 1: no DMA inside the loop (a timing experiment: wrong results); 3: stamps
__global__ __launch_bounds__(256, 1) void dense_scores_256w(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                            uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                            uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    static_assert(NBUF >= 3 && NBUF <= 5, "three to five 32-KiB buffers");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
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
    const uint32_t wm = (wave >> 1) * 128, wn = (wave & 1) * 128;
    float16v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // DMA plan: piece c (1 KiB = 16 rows x 64 B) of an operand's sub-step; wave w issues pieces w, w + 4, w + 8, w + 12
    const char* const qbase = reinterpret_cast<const char*>(Q + (uint64_t)q_blk * H);
    const char* const pbase = reinterpret_cast<const char*>(P + (uint64_t)d_blk * H);
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 16 * (wave + 4 * i) + (lane >> 2);
        const uint32_t seg = (lane & 3) ^ ((row >> 2) & 3);
        voff[i] = row * H * 2 + seg * 16;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const uint32_t KP = H / 32;
    auto issue = [&](uint32_t stage, uint32_t p) {
        const uint64_t k0 = (uint64_t)min(p, KP - 1) * 64;  // bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint8_t* const da = smem + stage * kGwStage + (wave + 4 * i) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(qbase + k0 + voff[i]), (lds_void*)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(pbase + k0 + voff[i]), (lds_void*)(da + 256 * 64), 16, 0, 0);
        }
    };
    const uint32_t f = (r >> 2) & 3;  // swizzle of this lane's fragment rows: ((w? + 32 i + r) >> 2) & 3 == (r >> 2) & 3
    const uint32_t fa = (wm + r) * 64, fb = 256 * 64 + (wn + r) * 64;
    const uint32_t slot0 = ((0 + h) ^ f) * 16, slot1 = ((2 + h) ^ f) * 16;
    auto read_frags = [&](half8 (&a)[2][4], half8 (&b)[2][4], uint32_t stage) {
        const uint8_t* const st = smem + stage * kGwStage;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 64 + slot0);
            b[0][i] = *reinterpret_cast<const half8*>(st + fb + i * 32 * 64 + slot0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[1][i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 64 + slot1);
            b[1][i] = *reinterpret_cast<const half8*>(st + fb + i * 32 * 64 + slot1);
        }
    };
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };
    // one sub-step: `cur` holds fragments(p); fragments(p + 1) go to `nxt`
    auto step = [&](uint32_t p, uint32_t stage, uint32_t stage_next, half8 (&ca)[2][4], half8 (&cb)[2][4], half8 (&na)[2][4],
                    half8 (&nb)[2][4]) {
        // own pieces of sub-step p + 1 have landed; the NBUF - 2 sub-steps after it stay in flight (8 DMAs each)
        if constexpr (LAB == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if constexpr (NBUF == 5) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
        else if constexpr (NBUF == 4) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        bar();
        read_frags(na, nb, stage_next);
        if constexpr (LAB != 1) issue(stage, p + NBUF);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca[kk][i], cb[kk][j], acc[i][j], 0, 0, 0);
        // issue order inside the sub-step: the first 16 MFMAs each followed by one fragment read of the NEXT sub-step
        // (they come back ~16 MFMAs before their first use), then 2 MFMAs per DMA (the DMA stores to LDS, so it stays
        // behind the fragment reads in program order anyway)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (the LDS-DMA)
        }
    };
    half8 a0[2][4], b0[2][4], a1[2][4], b1[2][4];
#pragma unroll
    for (int s = 0; s < NBUF; ++s) issue(s, s);
    // sub-step 0 has landed (own pieces; NBUF - 1 newer sub-steps in flight), for every wave after the barrier
    if constexpr (NBUF == 5) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    if constexpr (NBUF == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    if constexpr (NBUF == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    bar();
    read_frags(a0, b0, 0);
    uint32_t st = 0;  // buffer of sub-step p
    uint64_t t0 = 0, r0 = 0;
    if constexpr (LAB == 3) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t p = 0; p < KP; p += 2) {  // KP is even (H % 64 == 0)
        const uint32_t s1 = st + 1 == NBUF ? 0 : st + 1;
        const uint32_t s2 = s1 + 1 == NBUF ? 0 : s1 + 1;
        step(p, st, s1, a0, b0, a1, b1);
        step(p + 1, s1, s2, a1, b1, a0, b0);
        st = s2;
    }
    if constexpr (LAB == 3) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096) g_gemm_lab_stamps[8 * blockIdx.x] = t1 - t0, g_gemm_lab_stamps[8 * blockIdx.x + 1] = r1 - r0;
    }
    // (the over-issued DMAs are drained by the epilogue's vmcnt(0) before the LDS is reused)
    store_block_via_lds<NBUF * kGwStage>(acc, smem, out, M, N, ld, q_blk, d_blk, wave, lane, raw);
}

// dense_scores_256r: the same four-wave 128 x 128 tiling with the operands staged through REGISTERS
// (global_load_dwordx4 -> VGPRs -> ds_write_b128) instead of LDS-DMA. Measured on the w kernel: without its DMAs the
// loop runs at 1 166 TFLOP/s, with them at 890 — an LDS-DMA piece costs the issuing wave 60-185 cycles of issue
// (MI355X_MICROARCH.md), and with one wave per SIMD there is no partner wave to hide that behind. A plain load and a
// ds_write_b128 are ordinary short issues. Two register sets of 8 x 16 B per lane hold sub-steps p + 2 and p + 3 while
// sub-step p is multiplied; TWO LDS buffers suffice (sub-step p + 2 is written into the buffer whose fragments(p) are
// already in registers). Same 64-B rows and slot swizzle as the DMA kernels.
template <int LAB = 0>
__global__ __launch_bounds__(256, 1) void dense_scores_256r(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                            uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                            uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 buffers of kGwStage
    uint64_t life0 = 0;
    if constexpr (LAB == 3 || LAB == 4) life0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
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
    const uint32_t wm = (wave >> 1) * 128, wn = (wave & 1) * 128;
    float16v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // staging plan: chunk c = tid + 256 i (i = 0..3) of an operand's sub-step = 16 B: row c >> 2, segment c & 3; the
    // chunk goes to LDS slot seg ^ ((row >> 2) & 3) of its row (the fragment reads undo the same XOR)
    const char* const qbase = reinterpret_cast<const char*>(Q + (uint64_t)q_blk * H);
    const char* const pbase = reinterpret_cast<const char*>(P + (uint64_t)d_blk * H);
    uint32_t goff[4], loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = (tid >> 2) + 64 * i, seg = tid & 3;
        goff[i] = row * H * 2 + seg * 16;
        loff[i] = row * 64 + (seg ^ ((row >> 2) & 3)) * 16;
    }
    const uint32_t KP = H / 32;
    auto gload = [&](u32x4 (&ra)[4], u32x4 (&rb)[4], uint32_t p) {
        const uint64_t k0 = (uint64_t)min(p, KP - 1) * 64;  // bytes; past the end: the last sub-step again (never used)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(qbase + k0 + goff[i]);
            rb[i] = *reinterpret_cast<const u32x4*>(pbase + k0 + goff[i]);
        }
    };
    auto lstore = [&](const u32x4 (&ra)[4], const u32x4 (&rb)[4], uint32_t stage) {
        uint8_t* const st = smem + stage * kGwStage;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(st + loff[i]) = ra[i];
            *reinterpret_cast<u32x4*>(st + 256 * 64 + loff[i]) = rb[i];
        }
    };
    const uint32_t f = (r >> 2) & 3;
    const uint32_t fa = (wm + r) * 64, fb = 256 * 64 + (wn + r) * 64;
    const uint32_t slot0 = ((0 + h) ^ f) * 16, slot1 = ((2 + h) ^ f) * 16;
    auto read_frags = [&](half8 (&a)[2][4], half8 (&b)[2][4], uint32_t stage) {
        const uint8_t* const st = smem + stage * kGwStage;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[0][i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 64 + slot0);
            b[0][i] = *reinterpret_cast<const half8*>(st + fb + i * 32 * 64 + slot0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[1][i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 64 + slot1);
            b[1][i] = *reinterpret_cast<const half8*>(st + fb + i * 32 * 64 + slot1);
        }
    };
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };
    // one sub-step p (buffer `cur` = p & 1): fragments(p) are in (ca, cb); fragments(p + 1) go to (na, nb) from the other
    // buffer; the registers (ra, rb) hold sub-step p + 2, which goes into buffer `cur`, and are refilled with p + 4
    auto step = [&](uint32_t p, uint32_t cur, half8 (&ca)[2][4], half8 (&cb)[2][4], half8 (&na)[2][4], half8 (&nb)[2][4],
                    u32x4 (&ra)[4], u32x4 (&rb)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments(p) have arrived, own stores of p + 1 are done
        bar();                                              // ... everyone's: buffer cur ^ 1 is complete, buffer cur is free
        read_frags(na, nb, cur ^ 1);
        lstore(ra, rb, cur);
        if constexpr (LAB != 1 && LAB != 4) gload(ra, rb, p + 4);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ca[kk][i], cb[kk][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
    };
    half8 a0[2][4], b0[2][4], a1[2][4], b1[2][4];
    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    gload(ra0, rb0, 0);
    gload(ra1, rb1, 1);
    lstore(ra0, rb0, 0);
    lstore(ra1, rb1, 1);
    gload(ra0, rb0, 2);
    gload(ra1, rb1, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    read_frags(a0, b0, 0);
    uint64_t t0 = 0, r0 = 0;
    if constexpr (LAB == 3 || LAB == 4) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t p = 0; p < KP; p += 2) {  // KP is even (H % 64 == 0)
        step(p, 0, a0, b0, a1, b1, ra0, rb0);
        step(p + 1, 1, a1, b1, a0, b0, ra1, rb1);
    }
    if constexpr (LAB == 3 || LAB == 4) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x] = t1 - t0, g_gemm_lab_stamps[8 * blockIdx.x + 1] = r1 - r0, g_gemm_lab_stamps[8 * blockIdx.x + 4] = r0,
            g_gemm_lab_stamps[8 * blockIdx.x + 5] = r1;
    }
    store_block_via_lds<2 * kGwStage>(acc, smem, out, M, N, ld, q_blk, d_blk, wave, lane, raw);
    if constexpr (LAB == 3 || LAB == 4) {
        const uint64_t issued = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x + 2] = life0, g_gemm_lab_stamps[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime(),
            g_gemm_lab_stamps[8 * blockIdx.x + 6] = issued;
    }
}

// dense_scores_256r16: dense_scores_256r on v_mfma_f32_16x16x32_f16. Same FLOPs per cycle as the 32 x 32 x 16 shape, but
// the chip holds a higher clock under it (MI355X_MICROARCH.md, DVFS give-back (7): 1.12-1.15 x the FLOP/s on random data;
// the vendor library's kernels for this shape are MI16x16 too). Wave tile 128 x 128 = 8 x 8 tiles, one K step of 32 per
// sub-step: 64 MFMAs of 16 cycles, 8 + 8 fragment reads. Fragment map: lane (c = l & 15, g = l >> 4) holds k = 8g .. 8g+7
// of row c of A / of column c of B; C/D: column l & 15, rows 4 (l >> 4) + reg. LDS rows of 64 B with slot swizzle
// slot ^= (row >> 2) & 2 (conflict-free for the four 16-lane groups of ds_read_b128 under THIS lane -> (row, slot) map;
// found by enumeration).

template <int LAB = 0>
__global__ __launch_bounds__(256, 1) void dense_scores_256r16(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                              uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                              uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 buffers of kGwStage
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
    const char* const qbase = reinterpret_cast<const char*>(Q + (uint64_t)q_blk * H);
    const char* const pbase = reinterpret_cast<const char*>(P + (uint64_t)d_blk * H);
    uint32_t goff[4], loff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = (tid >> 2) + 64 * i, seg = tid & 3;
        goff[i] = row * H * 2 + seg * 16;
        loff[i] = row * 64 + (seg ^ ((row >> 2) & 2)) * 16;
    }
    const uint32_t KP = H / 32;
    auto gload = [&](u32x4 (&ra)[4], u32x4 (&rb)[4], uint32_t p) {
        const uint64_t k0 = (uint64_t)min(p, KP - 1) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const u32x4*>(qbase + k0 + goff[i]);
            rb[i] = *reinterpret_cast<const u32x4*>(pbase + k0 + goff[i]);
        }
    };
    auto lstore = [&](const u32x4 (&ra)[4], const u32x4 (&rb)[4], uint32_t stage) {
        uint8_t* const st = smem + stage * kGwStage;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(st + loff[i]) = ra[i];
            *reinterpret_cast<u32x4*>(st + 256 * 64 + loff[i]) = rb[i];
        }
    };
    // rows wm + 16 i + c: (row >> 2) & 2 == (c >> 2) & 2
    const uint32_t fslot = (g ^ ((c >> 2) & 2)) * 16;
    const uint32_t fa = (wm + c) * 64 + fslot, fb = 256 * 64 + (wn + c) * 64 + fslot;
    auto read_frags = [&](half8 (&a)[8], half8 (&b)[8], uint32_t stage) {
        const uint8_t* const st = smem + stage * kGwStage;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a[i] = *reinterpret_cast<const half8*>(st + fa + i * 16 * 64);
            b[i] = *reinterpret_cast<const half8*>(st + fb + i * 16 * 64);
        }
    };
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };
    auto step = [&](uint32_t p, uint32_t cur, half8 (&ca)[8], half8 (&cb)[8], half8 (&na)[8], half8 (&nb)[8], u32x4 (&ra)[4],
                    u32x4 (&rb)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        // The MFMAs are inline asm with the accumulator tied in place in AGPRs: left to the register allocator, 64
        // four-register accumulator tuples end in a storm of accvgpr copies and scratch spills. Inline asm is invisible
        // to the sched_group_barrier classes, so the issue order is pinned by hand with sched_barrier(0): 32 MFMAs with
        // one fragment read of the next sub-step after every second one, then 32 with a ds_write + a global load after
        // every fourth.
        const uint8_t* const stn = smem + (cur ^ 1) * kGwStage;
        uint8_t* const stw = smem + cur * kGwStage;
        const uint64_t k0 = (uint64_t)min(p + 4, KP - 1) * 64;
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            const int i = m >> 3, j = m & 7;
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(ca[i]), "v"(cb[j]));
            if (m < 32 && (m & 1)) {
                const int q = m >> 1;  // 0..15: a0 b0 a1 b1 ... in the order the next sub-step consumes them (b first rows)
                if (q < 8) nb[q] = *reinterpret_cast<const half8*>(stn + fb + q * 16 * 64);
                else na[q - 8] = *reinterpret_cast<const half8*>(stn + fa + (q - 8) * 16 * 64);
            }
            if (m >= 32 && (m & 3) == 3) {
                const int q = (m - 32) >> 2;  // 0..7
                if (q < 4) {
                    *reinterpret_cast<u32x4*>(stw + loff[q]) = ra[q];
                    ra[q] = *reinterpret_cast<const u32x4*>(qbase + k0 + goff[q]);
                } else {
                    *reinterpret_cast<u32x4*>(stw + 256 * 64 + loff[q - 4]) = rb[q - 4];
                    rb[q - 4] = *reinterpret_cast<const u32x4*>(pbase + k0 + goff[q - 4]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    half8 a0[8], b0[8], a1[8], b1[8];
    u32x4 ra0[4], rb0[4], ra1[4], rb1[4];
    gload(ra0, rb0, 0);
    gload(ra1, rb1, 1);
    lstore(ra0, rb0, 0);
    lstore(ra1, rb1, 1);
    gload(ra0, rb0, 2);
    gload(ra1, rb1, 3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    read_frags(a0, b0, 0);
    uint64_t t0 = 0, r0 = 0;
    if constexpr (LAB == 3) t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t p = 0; p < KP; p += 2) {  // KP is even (H % 64 == 0)
        step(p, 0, a0, b0, a1, b1, ra0, rb0);
        step(p + 1, 1, a1, b1, a0, b0, ra1, rb1);
    }
    if constexpr (LAB == 3) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && blockIdx.x < 4096)
            g_gemm_lab_stamps[8 * blockIdx.x] = t1 - t0, g_gemm_lab_stamps[8 * blockIdx.x + 1] = r1 - r0, g_gemm_lab_stamps[8 * blockIdx.x + 4] = r0,
            g_gemm_lab_stamps[8 * blockIdx.x + 5] = r1;
    }
    // The compiler does not know the asm statements are MFMAs: their write -> accvgpr-read hazard is covered by hand.
    // The empty asms tie every accumulator to a point AFTER the nops (volatile asms keep their order), so that no read
    // of a result can be scheduled above them.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    // ---- epilogue through LDS (see store_block_via_lds): pass t carries the tile rows i = 2t, 2t + 1 of every wave:
    // 64 query rows (2 wave rows x 32) x 256 docs
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bar();
    float* const stg = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t) bar();  // the previous pass has been read out
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
