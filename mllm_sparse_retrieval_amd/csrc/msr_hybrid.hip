// Hybrid path and encode-side sparsifier: the dense fp16 MFMA scorers (the production 256 x 256 kernel lives in
// msr_gemm_w4.hpp: dense_scores_256k; here the 128 x 128 kernel for small grids and two earlier 256 x 256 kernels kept
// as diagnostics), the fused per-query kernel of single-tile indexes (hybrid_tiles), the reference's min-max fusion
// for multi-tile indexes (fuse_tiles), the log1p-relu top-k sparsifier (sparsify_keys), and their C-ABI entry points.
// Selection and list merging reuse the search path's kernels (launch_select / launch_merge, msr_device.hip).
#include <chrono>
#include <cmath>

#include "msr_accumulate.hpp"
#include "msr_gemm_w4.hpp"

using namespace msr;

// ================================================================================================ dense (hybrid path)
// Flat inner-product search over fp16 passage vectors: the dense half of the reference's hybrid search
// (tevatron FaissFlatSearcher / faiss IndexFlatIP, fp16 storage on GPU: src/search.py:232-237,254-270; queries
// normalised at src/search.py:342, corpus at src/encode.py:301). Scores C[q][d] = sum_k Q[q][k] * P[d][k] on MFMA
// (v_mfma_f32_32x32x16_f16, f32 accumulate), written as order-preserving u32 keys into the accumulator layout of
// select_tiles, so that top-`depth` selection and the tile merge are the SAME kernels as on the sparse path.
namespace msr {


// One workgroup = 4 waves = a 128 (queries) x 128 (docs) block, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32.
// K runs in steps of 32 through a double-buffered LDS stage: 128 rows x 32 halves per operand, row stride 80 B
// (5 sixteen-byte slots: 5r mod 16 is a bijection, so the 16-lane groups of ds_read_b128 hit 16 distinct slots).
// The next K-step's global loads (2 x 16 B per operand per thread) are in flight while the current step's 8 MFMAs run.
// Fragment map of v_mfma_f32_32x32x16_f16: lane (r = l & 31, h = l >> 5) holds elements k = 8h .. 8h+7 of row r of A
// and of column r of B (= row r of P). Q has Mpad rows, P has Npad rows (multiples of 128, zero padded), H % 32 == 0.
constexpr int kGemmRowB = 80;                    // LDS row stride in bytes (64 B of data + 16 B pad)
constexpr int kGemmTileB = 128 * kGemmRowB;      // one operand stage

// raw != 0: the f32 bit patterns are written as they are (hybrid_tiles takes the row as floats) instead of keys
__global__ __launch_bounds__(256) void dense_scores(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                    uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                    uint64_t ld, uint32_t raw) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][2][kGemmTileB];  // [buffer][A|B]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const uint32_t q_blk = blockIdx.x * 128, d_blk = blockIdx.y * 128;
    if (d_blk >= N) {  // a block of padding docs that select_tiles still reads (its rounds are 4*NT docs wide): keys 0
        for (uint32_t i = tid; i < 128 * 32; i += 256) {
            const uint32_t q = q_blk + i / 32;
            if (q < M) reinterpret_cast<uint4*>(out + (uint64_t)q * ld + d_blk)[i % 32] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    float16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // global -> LDS assignment: 512 sixteen-byte segments per operand stage, two per thread (rows s/4, segment s%4)
    const uint32_t s0 = tid, s1 = tid + 256;
    const _Float16* ga0 = Q + (uint64_t)(q_blk + s0 / 4) * H + (s0 % 4) * 8;
    const _Float16* ga1 = Q + (uint64_t)(q_blk + s1 / 4) * H + (s1 % 4) * 8;
    const _Float16* gb0 = P + (uint64_t)(d_blk + s0 / 4) * H + (s0 % 4) * 8;
    const _Float16* gb1 = P + (uint64_t)(d_blk + s1 / 4) * H + (s1 % 4) * 8;
    const uint32_t l0 = (s0 / 4) * kGemmRowB + (s0 % 4) * 16, l1 = (s1 / 4) * kGemmRowB + (s1 % 4) * 16;
    uint4 ra0, ra1, rb0, rb1;
    auto g_load = [&](uint32_t k0) {
        ra0 = *reinterpret_cast<const uint4*>(ga0 + k0);
        ra1 = *reinterpret_cast<const uint4*>(ga1 + k0);
        rb0 = *reinterpret_cast<const uint4*>(gb0 + k0);
        rb1 = *reinterpret_cast<const uint4*>(gb1 + k0);
    };
    auto l_store = [&](int buf) {
        *reinterpret_cast<uint4*>(&stage[buf][0][l0]) = ra0;
        *reinterpret_cast<uint4*>(&stage[buf][0][l1]) = ra1;
        *reinterpret_cast<uint4*>(&stage[buf][1][l0]) = rb0;
        *reinterpret_cast<uint4*>(&stage[buf][1][l1]) = rb1;
    };
    const uint32_t fa = (wm + r) * kGemmRowB + 16 * h, fb = (wn + r) * kGemmRowB + 16 * h;
    g_load(0);
    l_store(0);
    __syncthreads();
    const uint32_t KT = H / 32;
    for (uint32_t kt = 0; kt < KT; ++kt) {
        const int cur = (int)(kt & 1);
        if (kt + 1 < KT) g_load((kt + 1) * 32);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 a0 = *reinterpret_cast<const half8*>(&stage[cur][0][fa + 32 * kk]);
            const half8 a1 = *reinterpret_cast<const half8*>(&stage[cur][0][fa + 32 * kGemmRowB + 32 * kk]);
            const half8 b0 = *reinterpret_cast<const half8*>(&stage[cur][1][fb + 32 * kk]);
            const half8 b1 = *reinterpret_cast<const half8*>(&stage[cur][1][fb + 32 * kGemmRowB + 32 * kk]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < KT) l_store(cur ^ 1);
        __syncthreads();
    }
    // C/D map of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t d = d_blk + wn + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t q = q_blk + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (q < M) out[(uint64_t)q * ld + d] = d < N ? (raw ? __float_as_uint(acc[i][j][e]) : f32_to_key(acc[i][j][e])) : 0u;
            }
        }
}


// ---- (diagnostic, MSR_GEMM_NO_PIPE) the same product on a 256 x 256 block: 8 waves (2 x 4), each 128 x 64 = 4 x 2 MFMA tiles of 32 x 32 (six fragment
// reads per eight MFMAs instead of four per four), K in steps of 64, two LDS stages of 2 x 32 KiB filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass), one barrier per K step. The DMA writes LDS
// lane-linear (wave-uniform base + 16 B x lane: eight 128-byte rows per instruction), so the bank swizzle goes on the
// SOURCE address: the lane that fills 16-byte slot s of row `row` fetches the row's segment s ^ ((row >> 1) & 7), and
// the fragment read of segment g of that row goes to slot g ^ ((row >> 1) & 7): 16 consecutive rows then touch 64
// distinct banks. Used when the grid fills the chip (>= 256 blocks) and H % 64 == 0; otherwise the 128 x 128 kernel.
constexpr int kG2Stage = 2 * 256 * 128;  // bytes per stage: A rows then B rows, 128 B (64 halves) each

// Block -> tile map: a 2-D grid (x = query block, y = doc block) when `db_n` is 0; else a 1-D grid that is XCD-aware:
// workgroup L runs on XCD L % 8 (round-robin dispatch) as that XCD's (L / 8)-th block, and every XCD walks its own
// contiguous EIGHTH of the block list in patch-major order (patches of 8 query blocks x 4 doc blocks). The 32 blocks
// resident together on an XCD (one per CU) are then one patch or the seam of two: per K step they share about 8 + 4
// operand slabs of 32 KB through that XCD's L2 instead of fetching 2 x 32 — and all XCDs get the same number of blocks
// (a patch grid dealt out patch by patch was 26 % SLOWER: 20 patches over 8 XCDs is 3 rounds for some, 2 for others).
__global__ __launch_bounds__(512, 1) void dense_scores_256(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                           uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                           uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 2 stages
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    uint32_t qb = blockIdx.x, db = blockIdx.y;
    if (db_n) {
        const uint32_t n_blk = qb_n * db_n, per = (n_blk + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const uint32_t t = xcd * per + slot;  // position in the patch-major order of all blocks
        if (slot >= per || t >= n_blk) return;
        const uint32_t pr = t / (8u * db_n), rem = t - pr * 8u * db_n;
        const uint32_t hp = min(8u, qb_n - 8u * pr);  // query blocks in this patch row (the last one may be short)
        const uint32_t pc = rem / (hp * 4u), rem2 = rem - pc * hp * 4u;
        qb = pr * 8u + rem2 % hp;
        db = pc * 4u + rem2 / hp;
    }
    const uint32_t q_blk = qb * 256, d_blk = db * 256;
    if (d_blk >= N) {  // padding docs that select_tiles still reads: keys 0
        for (uint32_t i = tid; i < 256 * 64; i += 512) {
            const uint32_t q = q_blk + i / 64;
            if (q < M) reinterpret_cast<uint4*>(out + (uint64_t)q * ld + d_blk)[i % 64] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const uint32_t wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // LDS-DMA plan: instruction i of wave w fills rows 8c .. 8c+7 (c = 8i + w) of an operand; lane = (row in chunk, slot)
    const _Float16* ga[4];
    const _Float16* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = 8 * (8 * i + wave) + (lane >> 3);
        const uint32_t seg = (lane & 7) ^ ((row >> 1) & 7);
        ga[i] = Q + (uint64_t)(q_blk + row) * H + seg * 8;
        gb[i] = P + (uint64_t)(d_blk + row) * H + seg * 8;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto issue = [&](int stage, uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint8_t* const da = smem + stage * kG2Stage + (8 * i + wave) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(ga[i] + k0), (lds_void*)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(gb[i] + k0), (lds_void*)(da + 256 * 128), 16, 0, 0);
        }
    };
    const uint32_t f = (r >> 1) & 7;  // swizzle of this lane's fragment rows ((wm + 32 i + r) >> 1) & 7 == (r >> 1) & 7
    const uint32_t fa = (wm + r) * 128, fb = 256 * 128 + (wn + r) * 128;
    const uint32_t KT = H / 64;
    issue(0, 0);
    for (uint32_t kt = 0; kt < KT; ++kt) {
        const int cur = (int)(kt & 1);
        __syncthreads();  // (hipcc drains the DMA with vmcnt(0) first) stage `cur` is filled, stage cur^1 is free again
        if (kt + 1 < KT) issue(cur ^ 1, (kt + 1) * 64);
        const uint8_t* const st = smem + cur * kG2Stage;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint32_t slot = ((2 * kk + h) ^ f) * 16;
            half8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 128 + slot);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const half8*>(st + fb + j * 32 * 128 + slot);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // C/D map of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t d = d_blk + wn + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t q = q_blk + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (q < M) out[(uint64_t)q * ld + d] = d < N ? (raw ? __float_as_uint(acc[i][j][e]) : f32_to_key(acc[i][j][e])) : 0u;
            }
        }
}

// ---- (diagnostic since dense_scores_256k, MSR_GEMM_PINGPONG) the same 256 x 256 block with a DEEPER staging pipeline: K in sub-steps of 32 through FOUR LDS buffers of 32 KiB
// (A: 256 rows x 64 B, B likewise), three sub-steps of LDS-DMA in flight while the fourth is consumed. The 2-buffer
// kernel above drains its DMA with vmcnt(0) at every barrier and issues a whole K step's 8 DMAs per wave up front — at
// 100-185 cycles of issue each that is ~1 200 cycles per wave and step during which neither wave of the SIMD feeds the
// MFMA pipe (measured 46 % MFMA-busy cycles at 74 % L2 hit rate: not bandwidth). Here each sub-step issues 4 DMAs
// per wave, waits with a COUNTED vmcnt (own pieces of the sub-step read next: 8 newer pieces stay in flight), uses
// raw s_barriers (no fence, no vmcnt(0)), and runs the two waves of every SIMD in PING-PONG (see the loop):
//     RAW: a sub-step's pieces are waited for by their issuing waves BEFORE the barrier the readers pass;
//     WAR: a buffer is refilled after both groups' reads of it have returned (lgkmcnt(0) before the barrier in between).
// Rows are 64 B = 4 slots of 16 B: the DMA writes LDS lane-linear (16 rows x 4 slots per instruction), so the bank
// swizzle slot ^= (row >> 2) & 3 goes on the SOURCE address and again on the fragment reads (the four 16-lane groups
// of a ds_read_b128 then touch 16 distinct 16-byte bank groups).
constexpr int kGpStage = 2 * 256 * 64;  // bytes per sub-step buffer: A rows then B rows
constexpr int kGpStages = 4;

__global__ __launch_bounds__(512, 1) void dense_scores_256p(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                            uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                            uint64_t ld, uint32_t qb_n, uint32_t db_n, uint32_t raw) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // 4 sub-step buffers (ONE LDS object)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = rfl(tid >> 6);
    const uint32_t r = lane & 31, h = lane >> 5;
    uint32_t qb, db;
    {   // XCD-aware patch-major order, as in dense_scores_256
        const uint32_t n_blk = qb_n * db_n, per = (n_blk + 7u) / 8u;
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        const uint32_t t = xcd * per + slot;
        if (slot >= per || t >= n_blk) return;
        const uint32_t pr = t / (8u * db_n), rem = t - pr * 8u * db_n;
        const uint32_t hp = min(8u, qb_n - 8u * pr);
        const uint32_t pc = rem / (hp * 4u), rem2 = rem - pc * hp * 4u;
        qb = pr * 8u + rem2 % hp;
        db = pc * 4u + rem2 / hp;
    }
    const uint32_t q_blk = qb * 256, d_blk = db * 256;
    if (d_blk >= N) {  // padding docs: keys 0
        for (uint32_t i = tid; i < 256 * 64; i += 512) {
            const uint32_t q = q_blk + i / 64;
            if (q < M) reinterpret_cast<uint4*>(out + (uint64_t)q * ld + d_blk)[i % 64] = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const uint32_t wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    float16v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // DMA plan: piece c (1 KiB = 16 rows x 64 B) of an operand's sub-step; wave w issues pieces w and w + 8. Source
    // address = (block base + k: scalar) + (row, swizzled segment: a 32-bit byte offset per lane), so that stepping k
    // costs scalar adds only (a block's rows span 256 x H x 2 B < 4 GiB)
    const char* const qbase = reinterpret_cast<const char*>(Q + (uint64_t)q_blk * H);
    const char* const pbase = reinterpret_cast<const char*>(P + (uint64_t)d_blk * H);
    uint32_t voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t row = 16 * (wave + 8 * i) + (lane >> 2);
        const uint32_t seg = (lane & 3) ^ ((row >> 2) & 3);
        voff[i] = row * H * 2 + seg * 16;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    auto issue = [&](uint32_t stage, uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint8_t* const da = smem + stage * kGpStage + (wave + 8 * i) * 1024;
            __builtin_amdgcn_global_load_lds((glb_void*)(qbase + (uint64_t)k0 * 2 + voff[i]), (lds_void*)da, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void*)(pbase + (uint64_t)k0 * 2 + voff[i]), (lds_void*)(da + 256 * 64), 16, 0, 0);
        }
    };
    const uint32_t f = (r >> 2) & 3;  // swizzle of this lane's fragment rows: (wm + 32 i + r) >> 2 & 3 == (r >> 2) & 3
    const uint32_t fa = (wm + r) * 64, fb = 256 * 64 + (wn + r) * 64;
    const uint32_t KP = H / 32;
    // PING-PONG: the workgroup's waves w and w + 4 share a SIMD (waves go to SIMDs cyclically); group 0 (waves 0-3) and
    // group 1 (waves 4-7) run the same loop half a sub-step apart, so in every barrier interval one wave of each SIMD
    // reads fragments / issues DMA while the other issues its 16 MFMAs:
    //     interval 2p   : group 0 MEM(p)   | group 1 MFMA(p-1)
    //     interval 2p+1 : group 0 MFMA(p)  | group 1 MEM(p)
    // MEM(p) = 12 fragment reads of buffer p & 3, then the wave's 4 DMAs of sub-step p + 3 into buffer (p - 1) & 3 (both
    // groups have read it: group 1 in interval 2p - 1, with lgkmcnt(0) before that interval's barrier). Before the barrier
    // that ends an odd interval every wave waits (counted) for its own pieces of sub-step p + 1, which group 0 reads next.
    // Measured with s_memtime laps (one block, 128 sub-steps, cycles per sub-step and wave): MEM 780 (fragment reads +
    // DMA issue + lgkmcnt), counted vmcnt wait 300, MFMA issue 575, barriers 360 — 2 020 per sub-step for 1 024 cycles of
    // MFMA per SIMD = the 52 % MFMA-busy the counters show, at a clock that sags to ~1.85 GHz under this load (2.55 GHz in
    // the integer kernels). Moving 1, 2 or all 4 DMAs between the MFMAs shortens MEM by < 10 % and lengthens MFMA (908
    // vs 949 TFLOP/s for all 4); without s_setprio 840; sixteen waves (4 x 4, 64 x 64 each, four per SIMD, one barrier
    // per sub-step, no ping-pong) 857. What is left is the fragment-read phase itself and the DMA latency under load
    // (~4 000 cycles: three sub-steps in flight do not quite cover it).
    const uint32_t grp = wave >> 2;
    auto bar = []() { asm volatile("s_barrier" ::: "memory"); };
    auto wait_landed = [&](uint32_t p_next) {  // own pieces of sub-step p_next landed; newer ones (<= 2 sub-steps) stay in flight
        const uint32_t newer = p_next + 2 < KP ? 2u : (p_next + 1 < KP ? 1u : 0u);
        if (newer == 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (newer == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    issue(0, 0);
    if (KP > 1) issue(1, 32);
    if (KP > 2) issue(2, 64);
    wait_landed(0);
    bar();
    if (grp) bar();  // group 1 starts half a sub-step later
    for (uint32_t p = 0; p < KP; ++p) {
        // ---- MEM(p)
        const uint8_t* const st = smem + (p & 3) * kGpStage;
        half8 a[2][4], b[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t slot = ((2 * kk + h) ^ f) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) a[kk][i] = *reinterpret_cast<const half8*>(st + fa + i * 32 * 64 + slot);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[kk][j] = *reinterpret_cast<const half8*>(st + fb + j * 32 * 64 + slot);
        }
        if (p + 3 < KP) issue((p + 3) & 3, (p + 3) * 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragments are in registers: the buffer may be refilled
        if (grp) wait_landed(p + 1);                        // (group 1's MEM is the odd interval)
        bar();
        // ---- MFMA(p)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (!grp) wait_landed(p + 1);                       // (group 0's MFMA is the odd interval)
        bar();
    }
    if (!grp) bar();  // every wave has passed the same number of barriers
    // C/D map of the 32x32 shapes: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t d = d_blk + wn + 32 * j + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t q = q_blk + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (q < M) out[(uint64_t)q * ld + d] = d < N ? (raw ? __float_as_uint(acc[i][j][e]) : f32_to_key(acc[i][j][e])) : 0u;
            }
        }
}

}  // namespace msr

struct msr_dense {
    int device = -1;
    hipStream_t stream = nullptr;
    _Float16* d_P = nullptr;  // [n_pad][h]
    uint64_t n = 0;
    uint64_t n_pad = 0;       // multiple of 128 and of tile_docs
    uint32_t h = 0;
    uint32_t tile_docs = 0;
    uint32_t n_tiles = 0;
    // hybrid path on single-tile indexes: the same rows in doc-ORDINAL order (P_ord[row2ord[r]] = P[r]), built on first
    // use and kept while the mapping stays the same
    _Float16* d_P_ord = nullptr;
    std::vector<uint32_t> ord_map;
    bool lds_attr_set = false;  // dense_scores_256 needs the 128 KiB dynamic-LDS opt-in once per device
    // device scratch of the fused hybrid call (query matrix, score rows, result lists), kept between calls and only ever
    // grown: six hipMalloc / hipFree pairs of up to 200 MB cost a call more than its kernels
    enum Slot { S_Q = 0, S_S, S_ORD, S_SF, S_N, S_SELF, S_CS, S_META, S_FLAGS, S_QLIST, S_PART, S_SU, S_COUNT };
    void* scratch[S_COUNT] = {};
    size_t scratch_bytes[S_COUNT] = {};
    uint64_t n_device_allocs = 0;  // hipMalloc calls made on behalf of this handle (steady-state calls make none)
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // msr_dense_search's laps, created on first use
    void* take(int slot, size_t bytes) {
        if (scratch_bytes[slot] >= bytes && scratch[slot]) return scratch[slot];
        if (scratch[slot]) (void)hipFree(scratch[slot]);
        scratch[slot] = nullptr;
        scratch_bytes[slot] = 0;
        ++n_device_allocs;
        if (hipMalloc(&scratch[slot], bytes) != hipSuccess) {
            scratch[slot] = nullptr;
            return nullptr;
        }
        scratch_bytes[slot] = bytes;
        return scratch[slot];
    }
};

extern "C" {

int msr_dense_open(const uint16_t* p_fp16, uint64_t n, uint32_t h, int device, msr_dense** out) {
    if (!out) {
        set_error("msr_dense_open: null output");
        return MSR_E_INVAL;
    }
    *out = nullptr;
    if ((!p_fp16 && n) || h == 0 || h % 32 != 0 || n >= (1ull << 31)) {
        set_error("msr_dense_open: need fp16 rows with a dimension that is a multiple of 32 (got n=%llu, h=%u)",
                  (unsigned long long)n, h);
        return MSR_E_INVAL;
    }
    int n_dev = 0;
    if (device < 0 || hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) {
        set_error("no usable HIP device %d; there is no CPU dense search path", device);
        return MSR_E_NODEVICE;
    }
    msr_dense* dx = new (std::nothrow) msr_dense;
    if (!dx) {
        set_error("out of host memory");
        return MSR_E_NOMEM;
    }
    dx->device = device;
    dx->n = n;
    dx->h = h;
    dx->tile_docs = n <= 4096 ? 4096 : 8192;
    dx->n_tiles = (uint32_t)std::max<uint64_t>((n + dx->tile_docs - 1) / dx->tile_docs, 1);
    dx->n_pad = (uint64_t)dx->n_tiles * dx->tile_docs;
    const size_t bytes = (size_t)dx->n_pad * h * 2;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&dx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&dx->d_P, bytes) != hipSuccess || hipMemset(dx->d_P, 0, bytes) != hipSuccess ||
        (n && hipMemcpy(dx->d_P, p_fp16, (size_t)n * h * 2, hipMemcpyHostToDevice) != hipSuccess)) {
        set_error("device setup of the dense index failed (%zu bytes)", bytes);
        if (dx->d_P) (void)hipFree(dx->d_P);
        if (dx->stream) (void)hipStreamDestroy(dx->stream);
        delete dx;
        return MSR_E_HIP;
    }
    *out = dx;
    return MSR_OK;
}

void msr_dense_close(msr_dense* dx) {
    if (!dx) return;
    (void)hipSetDevice(dx->device);
    if (dx->d_P_ord) (void)hipFree(dx->d_P_ord);
    if (dx->d_P) (void)hipFree(dx->d_P);
    for (void* p : dx->scratch)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : dx->ev)
        if (e) (void)hipEventDestroy(e);
    if (dx->stream) (void)hipStreamDestroy(dx->stream);
    delete dx;
}

// Row blocks (of 256 queries) per GEMM launch: among the chunk sizes that fit (<= rb_max, <= the batch), the one whose
// grid of row blocks x col_blocks fills its last round of the chip's 256 CUs best (2 x 98 blocks leave a quarter of the
// chip idle, 5 x 98 = 490 fill 1.91 rounds); ties go to the larger chunk.
static uint32_t pick_row_blocks(uint32_t col_blocks, uint64_t rb_max, uint32_t batch_row_blocks) {
    uint32_t best = 1;
    double best_eff = 0;
    for (uint64_t rb = 1; rb <= std::min<uint64_t>(rb_max, std::max<uint32_t>(batch_row_blocks, 1)); ++rb) {
        const uint64_t blocks = rb * std::max<uint32_t>(col_blocks, 1);
        const double eff = (double)blocks / (double)((blocks + 255) / 256 * 256);
        if (eff >= best_eff - 1e-9) best_eff = eff, best = (uint32_t)rb;
    }
    return best;
}

// C = Q * P^T as order-preserving keys into out[q][ld] for doc columns [0, n_cover); rows of Q padded to qn_pad (a
// multiple of 256), P rows readable up to n_cover. Doc blocks are launched in slices of at most 65535 (grid y limit).
static int launch_dense_gemm(msr_dense* dx, const _Float16* P, const _Float16* d_Q, uint32_t* d_S, uint32_t qn,
                             uint32_t qn_pad, uint64_t n_cover, uint64_t ld, hipStream_t st, bool force_256 = false,
                             bool raw = false) {
    const bool big = dx->h % 64 == 0 && n_cover % 256 == 0 &&
                     (force_256 || (uint64_t)(qn_pad / 256) * (n_cover / 256) >= 256);
    if (big && !dx->lds_attr_set) {  // 128 KiB of dynamic LDS needs the opt-in (per device: kept in the handle)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_scores_256),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kG2Stage));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_scores_256p),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kGpStages * kGpStage));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_scores_256k<0>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kGkStage));
        dx->lds_attr_set = true;
    }
    const uint64_t blk = big ? 256 : 128;
    for (uint64_t d0 = 0; d0 < n_cover; d0 += blk * kMaxGridY) {
        const uint64_t nd = std::min<uint64_t>(n_cover - d0, blk * kMaxGridY);
        const uint32_t n_left = dx->n > d0 ? (uint32_t)std::min<uint64_t>(dx->n - d0, 0xFFFFFFFFull) : 0u;
        static const bool no_patch = getenv("MSR_GEMM_NO_PATCH") != nullptr;  // diagnostic: the plain 2-D grid
        static const bool no_pipe = getenv("MSR_GEMM_NO_PIPE") != nullptr;    // diagnostic: the 2-buffer kernel
        static const bool pingpong = getenv("MSR_GEMM_PINGPONG") != nullptr;  // diagnostic: the 8-wave ping-pong kernel
        if (big && !no_patch && !no_pipe && !pingpong) {
            const uint32_t qb_n = qn_pad / 256, db_n = (uint32_t)(nd / 256);
            hipLaunchKernelGGL(dense_scores_256k<0>, dim3((qb_n * db_n + 7) / 8 * 8), dim3(256), 2 * kGkStage, st, d_Q,
                               P + d0 * dx->h, d_S + d0, qn, n_left, dx->h, ld, qb_n, db_n, raw ? 1u : 0u);
        } else if (big && !no_patch && !no_pipe) {
            const uint32_t qb_n = qn_pad / 256, db_n = (uint32_t)(nd / 256);
            hipLaunchKernelGGL(dense_scores_256p, dim3((qb_n * db_n + 7) / 8 * 8), dim3(512), kGpStages * kGpStage, st, d_Q,
                               P + d0 * dx->h, d_S + d0, qn, n_left, dx->h, ld, qb_n, db_n, raw ? 1u : 0u);
        } else if (big && !no_patch) {
            const uint32_t qb_n = qn_pad / 256, db_n = (uint32_t)(nd / 256);
            hipLaunchKernelGGL(dense_scores_256, dim3((qb_n * db_n + 7) / 8 * 8), dim3(512), 2 * kG2Stage, st, d_Q,
                               P + d0 * dx->h, d_S + d0, qn, n_left, dx->h, ld, qb_n, db_n, raw ? 1u : 0u);
        } else if (big)
            hipLaunchKernelGGL(dense_scores_256, dim3(qn_pad / 256, (uint32_t)(nd / 256)), dim3(512), 2 * kG2Stage, st, d_Q,
                               P + d0 * dx->h, d_S + d0, qn, n_left, dx->h, ld, 0u, 0u, raw ? 1u : 0u);
        else
            hipLaunchKernelGGL(dense_scores, dim3(qn_pad / 128, (uint32_t)(nd / 128)), dim3(256), 0, st, d_Q, P + d0 * dx->h,
                               d_S + d0, qn, n_left, dx->h, ld, raw ? 1u : 0u);
        HIP_TRY(hipGetLastError());
    }
    return MSR_OK;
}

// to_device = true: out_* are DEVICE buffers ([nq][k] / [nq]) filled on dx->stream (hybrid path); else host buffers.
static int dense_search_impl(msr_dense* dx, const uint16_t* q_fp16, int nq, int k, uint32_t* out_idx, uint32_t* out_key,
                             int32_t* out_n, float* gemm_ms, float* select_ms, bool to_device, bool unsorted = false) {
    if (!dx || nq < 0 || (nq && !q_fp16) || !out_idx || !out_key || !out_n) {
        set_error("msr_dense_search: bad argument");
        return MSR_E_INVAL;
    }
    if (k < 1 || k > MSR_KMAX) {
        set_error("k must be in [1, %d] (got %d)", MSR_KMAX, k);
        return MSR_E_RANGE;
    }
    HIP_TRY(hipSetDevice(dx->device));
    // queries per pass: score rows of at most ~256 MB (they are read back by select_tiles out of the Infinity Cache, and
    // the buffer stays with the handle), the GEMM grid filling its last round of 256 CUs as well as it can
    const uint32_t nq_pad_all = (uint32_t)((std::max(nq, 1) + 255) / 256 * 256);
    const uint64_t rb_max = std::max<uint64_t>(1, (256ull << 20) / (std::max<uint64_t>(dx->n_pad, 256) * 4 * 256));
    const uint32_t qt_pad = 256u * pick_row_blocks((uint32_t)(std::max<uint64_t>(dx->n_pad, 256) / 256), rb_max, nq_pad_all / 256);
    const uint32_t qt = (uint32_t)std::min<uint32_t>(qt_pad, std::max(nq, 1));
    int rc = MSR_OK;
    const size_t perq = std::max<size_t>((size_t)qt * k, 1);
    // device scratch and events live on the handle (only ever grown): at the reference's --batch_size 2
    // (scripts/search.sh:29) seven hipMalloc / hipFree pairs per call cost more than the search
    _Float16* d_Q = (_Float16*)dx->take(msr_dense::S_Q, (size_t)qt_pad * dx->h * 2);
    uint32_t* d_S = (uint32_t*)dx->take(msr_dense::S_S, (size_t)qt_pad * dx->n_pad * 4);
    uint64_t* d_part = (uint64_t*)dx->take(msr_dense::S_PART, (size_t)dx->n_tiles * perq * 8);
    uint32_t* d_ord = (uint32_t*)dx->take(msr_dense::S_ORD, perq * 4);
    uint32_t* d_su = (uint32_t*)dx->take(msr_dense::S_SU, perq * 4);
    float* d_sf = (float*)dx->take(msr_dense::S_SF, perq * 4);
    int32_t* d_n = (int32_t*)dx->take(msr_dense::S_N, (size_t)qt * 4);
    bool ok = d_Q && d_S && d_part && d_ord && d_su && d_sf && d_n;
    for (hipEvent_t& e : dx->ev)
        if (ok && !e) ok = hipEventCreate(&e) == hipSuccess;
    hipEvent_t e0 = dx->ev[0], e1 = dx->ev[1], e2 = dx->ev[2];
    if (!ok) {
        set_error("device allocation failed in msr_dense_search (%u queries per pass x %llu docs)", qt, (unsigned long long)dx->n_pad);
        rc = MSR_E_NOMEM;
    }
    double t_gemm = 0, t_sel = 0;
    for (int q0 = 0; q0 < nq && rc == MSR_OK; q0 += (int)qt) {
        const uint32_t qn = (uint32_t)std::min<int>((int)qt, nq - q0);
        const uint32_t qn_pad = (qn + 255) / 256 * 256;
        bool c = hipMemsetAsync(d_Q, 0, (size_t)qn_pad * dx->h * 2, dx->stream) == hipSuccess &&
                 hipMemcpyAsync(d_Q, q_fp16 + (size_t)q0 * dx->h, (size_t)qn * dx->h * 2, hipMemcpyHostToDevice,
                                dx->stream) == hipSuccess &&
                 hipEventRecord(e0, dx->stream) == hipSuccess;
        if (!c) {
            set_error("query upload failed in msr_dense_search");
            rc = MSR_E_HIP;
            break;
        }
        // doc blocks: up to the end of the last select round that holds a real doc (a round = 4 docs x threads of
        // the select instance: 1024 docs at 4096-doc tiles, 2048 at 8192); padding beyond that is never read
        const uint64_t round_docs = dx->tile_docs == 4096 ? 1024 : 2048;
        const uint64_t n_cover = std::min<uint64_t>(dx->n_pad, (dx->n + round_docs - 1) / round_docs * round_docs);
        rc = launch_dense_gemm(dx, dx->d_P, d_Q, d_S, qn, qn_pad, n_cover, dx->n_pad, dx->stream);
        if (rc != MSR_OK) break;
        (void)hipEventRecord(e1, dx->stream);
        SelectArgs se;
        se.unsorted = 0;
        se.src = d_S;
        se.part = d_part;
        se.n_docs = dx->n;
        se.n_tiles = dx->n_tiles;
        se.tpr = dx->n_tiles;
        se.rank = 0;
        se.nq = qn;
        se.q0 = 0;
        se.qn = qn;
        se.k = (uint32_t)k;
        se.unsorted = (unsorted && dx->n_tiles == 1) ? 1u : 0u;  // the fusion takes the list as a set
        rc = launch_select(dx->stream, dx->tile_docs, se);
        if (rc != MSR_OK) break;
        MergeArgs ma;
        ma.lists = d_part;
        ma.list_stride = (uint64_t)qn * k;
        ma.n_lists = dx->n_tiles;
        ma.nq = qn;
        ma.k = (uint32_t)k;
        ma.out_keys = nullptr;
        ma.out_ord = d_ord;
        ma.out_score_u32 = d_su;
        ma.out_score = d_sf;
        ma.out_n = d_n;
        rc = launch_merge(dx->stream, ma);
        if (rc != MSR_OK) break;
        (void)hipEventRecord(e2, dx->stream);
        const hipMemcpyKind kind = to_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        c = hipMemcpyAsync(out_idx + (size_t)q0 * k, d_ord, (size_t)qn * k * 4, kind, dx->stream) == hipSuccess &&
            hipMemcpyAsync(out_key + (size_t)q0 * k, d_su, (size_t)qn * k * 4, kind, dx->stream) == hipSuccess &&
            hipMemcpyAsync(out_n + q0, d_n, (size_t)qn * 4, kind, dx->stream) == hipSuccess &&
            hipStreamSynchronize(dx->stream) == hipSuccess;
        if (!c) {
            set_error("dense search kernels or result download failed: %s", hipGetErrorString(hipGetLastError()));
            rc = MSR_E_HIP;
            break;
        }
        float a = 0, b2 = 0;
        (void)hipEventElapsedTime(&a, e0, e1);
        (void)hipEventElapsedTime(&b2, e1, e2);
        t_gemm += a;
        t_sel += b2;
    }
    if (gemm_ms) *gemm_ms = (float)t_gemm;
    if (select_ms) *select_ms = (float)t_sel;
    return rc;
}

int msr_dense_stats(const msr_dense* dx, uint64_t out[4]) {
    if (!dx || !out) {
        set_error("msr_dense_stats: bad argument");
        return MSR_E_INVAL;
    }
    uint64_t bytes = 0;
    for (size_t b : dx->scratch_bytes) bytes += b;
    out[0] = dx->n_device_allocs;
    out[1] = bytes;
    out[2] = dx->n;
    out[3] = dx->h;
    return MSR_OK;
}

int msr_dense_search(msr_dense* dx, const uint16_t* q_fp16, int nq, int k, uint32_t* out_idx, uint32_t* out_key,
                     int32_t* out_n, float* gemm_ms, float* select_ms) {
    return dense_search_impl(dx, q_fp16, nq, k, out_idx, out_key, out_n, gemm_ms, select_ms, false);
}

}  // extern "C"

// ================================================================================================ hybrid fusion
// The reference's fuse() (src/hybrid.py:32-53) on the GPU: per query, over the union of the dense and the sparse
// top-`depth` lists,  fused(doc) = w_dense * (d - min_d) / max(max_d - min_d, 1e-9)   [if the dense list holds doc]
//                                + w_sparse * (s - min_s) / max(max_s - min_s, 1e-9)  [if the sparse list holds doc]
// with min/max over each UNFILTERED list (get_run_dict, src/search.py:76-81) and the query's own doc skipped when
// remove_query is set (src/search.py:72-74). Fused scores are built in an LDS accumulator tile over doc ordinals and
// the best k are selected by the same tile_select as everywhere else. f32 arithmetic (the reference mixes f32 and
// f64 depending on the numpy version): scores agree within the north star's 1e-5.
namespace msr {

struct FuseArgs {
    const uint64_t* s_keys;   // [nq][depth] sparse keys (score<<32 | ~ordinal), best first, 0 padded
    const uint32_t* d_idx;    // [nq][depth] dense row indices, best first
    const uint32_t* d_key;    // [nq][depth] order-preserving keys of the dense f32 scores, 0 padded
    const int32_t* d_n;       // [nq]
    const uint32_t* row2ord;  // dense row -> sparse doc ordinal
    const int32_t* self_ord;  // [nq] ordinal to skip (remove_query) or -1; may be null
    uint64_t* part;           // [n_tiles][nq][k]
    uint64_t n_docs;
    uint32_t nq, depth, k;
    float w_dense, w_sparse;
};


template <int TILE_DOCS, int NT, int CAND>
__global__ __launch_bounds__(NT) void fuse_tiles(const FuseArgs a) {
    using L = TileLds<TILE_DOCS, NT, CAND>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[L::kTotal];
    __shared__ uint8_t member[TILE_DOCS];
    __shared__ float mm[4];  // min_s, den_s, min_d, den_d
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    float* const facc = reinterpret_cast<float*>(lds);
    uint8_t* const un = lds + L::kAcc;
    uint64_t* const cand = reinterpret_cast<uint64_t*>(un);
    uint32_t* const tmax = reinterpret_cast<uint32_t*>(un + L::kUnion);
    uint32_t* const wmax = tmax + NT;
    SelectScratch& ss = *reinterpret_cast<SelectScratch*>(un + L::kUnion + L::kTmax);
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = blockIdx.x / a.nq, q = blockIdx.x % a.nq;
    const uint64_t doc0 = (uint64_t)tile * TILE_DOCS;
    const uint32_t ndocs_tile = (uint32_t)min((uint64_t)TILE_DOCS, a.n_docs - doc0);
    const int rounds = (int)((ndocs_tile + 4 * NT - 1) / (4 * NT));
    const uint64_t* sk = a.s_keys + (uint64_t)q * a.depth;
    const uint32_t* di = a.d_idx + (uint64_t)q * a.depth;
    const uint32_t* dk = a.d_key + (uint64_t)q * a.depth;
    const int32_t dn = a.d_n[q];
    const uint32_t self = a.self_ord ? (uint32_t)a.self_ord[q] : 0xFFFFFFFFu;

    for (int i = tid; i < rounds * 4 * NT; i += NT) {
        facc[i] = 0.f;
        member[i] = 0;
    }
    __shared__ uint32_t mmu[4];  // min / max sparse score, min / max dense key (the lists come in any order)
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
        mmu[0] = 0xFFFFFFFFu;
        mmu[1] = 0;
        mmu[2] = 0xFFFFFFFFu;
        mmu[3] = 0;
    }
    __syncthreads();
    {
        uint32_t lo_s = 0xFFFFFFFFu, hi_s = 0, lo_d = 0xFFFFFFFFu, hi_d = 0;
        for (uint32_t j = tid; j < a.depth; j += NT) {
            const uint64_t key = sk[j];
            if (key) {
                const uint32_t sc = (uint32_t)(key >> 32);
                lo_s = min(lo_s, sc);
                hi_s = max(hi_s, sc);
            }
            if ((int)j < dn) {
                lo_d = min(lo_d, dk[j]);
                hi_d = max(hi_d, dk[j]);
            }
        }
        // wave reductions (DPP) first, then one LDS atomic per wave and quantity (min x = ~max ~x)
        lo_s = ~wave_max_u32(~lo_s);
        hi_s = wave_max_u32(hi_s);
        lo_d = ~wave_max_u32(~lo_d);
        hi_d = wave_max_u32(hi_d);
        if ((tid & 63u) == 0) {
            if (hi_s) {
                atomicMin(&mmu[0], lo_s);
                atomicMax(&mmu[1], hi_s);
            }
            if (hi_d) {
                atomicMin(&mmu[2], lo_d);
                atomicMax(&mmu[3], hi_d);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        const bool has_s = mmu[1] != 0, has_d = mmu[3] != 0;  // (sparse scores and dense keys of hits are non-zero)
        const float smax = has_s ? (float)mmu[1] : 0.f, smin = has_s ? (float)mmu[0] : 0.f;
        const float dmax = has_d ? key_to_f32(mmu[3]) : 0.f, dmin = has_d ? key_to_f32(mmu[2]) : 0.f;
        mm[0] = smin;
        mm[1] = fmaxf(smax - smin, 1e-9f);
        mm[2] = dmin;
        mm[3] = fmaxf(dmax - dmin, 1e-9f);
    }
    __syncthreads();
    // dense pass first (the reference adds the dense term first), then the sparse pass; docs are unique per list
    for (int j = tid; j < dn; j += NT) {
        const uint32_t ord = a.row2ord[di[j]];
        if (ord != self && ord >= doc0 && ord < doc0 + ndocs_tile) {
            facc[ord - doc0] = a.w_dense * ((key_to_f32(dk[j]) - mm[2]) / mm[3]);
            member[ord - doc0] = 1;
        }
    }
    __syncthreads();
    for (uint32_t j = tid; j < a.depth; j += NT) {
        const uint64_t key = sk[j];
        if (!key) continue;
        const uint32_t ord = 0xFFFFFFFFu - (uint32_t)key;
        if (ord != self && ord >= doc0 && ord < doc0 + ndocs_tile) {
            facc[ord - doc0] += a.w_sparse * (((float)(uint32_t)(key >> 32) - mm[0]) / mm[1]);
            member[ord - doc0] = 1;
        }
    }
    __syncthreads();
    for (int i = tid; i < rounds * 4 * NT; i += NT) acc[i] = member[i] ? f32_to_key(facc[i]) : 0u;
    __syncthreads();
    tile_select<TILE_DOCS, NT, CAND>(reinterpret_cast<const uint4*>(lds), cand, tmax, wmax, ss, rounds, doc0, (int)a.k,
                                     a.part + ((uint64_t)tile * a.nq + q) * a.k, [](int) {}, threadIdx.x);
}

}  // namespace msr

// ================================================================================================ fused hybrid tile
// Single-tile indexes (n_docs <= 8192: BASELINE config 5 is COCO-5K's 5 000 images): ONE workgroup per query does the
// whole of  sparse top-`depth`  +  dense top-`depth`  +  fuse()  +  top-k  (src/search.py:55-63,85-99,455-461,
// src/hybrid.py:32-53) without a list ever leaving the CU:
//   1. accumulate_tile: the query's exact integer scores against the tile, in LDS (the search path's own phase);
//   2. every thread takes its 16 accumulators and the same 16 docs' dense scores (one row of the GEMM's output, written
//      in ORDINAL order because the GEMM ran on the ordinal-permuted passage matrix) into registers;
//   3. hist_threshold twice: the composite of the depth-th best sparse score and of the depth-th best dense score —
//      list MEMBERSHIP is all fuse() needs (min = the depth-th score, max = the best, "doc in run");
//   4. fused score per doc from registers -> LDS tile of order-preserving keys -> tile_select(k) -> result arrays.
// What this replaces: tile_select at k = 1000 inside score_tiles, select_tiles over the 500 MB score matrix,
// fuse_tiles re-reading 2 x 25 010 x 1000 keys, and three list merges.
namespace msr {

struct HybridArgs {
    const uint32_t* dkeys;    // [qn][ld] f32 bit patterns of the dense scores of queries q0 .. q0+qn-1, by ORDINAL
    uint64_t ld;
    const int32_t* self_ord;  // [nq] ordinal to skip (remove_query) or -1; may be null
    uint32_t depth, k;
    float w_dense, w_sparse;
    uint32_t* out_ord;        // [nq][k]
    float* out_score;         // [nq][k] fused scores
    int32_t* out_n;           // [nq]
    // ---- multi-tile indexes (MODE 1 of hybrid_tiles + hybrid_fuse_query): per (launch row, tile) candidate lists
    const uint32_t* qlist;    // launch row -> query of the batch (second round: the flagged queries), or null: q0 + row
    // candidate RECORDS, [rows][n_tiles][stride] each: a doc that is a candidate for either depth list of its tile, with
    // its sparse score (0: not a sparse candidate) and the key of its dense score (0: not a dense candidate)
    uint32_t* rec_ord;        // doc ordinal (0xFFFFFFFF = empty slot)
    uint32_t* rec_s;
    uint32_t* rec_d;
    uint4* tile_meta;         // [rows][n_tiles][2]: {weakest emitted sparse composite (lo, hi), present sparse scores, all
                              //   emitted?}, {weakest emitted dense composite (lo, hi), docs of the tile, all emitted?}
    uint32_t n_tiles;         // tiles of the index
    uint32_t stride;          // record slots per (row, tile): 2 x quota_full (the union of two candidate sets)
    uint32_t quota_full;      // candidates a full tile emits per side
    uint32_t quota_last;      // ... and the (shorter) last tile
    uint32_t* flags;          // [nq] hybrid_fuse_query: 1 = the candidate lists did not cover a depth list (second round)
    unsigned long long* fq_stamps = nullptr;  // diagnostic (MSR_DEBUG_HYBRID): phase clocks of hybrid_fuse_query
};

// The rare paths of hybrid_tiles, kept OUT OF LINE: inlined, the general selections' live ranges cost the common path
// ~100 spilled VGPRs (measured: 316 -> 20 bytes of scratch per lane). `keys` is the thread-interleaved tile of u32 keys
// (LDS accumulators, or the query's row of dense scores — f32 bit patterns — in global memory); dense scores become
// order-preserving keys, canonicalised like the fast path's floats (-0 = +0), docs past the corpus are masked out.
template <int TILE_DOCS, int NT>
__device__ __attribute__((noinline)) uint64_t threshold_general(const uint32_t* keys, int rounds, uint32_t ndocs, uint32_t k,
                                                                bool dense, uint32_t* hist, uint64_t* cand, HistScratch* hs,
                                                                uint32_t tid) {
    const HistResult g = hist_threshold<TILE_DOCS, NT>(
        [&](int r) {
            uint4 x = r < rounds ? reinterpret_cast<const uint4*>(keys)[r * NT + tid] : make_uint4(0, 0, 0, 0);
            if (dense) {
                uint32_t k4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
                    k4[e] = (r < rounds && local < ndocs) ? f32_to_key(__uint_as_float(k4[e]) + 0.0f) : 0u;
                }
                x = make_uint4(k4[0], k4[1], k4[2], k4[3]);
            }
            return x;
        },
        k, hist, cand, *hs, tid,
        [dense](uint32_t key, uint32_t lo) { return dense ? key_to_f32(key) - key_to_f32(lo) : (float)(key - lo); });
    return g.T > 1 ? g.T : ((uint64_t)g.kmin << 13);  // (everything present belongs: the smallest key, any doc)
}

template <int TILE_DOCS, int NT, int CAND>
__device__ __attribute__((noinline)) void tile_select_general(const uint4* a4, uint64_t* cand, uint32_t* tmax, uint32_t* wmax,
                                                              SelectScratch* ss, int rounds, int k, uint64_t* out,
                                                              uint32_t tid) {
    tile_select<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, *ss, rounds, 0ull, k, out, [](int) {}, tid);
}

// scratch of the dual selection (separate from the tile's LDS carve, which accumulate_tile and tile_select own)
struct DualScratch {
    uint32_t red[8];      // smax, ~smin', scount, key(dmax), ~key(dmin), member count, -, -   (LDS atomics, one per wave)
    uint32_t wtot[2][8];  // per-side wave totals of the bin scan
    uint32_t bin[2], above[2], cnt[2], ncand[2];
    uint64_t T[2];        // threshold composites  key << 13 | (TILE_DOCS - 1 - local)
    uint32_t fbin, fabove, fcnt, fn;  // final top-k: bin of the k-th best fused score, members above it, in it, collected
};

// MODE 0: the fused single-tile kernel described above, grid = (queries).
// MODE 1: multi-tile indexes, grid = (launch rows, tiles) like score_tiles: steps 1-3 for ONE tile with the tile's
//   candidate QUOTA in the place of the depth, then every element at or above the two quota thresholds is written to the
//   (row, tile) candidate lists; hybrid_fuse_query (below) finishes the query. Exact by verification: the quota is the
//   tile's expected share of a depth list plus five standard deviations; a query whose depth-th best lies below some
//   tile's weakest candidate is flagged and repeated with quota = depth (every tile's own top-depth: always enough).
template <int TILE_DOCS, int NT, int U, int MIN_WAVES, bool DBG = false, int MODE = 0>
__global__ __launch_bounds__(NT, MIN_WAVES) void hybrid_tiles(const ScoreArgs a, const HybridArgs h) {
    constexpr int CAND = 1024;
    constexpr int E = TILE_DOCS / NT, R = E / 4, NW = NT / 64, HW = NW / 2;
    constexpr int BPT = kHistBins / (HW * 64);  // bins per scanning thread (half of the workgroup scans each side)
    static_assert(BPT == 4 || BPT == 8, "the scan reads its bins as one or two 16-byte vectors");
    using L = TileLds<TILE_DOCS, NT, CAND>;
    static_assert(L::kUnion >= 2 * kHistBins * 4 && L::kUnion >= 2 * kHistCand * 8, "two histograms / candidate lists");
    static_assert(L::kTmax >= (int)sizeof(HistScratch), "the fallback selection's scratch lives in the maxima region");
    __shared__ __attribute__((aligned(16))) uint8_t lds[L::kTotal];
    __shared__ DualScratch ds;
    __shared__ uint64_t res[64];
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    uint8_t* const un = lds + L::kAcc;
    uint64_t* const cand = reinterpret_cast<uint64_t*>(un);
    uint32_t* const tmax = reinterpret_cast<uint32_t*>(un + L::kUnion);
    uint32_t* const wmax = tmax + NT;
    SelectScratch& ss = *reinterpret_cast<SelectScratch*>(un + L::kUnion + L::kTmax);
    uint32_t* const hist = reinterpret_cast<uint32_t*>(un);            // [2][kHistBins]: sparse, dense
    uint64_t* const cands = reinterpret_cast<uint64_t*>(un);           // [2][kHistCand] once the scan is over
    HistScratch& hs = *reinterpret_cast<HistScratch*>(tmax);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = rfl(tid >> 6);
    // diagnostic instance only (MSR_DEBUG_HYBRID): wave 0 of one workgroup in 64 adds its clock deltas per phase
    long long t_prev = 0;
    auto stamp = [&](int slot) {
        if (DBG && a.stamps && tid == 0 && (blockIdx.x & 63u) == 0) {
            const long long now = clock64();
            if (slot >= 0) atomicAdd(&a.stamps[slot], (unsigned long long)(now - t_prev));
            t_prev = now;
        }
    };
    stamp(-1);
    const uint32_t q = (MODE == 1 && h.qlist) ? h.qlist[blockIdx.x] : a.q0 + blockIdx.x;
    const uint32_t tile_l = MODE == 1 ? blockIdx.y : 0u;  // (the hybrid paths take whole indexes: local tile = global tile)
    const uint64_t doc0 = (uint64_t)tile_l * TILE_DOCS;
    const uint32_t ndocs = MODE == 1 ? (uint32_t)min((uint64_t)TILE_DOCS, a.n_docs - doc0) : (uint32_t)a.n_docs;
    const int rounds = (int)((ndocs + 4 * NT - 1) / (4 * NT));
    uint4* const a4 = reinterpret_cast<uint4*>(acc);
    if (tid < 8) ds.red[tid] = 0;  // (ordered before their first use by accumulate_tile's barriers)
    if (tid < 2) ds.ncand[tid] = 0;
    if (tid == 0) ds.fn = 0;
    accumulate_tile<TILE_DOCS, NT, U, false>(a, q, tile_l, rounds, lds, ss, [](int) {}, tid);
    stamp(0);

    // ---- the query's row of dense scores (the GEMM wrote the f32 values in ORDINAL order) in registers; docs past the
    // corpus become NaN: min / max skip them, every comparison with them is false
    float df[E];
    {
        const uint4* row = reinterpret_cast<const uint4*>(h.dkeys + (uint64_t)blockIdx.x * h.ld + doc0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 x = r < rounds ? row[r * NT + tid] : make_uint4(0, 0, 0, 0);
            const uint32_t k4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
                // (+ 0.0f: -0 and +0 are one score; they would be two keys)
                df[4 * r + e] = (r < rounds && local < ndocs) ? __uint_as_float(k4[e]) + 0.0f : __builtin_nanf("");
            }
        }
    }
    // zero both histograms (the staging arrays of the accumulation lived here)
    for (int i = tid; i < 2 * kHistBins / 4; i += NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
    // ---- pass A: present sparse scores (count, max, min) and dense scores (max, min)
    {
        uint32_t smx = 0, smn1 = 0xFFFFFFFFu, scnt = 0;
        float dmx = -__builtin_inff(), dmn = __builtin_inff();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < rounds) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t s4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    smx = max(smx, s4[e]);
                    smn1 = min(smn1, s4[e] - 1u);  // an absent score (0) wraps to the largest value
                    scnt += min(s4[e], 1u);
                    dmx = fmaxf(dmx, df[4 * r + e]);
                    dmn = fminf(dmn, df[4 * r + e]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        smx = wave_max_u32(smx);
        smn1 = wave_max_u32(~smn1);
        scnt = wave_sum_u32(scnt);
        const uint32_t kx = wave_max_u32(f32_to_key(dmx)), kn = wave_max_u32(~f32_to_key(dmn));
        if (lane == 0) {
            atomicMax(&ds.red[0], smx);
            atomicMax(&ds.red[1], smn1);
            atomicAdd(&ds.red[2], scnt);
            atomicMax(&ds.red[3], kx);
            atomicMax(&ds.red[4], kn);
        }
    }
    __syncthreads();
    const uint32_t smax_u = ds.red[0], n_s = ds.red[2];
    const uint32_t smin_u = n_s ? ~ds.red[1] + 1u : 0u;
    const uint32_t n_d = ndocs;  // every doc has a dense score
    const float dmax_f = key_to_f32(ds.red[3]), dmin_all = key_to_f32(~ds.red[4]);
    // list sizes: the depth best (MODE 1: the tile's quota of candidates), or everything that is present
    const uint32_t want = MODE == 1 ? (tile_l + 1 == h.n_tiles ? h.quota_last : h.quota_full) : h.depth;
    const uint32_t need_s = min(want, n_s), need_d = min(want, n_d);
    // ---- pass B: histograms, linear between the smallest and the largest present value of each side (an absent
    // sparse score, 0, saturates to bin 0, a doc past the corpus, NaN, converts to bin 0: harmless, the chosen bin's
    // candidates are filtered by presence). Branch-free: one fused multiply-add,
    // one conversion and one LDS add per element and side.
    const float sc_s = ((float)kHistBins - 0.5f) / (float)(smax_u - smin_u);
    const float sc_d = ((float)kHistBins - 0.5f) / (dmax_f - dmin_all);
    const bool flat_s = smax_u == smin_u, flat_d = !(dmax_f > dmin_all);  // all equal: no spread to bin on
    const uint32_t one_s = flat_s ? 0u : 1u, one_d = flat_d ? 0u : 1u;
    auto pack_u16 = [](uint32_t lo, uint32_t hi) -> uint32_t {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, (us2)__builtin_amdgcn_cvt_pk_u16(lo, hi));  // v_cvt_pk_u16_u32 saturates
    };
    uint32_t bins[E];  // sparse bin | dense bin << 16
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r < rounds) {
            const uint4 x = a4[r * NT + tid];
            const uint32_t s4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (x - min) * scale, not fma(x, scale, -min * scale): the rounded offset of a tight cluster far from 0
                // (sparse 2e7 +- 300, dense 138.5469 +- 3e-5) puts the top bin past the array; clamped besides
                const uint32_t bs = min((uint32_t)(((float)s4[e] - (float)smin_u) * sc_s), (uint32_t)(kHistBins - 1));  // absent (0) -> 0
                const uint32_t bd = min((uint32_t)((df[4 * r + e] - dmin_all) * sc_d), (uint32_t)(kHistBins - 1));      // NaN -> 0
                atomicAdd(&hist[flat_s ? 0u : bs], one_s);
                atomicAdd(&hist[kHistBins + (flat_d ? 0u : bd)], one_d);
                bins[4 * r + e] = pack_u16(bs, bd);
            }
            __builtin_amdgcn_sched_barrier(0);  // one round at a time: hoisting all four rounds' loads costs ~50 spills
        }
    __syncthreads();
    // ---- scan: waves [0, HW) walk the sparse bins from the top, waves [HW, NW) the dense bins; the thread whose bins
    // hold the need-th best publishes {bin, elements above it, elements in it}
    {
        const uint32_t side = wave >= (uint32_t)HW ? 1u : 0u;
        const uint32_t t2 = tid - side * (HW * 64);  // thread index inside its half
        const uint32_t need = side ? need_d : need_s;
        const uint32_t* hb_base = hist + side * kHistBins + (kHistBins - BPT) - BPT * t2;  // bins ascending in memory
        uint32_t hb[BPT];  // descending bin order
        {
            const uint4 v = *reinterpret_cast<const uint4*>(hb_base + (BPT - 4));
            hb[0] = v.w, hb[1] = v.z, hb[2] = v.y, hb[3] = v.x;
            if (BPT == 8) {
                const uint4 u = *reinterpret_cast<const uint4*>(hb_base);
                hb[BPT - 4] = u.w, hb[BPT - 3] = u.z, hb[BPT - 2] = u.y, hb[BPT - 1] = u.x;
            }
        }
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) s += hb[i];
        const uint32_t inc = wave_inclusive_scan_u32(s);
        if (lane == 63) ds.wtot[side][wave - side * HW] = inc;
        __syncthreads();
        uint32_t above = inc - s;
#pragma unroll
        for (int w = 0; w < HW; ++w) above += (uint32_t)w < wave - side * HW ? ds.wtot[side][w] : 0u;
        if (above < need && need <= above + s) {  // exactly one thread per side (none when the side is flat / empty)
            uint32_t b = 0, ab = above;
#pragma unroll
            for (int i = 0; i < BPT - 1; ++i)
                if (b == (uint32_t)i && ab + hb[i] < need) {
                    ab += hb[i];
                    b = (uint32_t)i + 1;
                }
            uint32_t cnt = hb[0];
#pragma unroll
            for (int i = 1; i < BPT; ++i) cnt = b == (uint32_t)i ? hb[i] : cnt;
            ds.bin[side] = (uint32_t)(kHistBins - 1) - (BPT * t2 + b);
            ds.above[side] = ab;
            ds.cnt[side] = cnt;
        }
    }
    __syncthreads();
    // ---- pass C: the chosen bins' elements as composites  key << 13 | (TILE_DOCS - 1 - local)  (ties: lower ordinal)
    const uint32_t bsel_s = ds.bin[0], bsel_d = ds.bin[1], cnt_s = ds.cnt[0], cnt_d = ds.cnt[1];
    const bool fast_s = !flat_s && n_s != 0 && cnt_s <= (uint32_t)kHistCand;
    const bool fast_d = !flat_d && cnt_d <= (uint32_t)kHistCand;
    {
        // which of this thread's elements sit in a chosen bin: bit j (sparse), bit 16 + j (dense); almost always none
        const uint32_t want = pack_u16(fast_s ? bsel_s : 0xFFFEu, fast_d ? bsel_d : 0xFFFEu);
        uint32_t hit = 0;
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t x = bins[j] ^ want;  // a zero half = that side's bin matches
            hit |= ((x & 0xFFFFu) == 0 ? 1u : 0u) << j;
            hit |= ((x >> 16) == 0 ? 1u : 0u) << (16 + j);
        }
        if (hit) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < rounds) {
                    const uint4 x = a4[r * NT + tid];
                    const uint32_t s4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = 4 * r + e;
                        const uint64_t inv = (uint64_t)(TILE_DOCS - 1 - (4u * ((uint32_t)r * NT + tid) + (uint32_t)e));
                        if ((hit >> j & 1u) && s4[e] != 0)
                            cands[atomicAdd(&ds.ncand[0], 1u)] = ((uint64_t)s4[e] << 13) | inv;
                        if ((hit >> (16 + j) & 1u) && df[j] == df[j])
                            cands[kHistCand + atomicAdd(&ds.ncand[1], 1u)] = ((uint64_t)f32_to_key(df[j]) << 13) | inv;
                    }
                }
        }
    }
    __syncthreads();
    // ---- exact rank inside the chosen bin: one wave per side while the bin holds <= 64 elements
    auto rank_side = [&](uint32_t side, uint32_t cnt, uint32_t need_in_bin) {
        const uint64_t* c = cands + side * kHistCand;
        if (cnt <= 64) {
            if (wave == side * HW) {
                const uint64_t me = lane < cnt ? c[lane] : 0ull;
                const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
                uint32_t rank = 0;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint64_t o = ((uint64_t)rdl(hi, i) << 32) | rdl(lo, i);
                    rank += o > me;
                }
                if (lane < cnt && rank == need_in_bin - 1) ds.T[side] = me;
            }
        } else {
            for (uint32_t i = tid; i < cnt; i += NT) {
                const uint64_t me = c[i];
                uint32_t rank = 0;
                for (uint32_t o = 0; o < cnt; ++o) rank += c[o] > me;
                if (rank == need_in_bin - 1) ds.T[side] = me;
            }
        }
    };
    // (the number actually collected: bin 0's count also holds the absent elements that were binned there)
    if (fast_s) rank_side(0, ds.ncand[0], need_s - ds.above[0]);
    if (fast_d) rank_side(1, ds.ncand[1], need_d - ds.above[1]);
    __syncthreads();
    uint64_t T_s = ds.T[0], T_d = ds.T[1];
    // ---- rare: a side without spread, or a bin with more than kHistCand elements (mass ties): the general selection
    if (__builtin_expect(n_s != 0 && !fast_s, 0))  // (uniform, rare)
        T_s = threshold_general<TILE_DOCS, NT>(acc, rounds, ndocs, want, false, hist, cand + kHistBins / 2, &hs, tid);
    if (__builtin_expect(!fast_d, 0))
        T_d = threshold_general<TILE_DOCS, NT>(h.dkeys + (uint64_t)blockIdx.x * h.ld + doc0, rounds, ndocs, want, true, hist,
                                               cand + kHistBins / 2, &hs, tid);
    stamp(1);
    if constexpr (MODE == 1) {
        // ---- the tile's candidates: every element at or above its side's threshold composite (exactly need_s / need_d
        // of them: composites are unique) becomes ONE record {ordinal, sparse score or 0, dense key or 0} — a doc that
        // is a candidate on both sides arrives at hybrid_fuse_query already joined
        __syncthreads();  // (everyone has read ds.ncand / ds.T of the selection)
        if (tid < 2) ds.ncand[tid] = 0;
        __syncthreads();
        const uint64_t slot0 = ((uint64_t)blockIdx.x * h.n_tiles + tile_l) * h.stride;
        uint32_t* const ro = h.rec_ord + slot0;
        uint32_t* const rs = h.rec_s + slot0;
        uint32_t* const rd = h.rec_d + slot0;
        const uint32_t ts_key = (uint32_t)(T_s >> 13), ts_inv = (uint32_t)T_s & 8191u;
        const uint32_t td_inv = (uint32_t)T_d & 8191u;
        const float td_f = key_to_f32((uint32_t)(T_d >> 13)) + 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < rounds) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t s4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
                    const uint32_t inv = (uint32_t)(TILE_DOCS - 1) - local;
                    const float d = df[4 * r + e];
                    const bool is_s = need_s && s4[e] != 0 && (s4[e] > ts_key || (s4[e] == ts_key && inv >= ts_inv));
                    const bool is_d = d > td_f || (d == td_f && inv >= td_inv);  // (NaN = past the corpus: never)
                    if (is_s || is_d) {
                        const uint32_t pos = atomicAdd(&ds.ncand[0], 1u);
                        if (pos < h.stride) {
                            ro[pos] = (uint32_t)(doc0 + local);
                            rs[pos] = is_s ? s4[e] : 0u;
                            rd[pos] = is_d ? f32_to_key(d) : 0u;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();
        for (uint32_t i = min(ds.ncand[0], h.stride) + tid; i < h.stride; i += NT) {  // empty slots
            ro[i] = 0xFFFFFFFFu;
            rs[i] = 0u;
            rd[i] = 0u;
        }
        if (tid == 0) {
            // weakest emitted element of each side as a global composite; "all emitted" = the list holds everything the
            // tile has (no threshold to respect)
            const uint32_t tl_s = (uint32_t)(TILE_DOCS - 1) - ts_inv, tl_d = (uint32_t)(TILE_DOCS - 1) - td_inv;
            const uint64_t ws = ((uint64_t)ts_key << 32) | (0xFFFFFFFFu - (uint32_t)(doc0 + tl_s));
            const uint64_t wd = ((uint64_t)(uint32_t)(T_d >> 13) << 32) | (0xFFFFFFFFu - (uint32_t)(doc0 + tl_d));
            uint4* m = h.tile_meta + ((uint64_t)blockIdx.x * h.n_tiles + tile_l) * 2;
            m[0] = make_uint4((uint32_t)ws, (uint32_t)(ws >> 32), n_s, need_s == n_s ? 1u : 0u);
            m[1] = make_uint4((uint32_t)wd, (uint32_t)(wd >> 32), n_d, need_d == n_d ? 1u : 0u);
        }
        stamp(2);
        return;
    }
    // ---- fusion (src/hybrid.py:32-53): min = the last member's score, max = the best, per side; fused keys go to the
    // accumulator tile (each thread rewrites only what it has read) and into a histogram over [fmin, fmax]
    const uint32_t ts_key = (uint32_t)(T_s >> 13), ts_inv = (uint32_t)T_s & 8191u;
    const uint32_t td_inv = (uint32_t)T_d & 8191u;
    const float td_f = key_to_f32((uint32_t)(T_d >> 13)) + 0.0f;
    const float smin = n_s ? (float)ts_key : 0.f, smax = n_s ? (float)smax_u : 0.f;
    const float dmin = td_f, dmax = dmax_f;
    const float sden = fmaxf(smax - smin, 1e-9f), dden = fmaxf(dmax - dmin, 1e-9f);
    const uint32_t self = h.self_ord ? (uint32_t)h.self_ord[q] : 0xFFFFFFFFu;
    // fused scores lie in [min(w,0) sums, max(w,0) sums]
    const float f_lo = fminf(h.w_dense, 0.f) + fminf(h.w_sparse, 0.f), f_hi = fmaxf(h.w_dense, 0.f) + fmaxf(h.w_sparse, 0.f);
    const float sc_f = ((float)kHistBins - 0.5f) / fmaxf(f_hi - f_lo, 1e-30f), of_f = -f_lo * sc_f;
    // w * ((x - min) / den) with the division as a multiplication by 1 / den (<= 1 ulp from the quotient, far inside the
    // 1e-5 the fused scores are held to) — but EXACTLY 1 at x = max, as the quotient is: with alpha = 0.5 the best
    // dense-only doc and the best sparse-only doc tie at exactly 0.5 in the reference, and that tie must stay a tie
    // (it is then broken like every other: lower ordinal first)
    const float inv_dden = 1.0f / dden, inv_sden = 1.0f / sden;
    const bool spread_d = dmax > dmin, spread_s = smax > smin;  // (max == min: every quotient is 0 / 1e-9 = 0)
    for (int i = tid; i < kHistBins / 4; i += NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
    }
    __syncthreads();  // histogram zeroed (the candidate lists that lived there are dead)
    uint32_t n_mem = 0;  // this thread's docs in the union of the two lists (without the query's own doc)
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (r < rounds) {
            const uint4 x = a4[r * NT + tid];
            const uint32_t s4[4] = {x.x, x.y, x.z, x.w};
            uint32_t fk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
                const uint32_t inv = (uint32_t)(TILE_DOCS - 1) - local;
                const float d = df[4 * r + e];
                const bool in_d = d > td_f || (d == td_f && inv >= td_inv);
                const bool in_s = s4[e] != 0 && (s4[e] > ts_key || (s4[e] == ts_key && inv >= ts_inv));
                float f = 0.f;  // the reference adds the dense term first (runs = [dense, sparse], src/search.py:459)
                if (in_d) f += h.w_dense * ((d == dmax && spread_d) ? 1.0f : (d - dmin) * inv_dden);
                if (in_s) f += h.w_sparse * ((s4[e] == smax_u && spread_s) ? 1.0f : ((float)s4[e] - smin) * inv_sden);
                const bool member = (in_d || in_s) && local != self;
                fk[e] = member ? f32_to_key(f) : 0u;
                if (member) {
                    atomicAdd(&hist[(uint32_t)__builtin_fmaf(f, sc_f, of_f)], 1u);
                    ++n_mem;
                }
            }
            a4[r * NT + tid] = make_uint4(fk[0], fk[1], fk[2], fk[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    n_mem = wave_sum_u32(n_mem);
    if (lane == 0) atomicAdd(&ds.red[5], n_mem);
    __syncthreads();
    stamp(2);
    // ---- top-k of the fused scores: the bin of the k-th best, then every member at or above that bin is ranked
    const uint32_t n_members = ds.red[5];
    const uint32_t need_f = min(h.k, n_members);
    {
        const bool scans = wave < (uint32_t)HW;  // the first half of the workgroup walks the bins
        const uint32_t t2 = scans ? tid : 0u;
        const uint32_t* hb_base = hist + (kHistBins - BPT) - BPT * t2;
        uint32_t hb[BPT];
        {
            const uint4 v = *reinterpret_cast<const uint4*>(hb_base + (BPT - 4));
            hb[0] = v.w, hb[1] = v.z, hb[2] = v.y, hb[3] = v.x;
            if (BPT == 8) {
                const uint4 u = *reinterpret_cast<const uint4*>(hb_base);
                hb[BPT - 4] = u.w, hb[BPT - 3] = u.z, hb[BPT - 2] = u.y, hb[BPT - 1] = u.x;
            }
        }
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < BPT; ++i) s += hb[i];
        const uint32_t inc = wave_inclusive_scan_u32(s);
        if (scans && lane == 63) ds.wtot[0][wave] = inc;
        __syncthreads();
        uint32_t above = inc - s;
#pragma unroll
        for (int w = 0; w < HW; ++w) above += (uint32_t)w < wave ? ds.wtot[0][w] : 0u;
        if (scans && need_f != 0 && above < need_f && need_f <= above + s) {
            uint32_t b = 0, ab = above;
#pragma unroll
            for (int i = 0; i < BPT - 1; ++i)
                if (b == (uint32_t)i && ab + hb[i] < need_f) {
                    ab += hb[i];
                    b = (uint32_t)i + 1;
                }
            uint32_t cnt = hb[0];
#pragma unroll
            for (int i = 1; i < BPT; ++i) cnt = b == (uint32_t)i ? hb[i] : cnt;
            const uint32_t fb = (uint32_t)(kHistBins - 1) - (BPT * tid + b);
            ds.fbin = fb;
            ds.fabove = ab;
            ds.fcnt = cnt + (fb ? hist[fb - 1] : 0u);  // ... and the bin below it: the collection's margin (see there)
        }
    }
    __syncthreads();
    // Members at or above the k-th best's bin, plus the bin below: the fused keys are re-read from the tile (nothing is
    // kept in registers across the scan) and compared with the KEY of the lower edge of bin fbin - 1 — one whole bin of
    // margin against the rounding of the bin function, so that every member of bins >= fbin is collected.
    const uint32_t n_top_est = need_f ? ds.fabove + ds.fcnt : 0u;
    uint64_t* const top = reinterpret_cast<uint64_t*>(tmax);  // 64 keys
    bool ranked = false;
    if (__builtin_expect(n_top_est <= 60, 1)) {
        const uint32_t fbin = ds.fbin;
        const uint32_t key_edge = fbin > 1 ? f32_to_key(((float)fbin - 1.0f) / sc_f + f_lo) : 1u;
        if (need_f) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < rounds) {
                    const uint4 x = a4[r * NT + tid];
                    const uint32_t k4[4] = {x.x, x.y, x.z, x.w};
                    if (max(max(k4[0], k4[1]), max(k4[2], k4[3])) >= key_edge) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (k4[e] != 0 && k4[e] >= key_edge) {
                                const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
                                const uint32_t pos = atomicAdd(&ds.fn, 1u);
                                if (pos < 64) top[pos] = ((uint64_t)k4[e] << 32) | (uint64_t)(0xFFFFFFFFu - local);
                            }
                    }
                }
        }
        __syncthreads();
        ranked = ds.fn <= 64;  // (a key on the very edge can add one or two to the estimate; more than 64: general path)
    }
    if (__builtin_expect(ranked, 1)) {
        const uint32_t n_top = ds.fn;
        if (tid < 64) {
            const uint64_t me = tid < n_top ? top[tid] : 0ull;
            const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
            uint32_t rank = 0;
            for (uint32_t i = 0; i < n_top; ++i) {
                const uint64_t o = ((uint64_t)rdl(hi, i) << 32) | rdl(lo, i);
                rank += o > me;
            }
            const uint64_t o = (uint64_t)q * h.k;
            if (tid < n_top && rank < h.k) {
                h.out_ord[o + rank] = 0xFFFFFFFFu - (uint32_t)me;
                h.out_score[o + rank] = key_to_f32((uint32_t)(me >> 32));
            }
            if (tid >= need_f && tid < h.k) {
                h.out_ord[o + tid] = 0xFFFFFFFFu;
                h.out_score[o + tid] = 0.f;
            }
            if (tid == 0) h.out_n[q] = (int32_t)need_f;
        }
        stamp(3);
        return;
    }
    // ---- rare (many equal fused scores around the k-th place): the general tile selection over the fused keys
    tile_select_general<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, &ss, rounds, (int)h.k, res, tid);
    __syncthreads();
    stamp(4);
    if (tid < h.k) {
        const uint64_t key = res[tid];
        const uint64_t o = (uint64_t)q * h.k + tid;
        h.out_ord[o] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
        h.out_score[o] = key ? key_to_f32((uint32_t)(key >> 32)) : 0.f;
    }
    if (tid < 64) {
        const uint32_t nh = (uint32_t)__popcll(__ballot(tid < h.k && res[tid] != 0));
        if (tid == 0) h.out_n[q] = (int32_t)nh;
    }
}

// ------------------------------------------------------------------------------------------------ multi-tile: per query
// need-th largest of the workgroup's non-zero u64 keys (unique; 0 = empty slot; each(f) calls f(key) for every key of the
// calling thread — registers, LDS or global memory, the caller's choice), by radix passes of 8 bits over the bits
// BELOW the keys' common prefix (scores of one list differ in their low 20 bits or so: the pass that matters comes first;
// a pass ends the search as soon as the chosen bin holds <= 64 keys, which are then ranked by one wave). Returns T with
// "key belongs to the need best  <=>  key != 0 and key >= T";  *present = non-zero keys, *kmax = the largest.
// need == 0: T = ~0 (nothing belongs); need >= present: T = the smallest key (everything belongs).
struct KthScratch {
    uint32_t hist[256];
    uint64_t wmin[16], wmax[16];
    uint32_t wcnt[16];
    uint64_t small[64];
    uint32_t n_small, pad;
    uint64_t T;
};

template <int NT, class Each>
__device__ __forceinline__ uint64_t kth_largest_u64(Each each, uint32_t need, KthScratch& ks, const uint32_t tid,
                                                    uint32_t* present, uint64_t* kmax) {
    constexpr int NW = NT / 64;
    const uint32_t lane = tid & 63, wave = rfl(tid >> 6);
    uint64_t mn = ~0ull, mx = 0;
    uint32_t c = 0;
    each([&](const uint64_t key) {
        if (key) {
            mn = key < mn ? key : mn;
            mx = key > mx ? key : mx;
            ++c;
        }
    });
    mn = ~wave_max_u64(~mn);
    mx = wave_max_u64(mx);
    c = wave_sum_u32(c);
    __syncthreads();  // previous users of the scratch are done
    if (lane == 0) {
        ks.wmin[wave] = mn;
        ks.wmax[wave] = mx;
        ks.wcnt[wave] = c;
    }
    if (tid == 0) ks.n_small = 0;
    __syncthreads();
    mn = ~0ull, mx = 0, c = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        mn = ks.wmin[w] < mn ? ks.wmin[w] : mn;
        mx = ks.wmax[w] > mx ? ks.wmax[w] : mx;
        c += ks.wcnt[w];
    }
    *present = c;
    *kmax = mx;
    if (need == 0 || c == 0) return ~0ull;
    if (need >= c) return mn;
    // (two or more keys, all different: mx != mn)
    const int top = 63 - __clzll((long long)(mx ^ mn));
    int shift = max(top - 7, 0);
    uint64_t hi_mask = shift + 8 >= 64 ? 0ull : ~0ull << (shift + 8);
    uint64_t prefix = mx & hi_mask;
#pragma unroll 1
    for (;;) {
        ks.hist[tid & 255u] = 0;  // (NT >= 256)
        __syncthreads();
        each([&](const uint64_t key) {
            if (key && (key & hi_mask) == prefix) atomicAdd(&ks.hist[(uint32_t)(key >> shift) & 255u], 1u);
        });
        __syncthreads();
        // every wave finds the bin of the need-th largest for itself: lane l owns bins 4l .. 4l+3
        const uint4 hb = reinterpret_cast<const uint4*>(ks.hist)[lane];
        const uint32_t sum4 = hb.x + hb.y + hb.z + hb.w;
        const uint32_t inc = wave_inclusive_scan_u32(sum4);
        const uint32_t suf = rdl(inc, 63) - (inc - sum4);  // keys in bins >= 4 * lane: non-increasing in lane
        const uint32_t L = 63u - (uint32_t)__clzll((long long)__ballot(suf >= need));
        uint32_t above = rdl(suf, L) - rdl(sum4, L);
        const uint32_t b3 = rdl(hb.w, L), b2 = rdl(hb.z, L), b1 = rdl(hb.y, L), b0 = rdl(hb.x, L);
        uint32_t bin = 4 * L + 3, cnt = b3;
        if (above + b3 < need) {
            above += b3, bin = 4 * L + 2, cnt = b2;
            if (above + b2 < need) {
                above += b2, bin = 4 * L + 1, cnt = b1;
                if (above + b1 < need) above += b1, bin = 4 * L, cnt = b0;
            }
        }
        need -= above;  // rank inside the chosen bin, 1-based from the top
        prefix |= (uint64_t)bin << shift;
        hi_mask |= 255ull << shift;
        if (cnt <= 64 || shift == 0) {
            each([&](const uint64_t key) {
                if (key && (key & hi_mask) == prefix) ks.small[atomicAdd(&ks.n_small, 1u)] = key;
            });
            __syncthreads();
            if (tid < 64) {
                const uint64_t me = lane < cnt ? ks.small[lane] : 0ull;
                const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
                uint32_t rank = 0;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint64_t o = ((uint64_t)rdl(hi, i) << 32) | rdl(lo, i);
                    rank += o > me;
                }
                if (lane < cnt && rank == need - 1) ks.T = me;
            }
            __syncthreads();
            return ks.T;
        }
        shift = max(shift - 8, 0);  // (the last window may overlap bits that are already fixed: harmless)
        __syncthreads();            // everyone has read the histogram
    }
}

// Multi-tile indexes, second kernel: ONE workgroup per query finishes what hybrid_tiles<MODE 1> prepared. The query's
// candidate records (a few thousand: 4 tiles x 836 at the reference's own shape) live in REGISTERS from the first load to
// the last pass: the depth-th best of each side among them (exact when every tile's weakest candidate lies at or below
// it: checked, else the query is flagged for the second round), the reference's fusion over the union of the two lists
// (src/hybrid.py:32-53: min = the depth-th score, max = the best, of the UNFILTERED lists, src/search.py:76-81; the
// query's own doc skipped, src/search.py:72-74) — a record carries both of its doc's scores, so there is nothing to
// join — and the top-k of the fused scores (ties: lower ordinal). k <= 1024, depth <= 1024.
template <int NT, int EPT /* records per thread held in registers */>
__global__ __launch_bounds__(NT) void hybrid_fuse_query(const HybridArgs h, const uint32_t q0) {
    static_assert(NT >= 256 && NT % 64 == 0 && NT / 64 <= 16, "scratch layout");
    __shared__ __attribute__((aligned(16))) uint64_t cand[kCandCap];
    __shared__ __attribute__((aligned(16))) uint64_t res[kCandCap];
    __shared__ __attribute__((aligned(16))) KthScratch ks;
    __shared__ uint32_t red[4];             // present sparse scores, docs, invalid flag, collected keys
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t row = blockIdx.x;
    const uint32_t q = h.qlist ? h.qlist[row] : q0 + row;
    const uint32_t n_slots = h.n_tiles * h.stride;
    const uint32_t* const g_ord = h.rec_ord + (uint64_t)row * n_slots;
    const uint32_t* const g_s = h.rec_s + (uint64_t)row * n_slots;
    const uint32_t* const g_d = h.rec_d + (uint64_t)row * n_slots;
    const uint4* const meta = h.tile_meta + (uint64_t)row * h.n_tiles * 2;
    long long t_prev = h.fq_stamps ? clock64() : 0;
    auto stamp = [&](int slot) {  // thread 0 of one workgroup in 16 adds its clock deltas per phase
        if (h.fq_stamps && tid == 0 && (blockIdx.x & 15u) == 0) {
            const long long now = clock64();
            atomicAdd(&h.fq_stamps[slot], (unsigned long long)(now - t_prev));
            t_prev = now;
        }
    };
    // Read in a loop from global memory, every pass below would pay a dependent L2 round trip per element (measured with
    // the first version of this kernel: 0.47 ms instead of 0.3 for 5 000 queries); queries with more records than the
    // registers hold (second round on many tiles) take that loop.
    const bool in_regs = n_slots <= (uint32_t)(EPT * NT);  // (uniform)
    uint32_t ro[EPT], rs[EPT], rd[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = tid + (uint32_t)e * NT;
        const bool live = in_regs && i < n_slots;
        ro[e] = live ? g_ord[i] : 0xFFFFFFFFu;
        rs[e] = live ? g_s[i] : 0u;
        rd[e] = live ? g_d[i] : 0u;
    }
    if (tid < 4) red[tid] = 0;
    __syncthreads();
    {
        uint32_t ns = 0, nd = 0;
        for (uint32_t t = tid; t < h.n_tiles; t += NT) {
            ns += meta[2 * t].z;
            nd += meta[2 * t + 1].z;
        }
        ns = wave_sum_u32(ns);
        nd = wave_sum_u32(nd);
        if (lane == 0 && (ns | nd)) {
            atomicAdd(&red[0], ns);
            atomicAdd(&red[1], nd);
        }
    }
    __syncthreads();
    const uint32_t need_s = min(h.depth, red[0]), need_d = min(h.depth, red[1]);
    // visit(f): f(ordinal, sparse score, dense key) for every record of this thread
    auto visit = [&](auto f) {
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) f(ro[e], rs[e], rd[e]);
        } else {
            for (uint32_t i = tid; i < n_slots; i += NT) f(g_ord[i], g_s[i], g_d[i]);
        }
    };
    auto comp = [](uint32_t v, uint32_t ord) -> uint64_t { return v ? ((uint64_t)v << 32) | (uint64_t)(0xFFFFFFFFu - ord) : 0ull; };
    auto each_s = [&](auto f) { visit([&](uint32_t o, uint32_t sv, uint32_t) { f(comp(sv, o)); }); };
    auto each_d = [&](auto f) { visit([&](uint32_t o, uint32_t, uint32_t dv) { f(comp(dv, o)); }); };
    uint32_t c_s, c_d;
    uint64_t mx_s, mx_d;
    stamp(0);
    const uint64_t T_s = kth_largest_u64<NT>(each_s, need_s, ks, tid, &c_s, &mx_s);
    stamp(1);
    const uint64_t T_d = kth_largest_u64<NT>(each_d, need_d, ks, tid, &c_d, &mx_d);
    stamp(2);
    // ---- are the lists complete? a tile that did not emit everything it has must have its weakest candidate at or
    // below the side's depth-th best
    {
        bool bad = c_s < need_s || c_d < need_d;
        for (uint32_t t = tid; t < h.n_tiles; t += NT) {
            const uint4 a = meta[2 * t], b = meta[2 * t + 1];
            if (need_s && !a.w && T_s < (((uint64_t)a.y << 32) | a.x)) bad = true;
            if (need_d && !b.w && T_d < (((uint64_t)b.y << 32) | b.x)) bad = true;
        }
        if (bad) red[2] = 1;  // (benign race: every writer stores 1)
    }
    __syncthreads();
    if (red[2]) {  // (uniform) second round: hybrid_tiles<MODE 1> with quota = depth for this query
        if (tid == 0) {
            h.flags[q] = 1;
            h.out_n[q] = 0;
        }
        return;
    }
    stamp(3);
    // ---- fusion (src/hybrid.py:32-53; the dense term first: runs = [dense, sparse], src/search.py:459)
    const uint32_t self = h.self_ord ? (uint32_t)h.self_ord[q] : 0xFFFFFFFFu;
    const uint32_t smax_u = (uint32_t)(mx_s >> 32), smin_u = need_s ? (uint32_t)(T_s >> 32) : 0u;
    const float smax = need_s ? (float)smax_u : 0.f, smin = (float)smin_u;
    const float dmax = need_d ? key_to_f32((uint32_t)(mx_d >> 32)) + 0.0f : 0.f;
    const float dmin = need_d ? key_to_f32((uint32_t)(T_d >> 32)) + 0.0f : 0.f;
    const float inv_sden = 1.0f / fmaxf(smax - smin, 1e-9f), inv_dden = 1.0f / fmaxf(dmax - dmin, 1e-9f);
    const bool spread_s = smax > smin, spread_d = dmax > dmin;
    auto fused_key = [&](uint32_t o, uint32_t sv, uint32_t dv) -> uint64_t {
        const bool in_s = need_s && sv && comp(sv, o) >= T_s;
        const bool in_d = need_d && dv && comp(dv, o) >= T_d;
        if (!(in_s || in_d) || o == self) return 0ull;
        float f = 0.f;
        if (in_d) {  // (EXACTLY 1 at the maximum, as the reference's quotient is: see hybrid_tiles)
            const float d = key_to_f32(dv) + 0.0f;
            f += h.w_dense * ((d == dmax && spread_d) ? 1.0f : (d - dmin) * inv_dden);
        }
        if (in_s) f += h.w_sparse * ((sv == smax_u && spread_s) ? 1.0f : ((float)sv - smin) * inv_sden);
        return ((uint64_t)f32_to_key(f) << 32) | (uint64_t)(0xFFFFFFFFu - o);
    };
    // the fused keys replace the records in the registers (16 x u64 for 48 x u32)
    uint64_t fk[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) fk[e] = in_regs ? fused_key(ro[e], rs[e], rd[e]) : 0ull;
    auto each_f = [&](auto f) {
        if (in_regs) {
#pragma unroll
            for (int e = 0; e < EPT; ++e) f(fk[e]);
        } else {
            for (uint32_t i = tid; i < n_slots; i += NT) f(fused_key(g_ord[i], g_s[i], g_d[i]));
        }
    };
    stamp(4);
    // ---- top-k of the fused scores over the union (ties: lower ordinal)
    uint32_t members;
    uint64_t mx_f;
    const uint64_t T_f = kth_largest_u64<NT>(each_f, h.k, ks, tid, &members, &mx_f);
    stamp(5);
    const uint32_t n_out = min(h.k, members);
    each_f([&](const uint64_t key) {
        if (key && key >= T_f) {
            const uint32_t pos = atomicAdd(&red[3], 1u);
            if (pos < (uint32_t)kCandCap) cand[pos] = key;
        }
    });
    __syncthreads();
    rank_and_emit<NT, true>(cand, (int)n_out, (int)h.k, res, tid);
    __syncthreads();
    for (uint32_t i = tid; i < h.k; i += NT) {
        const uint64_t key = res[i];
        const uint64_t o = (uint64_t)q * h.k + i;
        h.out_ord[o] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
        h.out_score[o] = key ? key_to_f32((uint32_t)(key >> 32)) : 0.f;
    }
    if (tid == 0) {
        h.out_n[q] = (int32_t)n_out;
        h.flags[q] = 0;
    }
    stamp(6);
}

// rows of the passage matrix in ORDINAL order: P_ord[row2ord[r]] = P[r]
__global__ __launch_bounds__(256) void permute_rows(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                    const uint32_t* __restrict__ row2ord, uint32_t vecs_per_row) {
    const uint32_t r = blockIdx.x;
    const uint4* s = src + (uint64_t)r * vecs_per_row;
    uint4* d = dst + (uint64_t)row2ord[r] * vecs_per_row;
    for (uint32_t i = threadIdx.x; i < vecs_per_row; i += 256) d[i] = s[i];
}

}  // namespace msr

// Passage rows in doc-ORDINAL order (P_ord[row2ord[r]] = P[r]), so that a query's row of GEMM output lines up with the
// sparse accumulator tiles; built on first use, cached on the dense handle while the mapping is unchanged.
static int ensure_ordinal_rows(msr_dense* dx, hipStream_t stream, const uint32_t* row2ord, const uint64_t n) {
    if (!dx->d_P_ord || dx->ord_map.size() != n || memcmp(dx->ord_map.data(), row2ord, (size_t)n * 4) != 0) {
        std::vector<uint8_t> seen((size_t)n, 0);
        for (uint64_t r = 0; r < n; ++r) {
            if (row2ord[r] >= n || seen[row2ord[r]]) {
                set_error("row2ord is not a permutation of the doc ordinals (row %llu -> %u)", (unsigned long long)r, row2ord[r]);
                return MSR_E_INVAL;
            }
            seen[row2ord[r]] = 1;
        }
        uint32_t* d_map = nullptr;
        const size_t bytes = (size_t)dx->n_pad * dx->h * 2;
        if (!dx->d_P_ord && hipMalloc(&dx->d_P_ord, bytes) != hipSuccess) {
            dx->d_P_ord = nullptr;
            set_error("hipMalloc of %zu bytes for the ordinal-ordered passage matrix failed", bytes);
            return MSR_E_NOMEM;
        }
        bool ok = hipMalloc(&d_map, std::max<size_t>((size_t)n, 1) * 4) == hipSuccess &&
                  hipMemsetAsync(dx->d_P_ord, 0, bytes, stream) == hipSuccess &&
                  hipMemcpyAsync(d_map, row2ord, (size_t)n * 4, hipMemcpyHostToDevice, stream) == hipSuccess;
        if (ok && n) {
            hipLaunchKernelGGL(permute_rows, dim3((uint32_t)n), dim3(256), 0, stream, reinterpret_cast<const uint4*>(dx->d_P),
                               reinterpret_cast<uint4*>(dx->d_P_ord), d_map, dx->h / 8);
            ok = hipGetLastError() == hipSuccess;
        }
        ok = ok && hipStreamSynchronize(stream) == hipSuccess;
        if (d_map) (void)hipFree(d_map);
        if (!ok) {
            dx->ord_map.clear();
            set_error("building the ordinal-ordered passage matrix failed: %s", hipGetErrorString(hipGetLastError()));
            return MSR_E_HIP;
        }
        dx->ord_map.assign(row2ord, row2ord + n);
    }
    return MSR_OK;
}

// The fused path of msr_hybrid_search (single-tile indexes, k <= 64): per chunk of queries (normally ONE chunk) a GEMM
// launch on the ordinal-permuted passage matrix, then hybrid_tiles. ms = {fused scoring + selection + fusion kernel, dense GEMM, 0, 0}.
static int hybrid_search_fused(msr_index* ix, msr_dense* dx, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w,
                               const uint16_t* q_fp16, int nq, int depth, int k, float alpha, uint32_t flags,
                               const uint32_t* row2ord, const int32_t* self_ord, uint32_t* out_ord, float* out_score,
                               int32_t* out_n, float ms[4]) {
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    const uint64_t n = h->n_docs;
    HIP_TRY(hipSetDevice(d->device));
    // diagnostic (MSR_DEBUG_HYBRID): host-side laps of the call, printed with the kernel span
    static const bool dbg_host = getenv("MSR_DEBUG_HYBRID") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    char laps[512];
    size_t laps_n = 0;
    laps[0] = 0;
    auto lap = [&](const char* what) {
        if (!dbg_host) return;
        const auto now = std::chrono::steady_clock::now();
        const double msx = std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
        if (laps_n < sizeof(laps)) laps_n += (size_t)snprintf(laps + laps_n, sizeof(laps) - laps_n, " %s %.2f", what, msx);
    };
    if (int rc0 = ensure_ordinal_rows(dx, d->stream, row2ord, n)) return rc0;
    lap("ordinal map");
    msr_batch* b = nullptr;
    int rc = msr_batch_create(ix, q_ptr, q_term, q_w, nq, k, flags, &b);
    if (rc != MSR_OK) return rc;
    lap("sparse batch");
    const uint32_t nt_threads = h->tile_docs == 4096 ? 256u : 512u;
    const uint64_t ld = (n + 4 * nt_threads - 1) / (4 * nt_threads) * (4 * nt_threads);  // whole select rounds
    const uint64_t n_cover = std::min<uint64_t>(dx->n_pad, (n + 255) / 256 * 256);
    const uint32_t col_blocks = (uint32_t)(n_cover / 256);
    // queries per chunk: everything when the score rows fit 2 GiB (one GEMM launch: the least tail and launch gaps; the
    // round trip of the rows through HBM is ~0.25 GB per 6 400 queries — noise next to either kernel), else chunks of
    // whole rounds of the chip's 256 CUs worth of 256 x 256 blocks
    const uint32_t nq_pad = (uint32_t)((nq + 255) / 256 * 256);
    // queries per chunk: whole rounds of the chip's 256 CUs worth of 256 x 256 GEMM blocks, score rows <= ~160 MB so
    // that hybrid_tiles finds its row in the Infinity Cache (the row load is a dependent fetch at the start of the
    // epilogue: from HBM the same kernel measured 1.66 ms instead of 1.30 ms per 25 010 queries)
    static const uint64_t chunk_mb = getenv("MSR_HYBRID_CHUNK_MB") ? strtoull(getenv("MSR_HYBRID_CHUNK_MB"), nullptr, 0) : 160ull;  // diagnostic
    uint64_t row_blocks = std::max<uint64_t>(1, (std::max<uint64_t>(chunk_mb, 1) << 20) / (ld * 4 * 256));
    if (col_blocks && row_blocks * col_blocks >= 256) row_blocks = row_blocks * col_blocks / 256 * 256 / col_blocks;
    const uint32_t qc = (uint32_t)std::min<uint64_t>(row_blocks * 256, std::max<uint32_t>(nq_pad, 256u));
    _Float16* d_Q = nullptr;
    uint32_t* d_S = nullptr;
    uint32_t* d_ord = nullptr;
    float* d_sf = nullptr;
    int32_t *d_n = nullptr, *d_self = nullptr;
    std::vector<hipEvent_t> ev;
    unsigned long long* d_stamps = nullptr;  // diagnostic (MSR_DEBUG_HYBRID): phase clocks of wave 0
    if (getenv("MSR_DEBUG_HYBRID") && hipMalloc(&d_stamps, 8 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemsetAsync(d_stamps, 0, 8 * sizeof(unsigned long long), d->stream);
    const size_t perk = std::max<size_t>((size_t)nq * k, 1);
    d_Q = (_Float16*)dx->take(msr_dense::S_Q, std::max<size_t>((size_t)nq_pad * dx->h * 2, 16));
    d_S = (uint32_t*)dx->take(msr_dense::S_S, (size_t)qc * ld * 4);
    d_ord = (uint32_t*)dx->take(msr_dense::S_ORD, perk * 4);
    d_sf = (float*)dx->take(msr_dense::S_SF, perk * 4);
    d_n = (int32_t*)dx->take(msr_dense::S_N, std::max<size_t>(nq, 1) * 4);
    bool ok = d_Q && d_S && d_ord && d_sf && d_n;
    if (ok && self_ord)
        ok = (d_self = (int32_t*)dx->take(msr_dense::S_SELF, std::max<size_t>(nq, 1) * 4)) != nullptr &&
             hipMemcpyAsync(d_self, self_ord, (size_t)nq * 4, hipMemcpyHostToDevice, d->stream) == hipSuccess;
    if (ok && nq)
        ok = hipMemsetAsync(d_Q + (size_t)nq * dx->h, 0, (size_t)(nq_pad - nq) * dx->h * 2, d->stream) == hipSuccess &&
             hipMemcpyAsync(d_Q, q_fp16, (size_t)nq * dx->h * 2, hipMemcpyHostToDevice, d->stream) == hipSuccess;
    if (!ok) {
        set_error("device allocation or query upload failed in msr_hybrid_search (%u queries per chunk)", qc);
        rc = MSR_E_NOMEM;
    }
    ScoreArgs sa;
    sa.seg_ptr = d->d_seg_ptr;
    sa.postings = d->d_postings;
    sa.q_meta = reinterpret_cast<const uint4*>(b->d_qptr);
    sa.q_term = b->d_qterm;
    sa.q_w = b->d_qw;
    sa.dense = d->d_dense;
    sa.q_dense = b->d_qdense;
    sa.n_pairs = d->n_pairs;
    sa.part = nullptr;
    sa.theta = nullptr;
    sa.unsorted = 0;
    sa.n_docs = n;
    sa.vec_base = d->vec_base;
    sa.n_terms = d->seg_terms;
    sa.tile0 = 0;
    sa.tl0 = 0;
    sa.nq = (uint32_t)nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.tpr = 1;
    sa.dump_add = 0;
    sa.dbg = 0;
    sa.stamps = nullptr;
    sa.light = 0;
    HybridArgs ha;
    ha.dkeys = d_S;
    ha.ld = ld;
    ha.self_ord = d_self;
    ha.depth = (uint32_t)depth;
    ha.k = (uint32_t)k;
    ha.w_dense = alpha;
    ha.w_sparse = 1.0f - alpha;
    ha.out_ord = d_ord;
    ha.out_score = d_sf;
    ha.out_n = d_n;
    static const bool inner_events = getenv("MSR_HYBRID_NO_INNER_EVENTS") == nullptr;
    lap("alloc + enqueue uploads");
    hipEvent_t e_all[2] = {nullptr, nullptr};
    if (rc == MSR_OK && (hipEventCreate(&e_all[0]) != hipSuccess || hipEventCreate(&e_all[1]) != hipSuccess)) {
        set_error("hipEventCreate failed");
        rc = MSR_E_HIP;
    }
    if (rc == MSR_OK) (void)hipEventRecord(e_all[0], d->stream);
    for (uint32_t q0 = 0; q0 < (uint32_t)nq && rc == MSR_OK; q0 += qc) {
        const uint32_t qn = std::min<uint32_t>(qc, (uint32_t)nq - q0);
        const uint32_t qn_pad = (qn + 255) / 256 * 256;
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        for (auto& x : e) {
            if (hipEventCreate(&x) != hipSuccess) rc = MSR_E_HIP;
            ev.push_back(x);
        }
        if (rc != MSR_OK) {
            set_error("hipEventCreate failed");
            break;
        }
        if (inner_events) (void)hipEventRecord(e[0], d->stream);
        rc = launch_dense_gemm(dx, dx->d_P_ord, d_Q + (size_t)q0 * dx->h, d_S, qn, qn_pad, n_cover, ld, d->stream, true, true);
        if (rc != MSR_OK) break;
        if (inner_events) (void)hipEventRecord(e[1], d->stream);
        sa.q0 = q0;
        sa.qn = qn;
        sa.stamps = d_stamps;
        if (d_stamps && h->tile_docs == 8192)
            hipLaunchKernelGGL((hybrid_tiles<8192, 512, 4, 6, true>), dim3(qn), dim3(512), 0, d->stream, sa, ha);
        else if (h->tile_docs == 4096)
            hipLaunchKernelGGL((hybrid_tiles<4096, 256, 4, 5>), dim3(qn), dim3(256), 0, d->stream, sa, ha);
        else
            hipLaunchKernelGGL((hybrid_tiles<8192, 512, 4, 6>), dim3(qn), dim3(512), 0, d->stream, sa, ha);
        if (hipGetLastError() != hipSuccess) {
            set_error("hybrid_tiles launch failed: %s", hipGetErrorString(hipGetLastError()));
            rc = MSR_E_HIP;
            break;
        }
        if (inner_events) (void)hipEventRecord(e[2], d->stream);
    }
    if (rc == MSR_OK) (void)hipEventRecord(e_all[1], d->stream);
    if (rc == MSR_OK && hipStreamSynchronize(d->stream) != hipSuccess) {
        set_error("hybrid kernels failed: %s", hipGetErrorString(hipGetLastError()));
        rc = MSR_E_HIP;
    }
    float t_gemm = 0, t_fused = 0, t_all = 0;
    if (rc == MSR_OK) (void)hipEventElapsedTime(&t_all, e_all[0], e_all[1]);
    if (getenv("MSR_DEBUG_HYBRID")) fprintf(stderr, "[msr] hybrid pipeline span %.3f ms (inner events %d)\n", t_all, (int)inner_events);
    lap("uploads + kernels");
    if (rc == MSR_OK) {
        // (only the per-chunk laps depend on the inner events; the results are downloaded either way)
        for (size_t i = 0; inner_events && i + 2 < ev.size(); i += 3) {
            float a = 0, c = 0;
            (void)hipEventElapsedTime(&a, ev[i], ev[i + 1]);
            (void)hipEventElapsedTime(&c, ev[i + 1], ev[i + 2]);
            t_gemm += a;
            t_fused += c;
        }
        if (!inner_events) t_fused = t_all;  // the span is all there is: reported as the fused kernel's slot
        if (nq && (hipMemcpy(out_ord, d_ord, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                   hipMemcpy(out_score, d_sf, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                   hipMemcpy(out_n, d_n, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
            set_error("download failed in msr_hybrid_search");
            rc = MSR_E_HIP;
        }
    }
    lap("download");
    if (ms) {
        ms[0] = t_fused;
        ms[1] = t_gemm;
        ms[2] = 0.f;
        ms[3] = 0.f;
    }
    if (d_stamps) {
        unsigned long long st[8] = {0};
        (void)hipMemcpy(st, d_stamps, sizeof(st), hipMemcpyDeviceToHost);
        fprintf(stderr, "[msr] hybrid_tiles wave-0 clocks: accumulate %llu, sparse select %llu, dense select %llu, fuse %llu, "
                        "top-k %llu (sums over 1 workgroup in 64)\n", st[0], st[1], st[2], st[3], st[4]);
        (void)hipFree(d_stamps);
    }
    for (hipEvent_t x : ev)
        if (x) (void)hipEventDestroy(x);
    for (hipEvent_t x : e_all)
        if (x) (void)hipEventDestroy(x);
    // (d_Q, d_S, d_ord, d_sf, d_n, d_self stay with the dense handle for the next call)
    batch_free(b);
    lap("free");
    if (dbg_host) fprintf(stderr, "[msr] hybrid host laps (ms):%s\n", laps);
    return rc;
}

// Candidate quota of a tile that holds `tile_docs` of the corpus's `n` docs: its expected share of a depth list plus
// five standard deviations (docs are numbered by id string, so a tile's share of any query's best docs is a draw without
// replacement) plus a constant; never more than the depth. A query for which some tile needed more is repeated with
// quota = depth by the caller, so the result never depends on this number.
static uint32_t tile_quota(uint64_t tile_docs, uint64_t n, uint32_t depth) {
    const char* env = getenv("MSR_HYBRID_QUOTA");  // diagnostic / tests (read per call: a test flips it)
    const long forced = env ? atol(env) : 0;
    if (forced > 0) return (uint32_t)std::min<long>(forced, depth);
    if (n == 0 || tile_docs >= n) return depth;
    const double share = (double)tile_docs / (double)n, mean = depth * share;
    const double q = mean + 5.0 * std::sqrt(mean * (1.0 - share)) + 16.0;
    return (uint32_t)std::min<double>(depth, std::ceil(q));
}

// Multi-tile indexes (and k > 64 on one tile): per chunk of queries a GEMM launch on the ordinal-permuted passage matrix,
// hybrid_tiles<MODE 1> over (query, tile) and hybrid_fuse_query over the queries; flagged queries (normally none) are
// repeated with quota = depth. ms = {candidate kernel, dense GEMM, 0, fusion kernel}.
static int hybrid_search_multitile(msr_index* ix, msr_dense* dx, const int64_t* q_ptr, const int32_t* q_term,
                                   const int32_t* q_w, const uint16_t* q_fp16, int nq, int depth, int k, float alpha,
                                   uint32_t flags, const uint32_t* row2ord, const int32_t* self_ord, uint32_t* out_ord,
                                   float* out_score, int32_t* out_n, float ms[4]) {
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    const uint64_t n = h->n_docs;
    const uint32_t tile = h->tile_docs, n_tiles = h->n_tiles;
    HIP_TRY(hipSetDevice(d->device));
    if (int rc0 = ensure_ordinal_rows(dx, d->stream, row2ord, n)) return rc0;
    msr_batch* b = nullptr;
    int rc = msr_batch_create(ix, q_ptr, q_term, q_w, nq, 1, flags, &b);  // (only the query arrays are used)
    if (rc != MSR_OK) return rc;
    const uint64_t ld = (uint64_t)n_tiles * tile;  // a query's score row: tile t starts at column t * tile
    const uint64_t n_cover = std::min<uint64_t>(dx->n_pad, (n + 255) / 256 * 256);
    const uint32_t col_blocks = (uint32_t)(n_cover / 256);
    const uint32_t nq_pad = (uint32_t)((nq + 255) / 256 * 256);
    // queries per chunk: score rows of ~160 MB (they are read back out of the Infinity Cache), whole rounds of the
    // chip's 256 CUs worth of GEMM blocks
    static const uint64_t chunk_mb = getenv("MSR_HYBRID_CHUNK_MB") ? strtoull(getenv("MSR_HYBRID_CHUNK_MB"), nullptr, 0) : 160ull;
    // (the bytes that travel are the n_cover written columns of a row, not its padded stride). Among the chunk sizes
    // that fit, the one whose GEMM grid fills its last round of 256 CUs best: 2 x 98 blocks leave a quarter of the chip
    // idle, 5 x 98 = 490 fill 1.91 rounds
    const uint64_t rb_max = std::max<uint64_t>(1, (std::max<uint64_t>(chunk_mb, 1) << 20) / (std::max<uint64_t>(n_cover, 256) * 4 * 256));
    const uint64_t row_blocks = pick_row_blocks(col_blocks, rb_max, nq_pad / 256);
    const uint32_t qc = (uint32_t)std::min<uint64_t>(row_blocks * 256, std::max<uint32_t>(nq_pad, 256u));
    const uint64_t last_docs = n - (uint64_t)(n_tiles - 1) * tile;
    const uint32_t quota_full = n_tiles > 1 ? tile_quota(tile, n, (uint32_t)depth) : tile_quota(n, n, (uint32_t)depth);
    const uint32_t quota_last = n_tiles > 1 ? std::min(quota_full, tile_quota(last_docs, n, (uint32_t)depth)) : quota_full;
    // candidate slots: a chunk of the first round (qc rows x quota_full) or at least ONE row of the second (depth)
    const size_t per_tile_row = (size_t)n_tiles;
    // record slots (a tile's union of two candidate sets: 2 x quota): a chunk of the first round or at least ONE row of the second
    const size_t cand_slots = 2 * std::max<size_t>((size_t)qc * per_tile_row * quota_full, per_tile_row * (size_t)depth);
    const size_t perk = std::max<size_t>((size_t)nq * k, 1);
    _Float16* d_Q = (_Float16*)dx->take(msr_dense::S_Q, std::max<size_t>((size_t)nq_pad * dx->h * 2, 16));
    uint32_t* d_S = (uint32_t*)dx->take(msr_dense::S_S, (size_t)qc * ld * 4);
    uint32_t* d_ord = (uint32_t*)dx->take(msr_dense::S_ORD, perk * 4);
    float* d_sf = (float*)dx->take(msr_dense::S_SF, perk * 4);
    int32_t* d_n = (int32_t*)dx->take(msr_dense::S_N, std::max<size_t>(nq, 1) * 4);
    uint32_t* d_rec = (uint32_t*)dx->take(msr_dense::S_CS, cand_slots * 12);  // ordinals | sparse scores | dense keys
    uint4* d_meta = (uint4*)dx->take(msr_dense::S_META, std::max<size_t>((size_t)qc * per_tile_row * 2 * 16, 32));
    uint32_t* d_flags = (uint32_t*)dx->take(msr_dense::S_FLAGS, std::max<size_t>(nq, 1) * 4);
    uint32_t* d_qlist = (uint32_t*)dx->take(msr_dense::S_QLIST, std::max<size_t>(nq, 1) * 4);
    int32_t* d_self = nullptr;
    bool ok = d_Q && d_S && d_ord && d_sf && d_n && d_rec && d_meta && d_flags && d_qlist;
    if (ok && self_ord)
        ok = (d_self = (int32_t*)dx->take(msr_dense::S_SELF, std::max<size_t>(nq, 1) * 4)) != nullptr &&
             hipMemcpyAsync(d_self, self_ord, (size_t)nq * 4, hipMemcpyHostToDevice, d->stream) == hipSuccess;
    if (ok && nq)
        ok = hipMemsetAsync(d_Q + (size_t)nq * dx->h, 0, (size_t)(nq_pad - nq) * dx->h * 2, d->stream) == hipSuccess &&
             hipMemcpyAsync(d_Q, q_fp16, (size_t)nq * dx->h * 2, hipMemcpyHostToDevice, d->stream) == hipSuccess;
    if (!ok) {
        set_error("device allocation or query upload failed in msr_hybrid_search (%u queries per chunk, %u tiles)", qc, n_tiles);
        rc = MSR_E_NOMEM;
    }
    ScoreArgs sa;
    sa.seg_ptr = d->d_seg_ptr;
    sa.postings = d->d_postings;
    sa.q_meta = reinterpret_cast<const uint4*>(b->d_qptr);
    sa.q_term = b->d_qterm;
    sa.q_w = b->d_qw;
    sa.dense = d->d_dense;
    sa.q_dense = b->d_qdense;
    sa.n_pairs = d->n_pairs;
    sa.part = nullptr;
    sa.theta = nullptr;
    sa.unsorted = 0;
    sa.n_docs = n;
    sa.vec_base = d->vec_base;
    sa.n_terms = d->seg_terms;
    sa.tile0 = 0;
    sa.tl0 = 0;
    sa.nq = (uint32_t)nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.tpr = 1;
    sa.dump_add = 0;
    sa.dbg = 0;
    sa.stamps = nullptr;
    sa.light = 0;
    HybridArgs ha;
    ha.dkeys = d_S;
    ha.ld = ld;
    ha.self_ord = d_self;
    ha.depth = (uint32_t)depth;
    ha.k = (uint32_t)k;
    ha.w_dense = alpha;
    ha.w_sparse = 1.0f - alpha;
    ha.out_ord = d_ord;
    ha.out_score = d_sf;
    ha.out_n = d_n;
    ha.qlist = nullptr;
    ha.rec_ord = d_rec;
    ha.rec_s = d_rec + cand_slots;
    ha.rec_d = d_rec + 2 * cand_slots;
    ha.tile_meta = d_meta;
    ha.n_tiles = n_tiles;
    ha.stride = 2 * quota_full;
    ha.quota_full = quota_full;
    ha.quota_last = quota_last;
    ha.flags = d_flags;
    unsigned long long* d_fq = nullptr;
    if (getenv("MSR_DEBUG_HYBRID") && hipMalloc(&d_fq, 16 * sizeof(unsigned long long)) == hipSuccess) {
        (void)hipMemsetAsync(d_fq, 0, 16 * sizeof(unsigned long long), d->stream);
        ha.fq_stamps = d_fq;
    }
    // one pass of the three kernels over `rows` launch rows (queries q0 .. or the listed ones), dense rows from d_Qrows
    auto launch_rows = [&](const _Float16* d_Qrows, uint32_t rows, uint32_t q0, hipEvent_t* e) -> int {
        const uint32_t rows_pad = (rows + 255) / 256 * 256;
        if (e) (void)hipEventRecord(e[0], d->stream);
        int r2 = launch_dense_gemm(dx, dx->d_P_ord, d_Qrows, d_S, rows, rows_pad, n_cover, ld, d->stream, true, true);
        if (r2 != MSR_OK) return r2;
        if (e) (void)hipEventRecord(e[1], d->stream);
        sa.q0 = q0;
        sa.qn = rows;
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += kMaxGridY) {  // (grid y limit; tile_l = blockIdx.y needs t0 == 0)
            if (t0) {
                set_error("msr_hybrid_search: more than %u tiles", kMaxGridY);
                return MSR_E_RANGE;
            }
            const dim3 grid(rows, n_tiles);
            sa.stamps = reinterpret_cast<unsigned long long*>(ha.fq_stamps ? ha.fq_stamps + 8 : nullptr);
            if (tile == 4096)
                hipLaunchKernelGGL((hybrid_tiles<4096, 256, 4, 5, false, 1>), grid, dim3(256), 0, d->stream, sa, ha);
            else if (sa.stamps)  // diagnostic (MSR_DEBUG_HYBRID): wave 0 of one workgroup in 64 adds its phase clocks
                hipLaunchKernelGGL((hybrid_tiles<8192, 512, 4, 6, true, 1>), grid, dim3(512), 0, d->stream, sa, ha);
            else
                hipLaunchKernelGGL((hybrid_tiles<8192, 512, 4, 6, false, 1>), grid, dim3(512), 0, d->stream, sa, ha);
        }
        if (hipGetLastError() != hipSuccess) {
            set_error("hybrid_tiles launch failed: %s", hipGetErrorString(hipGetLastError()));
            return MSR_E_HIP;
        }
        if (e) (void)hipEventRecord(e[2], d->stream);
        // (a query's records stay in registers while they fit: 16 per thread)
        if ((uint64_t)ha.n_tiles * ha.stride <= 256u * 16u)
            hipLaunchKernelGGL((hybrid_fuse_query<256, 16>), dim3(rows), dim3(256), 0, d->stream, ha, q0);
        else
            hipLaunchKernelGGL((hybrid_fuse_query<512, 16>), dim3(rows), dim3(512), 0, d->stream, ha, q0);
        if (hipGetLastError() != hipSuccess) {
            set_error("hybrid_fuse_query launch failed: %s", hipGetErrorString(hipGetLastError()));
            return MSR_E_HIP;
        }
        if (e) (void)hipEventRecord(e[3], d->stream);
        return MSR_OK;
    };
    std::vector<hipEvent_t> ev;
    auto four_events = [&]() -> hipEvent_t* {
        const size_t at = ev.size();
        for (int i = 0; i < 4; ++i) {
            hipEvent_t x = nullptr;
            if (!d->spare_events.empty()) {
                x = d->spare_events.back();
                d->spare_events.pop_back();
            } else if (hipEventCreate(&x) != hipSuccess) {
                x = nullptr;
            }
            ev.push_back(x);
        }
        for (size_t i = at; i < ev.size(); ++i)
            if (!ev[i]) return nullptr;
        return ev.data() + at;
    };
    ev.reserve(4 * ((size_t)nq / std::max<uint32_t>(qc, 1) + 4) + 64);  // (pointers into ev stay valid)
    for (uint32_t q0 = 0; q0 < (uint32_t)nq && rc == MSR_OK; q0 += qc) {
        const uint32_t qn = std::min<uint32_t>(qc, (uint32_t)nq - q0);
        hipEvent_t* e = four_events();
        if (!e) {
            set_error("hipEventCreate failed");
            rc = MSR_E_HIP;
            break;
        }
        rc = launch_rows(d_Q + (size_t)q0 * dx->h, qn, q0, e);
    }
    std::vector<uint32_t> hflags((size_t)nq, 0);
    if (rc == MSR_OK && nq &&
        (hipMemcpyAsync(hflags.data(), d_flags, (size_t)nq * 4, hipMemcpyDeviceToHost, d->stream) != hipSuccess ||
         hipStreamSynchronize(d->stream) != hipSuccess)) {
        set_error("hybrid kernels failed: %s", hipGetErrorString(hipGetLastError()));
        rc = MSR_E_HIP;
    }
    // ---- second round (normally empty): the flagged queries again, every tile emitting its own top-depth
    std::vector<uint32_t> redo;
    for (int i = 0; i < nq && rc == MSR_OK; ++i)
        if (hflags[(size_t)i]) redo.push_back((uint32_t)i);
    static const bool dbg = getenv("MSR_DEBUG_HYBRID") != nullptr;
    if (dbg) fprintf(stderr, "[msr] hybrid multi-tile: %u tiles, quota %u / %u of depth %d, %u queries per chunk, %zu of %d "
                             "queries repeated with quota = depth\n", n_tiles, quota_full, quota_last, depth, qc, redo.size(), nq);
    if (rc == MSR_OK && !redo.empty()) {
        const uint32_t rows_cap = (uint32_t)std::min<size_t>(std::min<size_t>(cand_slots / (2 * per_tile_row * (size_t)depth), qc), redo.size());
        std::vector<uint16_t> qrows((size_t)rows_cap * dx->h);
        ha.quota_full = ha.quota_last = (uint32_t)depth;
        ha.stride = 2 * (uint32_t)depth;
        ha.qlist = d_qlist;
        for (size_t at = 0; at < redo.size() && rc == MSR_OK; at += rows_cap) {
            const uint32_t rows = (uint32_t)std::min<size_t>(rows_cap, redo.size() - at);
            const uint32_t rows_pad = (rows + 255) / 256 * 256;
            for (uint32_t r = 0; r < rows; ++r)
                memcpy(qrows.data() + (size_t)r * dx->h, q_fp16 + (size_t)redo[at + r] * dx->h, (size_t)dx->h * 2);
            // (d_Q's first rows are reused: the first round is over)
            ok = hipMemsetAsync(d_Q, 0, (size_t)rows_pad * dx->h * 2, d->stream) == hipSuccess &&
                 hipMemcpyAsync(d_Q, qrows.data(), (size_t)rows * dx->h * 2, hipMemcpyHostToDevice, d->stream) == hipSuccess &&
                 hipMemcpyAsync(d_qlist, redo.data() + at, (size_t)rows * 4, hipMemcpyHostToDevice, d->stream) == hipSuccess;
            if (!ok) {
                set_error("upload failed in the second round of msr_hybrid_search");
                rc = MSR_E_HIP;
                break;
            }
            hipEvent_t* e = ev.size() + 4 <= ev.capacity() ? four_events() : nullptr;
            rc = launch_rows(d_Q, rows, 0, e);
            if (rc == MSR_OK && hipStreamSynchronize(d->stream) != hipSuccess) {  // (qrows / redo are re-filled next)
                set_error("hybrid kernels failed in the second round: %s", hipGetErrorString(hipGetLastError()));
                rc = MSR_E_HIP;
            }
        }
        if (rc == MSR_OK) {  // with quota = depth every list is complete by construction
            (void)hipMemcpy(hflags.data(), d_flags, (size_t)nq * 4, hipMemcpyDeviceToHost);
            for (uint32_t i : redo)
                if (hflags[i]) {
                    set_error("internal error: query %u still incomplete after the second round", i);
                    rc = MSR_E_HIP;
                    break;
                }
        }
    }
    float t_gemm = 0, t_cand = 0, t_fuse = 0;
    if (rc == MSR_OK) {
        for (size_t i = 0; i + 3 < ev.size(); i += 4) {
            float a = 0, c = 0, f = 0;
            (void)hipEventElapsedTime(&a, ev[i], ev[i + 1]);
            (void)hipEventElapsedTime(&c, ev[i + 1], ev[i + 2]);
            (void)hipEventElapsedTime(&f, ev[i + 2], ev[i + 3]);
            t_gemm += a;
            t_cand += c;
            t_fuse += f;
        }
        if (nq && (hipMemcpy(out_ord, d_ord, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                   hipMemcpy(out_score, d_sf, (size_t)nq * k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                   hipMemcpy(out_n, d_n, (size_t)nq * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
            set_error("download failed in msr_hybrid_search");
            rc = MSR_E_HIP;
        }
    }
    if (ms) {
        ms[0] = t_cand;
        ms[1] = t_gemm;
        ms[2] = 0.f;
        ms[3] = t_fuse;
    }
    if (d_fq) {
        unsigned long long st[16] = {0};
        (void)hipMemcpy(st, d_fq, sizeof(st), hipMemcpyDeviceToHost);
        fprintf(stderr, "[msr] hybrid_tiles<MODE 1> wave-0 clocks (1 workgroup in 64): accumulate + dense row %llu, dual quota "
                        "selection %llu, emission %llu\n", st[8], st[9], st[10]);
        fprintf(stderr, "[msr] hybrid_fuse_query thread-0 clocks (1 workgroup in 16): load+init %llu, sparse depth-th %llu, dense "
                        "depth-th %llu, verify %llu, fusion %llu, fused k-th %llu, collect+rank+store %llu\n",
                st[0], st[1], st[2], st[3], st[4], st[5], st[6]);
        (void)hipFree(d_fq);
    }
    for (hipEvent_t x : ev) {
        if (!x) continue;
        if (d->spare_events.size() < 64)
            d->spare_events.push_back(x);
        else
            (void)hipEventDestroy(x);
    }
    batch_free(b);
    return rc;
}

extern "C" {

int msr_hybrid_search(msr_index* ix, msr_dense* dx, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w,
                      const uint16_t* q_fp16, int nq, int depth, int k, float alpha, uint32_t flags,
                      const uint32_t* row2ord, const int32_t* self_ord, uint32_t* out_ord, float* out_score, int32_t* out_n,
                      float ms[4]) {
    if (!ix || !dx || !row2ord || !out_ord || !out_score || !out_n || nq < 0) {
        set_error("msr_hybrid_search: bad argument");
        return MSR_E_INVAL;
    }
    if (!ix->dev) {
        set_error("index handle has no HIP device bound; there is no CPU scoring path");
        return MSR_E_NODEVICE;
    }
    if (ix->dev->device != dx->device) {
        set_error("the sparse and the dense index live on different devices");
        return MSR_E_INVAL;
    }
    if (depth < 1 || depth > MSR_KMAX || k < 1 || k > MSR_KMAX) {
        set_error("depth and k must be in [1, %d]", MSR_KMAX);
        return MSR_E_RANGE;
    }
    if (dx->n != ix->host.h->n_docs) {
        set_error("the dense index holds %llu rows but the sparse index %llu docs", (unsigned long long)dx->n,
                  (unsigned long long)ix->host.h->n_docs);
        return MSR_E_INVAL;
    }
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    static const bool no_fused = getenv("MSR_NO_FUSED_HYBRID") != nullptr;  // diagnostic: force the list-based path
    if (!no_fused && h->n_tiles == 1 && ix->shard_ntiles == 1 && ix->term_nshards == 0 && k <= 64 && dx->h % 8 == 0 &&
        (h->tile_docs == 4096 || h->tile_docs == 8192))
        return hybrid_search_fused(ix, dx, q_ptr, q_term, q_w, q_fp16, nq, depth, k, alpha, flags, row2ord, self_ord,
                                   out_ord, out_score, out_n, ms);
    if (!no_fused && ix->shard_ntiles == h->n_tiles && ix->term_nshards == 0 && dx->h % 8 == 0 && h->n_tiles <= kMaxGridY &&
        (h->tile_docs == 4096 || h->tile_docs == 8192))
        return hybrid_search_multitile(ix, dx, q_ptr, q_term, q_w, q_fp16, nq, depth, k, alpha, flags, row2ord, self_ord,
                                       out_ord, out_score, out_n, ms);
    msr_batch* b = nullptr;
    int rc = msr_batch_create(ix, q_ptr, q_term, q_w, nq, depth, flags, &b);
    if (rc != MSR_OK) return rc;
    uint32_t *d_didx = nullptr, *d_dkey = nullptr, *d_map = nullptr, *d_ord = nullptr, *d_su = nullptr;
    int32_t *d_dn = nullptr, *d_self = nullptr, *d_n = nullptr;
    float* d_sf = nullptr;
    uint64_t* d_part = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const uint32_t ftile = h->n_docs <= 4096 ? 4096 : 8192;
    const uint32_t ftiles = (uint32_t)std::max<uint64_t>((h->n_docs + ftile - 1) / ftile, 1);
    const size_t per = std::max<size_t>((size_t)nq * depth, 1), perk = std::max<size_t>((size_t)nq * k, 1);
    bool ok = hipSetDevice(d->device) == hipSuccess && hipMalloc(&d_didx, per * 4) == hipSuccess &&
              hipMalloc(&d_dkey, per * 4) == hipSuccess && hipMalloc(&d_dn, std::max<size_t>(nq, 1) * 4) == hipSuccess &&
              hipMalloc(&d_map, std::max<size_t>(h->n_docs, 1) * 4) == hipSuccess &&
              hipMalloc(&d_part, (size_t)ftiles * perk * 8) == hipSuccess && hipMalloc(&d_ord, perk * 4) == hipSuccess &&
              hipMalloc(&d_su, perk * 4) == hipSuccess && hipMalloc(&d_sf, perk * 4) == hipSuccess &&
              hipMalloc(&d_n, std::max<size_t>(nq, 1) * 4) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
              hipEventCreate(&e1) == hipSuccess &&
              hipMemcpy(d_map, row2ord, (size_t)h->n_docs * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && self_ord)
        ok = hipMalloc(&d_self, std::max<size_t>(nq, 1) * 4) == hipSuccess &&
             hipMemcpy(d_self, self_ord, (size_t)nq * 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        set_error("device allocation failed in msr_hybrid_search");
        rc = MSR_E_NOMEM;
    }
    float t_gemm = 0, t_sel = 0, t_sparse = 0, t_merge = 0, t_fuse = 0;
    // fuse_tiles takes both top-depth lists as SETS (min, max, membership): single-tile indexes skip the ranking
    b->unsorted_ok = true;
    if (rc == MSR_OK) rc = batch_search_local(b, depth, false);  // sparse top-depth keys -> b->d_keys
    if (rc == MSR_OK) rc = dense_search_impl(dx, q_fp16, nq, depth, d_didx, d_dkey, d_dn, &t_gemm, &t_sel, true, true);
    if (rc == MSR_OK) {
        (void)hipEventRecord(e0, d->stream);
        FuseArgs fa;
        fa.s_keys = b->d_keys;
        fa.d_idx = d_didx;
        fa.d_key = d_dkey;
        fa.d_n = d_dn;
        fa.row2ord = d_map;
        fa.self_ord = d_self;
        fa.part = d_part;
        fa.n_docs = h->n_docs;
        fa.nq = (uint32_t)nq;
        fa.depth = (uint32_t)depth;
        fa.k = (uint32_t)k;
        fa.w_dense = alpha;
        fa.w_sparse = 1.0f - alpha;
        if (nq) {
            if (ftile == 4096)
                hipLaunchKernelGGL((fuse_tiles<4096, 256, 1024>), dim3(ftiles * (uint32_t)nq), dim3(256), 0, d->stream, fa);
            else
                hipLaunchKernelGGL((fuse_tiles<8192, 512, 1024>), dim3(ftiles * (uint32_t)nq), dim3(512), 0, d->stream, fa);
        }
        MergeArgs ma;
        ma.lists = d_part;
        ma.list_stride = (uint64_t)nq * k;
        ma.n_lists = ftiles;
        ma.nq = (uint32_t)nq;
        ma.k = (uint32_t)k;
        ma.out_keys = nullptr;
        ma.out_ord = d_ord;
        ma.out_score_u32 = d_su;
        ma.out_score = d_sf;
        ma.out_n = d_n;
        rc = launch_merge(d->stream, ma);
        (void)hipEventRecord(e1, d->stream);
    }
    if (rc == MSR_OK && hipStreamSynchronize(d->stream) != hipSuccess) {
        set_error("hybrid kernels failed: %s", hipGetErrorString(hipGetLastError()));
        rc = MSR_E_HIP;
    }
    if (rc == MSR_OK) {
        (void)msr_batch_kernel_ms(b, &t_sparse, &t_merge);
        (void)hipEventElapsedTime(&t_fuse, e0, e1);
        // out_score: the fused f32 score is carried as an order-preserving key in the u32 score slot
        std::vector<uint32_t> keys((size_t)nq * k);
        bool c = (!nq) || (hipMemcpy(out_ord, d_ord, (size_t)nq * k * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                           hipMemcpy(keys.data(), d_su, (size_t)nq * k * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                           hipMemcpy(out_n, d_n, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        if (!c) {
            set_error("download failed in msr_hybrid_search");
            rc = MSR_E_HIP;
        } else {
            for (size_t i = 0; i < keys.size(); ++i) {
                const uint32_t key = keys[i];
                uint32_t bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
                float f;
                memcpy(&f, &bits, 4);
                out_score[i] = key ? f : 0.f;
            }
        }
    }
    if (ms) {
        ms[0] = t_sparse + t_merge;
        ms[1] = t_gemm;
        ms[2] = t_sel;
        ms[3] = t_fuse;
    }
    void* ptrs[] = {d_didx, d_dkey, d_dn, d_map, d_self, d_part, d_ord, d_su, d_sf, d_n};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    batch_free(b);
    return rc;
}

}  // extern "C"

// ================================================================================================ encode-side sparsifier
// The step immediately upstream of the index / the query encoder: per row of next-token logits
//     v = log(1 + relu(logit))                       src/model.py:104
//     top-k of v (k = 128 or --sparse_length)        src/encode.py:69-72
//     weight = rint(v * 100) as int                  src/encode.py:75
// One elementwise kernel turns the logits into order-preserving keys of v in the select_tiles layout; selection and
// the tile merge are the kernels of the search path. fp16_math = 1 reproduces a model that runs in fp16 (1 + relu and
// the log are rounded to half before the f32 multiplication by 100), 0 keeps f32 throughout.
namespace msr {

__global__ __launch_bounds__(256) void sparsify_keys(const void* __restrict__ logits, int is_f16, int fp16_math,
                                                     uint32_t* __restrict__ out, uint32_t V, uint64_t ld) {
    const uint32_t row = blockIdx.y;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < V; i += gridDim.x * 256) {
        float x = is_f16 ? (float)reinterpret_cast<const _Float16*>(logits)[(uint64_t)row * V + i]
                         : reinterpret_cast<const float*>(logits)[(uint64_t)row * V + i];
        float v;
        if (fp16_math) {
            const _Float16 y = (_Float16)((_Float16)1.0f + (_Float16)fmaxf(x, 0.f));  // half add, round to nearest even
            v = (float)(_Float16)logf((float)y);
        } else {
            v = logf(1.0f + fmaxf(x, 0.f));
        }
        out[(uint64_t)row * ld + i] = f32_to_key(v);
    }
}

}  // namespace msr

extern "C" int msr_sparsify(const void* logits, int is_f16, int fp16_math, int rows, uint32_t vocab, int k, int device,
                            uint32_t* out_idx, float* out_val, int32_t* out_weight) {
    if (!logits || rows < 0 || vocab == 0 || !out_idx || !out_val || !out_weight) {
        set_error("msr_sparsify: bad argument");
        return MSR_E_INVAL;
    }
    if (k < 1 || k > MSR_KMAX) {
        set_error("k must be in [1, %d] (got %d)", MSR_KMAX, k);
        return MSR_E_RANGE;
    }
    int n_dev = 0;
    if (device < 0 || hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) {
        set_error("no usable HIP device %d; there is no CPU sparsifier path", device);
        return MSR_E_NODEVICE;
    }
    if (rows == 0) return MSR_OK;
    HIP_TRY(hipSetDevice(device));
    const uint32_t tile = vocab <= 4096 ? 4096 : 8192;
    const uint32_t n_tiles = (vocab + tile - 1) / tile;
    const uint64_t ld = (uint64_t)n_tiles * tile;
    const size_t in_bytes = (size_t)rows * vocab * (is_f16 ? 2 : 4);
    void* d_in = nullptr;
    uint32_t *d_keys = nullptr, *d_ord = nullptr, *d_su = nullptr;
    uint64_t* d_part = nullptr;
    float* d_sf = nullptr;
    int32_t* d_n = nullptr;
    hipStream_t st = nullptr;
    const size_t per = (size_t)rows * k;
    int rc = MSR_OK;
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d_in, in_bytes) == hipSuccess &&
              hipMalloc(&d_keys, (size_t)rows * ld * 4) == hipSuccess && hipMalloc(&d_part, (size_t)n_tiles * per * 8) == hipSuccess &&
              hipMalloc(&d_ord, per * 4) == hipSuccess && hipMalloc(&d_su, per * 4) == hipSuccess &&
              hipMalloc(&d_sf, per * 4) == hipSuccess && hipMalloc(&d_n, (size_t)rows * 4) == hipSuccess &&
              hipMemcpyAsync(d_in, logits, in_bytes, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemsetAsync(d_keys, 0, (size_t)rows * ld * 4, st) == hipSuccess;
    if (!ok) {
        set_error("device setup failed in msr_sparsify");
        rc = MSR_E_NOMEM;
    }
    if (rc == MSR_OK) {
        hipLaunchKernelGGL(sparsify_keys, dim3(std::min<uint32_t>((vocab + 255) / 256, 1024), (uint32_t)rows), dim3(256), 0, st,
                           d_in, is_f16, fp16_math, d_keys, vocab, ld);
        SelectArgs se;
        se.unsorted = 0;
        se.src = d_keys;
        se.part = d_part;
        se.n_docs = vocab;
        se.n_tiles = n_tiles;
        se.tpr = n_tiles;
        se.rank = 0;
        se.nq = (uint32_t)rows;
        se.q0 = 0;
        se.qn = (uint32_t)rows;
        se.k = (uint32_t)k;
        rc = launch_select(st, tile, se);
    }
    if (rc == MSR_OK) {
        MergeArgs ma;
        ma.lists = d_part;
        ma.list_stride = per;
        ma.n_lists = n_tiles;
        ma.nq = (uint32_t)rows;
        ma.k = (uint32_t)k;
        ma.out_keys = nullptr;
        ma.out_ord = d_ord;
        ma.out_score_u32 = d_su;
        ma.out_score = d_sf;
        ma.out_n = d_n;
        rc = launch_merge(st, ma);
    }
    std::vector<uint32_t> keys(per);
    if (rc == MSR_OK) {
        bool c = hipMemcpyAsync(out_idx, d_ord, per * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipMemcpyAsync(keys.data(), d_su, per * 4, hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipStreamSynchronize(st) == hipSuccess;
        if (!c) {
            set_error("sparsifier kernels or download failed: %s", hipGetErrorString(hipGetLastError()));
            rc = MSR_E_HIP;
        }
    }
    if (rc == MSR_OK)
        for (size_t i = 0; i < per; ++i) {
            const uint32_t key = keys[i];
            const uint32_t bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
            float v;
            memcpy(&v, &bits, 4);
            if (!key) v = 0.f;
            out_val[i] = v;
            out_weight[i] = (int32_t)nearbyintf(v * 100.0f);  // np.rint(v * 100).astype(int), src/encode.py:75
        }
    void* ptrs[] = {d_in, d_keys, d_part, d_ord, d_su, d_sf, d_n};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}
