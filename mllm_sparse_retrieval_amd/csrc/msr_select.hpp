// Device-side building blocks shared by every kernel: cross-lane reductions (DPP), the order-preserving f32 <-> u32
// key map, the LDS carve of an accumulator tile and the exact per-tile top-k (tile_select).
#pragma once

#include "msr_device_internal.hpp"

namespace msr {

// ------------------------------------------------------------------------------------------------ device helpers
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Unsigned max across lanes with DPP (VALU cross-lane operands, no LDS round trip like ds_bpermute):
// quad_perm [1,0,3,2] -> quad_perm [2,3,0,1] -> row_half_mirror -> row_mirror leave every lane of a 4 / 8 / 16-lane
// group with the group's maximum; row_bcast:15 / row_bcast:31 then carry the row maxima into lane 63.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_umax(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);  // unwritten lanes: 0
    return max(v, o);
}
template <int G>  // maximum of every aligned group of G = 1, 2, 4, 8 or 16 lanes, in all lanes of the group
__device__ __forceinline__ uint32_t group_max_u32(uint32_t v) {
    if (G >= 2) v = dpp_umax<0xB1>(v);
    if (G >= 4) v = dpp_umax<0x4E>(v);
    if (G >= 8) v = dpp_umax<0x141>(v);
    if (G >= 16) v = dpp_umax<0x140>(v);
    return v;
}
// Inclusive prefix sum over the 64 lanes with DPP: row_shr 1, 2, 4, 8 build the scan inside every 16-lane row, then
// row_bcast:15 / row_bcast:31 add the totals of the preceding rows.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);  // unwritten lanes add 0
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    v = dpp_add<0x111>(v);             // row_shr:1
    v = dpp_add<0x112>(v);             // row_shr:2
    v = dpp_add<0x114, 0xF, 0xE>(v);   // row_shr:4
    v = dpp_add<0x118, 0xF, 0xC>(v);   // row_shr:8
    v = dpp_add<0x142, 0xA>(v);        // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);        // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = group_max_u32<16>(v);
    v = dpp_umax<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_umax<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t other = __shfl_xor(v, o, 64);
        v = other > v ? other : v;
    }
    return v;
}

// Buffer loads: a 128-bit resource (base, size) in SGPRs + a 32-bit byte offset per lane + a scalar byte offset. No
// 64-bit address arithmetic in VGPRs (the compiler otherwise keeps zero-extended offset PAIRS alive), and a read past
// `bytes` returns 0 instead of faulting. Word 3 = 0x00020000: raw 32-bit dwords on gfx9-family targets (gfx950).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// Rank-by-counting over `n` UNIQUE non-zero keys in LDS: key with rank r < k goes to out[r]; slots [n, k) get 0.
// BITONIC: compile the sort path in (kernel instances for k > 512; the k <= 512 instances of score_tiles sit at the
// 64-VGPR edge and must not carry it).
template <int NT, bool BITONIC = true>
__device__ __forceinline__ void rank_and_emit(uint64_t* cand, int n, int k, uint64_t* __restrict__ out,
                                              const uint32_t tid_) {
    const int tidx = (int)tid_;  // (a parameter, so that a caller's per-iteration copy of the thread id is used)
    if (BITONIC && n > 256) {
        // many keys (k in the hundreds): bitonic sort in LDS, descending, O(n log^2 n) instead of O(n^2) counting.
        // `cand` has room for the next power of two (capacity is a power of two >= n).
        int P = 512;
        while (P < n) P <<= 1;
        for (int i = n + tidx; i < P; i += NT) cand[i] = 0;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tidx; i < P / 2; i += NT) {
                    const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                    const uint64_t x = cand[lo], y = cand[hi];
                    if ((x < y) == ((lo & size) == 0)) {
                        cand[lo] = y;
                        cand[hi] = x;
                    }
                }
                __syncthreads();
            }
        for (int i = tidx; i < k; i += NT) out[i] = i < n ? cand[i] : 0;
        return;
    }
    if (n <= 64) {
        // one wave, keys in registers, partner keys broadcast with v_readlane (no LDS round trips)
        if (tidx < 64) {
            const int lane = tidx;
            const uint64_t me = lane < n ? cand[lane] : 0ull;
            const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
            int rank = 0;
            for (int j = 0; j < n; j += 4) {  // lanes n..63 hold key 0, which outranks nobody: no tail guard
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j + u) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)lo, j + u);
                    rank += o > me;
                }
            }
            if (lane < n && rank < k) out[rank] = me;
            for (int i = n + lane; i < k; i += 64) out[i] = 0;
        }
        return;
    }
    for (int i = tidx; i < n; i += NT) {
        const uint64_t me = cand[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += cand[j] > me;
        if (rank < k) out[rank] = me;
    }
    for (int i = n + tidx; i < k; i += NT) out[i] = 0;
}

struct SelectScratch {
    uint32_t cnt[64];  // one counter per bisection step
    uint32_t n_cand;
    uint32_t tau0;
    uint32_t smax;
    uint32_t pad;
};

__device__ __forceinline__ uint32_t f32_to_key(float f) {  // monotone: a < b  <=>  key(a) < key(b); never 0 for finite f
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_to_f32(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}



// LDS carve (bytes). The staging arrays of the streaming phase and the candidate keys of the select phase are
// never live together, so they share one region.
template <int TILE_DOCS, int NT, int CAND>
struct TileLds {
    static constexpr int kAcc = TILE_DOCS * 4;
    static constexpr int kStage = kQtBlock * 4 * 3 + (kQtBlock + 4) * 4 + 8 * 4;  // seg_start/len/w, pref, wsum
    static constexpr int kCand = CAND * 8;
    static constexpr int kUnion = (kStage > kCand ? kStage : kCand);
    static constexpr int kTmax = NT * 4 + 64 * 4;  // per-thread maxima (k > waves) + per-wave maxima
    static constexpr int kTotal = kAcc + kUnion + kTmax + (int)sizeof(SelectScratch);
};


}  // namespace msr

#include "msr_hist_select.hpp"

namespace msr {

// Exact top-k of one accumulator tile held in LDS (shared by score_tiles and select_tiles).
// Thread `tid` owns vec r*NT + tid of the accumulators in round r (conflict-free ds_read_b128); the accumulators are
// re-read from LDS in every pass instead of being held in registers. Writes k keys best-first (0 = empty slot).
// `unsorted`: the k keys may be written in any order (see the end of the function).
template <int TILE_DOCS, int NT, int CAND, class Stamp>
__device__ __forceinline__ void tile_select(const uint4* a4, uint64_t* cand, uint32_t* tmax, uint32_t* wmax,
                                            SelectScratch& ss, int rounds, uint64_t doc0, int k,
                                            uint64_t* __restrict__ out, Stamp stamp, const uint32_t tid,
                                            const bool unsorted = false) {
    constexpr int NW = NT / 64;
    const uint32_t lane = tid & 63;
    const uint32_t wave = rfl(tid >> 6);
    uint32_t mymax = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint4 x = a4[r * NT + tid];
        mymax = max(max(mymax, max(x.x, x.y)), max(x.z, x.w));
    }
    // ---- tau0: a lower bound with at least k accumulators at or above it = the k-th largest GROUP maximum.
    //   k <= 64 : 64 groups of NT/64 consecutive threads (DPP reductions), wave 0 bisects the 64 group maxima with
    //             ballots;
    //   k >  64 : groups are single threads (NT maxima, NT/64 per lane of wave 0).
    constexpr int G = NT / 64;  // threads per group
    static_assert(G == 4 || G == 8 || G == 16, "group maxima use the 4/8/16-lane DPP reductions");
    const uint32_t gm = group_max_u32<G>(mymax);
    if ((lane & (G - 1)) == 0) wmax[tid / G] = gm;
    if (k > 64) tmax[tid] = mymax;
    __syncthreads();

    uint32_t tau0 = 1, smax;
    {
        const uint32_t v = wmax[lane];  // 64 group maxima, one per lane, in every wave
        smax = wave_max_u32(v);
        if (smax == 0) {  // nothing matched in this tile
            for (int i = tid; i < k; i += NT) out[i] = 0;
            return;
        }
        if (k <= NT) {
            // Wave 0 bisects for all: a ballot step is ~8 scalar instructions, and the scalar ALU is shared by the
            // CU's 32 waves (eight waves repeating the search cost more than the extra barrier). The bisection stops
            // after kTauBits significant bits: tau0 is then a slightly lower bound (a few more candidates), still
            // with at least k accumulators at or above it.
            constexpr int kTauBits = 8;
            if (wave == 0) {
                const int top = 31 - __clz(smax);
                const int low = max(top - (kTauBits - 1), 0);
                uint32_t tau = 0;
                if (k <= 64) {
                    for (int bit = top; bit >= low; --bit) {
                        const uint32_t t2 = tau | (1u << bit);
                        if (__popcll(__ballot(v >= t2)) >= k) tau = t2;
                    }
                } else {
                    uint32_t mine[NW];
#pragma unroll
                    for (int i = 0; i < NW; ++i) mine[i] = tmax[i * 64 + lane];
                    for (int bit = top; bit >= low; --bit) {
                        const uint32_t t2 = tau | (1u << bit);
                        uint32_t c = 0;
#pragma unroll
                        for (int i = 0; i < NW; ++i) c += (uint32_t)__popcll(__ballot(mine[i] >= t2));
                        if (c >= (uint32_t)k) tau = t2;
                    }
                }
                if (lane == 0) ss.tau0 = max(tau, 1u);
            }
            __syncthreads();
            tau0 = ss.tau0;
        }
    }

    stamp(4);  // thread / wave maxima, tau0
    // ---- candidates: accumulators >= tau0 as unique global keys (score << 32 | ~ordinal)
    if (mymax >= tau0) {
        for (int r = 0; r < rounds; ++r) {
            const uint4 x = a4[r * NT + tid];
            const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sc4[e] >= tau0) {
                    const uint32_t local = 4 * (r * NT + tid) + e;
                    const uint32_t pos = atomicAdd(&ss.n_cand, 1u);
                    if (pos < CAND)
                        cand[pos] = ((uint64_t)sc4[e] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(doc0 + local));
                }
        }
    }
    __syncthreads();
    stamp(5);  // candidate collection
    uint32_t n_cand = ss.n_cand;

    if (TILE_DOCS <= 8192 && CAND >= 2 * kHistCand && n_cand > CAND) {
        // ---- k in the hundreds (the reference's default depth is 1000, src/arguments.py:59), or mass ties: ONE
        // histogram pass finds the composite threshold of the k best (hist_threshold, msr_hist_select.hpp); the
        // accumulators at or above it are exactly min(k, #positive) <= CAND keys. (Instances with a 512-key candidate
        // buffer or tiles above 8192 docs keep the byte-wise radix passes below.)
        if constexpr (TILE_DOCS <= 8192 && CAND >= 2 * kHistCand) {
            static_assert(TileLds<TILE_DOCS, NT, CAND>::kTmax >= (int)sizeof(HistScratch), "selection scratch");
            uint32_t* const hist = reinterpret_cast<uint32_t*>(cand);
            HistScratch& hs = *reinterpret_cast<HistScratch*>(tmax);  // (tmax is dead since tau0 was published)
            const HistResult hr = hist_threshold<TILE_DOCS, NT>(
                [&](int r) { return r < rounds ? a4[r * NT + tid] : make_uint4(0, 0, 0, 0); }, (uint32_t)k, hist,
                cand + kHistBins / 2, hs, tid, [](uint32_t key, uint32_t lo) { return (float)(key - lo); });
            if (tid == 0) ss.n_cand = 0;
            __syncthreads();  // the histogram / candidate area becomes the key buffer
            for (int r = 0; r < rounds; ++r) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t local = 4 * (r * NT + tid) + e;
                    if (sc4[e] != 0 && (((uint64_t)sc4[e] << 13) | (uint64_t)(TILE_DOCS - 1 - local)) >= hr.T)
                        cand[atomicAdd(&ss.n_cand, 1u)] =
                            ((uint64_t)sc4[e] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(doc0 + local));
                }
            }
            __syncthreads();
            n_cand = min(ss.n_cand, (uint32_t)CAND);  // == min(k, #positive) <= CAND by construction
        }
    } else if (n_cand > CAND) {
        // ---- fallback (mass ties, or k in the hundreds): exact selection by radix passes over the score.
        //   1. tau = k-th largest SCORE of the tile, one BYTE per pass, most significant first: a 256-bin LDS histogram
        //      of the accumulators that match the bytes fixed so far, then every wave finds the bin that holds the
        //      need-th largest (suffix sums over the bins: DPP scan + ballot) — 1-4 passes instead of 32 bisection steps;
        //   2. every accumulator above tau is in; of the c_eq accumulators equal to tau the `need` lowest ordinals are
        //      in: a tie's rank among the ties is a prefix count in (round, thread, element) order = ascending ordinal
        //      (wave scans + per-(round, wave) totals), no search.
        // Exactly min(k, #positive) <= CAND keys survive.
        constexpr int R = TILE_DOCS / (4 * NT);  // rounds of a full tile
        static_assert(NT >= 256 && R * NW <= NT, "the tie totals live in the NT-word tmax region");
        // histogram: 256 bins x SUB lane-interleaved copies in the (still unused) candidate buffer — scores cluster, and
        // 64 lanes adding to ONE LDS word serialise; with SUB copies on different banks a wave's adds to a hot bin
        // spread over SUB words
        constexpr int SUB = CAND / 128;  // 8 (k > 512 instances) or 4: 256 * SUB words = the CAND 8-byte keys
        static_assert(SUB == 4 || SUB == 8, "candidate buffer of 512 or 1024 keys");
        uint32_t* const hist = reinterpret_cast<uint32_t*>(cand);
        uint32_t* const ties = tmax;  // per-(round, wave) tie totals (tmax is dead since tau0 was published)
        __syncthreads();  // everyone has read n_cand
        if (tid == 0) ss.n_cand = 0;
        uint32_t prefix = 0, hi_mask = 0, need = (uint32_t)k;
        // first window = the 8 most significant bits of the tile's maximum (spreads the scores over up to 128 bins),
        // then whole bytes below it; the last window may overlap bits that are already fixed (harmless)
        for (int shift = max(31 - __clz(smax) - 7, 0);; shift = max(shift - 8, 0)) {
            for (int i = tid; i < 256 * SUB / 4; i += NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
            __syncthreads();
            for (int r = 0; r < rounds; ++r) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((sc4[e] & hi_mask) == prefix)
                        atomicAdd(&hist[((sc4[e] >> shift) & 255u) * SUB + (lane & (SUB - 1))], 1u);
            }
            __syncthreads();
            uint4 hb = make_uint4(0, 0, 0, 0);  // bins 4*lane .. 4*lane+3, summed over their SUB copies
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
                const uint4 t = reinterpret_cast<const uint4*>(hist)[lane * SUB + c];  // words 4*SUB*lane + 4c ..
                // word index = bin * SUB + copy: the 4*SUB words of this lane hold bin 4*lane + (4c + j) / SUB
                const uint32_t w4[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b = (4 * c + j) / SUB;
                    if (b == 0) hb.x += w4[j];
                    if (b == 1) hb.y += w4[j];
                    if (b == 2) hb.z += w4[j];
                    if (b == 3) hb.w += w4[j];
                }
            }
            const uint32_t s = hb.x + hb.y + hb.z + hb.w;
            const uint32_t inc = wave_inclusive_scan_u32(s);
            const uint32_t suf = rdl(inc, 63) - (inc - s);  // elements in bins >= 4*lane: non-increasing in lane
            // (the matching elements number at least `need`: more than CAND >= k accumulators are positive here)
            const uint32_t L = 63u - (uint32_t)__clzll((long long)__ballot(suf >= need));
            uint32_t above = rdl(suf, L) - rdl(s, L);  // elements in the bins above lane L's four
            const uint32_t b3 = rdl(hb.w, L), b2 = rdl(hb.z, L), b1 = rdl(hb.y, L);
            uint32_t bin = 4 * L + 3;
            if (above + b3 < need) {
                above += b3;
                bin = 4 * L + 2;
                if (above + b2 < need) {
                    above += b2;
                    bin = 4 * L + 1;
                    if (above + b1 < need) {
                        above += b1;
                        bin = 4 * L;
                    }
                }
            }
            need -= above;  // rank of the wanted element inside the chosen bin
            prefix |= bin << shift;
            hi_mask |= 255u << shift;
            __syncthreads();  // the histogram is rewritten next (or becomes the candidate buffer again)
            if (shift == 0) break;
        }
        const uint32_t tau = prefix;   // k-th largest score; 0: fewer than k positive accumulators (take them all)
        if (tau == 0) need = 0;        // ... and none of the zeros
        // rank of every tie among the ties, in ascending ordinal order
        uint32_t first[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            first[r] = 0;
            if (r < rounds) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t c = (uint32_t)(x.x == tau) + (uint32_t)(x.y == tau) + (uint32_t)(x.z == tau) + (uint32_t)(x.w == tau);
                const uint32_t inc = wave_inclusive_scan_u32(c);
                if (lane == 63) ties[r * NW + wave] = inc;
                first[r] = inc - c;  // ties of lower lanes of this wave in this round
            }
        }
        __syncthreads();
        {
            uint32_t run = 0;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < rounds) {
                    uint32_t row = 0, before = 0;
                    for (int w = 0; w < NW; ++w) {
                        const uint32_t v = ties[r * NW + w];
                        before += (uint32_t)w < wave ? v : 0u;
                        row += v;
                    }
                    first[r] += run + before;
                    run += row;
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < rounds) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
                uint32_t tr = first[r];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bool take = sc4[e] > tau;
                    if (sc4[e] == tau) take = tr++ < need;
                    if (take) {
                        const uint32_t local = 4 * (r * NT + tid) + e;
                        const uint32_t pos = atomicAdd(&ss.n_cand, 1u);
                        if (pos < CAND)
                            cand[pos] = ((uint64_t)sc4[e] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(doc0 + local));
                    }
                }
            }
        __syncthreads();
        n_cand = min(ss.n_cand, (uint32_t)CAND);  // == min(k, #positive) <= CAND by construction
    }
    if (unsorted && n_cand <= (uint32_t)k) {
        // the consumer takes the top-k as a SET (hybrid fusion: min / max / membership only): no ranking
        for (int i = tid; i < k; i += NT) out[i] = i < (int)n_cand ? cand[i] : 0ull;
        return;
    }
    rank_and_emit<NT, (CAND > 512)>(cand, (int)n_cand, k, out, tid);
}


// Top-k of a tile when a lower bound `th` on the query's GLOBAL k-th key is already known (the k-th best key over the
// tiles scored by an earlier launch, msr_device.hip: staged search). A key below `th` cannot reach the final top-k,
// so the tile's survivors are the accumulators with key >= th — usually none or a handful: one pass over the
// accumulators, one barrier, and the (tiny) ranking; no group maxima, no threshold bisection. The list may hold fewer
// than k keys (the merge treats 0 as an empty slot). Returns false, with the scratch reset, when more than CAND keys
// survive (the caller then runs the full tile_select).
template <int TILE_DOCS, int NT, int CAND>
__device__ __forceinline__ bool theta_select(const uint4* a4, uint64_t* cand, SelectScratch& ss, int rounds,
                                             uint64_t doc0, int k, uint64_t* __restrict__ out, const uint64_t th,
                                             const uint32_t tid) {
    const uint32_t ts = (uint32_t)(th >> 32);
    uint32_t mymax = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint4 x = a4[r * NT + tid];
        mymax = max(max(mymax, max(x.x, x.y)), max(x.z, x.w));
    }
    if (mymax >= ts) {
        for (int r = 0; r < rounds; ++r) {
            const uint4 x = a4[r * NT + tid];
            const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (sc4[e] >= ts) {
                    const uint32_t local = 4 * (r * NT + tid) + e;
                    const uint64_t key = ((uint64_t)sc4[e] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(doc0 + local));
                    if (key >= th) {
                        const uint32_t pos = atomicAdd(&ss.n_cand, 1u);
                        if (pos < CAND) cand[pos] = key;
                    }
                }
        }
    }
    __syncthreads();
    const uint32_t n_cand = ss.n_cand;
    if (n_cand > CAND) {
        __syncthreads();  // everyone has read n_cand
        if (tid == 0) ss.n_cand = 0;
        __syncthreads();
        return false;
    }
    rank_and_emit<NT, (CAND > 512)>(cand, (int)n_cand, k, out, tid);
    return true;
}

}  // namespace msr
