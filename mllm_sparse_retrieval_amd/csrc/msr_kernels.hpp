// The search-path kernels (included by msr_device.hip only). Roofline class: HBM by the task's definition (SURVEY.md
// §8d); no MFMA — this is gather + integer reduce, and what it actually runs into is VALU issue and the L2 -> CU path
// (DESIGN.md §7).
//
// score_tiles<TILE_DOCS, NT, U, MIN_WAVES, CAND, DBG, MODE>   one workgroup per (doc tile, query); grid = (query, tile)
//     - TILE_DOCS u32 accumulators in LDS (32 KiB at the default 8192 docs: four workgroups = 32 waves per CU),
//     - dense-head terms: doc-major rows scored by the accumulator's owner with v_dot2_u32_u16 (this is also the
//       accumulator init),
//     - the query's other (term, tile) segments are cut into 1-KiB chunks (64 lanes x 16 B = 256 postings); waves take
//       chunks round-robin, resolve 64 chunks lane-parallel, then walk them with v_readlane broadcasts: one 16-byte
//       buffer load (chunk end in the resource's size word) and four ds_add_u32 per lane and chunk, two register
//       banks in flight,
//     - exact per-tile top-k: tile_select, or theta_select when an earlier launch's tiles gave the query a threshold
//       (staged search, msr_device.hip) — both in msr_select.hpp.
//     x (query) runs fastest in dispatch order, so the ~1000 workgroups in flight score the SAME tile for different
//     queries and the tile's rows and hot segments are served from the XCDs' L2s, not from HBM.
// select_tiles   top-k of an accumulator tile that already sits in HBM (term shards, dense scores, sparsifier).
// merge_small    one WAVE per query: exact top-k of at most 64 keys (a few tiles x small k).
// merge_lists    one workgroup per query: exact top-k of best-first lists (per tile or per shard).
#pragma once

#include "msr_accumulate.hpp"

namespace msr {

// At most 64 keys per query (e.g. 4 tiles x top-10): one WAVE ranks them, keys in registers, ranks by v_readlane
// broadcasts — no LDS, no barrier. (merge_small's body.)
__device__ __forceinline__ void merge_wave(const MergeArgs& a, const uint32_t q, const uint32_t lane) {
    const uint32_t n_items = a.n_lists * a.k;  // <= 64
    uint64_t me = 0;
    if (lane < n_items) {
        const uint32_t l = lane / a.k, j = lane - l * a.k;
        me = a.lists[(uint64_t)l * a.list_stride + (uint64_t)q * a.k + j];
    }
    const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_items; j += 4) {  // lanes past n_items hold key 0, which outranks nobody
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
            const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, (int)(j + u)) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)(j + u));
            rank += o > me;
        }
    }
    const uint32_t n_hit = min((uint32_t)__popcll(__ballot(me != 0)), a.k);
    // keys are unique, so the non-empty keys take ranks 0 .. n-1; slots [n_hit, k) are empty
    const uint64_t o = (uint64_t)q * a.k;
    auto emit = [&](uint32_t i, uint64_t key) {
        if (a.out_keys) a.out_keys[o + i] = key;
        if (a.out_ord) {
            const uint32_t sc = (uint32_t)(key >> 32);
            a.out_ord[o + i] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
            a.out_score_u32[o + i] = sc;
            a.out_score[o + i] = (float)sc;  // round-to-nearest-even, exact below 2^24 (contract T5)
        }
    };
    if (me != 0 && rank < a.k) emit(rank, me);
    if (lane >= n_hit && lane < a.k) emit(lane, 0ull);
    if (lane == 0 && a.out_n) a.out_n[q] = (int32_t)n_hit;
}

// ------------------------------------------------------------------------------------------------ kernel 1
// <docs per tile, threads, 1-KiB chunk loads per register bank, min waves per SIMD, candidate-key capacity (>= k), diag>
// LIGHT: the batch's queries hold at most 64 sparse terms each -> accumulate_tile_light (msr_accumulate.hpp)
template <int TILE_DOCS, int NT, int U, int MIN_WAVES, int CAND, bool DBG, int MODE = 0, bool LIGHT = false>
__global__ __launch_bounds__(NT, MIN_WAVES) void score_tiles(const ScoreArgs a) {
    using L = TileLds<TILE_DOCS, NT, CAND>;

    __shared__ __attribute__((aligned(16))) uint8_t lds[L::kTotal];
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    uint8_t* const un = lds + L::kAcc;
    // select-phase view of the union (the accumulation phase's staging arrays live in the same bytes)
    uint64_t* const cand = reinterpret_cast<uint64_t*>(un);
    uint32_t* const tmax = reinterpret_cast<uint32_t*>(un + L::kUnion);
    uint32_t* const wmax = tmax + NT;
    SelectScratch& ss = *reinterpret_cast<SelectScratch*>(un + L::kUnion + L::kTmax);

    const uint32_t tid = threadIdx.x;
    // diagnostic build only: wave 0 stamps s_memtime at phase boundaries and adds the deltas to a side buffer
    long long t_prev = 0;
    auto stamp = [&](int slot) {
        if (DBG && (a.dbg & 8u) && a.stamps && tid == 0 && (blockIdx.x & 63u) == 0) {  // 1 workgroup in 64
            const long long now = clock64();
            if (slot >= 0) atomicAdd(&a.stamps[slot], (unsigned long long)(now - t_prev));
            t_prev = now;
        }
    };
    stamp(-1);
    if (DBG && (a.dbg & 128u)) return;  // ablation: workgroup launch cost only
    // 2-D grid (x = query, y = tile): x runs fastest in dispatch order, so neighbours share the tile (tile-major)
    // without a division in every wave's prologue
    const uint32_t tile_l = a.tl0 + blockIdx.y;
    const uint32_t q = a.q0 + blockIdx.x;
    const uint32_t tile_g = a.tile0 + tile_l;
    const uint64_t doc0 = (uint64_t)tile_g * TILE_DOCS;
    const uint32_t ndocs_tile = (uint32_t)min((uint64_t)TILE_DOCS, a.n_docs - doc0);
    // rounds of 4*NT accumulators that hold real docs
    const int rounds = (int)((ndocs_tile + 4 * NT - 1) / (4 * NT));
    uint4* const a4 = reinterpret_cast<uint4*>(acc);

    // staged search: the query's k-th best key over the tiles of an earlier launch (0: fewer than k hits so far)
    const uint64_t theta = (MODE == 0 && a.theta) ? a.theta[(uint64_t)q * a.k + (a.k - 1)] : 0ull;
    if constexpr (LIGHT)
        accumulate_tile_light<TILE_DOCS, NT>(a, q, tile_l, rounds, lds, ss, stamp, tid);
    else
        accumulate_tile<TILE_DOCS, NT, U, DBG>(a, q, tile_l, rounds, lds, ss, stamp, tid);
    stamp(3);  // waiting for the slowest wave

    // =============================================================== exact top-k of this tile
    // Thread `tid` owns vec r*NT + tid of the accumulators in round r (conflict-free ds_read_b128); the
    // accumulators are re-read from LDS in every pass instead of being held in registers.
    if (MODE == 1) {  // term-sharded search: hand the partial sums of this tile to the reduction
        const uint32_t g = tile_g / a.tpr, t = tile_g % a.tpr;
        uint4* dst = reinterpret_cast<uint4*>(a.dump + (((uint64_t)g * a.qn + (q - a.q0)) * a.tpr + t) * TILE_DOCS);
        for (int r = 0; r < rounds; ++r) {
            uint4 x = a4[r * NT + tid];
            if (a.dump_add) {
                const uint4 o = dst[r * NT + tid];
                x = make_uint4(x.x + o.x, x.y + o.y, x.z + o.z, x.w + o.w);
            }
            dst[r * NT + tid] = x;
        }
        return;
    }
    uint64_t* out = a.part + ((uint64_t)tile_l * a.nq + q) * a.k;
    const int k = (int)a.k;
    if (DBG && (a.dbg & 4u)) {  // ablation: no select phase
        for (int i = tid; i < k; i += NT) out[i] = 0;
        return;
    }

    if (theta && theta_select<TILE_DOCS, NT, CAND>(a4, cand, ss, rounds, doc0, k, out, theta, tid)) {
        stamp(6);
        return;
    }
    tile_select<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, ss, rounds, doc0, k, out, stamp, tid,
                                     a.unsorted != 0);
    stamp(6);  // ranking + output
}

// ------------------------------------------------------------------------------------------------ kernel 1b
// Term-sharded search, after the reduce-scatter: rank `rank` holds the SUMMED accumulators of its doc range,
// src[(qi * tpr + t) * TILE_DOCS + i]; one workgroup per (tile of the range, query) selects the exact tile top-k.


template <int TILE_DOCS, int NT, int CAND>
__global__ __launch_bounds__(NT) void select_tiles(const SelectArgs a) {
    using L = TileLds<TILE_DOCS, NT, CAND>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[L::kTotal];
    uint4* const a4 = reinterpret_cast<uint4*>(lds);
    uint8_t* const un = lds + L::kAcc;
    uint64_t* const cand = reinterpret_cast<uint64_t*>(un);
    uint32_t* const tmax = reinterpret_cast<uint32_t*>(un + L::kUnion);
    uint32_t* const wmax = tmax + NT;
    SelectScratch& ss = *reinterpret_cast<SelectScratch*>(un + L::kUnion + L::kTmax);
    const uint32_t tid = threadIdx.x;
    const uint32_t t = blockIdx.y, qi = blockIdx.x;
    const uint32_t tile_g = a.rank * a.tpr + t;
    uint64_t* out = a.part + ((uint64_t)t * a.nq + a.q0 + qi) * a.k;
    if (tile_g >= a.n_tiles) {  // padding tile of the last rank
        for (uint32_t i = tid; i < a.k; i += NT) out[i] = 0;
        return;
    }
    const uint64_t doc0 = (uint64_t)tile_g * TILE_DOCS;
    const uint32_t ndocs_tile = (uint32_t)min((uint64_t)TILE_DOCS, a.n_docs - doc0);
    const int rounds = (int)((ndocs_tile + 4 * NT - 1) / (4 * NT));
    const uint4* src = reinterpret_cast<const uint4*>(a.src + ((uint64_t)qi * a.tpr + t) * TILE_DOCS);
    for (int r = 0; r < rounds; ++r) a4[r * NT + tid] = src[r * NT + tid];
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
    }
    __syncthreads();
    tile_select<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, ss, rounds, doc0, (int)a.k, out, [](int) {}, tid,
                                     a.unsorted != 0);
}

// ------------------------------------------------------------------------------------------------ kernel 2


// merge_small: one wave per query, four queries per workgroup.
__global__ __launch_bounds__(256) void merge_small(const MergeArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.nq) return;
    merge_wave(a, q, lane);
}

template <int NT>
__global__ __launch_bounds__(NT) void merge_lists(const MergeArgs a) {
    __shared__ __attribute__((aligned(16))) uint64_t cand[kCandCap];
    __shared__ uint64_t res[kCandCap];
    __shared__ SelectScratch ss;
    __shared__ uint64_t wmax[NT / 64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t q = blockIdx.x;
    const int k = (int)a.k;
    const uint32_t n_items = a.n_lists * a.k;
    auto key_at = [&](uint32_t i) -> uint64_t {
        const uint32_t l = i / a.k, j = i - l * a.k;
        return a.lists[(uint64_t)l * a.list_stride + (uint64_t)q * a.k + j];
    };
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) ss.n_cand = 0;
    __syncthreads();

    uint32_t n_cand;
    if (a.n_lists == 1) {
        // a single list is already the answer, best first (single-tile indexes, e.g. the hybrid path's 5 000 docs at
        // depth 1000): copy instead of ranking it again
        for (int i = tid; i < k; i += NT) res[i] = key_at((uint32_t)i);
        __syncthreads();
        uint32_t c = 0;
        for (int i = tid; i < k; i += NT) c += res[i] != 0;
        c = wave_sum_u32(c);
        if (lane == 0 && c) atomicAdd(&ss.n_cand, c);
        __syncthreads();
        n_cand = ss.n_cand;
    } else if (n_items <= kCandCap) {
        for (uint32_t i = tid; i < n_items; i += NT) {
            const uint64_t key = key_at(i);
            if (key) cand[atomicAdd(&ss.n_cand, 1u)] = key;
        }
        __syncthreads();
        n_cand = ss.n_cand;
    } else {
        // More keys than the LDS buffer holds. Every list is sorted best-first, so the k-th largest of the lists'
        // HEAD keys is a lower bound on the global k-th key (when there are at least k non-empty lists); keys at or
        // above it are few. If they do not fit either (or there are fewer than k lists), bisect over all keys.
        auto bisect_kth = [&](uint32_t n, auto key_of) -> uint64_t {  // k-th largest of n keys (0 if fewer than k > 0)
            uint64_t m = 0;
            for (uint32_t i = tid; i < n; i += NT) {
                const uint64_t key = key_of(i);
                m = key > m ? key : m;
            }
            m = wave_max_u64(m);
            __syncthreads();  // previous users of wmax / cnt are done
            if (lane == 0) wmax[wave] = m;
            if (tid < 64) ss.cnt[tid] = 0;
            __syncthreads();
            m = 0;
            for (int w = 0; w < NT / 64; ++w) m = wmax[w] > m ? wmax[w] : m;
            uint64_t tau = 0;
            if (m) {
                int step = 0;
                for (int bit = 63 - __clzll((long long)m); bit >= 0; --bit, ++step) {
                    const uint64_t t2 = tau | (1ull << bit);
                    uint32_t c = 0;
                    for (uint32_t i = tid; i < n; i += NT) c += key_of(i) >= t2;
                    c = wave_sum_u32(c);
                    if (lane == 0 && c) atomicAdd(&ss.cnt[step], c);  // at most 64 steps: one slot each
                    __syncthreads();
                    if (ss.cnt[step] >= (uint32_t)k) tau = t2;
                }
            }
            return tau;
        };
        auto collect = [&](uint64_t tau) {
            __syncthreads();
            if (tid == 0) ss.n_cand = 0;
            __syncthreads();
            for (uint32_t i = tid; i < n_items; i += NT) {
                const uint64_t key = key_at(i);
                if (key && key >= tau) {
                    const uint32_t pos = atomicAdd(&ss.n_cand, 1u);
                    if (pos < kCandCap) cand[pos] = key;
                }
            }
            __syncthreads();
            return ss.n_cand;
        };
        // The lists of a staged search are mostly EMPTY (tiles of the second stage report only keys at or above the
        // query's threshold): when all non-empty keys fit the buffer, take them as they are — no threshold search.
        uint32_t got = collect(1);  // every non-zero key (collect() stops storing at kCandCap but keeps counting)
        if (got > kCandCap && a.n_lists >= (uint32_t)k) {
            const uint64_t tau_heads = bisect_kth(a.n_lists, [&](uint32_t l) { return key_at(l * a.k); });
            if (tau_heads) got = collect(tau_heads);
        }
        if (got > kCandCap) got = collect(bisect_kth(n_items, key_at));  // exactly min(k, #keys) <= kCandCap survive
        n_cand = min(got, (uint32_t)kCandCap);
    }
    if (a.n_lists != 1) rank_and_emit<NT>(cand, (int)n_cand, k, res, tid);
    __syncthreads();
    const int n_hit = min((int)n_cand, k);
    for (int i = tid; i < k; i += NT) {
        const uint64_t key = res[i];
        const uint64_t o = (uint64_t)q * a.k + i;
        if (a.out_keys) a.out_keys[o] = key;
        if (a.out_ord) {
            const uint32_t sc = (uint32_t)(key >> 32);
            a.out_ord[o] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
            a.out_score_u32[o] = sc;
            a.out_score[o] = (float)sc;  // round-to-nearest-even, exact below 2^24 (contract T5)
        }
    }
    if (tid == 0 && a.out_n) a.out_n[q] = n_hit;
}

}  // namespace msr
