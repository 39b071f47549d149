// HIP search path for gfx950 (MI355X): device residency of the tile-major inverted index, the scoring kernel,
// the top-k merge kernel, resident query batches and the RCCL exchange for doc-range shards.
//
// Replaces what runs below `LuceneImpactSearcher.batch_search(queries, qids, k, threads)` in the reference
// (call site src/search.py:86-87; Lucene impact scoring):
//     score(d) = sum over query terms t of  q_w(t) * tf(t, d),   top-k of the docs with score > 0,
//     ties broken by external doc id ascending (= lower ordinal).
//
// Kernel 1  score_tiles<TILE_DOCS, NT>   one workgroup per (doc tile, query)
//     - TILE_DOCS u32 accumulators in LDS (128 KiB at 32768 docs),
//     - the query's (term, tile) segments are cut into 1-KiB "chunks" (64 lanes x 16 B = 256 postings);
//       waves take chunks round-robin, each lane loads one uint4 (4 postings) and issues 4 ds_add_u32,
//     - exact per-tile top-k: a lower bound from the per-thread maxima prunes the tile to a few dozen
//       candidates, which are ranked by counting on unique 64-bit keys (score << 32 | ~ordinal);
//       a bisection over (score, ordinal) keys is the always-correct fallback (many ties / large k).
//     Workgroups are ordered tile-major, so the ~512 workgroups in flight score the SAME tile for
//     different queries and the tile's hot segments are served from the XCDs' L2s, not from HBM.
// Kernel 2  merge_lists<NT>              one workgroup per query: exact top-k of the per-tile (or per-shard) lists.
//
// Roofline: HBM; no MFMA anywhere — this is gather / integer reduce.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "msr_internal.h"

namespace msr {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MSR_E_HIP;                                                                  \
        }                                                                                      \
    } while (0)

// ------------------------------------------------------------------------------------------------ constants
constexpr int kQtBlock = 256;    // query terms staged in LDS per round
constexpr int kCandCap = 1024;   // candidate keys per workgroup (>= MSR_KMAX)
constexpr int kChunkVecs = 64;   // one chunk = one wave-wide uint4 load = 256 postings = 1 KiB
static_assert(kCandCap >= MSR_KMAX, "candidate buffer must hold k keys");

struct DeviceIndex {
    int device = -1;
    hipStream_t stream = nullptr;
    uint32_t* d_seg_ptr = nullptr;   // [shard_ntiles][n_terms+1] absolute vec index
    uint32_t* d_postings = nullptr;  // the shard's vecs; vec v of the index lives at d_postings + (v - vec_base)*4
    uint32_t* d_dense = nullptr;     // [shard_ntiles][n_pairs][tile_docs] dense head of the shard's tiles
    uint32_t n_pairs = 0;
    uint32_t vec_base = 0;
    uint64_t shard_vecs = 0;
    std::vector<uint32_t> df_shard;  // postings of each term inside this shard (for algorithmic bytes)
    bool df_shard_ready = false;
    // exchange
    ncclComm_t comm = nullptr;
    int n_ranks = 1;
    int rank = 0;
};

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

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// Rank-by-counting over `n` UNIQUE non-zero keys in LDS: key with rank r < k goes to out[r]; slots [n, k) get 0.
template <int NT>
__device__ __forceinline__ void rank_and_emit(const uint64_t* cand, int n, int k, uint64_t* __restrict__ out) {
    if (n <= 64) {
        // one wave, keys in registers, partner keys broadcast with v_readlane (no LDS round trips)
        if (threadIdx.x < 64) {
            const int lane = (int)threadIdx.x;
            const uint64_t me = lane < n ? cand[lane] : 0ull;
            const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
            int rank = 0;
            for (int j = 0; j < n; ++j) {
                const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32) |
                                   (uint32_t)__builtin_amdgcn_readlane((int)lo, j);
                rank += o > me;
            }
            if (lane < n && rank < k) out[rank] = me;
            for (int i = n + lane; i < k; i += 64) out[i] = 0;
        }
        return;
    }
    for (int i = threadIdx.x; i < n; i += NT) {
        const uint64_t me = cand[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += cand[j] > me;
        if (rank < k) out[rank] = me;
    }
    for (int i = n + (int)threadIdx.x; i < k; i += NT) out[i] = 0;
}

struct SelectScratch {
    uint32_t cnt[64];  // one counter per bisection step
    uint32_t n_cand;
    uint32_t tau0;
    uint32_t smax;
    uint32_t pad;
};

// ------------------------------------------------------------------------------------------------ kernel 1
struct ScoreArgs {
    const uint32_t* seg_ptr;   // [ntiles][n_terms+1]
    const uint32_t* postings;  // shard base
    const uint32_t* q_ptr;     // [nq+1]
    const uint32_t* q_term;
    const uint32_t* q_w;
    const uint32_t* dense;     // [ntiles][n_pairs][TILE_DOCS] dense head (weights of term 2p+1 << 16 | term 2p)
    const uint32_t* q_dense;   // [nq][n_pairs] packed query weights of the dense-head terms (0 = absent)
    uint32_t n_pairs;
    uint64_t* part;            // [ntiles][nq][k] keys
    uint64_t n_docs;           // whole index
    uint32_t vec_base;
    uint32_t n_terms;
    uint32_t tile0;            // first (global) tile of the shard
    uint32_t nq;               // queries of the batch (row stride of `part`)
    uint32_t q0;               // this launch scores queries [q0, q0 + qn)
    uint32_t qn;
    uint32_t k;
    // term-sharded search (MODE 1): instead of selecting, the accumulator tile is written (or added) to
    // dump[((g * qn + (q - q0)) * tpr + t) * TILE_DOCS + i] with g = tile / tpr, t = tile % tpr  — the layout whose
    // G equal chunks are the doc ranges that ncclReduceScatter hands to the G ranks
    uint32_t* dump;
    uint32_t tpr;              // tiles per rank = ceil(n_tiles / G)
    uint32_t dump_add;         // 1: dump[...] += tile (single-GPU emulation of the reduction), 0: store
    uint32_t dbg;              // MSR_DEBUG_FLAGS (timing ablations only; results are wrong when bits 0-2 are set)
    unsigned long long* stamps;  // [8] summed s_memtime deltas of wave 0 per phase (dbg bit 3), else null
};

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


// Exact top-k of one accumulator tile held in LDS (shared by score_tiles and select_tiles).
// Thread `tid` owns vec r*NT + tid of the accumulators in round r (conflict-free ds_read_b128); the accumulators are
// re-read from LDS in every pass instead of being held in registers. Writes k keys best-first (0 = empty slot).
template <int TILE_DOCS, int NT, int CAND, class Stamp>
__device__ __forceinline__ void tile_select(const uint4* a4, uint64_t* cand, uint32_t* tmax, uint32_t* wmax,
                                            SelectScratch& ss, int rounds, uint64_t doc0, int k,
                                            uint64_t* __restrict__ out, Stamp stamp) {
    constexpr int NW = NT / 64;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = rfl(tid >> 6);
    uint32_t mymax = 0;
    for (int r = 0; r < rounds; ++r) {
        const uint4 x = a4[r * NT + tid];
        mymax = max(max(mymax, max(x.x, x.y)), max(x.z, x.w));
    }
    // ---- tau0: a lower bound with at least k accumulators at or above it = the k-th largest GROUP maximum.
    //   k <= 64 : 64 groups of NT/64 consecutive threads (log2(NT/64) shuffle steps), then every wave bisects the
    //             64 group maxima with ballots (no further barrier);
    //   k >  64 : groups are single threads (NT maxima, NT/64 per lane of wave 0), one more barrier.
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
        if (k <= 64) {
            uint32_t tau = 0;
            for (int bit = 31 - __clz(smax); bit >= 0; --bit) {
                const uint32_t t2 = tau | (1u << bit);
                if (__popcll(__ballot(v >= t2)) >= k) tau = t2;
            }
            tau0 = max(tau, 1u);
        }
    }
    if (k > 64 && k <= NT) {
        if (wave == 0) {
            uint32_t mine[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) mine[i] = tmax[i * 64 + lane];
            uint32_t tau = 0;
            for (int bit = 31 - __clz(smax); bit >= 0; --bit) {
                const uint32_t t2 = tau | (1u << bit);
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) c += (uint32_t)__popcll(__ballot(mine[i] >= t2));
                if (c >= (uint32_t)k) tau = t2;
            }
            if (lane == 0) ss.tau0 = max(tau, 1u);
        }
        __syncthreads();
        tau0 = ss.tau0;
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

    if (n_cand > CAND) {
        // ---- fallback (mass ties, or k in the hundreds): exact selection by bisection.
        //   1. tau = k-th largest SCORE of the tile: one bit per step, counted with ballots (scalar popcounts);
        //   2. every accumulator above tau is in; of the c_eq accumulators equal to tau the `need` lowest ordinals are
        //      in — found by a second bisection over the local ordinal only when there are more ties than needed.
        // Exactly min(k, #positive) <= CAND keys survive.
        __syncthreads();  // everyone has read n_cand
        if (tid == 0) ss.n_cand = 0;
        auto count_if = [&](auto pred) -> uint32_t {  // wave-level count over this wave's accumulators (uniform)
            uint32_t c = 0;
            for (int r = 0; r < rounds; ++r) {
                const uint4 x = a4[r * NT + tid];
                const uint32_t base = 4 * (r * NT + tid);
                c += (uint32_t)__popcll(__ballot(pred(x.x, base))) + (uint32_t)__popcll(__ballot(pred(x.y, base + 1))) +
                     (uint32_t)__popcll(__ballot(pred(x.z, base + 2))) + (uint32_t)__popcll(__ballot(pred(x.w, base + 3)));
            }
            return c;
        };
        int step = 0;
        auto block_count = [&](uint32_t c) -> uint32_t {  // sum of the waves' counts, the same value on every thread
            if (lane == 0 && c) atomicAdd(&ss.cnt[step], c);
            __syncthreads();
            return ss.cnt[step++];
        };
        uint32_t tau = 0;
        for (int bit = 31 - __clz(smax); bit >= 0; --bit) {
            const uint32_t t2 = tau | (1u << bit);
            if (block_count(count_if([&](uint32_t sc, uint32_t) { return sc >= t2; })) >= (uint32_t)k) tau = t2;
        }
        uint32_t o_star = 0xFFFFFFFFu;  // ties at tau with local ordinal <= o_star are selected
        if (tau == 0) {
            tau = 1;  // fewer than k positive accumulators: all of them
        } else {
            const uint32_t c_gt = block_count(count_if([&](uint32_t sc, uint32_t) { return sc > tau; }));
            const uint32_t c_eq = block_count(count_if([&](uint32_t sc, uint32_t) { return sc == tau; }));
            const uint32_t need = (uint32_t)k - c_gt;  // >= 1 by the definition of tau
            if (c_eq > need) {
                uint32_t lo = 0, hi = TILE_DOCS - 1;  // smallest o with #(ties, local <= o) >= need
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint32_t c = block_count(count_if([&](uint32_t sc, uint32_t loc) { return sc == tau && loc <= mid; }));
                    if (c >= need)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                o_star = lo;
            }
        }
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            const uint4 x = a4[r * NT + tid];
            const uint32_t sc4[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t local = 4 * (r * NT + tid) + e;
                if (sc4[e] > tau || (sc4[e] == tau && local <= o_star)) {
                    const uint32_t pos = atomicAdd(&ss.n_cand, 1u);
                    if (pos < CAND)
                        cand[pos] = ((uint64_t)sc4[e] << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)(doc0 + local));
                }
            }
        }
        __syncthreads();
        n_cand = min(ss.n_cand, (uint32_t)CAND);  // == min(k, #positive) <= CAND by construction
    }
    rank_and_emit<NT>(cand, (int)n_cand, k, out);
}

// <docs per tile, threads, 1-KiB chunk loads per register bank, min waves per SIMD, candidate-key capacity (>= k), diag>
template <int TILE_DOCS, int NT, int U, int MIN_WAVES, int CAND, bool DBG, int MODE = 0>
__global__ __launch_bounds__(NT, MIN_WAVES) void score_tiles(const ScoreArgs a) {
    constexpr int NW = NT / 64;
    static_assert(TILE_DOCS % (4 * NT) == 0, "tile must be a multiple of 4*NT");
    static_assert(NT >= kQtBlock, "the staging scan uses the first 256 threads");
    using L = TileLds<TILE_DOCS, NT, CAND>;

    __shared__ __attribute__((aligned(16))) uint8_t lds[L::kTotal];
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    uint8_t* const un = lds + L::kAcc;
    // streaming-phase view of the union
    uint32_t* const seg_start = reinterpret_cast<uint32_t*>(un);
    uint32_t* const seg_len = seg_start + kQtBlock;
    uint32_t* const seg_w = seg_len + kQtBlock;
    uint32_t* const pref = seg_w + kQtBlock;
    uint32_t* const wsum = pref + kQtBlock + 4;
    // select-phase view of the union
    uint64_t* const cand = reinterpret_cast<uint64_t*>(un);
    uint32_t* const tmax = reinterpret_cast<uint32_t*>(un + L::kUnion);
    uint32_t* const wmax = tmax + NT;
    SelectScratch& ss = *reinterpret_cast<SelectScratch*>(un + L::kUnion + L::kTmax);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = rfl(tid >> 6);        // provably wave-uniform for the compiler
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
    const uint32_t tile_l = blockIdx.x / a.qn;  // tile-major: neighbours in dispatch order share the tile
    const uint32_t q = a.q0 + blockIdx.x % a.qn;
    const uint32_t tile_g = a.tile0 + tile_l;
    const uint64_t doc0 = (uint64_t)tile_g * TILE_DOCS;
    const uint32_t ndocs_tile = (uint32_t)min((uint64_t)TILE_DOCS, a.n_docs - doc0);
    // rounds of 4*NT accumulators that hold real docs
    const int rounds = (int)((ndocs_tile + 4 * NT - 1) / (4 * NT));
    uint4* const a4 = reinterpret_cast<uint4*>(acc);

    const uint32_t qb = a.q_ptr[q], qe = a.q_ptr[q + 1];
    const uint32_t* seg_row = a.seg_ptr + (uint64_t)tile_l * (a.n_terms + 1);
    const uint4* post4 = reinterpret_cast<const uint4*>(a.postings);

    // ---- first round's (term -> segment) lookups: two dependent global loads, issued before the zeroing so that
    // their latency hides behind it
    uint32_t pre_w = 0, pre_s0 = 0, pre_s1 = 0;
    if (tid < min((uint32_t)kQtBlock, qe - qb)) {
        const uint32_t t = a.q_term[qb + tid];
        pre_w = a.q_w[qb + tid];
        pre_s0 = seg_row[t];
        pre_s1 = seg_row[t + 1];
    }

    // ---- initialise the accumulators: zero, or — when the query holds dense-head terms — their whole contribution.
    // Thread `tid` owns vecs r*NT + tid (4 consecutive docs each); the dense head is doc-major, one dword per doc and
    // term pair, so the owner scores two postings per v_dot2_u32_u16 and stores the sums with a plain ds_write_b128:
    // no atomics, and no separate zeroing pass. Term pairs the query does not hold are skipped (wave-uniform bit
    // mask); the rows of the next pair are in flight while the current pair is accumulated (two register banks).
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        constexpr int RG = 4;  // rounds per register group
        const uint32_t qv = lane < a.n_pairs ? a.q_dense[(uint64_t)q * a.n_pairs + lane] : 0u;
        const unsigned long long pmask = (DBG && (a.dbg & 16u)) ? 0ull : __ballot(qv != 0);
        const uint4* dblk = reinterpret_cast<const uint4*>(a.dense) + (uint64_t)tile_l * a.n_pairs * (TILE_DOCS / 4);
        for (int r0 = 0; r0 < rounds; r0 += RG) {
            uint4 sacc[RG];
#pragma unroll
            for (int i = 0; i < RG; ++i) sacc[i] = make_uint4(0, 0, 0, 0);
            if (pmask) {
                auto load_rows = [&](uint4 (&x)[RG], uint32_t p) {
#pragma unroll
                    for (int i = 0; i < RG; ++i)  // rows past `rounds` re-read the last real round (result unused)
                        x[i] = dblk[(uint64_t)p * (TILE_DOCS / 4) + (uint32_t)min(r0 + i, rounds - 1) * NT + tid];
                };
                auto add_rows = [&](const uint4 (&x)[RG], uint32_t qp) {
                    const us2 qq = __builtin_bit_cast(us2, qp);
#pragma unroll
                    for (int i = 0; i < RG; ++i) {
                        sacc[i].x = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x[i].x), qq, sacc[i].x, false);
                        sacc[i].y = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x[i].y), qq, sacc[i].y, false);
                        sacc[i].z = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x[i].z), qq, sacc[i].z, false);
                        sacc[i].w = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, x[i].w), qq, sacc[i].w, false);
                    }
                };
                unsigned long long m = pmask;
                uint4 xa[RG], xb[RG];
                uint32_t pa = (uint32_t)__builtin_ctzll(m), pb = 0;
                m &= m - 1;
                load_rows(xa, pa);
                for (;;) {
                    const bool more_b = m != 0;
                    if (more_b) {
                        pb = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        load_rows(xb, pb);
                    }
                    add_rows(xa, rdl(qv, pa));
                    if (!more_b) break;
                    const bool more_a = m != 0;
                    if (more_a) {
                        pa = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        load_rows(xa, pa);
                    }
                    add_rows(xb, rdl(qv, pb));
                    if (!more_a) break;
                }
            }
#pragma unroll
            for (int i = 0; i < RG; ++i)
                if (r0 + i < rounds) a4[(r0 + i) * NT + tid] = sacc[i];
        }
    }
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
    }

    for (uint32_t base = qb; base < qe; base += kQtBlock) {
        const uint32_t cnt = min((uint32_t)kQtBlock, qe - base);
        if (base != qb) __syncthreads();  // the previous round's readers of seg_* / pref are done (the accumulator
                                          // init is ordered before the atomics by the two barriers below)
        stamp(0);  // zeroing (+ q_ptr fetch)
        // ---- stage the round's segments and an exclusive prefix sum of their chunk counts
        uint32_t nch = 0;
        if (tid < kQtBlock) {
            if (tid < cnt) {
                uint32_t s0 = pre_s0, s1 = pre_s1, w = pre_w;
                if (base != qb) {
                    const uint32_t t = a.q_term[base + tid];
                    s0 = seg_row[t];
                    s1 = seg_row[t + 1];
                    w = a.q_w[base + tid];
                }
                seg_start[tid] = s0 - a.vec_base;
                seg_len[tid] = s1 - s0;
                seg_w[tid] = w;
                nch = (s1 - s0 + kChunkVecs - 1) / kChunkVecs;
            }
            const uint32_t inc = wave_inclusive_scan_u32(nch);
            if (lane == 63) wsum[wave] = inc;
            nch = inc - nch;  // exclusive within the wave
        }
        __syncthreads();
        if (tid < kQtBlock) {
            uint32_t off = 0;
            for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
            pref[tid] = nch + off;
            if (tid == kQtBlock - 1) pref[kQtBlock] = off + wsum[wave];
        }
        __syncthreads();
        // ---- the round's chunks are dealt round-robin to the waves (chunk c -> wave c % NW), which spreads the
        // dense head terms and the one-chunk tail terms evenly. The (term, offset) of a wave's next 64 chunks is
        // resolved lane-parallel (one binary search per lane), then broadcast chunk by chunk with v_readlane, so
        // the inner loop is scalar control + one 16-byte load and four LDS atomics per lane.
        stamp(1);  // staging: segment table + chunk-count scan
        if (DBG && (a.dbg & 64u)) break;  // ablation: stop after staging
        const uint32_t total = rfl(pref[kQtBlock]);
        const uint32_t c_end = total > wave ? (total - wave + NW - 1) / NW : 0u;  // chunks of this wave
        for (uint32_t cb = 0; cb < c_end; cb += 64) {
            const uint32_t my_i = cb + lane;
            uint32_t m_base = 0, m_n = 0, m_w = 0;
            if (my_i < c_end) {
                const uint32_t my_c = wave + my_i * NW;
                uint32_t lo = 0, hi = cnt;  // largest lo with pref[lo] <= my_c
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (pref[mid] <= my_c)
                        lo = mid;
                    else
                        hi = mid;
                }
                const uint32_t voff = (my_c - pref[lo]) * kChunkVecs;
                m_base = seg_start[lo] + voff;
                m_n = min((uint32_t)kChunkVecs, seg_len[lo] - voff);
                m_w = seg_w[lo];
            }
            stamp(7);  // lane-parallel chunk resolution (binary search)
            if (DBG && (a.dbg & 256u)) break;  // ablation: stop after the first chunk resolution
            const uint32_t nchunk = min(64u, c_end - cb);
            // Software pipeline, two register banks of U chunks: the next bank's 1-KiB loads are in flight while
            // the current bank's LDS atomics issue. Loads are unconditional (lanes past a chunk's end, and chunk
            // slots past nchunk, re-read the chunk's / the shard's first vec) so that the compiler can count them
            // with s_waitcnt vmcnt(N) instead of draining to vmcnt(0); only the atomics are predicated.
            auto load_bank = [&](uint4 (&v)[U], uint32_t u0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t idx = (u0 + u) & 63u;
                    const uint32_t b = rdl(m_base, idx);
                    const uint32_t n = rdl(m_n, idx);
                    if (DBG && (a.dbg & 2u)) {  // ablation: no global loads, synthetic postings
                        const uint32_t hsh = ((cb + idx) * 64u + lane) * 2654435761u;
                        v[u] = make_uint4((1u << 16) | (hsh >> 17), (1u << 16) | ((hsh * 31u) >> 17),
                                          (1u << 16) | ((hsh * 131u) >> 17), (1u << 16) | ((hsh * 1031u) >> 17));
                    } else {
                        v[u] = post4[b + (lane < n ? lane : 0u)];
                    }
                }
            };
            auto add_bank = [&](const uint4 (&v)[U], uint32_t u0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t idx = u0 + u;
                    const uint32_t n = idx < nchunk ? rdl(m_n, idx & 63u) : 0u;
                    const uint32_t w = rdl(m_w, idx & 63u);
                    // lanes past the chunk's last vec must not touch LDS (64 lanes adding to one accumulator would
                    // serialise); padding INSIDE a vec has weight 0 and a lane-distinct ordinal
                    if (lane < n) {
                        const uint32_t p[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        if (DBG && (a.dbg & 1u)) {  // ablation: no LDS atomics (keep the loads alive)
                            if ((p[0] ^ p[1] ^ p[2] ^ p[3]) == 0xDEADBEEFu) atomicAdd(&acc[0], 1u);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                atomicAdd(&acc[DBG ? (p[e] & 0xFFFFu) % TILE_DOCS : (p[e] & 0xFFFFu)],
                                          __umul24(p[e] >> 16, w));
                        }
                    }
                }
            };
            uint4 va[U], vb[U];
            load_bank(va, 0);
            for (uint32_t u0 = 0; u0 < nchunk; u0 += 2 * U) {
                const bool more = u0 + U < nchunk;  // wave-uniform
                if (more) load_bank(vb, u0 + U);
                add_bank(va, u0);
                if (more) {
                    if (u0 + 2 * U < nchunk) load_bank(va, u0 + 2 * U);
                    add_bank(vb, u0 + U);
                }
            }
        }
    }
    stamp(2);  // wave 0's own streaming
    __syncthreads();  // accumulation complete; the staging view of the union is dead from here on
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

    tile_select<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, ss, rounds, doc0, k, out, stamp);
    stamp(6);  // ranking + output
}

// ------------------------------------------------------------------------------------------------ kernel 1b
// Term-sharded search, after the reduce-scatter: rank `rank` holds the SUMMED accumulators of its doc range,
// src[(qi * tpr + t) * TILE_DOCS + i]; one workgroup per (tile of the range, query) selects the exact tile top-k.
struct SelectArgs {
    const uint32_t* src;
    uint64_t* part;   // [tpr][nq][k]
    uint64_t n_docs;
    uint32_t n_tiles; // tiles of the whole index
    uint32_t tpr;
    uint32_t rank;
    uint32_t nq, q0, qn, k;
};

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
    const uint32_t t = blockIdx.x / a.qn, qi = blockIdx.x % a.qn;
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
    tile_select<TILE_DOCS, NT, CAND>(a4, cand, tmax, wmax, ss, rounds, doc0, (int)a.k, out, [](int) {});
}

// ------------------------------------------------------------------------------------------------ kernel 2
struct MergeArgs {
    const uint64_t* lists;   // key(list, q, j) = lists[list*list_stride + q*k + j]; 0 = empty slot
    uint64_t list_stride;
    uint32_t n_lists;
    uint32_t nq;
    uint32_t k;
    uint64_t* out_keys;      // [nq][k] (may be null)
    uint32_t* out_ord;       // [nq][k] (may be null)
    uint32_t* out_score_u32;
    float* out_score;
    int32_t* out_n;
};

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
    if (n_items <= kCandCap) {
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
        uint32_t got = kCandCap + 1;
        if (a.n_lists >= (uint32_t)k) {
            const uint64_t tau_heads = bisect_kth(a.n_lists, [&](uint32_t l) { return key_at(l * a.k); });
            if (tau_heads) got = collect(tau_heads);
        }
        if (got > kCandCap) got = collect(bisect_kth(n_items, key_at));  // exactly min(k, #keys) <= kCandCap survive
        n_cand = min(got, (uint32_t)kCandCap);
    }
    rank_and_emit<NT>(cand, (int)n_cand, k, res);
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

// ------------------------------------------------------------------------------------------------ launch
static int launch_score(hipStream_t st, uint32_t tile_docs, uint32_t ntiles, const ScoreArgs& a, bool dump = false) {
    const uint64_t blocks = (uint64_t)ntiles * a.qn;
    if (blocks == 0) return MSR_OK;
    if (blocks > 0x7FFFFFFFull) {
        set_error("too many workgroups (%llu tiles x queries); split the batch", (unsigned long long)blocks);
        return MSR_E_RANGE;
    }
    switch (tile_docs) {
#define MSR_LAUNCH(T, N, UU, W, WR)                                                                           \
    if (dump)                                                                                                 \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 512, false, 1>), dim3((uint32_t)blocks), dim3(N), 0, st, a); \
    else if (a.dbg == 8u && a.k <= 512) /* stamps only: same shape as the production instance */             \
        hipLaunchKernelGGL((score_tiles<T, N, UU, W, 512, true>), dim3((uint32_t)blocks), dim3(N), 0, st, a);   \
    else if (a.dbg)                                                                                           \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 1024, true>), dim3((uint32_t)blocks), dim3(N), 0, st, a); \
    else if (a.k <= 512)                                                                                      \
        hipLaunchKernelGGL((score_tiles<T, N, UU, W, 512, false>), dim3((uint32_t)blocks), dim3(N), 0, st, a);  \
    else                                                                                                      \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 1024, false>), dim3((uint32_t)blocks), dim3(N), 0, st, a); \
    break;
        // <tile, threads, chunk loads per bank, min waves/SIMD of the k <= 512 instance, of the other instances>
        // LDS per workgroup = 4 B x tile + 4-8 KiB candidates + maxima -> workgroups (waves) per CU:
        case 32768: MSR_LAUNCH(32768, 1024, 8, 4, 4)  // 1 (16)
        case 16384: MSR_LAUNCH(16384, 512, 8, 4, 4)   // 2 (16)
        case 12288: MSR_LAUNCH(12288, 512, 4, 4, 4)   // 2 (16)
        case 8192: MSR_LAUNCH(8192, 512, 4, 8, 6)     // 4 (32) with k <= 512, 3 (24) above
        case 4096: MSR_LAUNCH(4096, 256, 4, 6, 5)     // 7 (28)
#undef MSR_LAUNCH
        default:
            set_error("no kernel instance for tile_docs=%u (supported: 4096, 8192, 12288, 16384, 32768)", tile_docs);
            return MSR_E_RANGE;
    }
    HIP_TRY(hipGetLastError());
    return MSR_OK;
}

static int launch_select(hipStream_t st, uint32_t tile_docs, const SelectArgs& a) {
    const uint64_t blocks = (uint64_t)a.tpr * a.qn;
    if (blocks == 0) return MSR_OK;
    if (blocks > 0x7FFFFFFFull) {
        set_error("too many workgroups in select_tiles");
        return MSR_E_RANGE;
    }
    switch (tile_docs) {
        case 32768: hipLaunchKernelGGL((select_tiles<32768, 1024, 1024>), dim3((uint32_t)blocks), dim3(1024), 0, st, a); break;
        case 16384: hipLaunchKernelGGL((select_tiles<16384, 512, 1024>), dim3((uint32_t)blocks), dim3(512), 0, st, a); break;
        case 12288: hipLaunchKernelGGL((select_tiles<12288, 512, 1024>), dim3((uint32_t)blocks), dim3(512), 0, st, a); break;
        case 8192: hipLaunchKernelGGL((select_tiles<8192, 512, 1024>), dim3((uint32_t)blocks), dim3(512), 0, st, a); break;
        case 4096: hipLaunchKernelGGL((select_tiles<4096, 256, 1024>), dim3((uint32_t)blocks), dim3(256), 0, st, a); break;
        default:
            set_error("no kernel instance for tile_docs=%u", tile_docs);
            return MSR_E_RANGE;
    }
    HIP_TRY(hipGetLastError());
    return MSR_OK;
}

static int launch_merge(hipStream_t st, const MergeArgs& a) {
    if (a.nq == 0) return MSR_OK;
    hipLaunchKernelGGL((merge_lists<256>), dim3(a.nq), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return MSR_OK;
}

// ------------------------------------------------------------------------------------------------ residency
int device_attach(msr_index* ix, int device) {
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        set_error("no usable HIP device (%s); this library has no CPU scoring path",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return MSR_E_NODEVICE;
    }
    if (device >= n_dev) {
        set_error("HIP device %d requested but only %d present", device, n_dev);
        return MSR_E_NODEVICE;
    }
    const IndexHeader* h = ix->host.h;
    switch (h->tile_docs) {
        case 4096: case 8192: case 12288: case 16384: case 32768: break;
        default:
            set_error("index tile_docs=%u has no kernel instance (supported: 4096, 8192, 12288, 16384, 32768)", h->tile_docs);
            return MSR_E_RANGE;
    }
    DeviceIndex* d = new (std::nothrow) DeviceIndex;
    if (!d) {
        set_error("out of host memory");
        return MSR_E_NOMEM;
    }
    d->device = device;
    ix->dev = d;
    ix->device = device;
    auto fail = [&](int rc) {
        device_detach(ix);
        return rc;
    };
    if (hipSetDevice(device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", device);
        return fail(MSR_E_HIP);
    }
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        return fail(MSR_E_HIP);
    }
    const uint64_t stride = (uint64_t)h->n_terms + 1;
    const uint32_t t0 = ix->shard_tile0, nt = ix->shard_ntiles;
    if (nt) {
        const uint32_t* sp = ix->host.seg_ptr + (uint64_t)t0 * stride;
        d->vec_base = sp[0];
        d->shard_vecs = (uint64_t)sp[(uint64_t)(nt - 1) * stride + h->n_terms] - d->vec_base;
        const size_t seg_bytes = (size_t)nt * stride * 4;
        // the kernel's unconditional loads may read the first 64 vecs of the shard even when it holds fewer
        const size_t post_bytes = std::max<size_t>((size_t)d->shard_vecs * 16, 64 * 16);
        if (hipMalloc(&d->d_seg_ptr, seg_bytes) != hipSuccess || hipMalloc(&d->d_postings, post_bytes) != hipSuccess) {
            set_error("hipMalloc of %zu + %zu bytes for the index shard failed", seg_bytes, post_bytes);
            return fail(MSR_E_NOMEM);
        }
        d->n_pairs = h->n_dense / 2;
        const size_t dense_bytes = std::max<size_t>((size_t)nt * d->n_pairs * h->tile_docs * 4, 16);
        if (hipMalloc(&d->d_dense, dense_bytes) != hipSuccess) {
            set_error("hipMalloc of %zu bytes for the dense head failed", dense_bytes);
            return fail(MSR_E_NOMEM);
        }
        if (d->n_pairs && hipMemcpy(d->d_dense, ix->host.dense + (uint64_t)t0 * d->n_pairs * h->tile_docs,
                                    (size_t)nt * d->n_pairs * h->tile_docs * 4, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("upload of the dense head failed");
            return fail(MSR_E_HIP);
        }
        if (hipMemcpy(d->d_seg_ptr, sp, seg_bytes, hipMemcpyHostToDevice) != hipSuccess ||
            (d->shard_vecs && hipMemcpy(d->d_postings, ix->host.postings + (uint64_t)d->vec_base * 4,
                                        (size_t)d->shard_vecs * 16, hipMemcpyHostToDevice) != hipSuccess)) {
            set_error("upload of the index shard failed");
            return fail(MSR_E_HIP);
        }
    }
    return MSR_OK;
}

void device_detach(msr_index* ix) {
    DeviceIndex* d = ix->dev;
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->comm) ncclCommDestroy(d->comm);
    if (d->d_seg_ptr) (void)hipFree(d->d_seg_ptr);
    if (d->d_postings) (void)hipFree(d->d_postings);
    if (d->d_dense) (void)hipFree(d->d_dense);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
    ix->dev = nullptr;
}

// postings of every term inside the shard: 4*vecs minus the zero padding of each segment's last vec
static void compute_df_shard(msr_index* ix) {
    DeviceIndex* d = ix->dev;
    if (d->df_shard_ready) return;
    const IndexHeader* h = ix->host.h;
    if (ix->shard_ntiles == h->n_tiles) {
        d->df_shard.assign(ix->host.df, ix->host.df + h->n_terms);
    } else {
        d->df_shard.assign(h->n_terms, 0);
        const uint64_t stride = (uint64_t)h->n_terms + 1;
        for (uint32_t t = ix->shard_tile0; t < ix->shard_tile0 + ix->shard_ntiles; ++t) {
            const uint32_t* sp = ix->host.seg_ptr + (uint64_t)t * stride;
            for (uint32_t v = 0; v < h->n_terms; ++v) {
                const uint32_t len = sp[v + 1] - sp[v];
                if (!len) continue;
                // zero padding lives in the segment's last chunk (chunk-interleaved layout, msr_internal.h)
                const uint32_t tail = len % kChunkVecs ? len % kChunkVecs : (uint32_t)kChunkVecs;
                const uint32_t* last = ix->host.postings + ((uint64_t)sp[v + 1] - tail) * 4;
                uint32_t zeros = 0;
                for (uint32_t i = 0; i < tail * 4; ++i) zeros += (last[i] >> 16) == 0;
                d->df_shard[v] += len * 4 - zeros;
            }
            const uint32_t np = h->n_dense / 2;
            const uint32_t* dt = ix->host.dense + (uint64_t)t * np * h->tile_docs;
            for (uint32_t s2 = 0; s2 < h->n_dense; ++s2) {
                const uint32_t term = ix->host.dense_terms[s2];
                if (term == 0xFFFFFFFFu) continue;
                const uint32_t* row = dt + (uint64_t)(s2 >> 1) * h->tile_docs;
                const uint32_t sh = 16 * (s2 & 1);
                uint32_t c = 0;
                for (uint32_t i = 0; i < h->tile_docs; ++i) c += ((row[i] >> sh) & 0xFFFFu) != 0;
                d->df_shard[term] += c;
            }
        }
    }
    d->df_shard_ready = true;
}

}  // namespace msr

// ================================================================================================ batches
using namespace msr;

struct msr_batch {
    msr_index* ix = nullptr;
    int nq = 0;
    int kmax = 0;
    int last_k = 0;
    uint64_t nnz = 0;             // kept query entries
    uint64_t sum_df = 0;          // sum over kept entries of df_shard(term)
    uint32_t* d_qptr = nullptr;
    uint32_t* d_qterm = nullptr;
    uint32_t* d_qw = nullptr;
    uint32_t* d_qdense = nullptr; // [nq][n_pairs]
    uint64_t* d_part = nullptr;   // [ntiles][nq][kmax]
    uint64_t* d_keys = nullptr;   // [nq][kmax] local top-k keys
    uint64_t* d_gather = nullptr; // [n_ranks][nq][kmax] (sharded search)
    uint32_t* d_ord = nullptr;
    uint32_t* d_su32 = nullptr;
    float* d_sf32 = nullptr;
    int32_t* d_n = nullptr;
    unsigned long long* d_stamps = nullptr;  // diagnostic (MSR_DEBUG_FLAGS bit 3)
    // term-sharded search
    uint32_t* d_S = nullptr;      // [G][Qt][tpr*tile] partial accumulators of one query tile (reduce-scatter send buffer)
    uint32_t* d_R = nullptr;      // [Qt][tpr*tile] summed accumulators of this rank's doc range
    uint64_t* d_tpart = nullptr;  // [tpr][nq][kmax] per-tile keys of this rank's doc range
    size_t S_elems = 0, R_elems = 0, tpart_elems = 0;
    int term_shard = -1, term_nshards = 0;  // >= 0: the batch holds only the query terms of that term range
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;  // the current call's events (borrowed from `events`)
    std::vector<hipEvent_t> events;  // 3 per recorded search call since the last timing reset
    size_t calls = 0;                // recorded calls
    bool timed = false;
};

static void batch_free(msr_batch* b) {
    if (!b) return;
    if (b->ix && b->ix->dev) (void)hipSetDevice(b->ix->dev->device);
    void* ptrs[] = {b->d_qptr, b->d_qterm, b->d_qw, b->d_qdense, b->d_part, b->d_keys, b->d_gather, b->d_ord, b->d_su32, b->d_sf32, b->d_n,
                    b->d_stamps, b->d_S, b->d_R, b->d_tpart};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : b->events)
        if (e) (void)hipEventDestroy(e);
    delete b;
}

extern "C" {

}  // extern "C"

// Term ownership for term-range sharding: G contiguous term-id ranges balanced by postings (cumulative df), the
// same on every rank because it depends only on the index. bounds has G+1 entries.
static void term_bounds(const msr::HostIndex& hx, int G, std::vector<uint32_t>& bounds) {
    const uint32_t V = hx.h->n_terms;
    uint64_t total = 0;
    for (uint32_t v = 0; v < V; ++v) total += hx.df[v];
    bounds.assign((size_t)G + 1, V);
    bounds[0] = 0;
    uint64_t acc = 0;
    int g = 1;
    for (uint32_t v = 0; v < V && g < G; ++v) {
        acc += hx.df[v];
        while (g < G && acc * (uint64_t)G >= total * (uint64_t)g) bounds[g++] = v + 1;
    }
}

static int batch_create_impl(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                             int kmax, uint32_t flags, int shard, int n_shards, msr_batch** out) {
    if (!out) {
        set_error("msr_batch_create: null output");
        return MSR_E_INVAL;
    }
    *out = nullptr;
    if (!ix || nq < 0 || !q_ptr) {
        set_error("msr_batch_create: bad argument");
        return MSR_E_INVAL;
    }
    if (!ix->dev) {
        set_error("index handle has no HIP device bound (opened with device < 0); there is no CPU scoring path");
        return MSR_E_NODEVICE;
    }
    if (kmax < 1 || kmax > MSR_KMAX) {
        set_error("k must be in [1, %d] (got %d)", MSR_KMAX, kmax);
        return MSR_E_RANGE;
    }
    const IndexHeader* h = ix->host.h;
    DeviceIndex* d = ix->dev;
    compute_df_shard(ix);
    uint32_t term_lo = 0, term_hi = h->n_terms;
    if (n_shards > 0) {
        if (shard < 0 || shard >= n_shards) {
            set_error("bad term shard %d of %d", shard, n_shards);
            return MSR_E_INVAL;
        }
        std::vector<uint32_t> tb;
        term_bounds(ix->host, n_shards, tb);
        term_lo = tb[shard];
        term_hi = tb[shard + 1];
    }

    // ---- host-side query normalisation (what pyserini does before handing the query to Lucene):
    // OOV (term < 0) and non-positive weights vanish, terms present in every doc are dropped when asked,
    // and the worst-case score must fit the u32 accumulators.
    std::vector<uint32_t> qptr((size_t)nq + 1, 0), qterm, qw;
    const int64_t total_in = q_ptr[nq];
    if (total_in < 0 || (total_in && (!q_term || !q_w))) {
        set_error("msr_batch_create: bad CSR arrays");
        return MSR_E_INVAL;
    }
    qterm.reserve((size_t)total_in);
    qw.reserve((size_t)total_in);
    const uint32_t n_pairs = h->n_dense / 2;
    std::vector<uint32_t> qdense((size_t)nq * n_pairs, 0u);  // packed 16-bit query weights of the dense-head terms
    std::vector<uint32_t> dsum(h->n_dense);
    uint64_t n_kept = 0;
    uint64_t sum_df = 0;
    for (int i = 0; i < nq; ++i) {
        if (q_ptr[i + 1] < q_ptr[i] || q_ptr[i + 1] > total_in) {
            set_error("q_ptr is not monotone at query %d", i);
            return MSR_E_INVAL;
        }
        uint64_t bound = 0;
        std::fill(dsum.begin(), dsum.end(), 0u);
        for (int64_t e = q_ptr[i]; e < q_ptr[i + 1]; ++e) {
            const int32_t t = q_term[e];
            const int32_t w = q_w[e];
            if (t < 0 || w <= 0) continue;
            if ((uint32_t)t >= h->n_terms) {
                set_error("query %d: term id %d is outside the dictionary (%u terms)", i, t, h->n_terms);
                return MSR_E_RANGE;
            }
            if (w > 0xFFFFFF) {  // the kernel multiplies with v_mul_u32_u24
                set_error("query %d: weight %d of term %d exceeds the supported maximum 16777215", i, w, t);
                return MSR_E_RANGE;
            }
            if ((flags & MSR_F_DROP_DF_EQ_N) && ix->host.df[t] == h->n_docs) continue;
            if (ix->host.df[t] == 0) continue;
            bound += (uint64_t)w * ix->host.maxw[t];  // the bound covers the WHOLE query, whoever owns the term
            if ((uint32_t)t < term_lo || (uint32_t)t >= term_hi) continue;
            sum_df += d->df_shard[t];
            ++n_kept;
            const int ds = ix->host.dense_slot[t];
            if (ds >= 0) {  // dense-head term: repeated entries add up; v_dot2_u32_u16 takes 16-bit weights
                if ((uint64_t)dsum[ds] + (uint64_t)w > 0xFFFFull) {
                    set_error("query %d: weight of dense-head term %d exceeds the supported maximum 65535", i, t);
                    return MSR_E_RANGE;
                }
                dsum[ds] += (uint32_t)w;
            } else {
                qterm.push_back((uint32_t)t);
                qw.push_back((uint32_t)w);
            }
        }
        for (uint32_t s2 = 0; s2 < h->n_dense; ++s2)
            qdense[(size_t)i * n_pairs + (s2 >> 1)] |= dsum[s2] << (16 * (s2 & 1));
        if (bound > 0xFFFFFFFFull) {
            set_error("query %d: worst-case score %llu exceeds the exact u32 range", i, (unsigned long long)bound);
            return MSR_E_OVERFLOW;
        }
        if (qterm.size() > 0xFFFFFFF0ull) {
            set_error("query batch too large");
            return MSR_E_RANGE;
        }
        qptr[i + 1] = (uint32_t)qterm.size();
    }

    msr_batch* b = new (std::nothrow) msr_batch;
    if (!b) {
        set_error("out of host memory");
        return MSR_E_NOMEM;
    }
    b->ix = ix;
    b->nq = nq;
    b->kmax = kmax;
    b->nnz = n_kept;
    b->term_shard = n_shards > 0 ? shard : -1;
    b->term_nshards = n_shards > 0 ? n_shards : 0;
    b->sum_df = sum_df;
    auto fail = [&](int rc) {
        batch_free(b);
        return rc;
    };
    if (hipSetDevice(d->device) != hipSuccess) {
        set_error("hipSetDevice failed");
        return fail(MSR_E_HIP);
    }
    const size_t nqk = std::max<size_t>((size_t)nq * kmax, 1);
    const size_t ntiles = std::max<uint32_t>(ix->shard_ntiles, 1);
    bool ok = hipMalloc(&b->d_qptr, ((size_t)nq + 1) * 4) == hipSuccess &&
              hipMalloc(&b->d_qterm, std::max<size_t>(qterm.size(), 1) * 4) == hipSuccess &&
              hipMalloc(&b->d_qw, std::max<size_t>(qw.size(), 1) * 4) == hipSuccess &&
              hipMalloc(&b->d_qdense, std::max<size_t>(qdense.size(), 1) * 4) == hipSuccess &&
              hipMalloc(&b->d_part, ntiles * nqk * 8) == hipSuccess && hipMalloc(&b->d_keys, nqk * 8) == hipSuccess &&
              hipMalloc(&b->d_ord, nqk * 4) == hipSuccess && hipMalloc(&b->d_su32, nqk * 4) == hipSuccess &&
              hipMalloc(&b->d_sf32, nqk * 4) == hipSuccess &&
              hipMalloc(&b->d_n, std::max<size_t>(nq, 1) * 4) == hipSuccess;
    if (!ok) {
        set_error("hipMalloc for the query batch failed (%d queries, kmax %d, %zu tiles)", nq, kmax, ntiles);
        return fail(MSR_E_NOMEM);
    }
    ok = (qdense.empty() || hipMemcpy(b->d_qdense, qdense.data(), qdense.size() * 4, hipMemcpyHostToDevice) == hipSuccess) &&
         hipMemcpy(b->d_qptr, qptr.data(), ((size_t)nq + 1) * 4, hipMemcpyHostToDevice) == hipSuccess &&
         (qterm.empty() || (hipMemcpy(b->d_qterm, qterm.data(), qterm.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                            hipMemcpy(b->d_qw, qw.data(), qw.size() * 4, hipMemcpyHostToDevice) == hipSuccess));
    if (!ok) {
        set_error("upload of the query batch failed");
        return fail(MSR_E_HIP);
    }
    *out = b;
    return MSR_OK;
}

extern "C" {

int msr_batch_create(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int kmax,
                     uint32_t flags, msr_batch** out) {
    return batch_create_impl(ix, q_ptr, q_term, q_w, nq, kmax, flags, 0, 0, out);
}

int msr_batch_create_termshard(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                               int kmax, uint32_t flags, int shard, int n_shards, msr_batch** out) {
    if (n_shards < 1) {
        set_error("msr_batch_create_termshard: n_shards must be >= 1");
        return MSR_E_INVAL;
    }
    return batch_create_impl(ix, q_ptr, q_term, q_w, nq, kmax, flags, shard, n_shards, out);
}

static int batch_search_local(msr_batch* b, int k, bool final_arrays) {
    msr_index* ix = b->ix;
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    HIP_TRY(hipSetDevice(d->device));
    // every call gets its own event triple so that a whole timed region can be summed afterwards
    if (b->calls >= 4096) b->calls = 0;  // bounded: callers that never reset keep only the recent calls
    while (b->events.size() < (b->calls + 1) * 3) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        b->events.push_back(e);
    }
    b->ev0 = b->events[b->calls * 3 + 0];
    b->ev1 = b->events[b->calls * 3 + 1];
    b->ev2 = b->events[b->calls * 3 + 2];
    b->calls++;
    HIP_TRY(hipEventRecord(b->ev0, d->stream));
    ScoreArgs sa;
    sa.seg_ptr = d->d_seg_ptr;
    sa.postings = d->d_postings;
    sa.q_ptr = b->d_qptr;
    sa.q_term = b->d_qterm;
    sa.q_w = b->d_qw;
    sa.dense = d->d_dense;
    sa.q_dense = b->d_qdense;
    sa.n_pairs = d->n_pairs;
    sa.part = b->d_part;
    sa.n_docs = h->n_docs;
    sa.vec_base = d->vec_base;
    sa.n_terms = h->n_terms;
    sa.tile0 = ix->shard_tile0;
    sa.nq = (uint32_t)b->nq;
    sa.q0 = 0;
    sa.qn = (uint32_t)b->nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.tpr = 1;
    sa.dump_add = 0;
    {
        const char* dbg = getenv("MSR_DEBUG_FLAGS");
        sa.dbg = dbg ? (uint32_t)strtoul(dbg, nullptr, 0) : 0u;
        sa.stamps = nullptr;
        if (sa.dbg & 8u) {
            if (!b->d_stamps) {
                HIP_TRY(hipMalloc(&b->d_stamps, 8 * sizeof(unsigned long long)));
                HIP_TRY(hipMemsetAsync(b->d_stamps, 0, 8 * sizeof(unsigned long long), d->stream));
            }
            sa.stamps = b->d_stamps;
        }
    }
    // one launch holds at most 2^31-1 workgroups (tiles x queries): larger batches go in query ranges
    int rc = MSR_OK;
    const uint32_t q_step = ix->shard_ntiles ? std::max<uint32_t>(0x7FFFFFFFu / ix->shard_ntiles, 1u) : (uint32_t)b->nq;
    for (uint32_t q0 = 0; q0 < (uint32_t)b->nq; q0 += q_step) {
        sa.q0 = q0;
        sa.qn = std::min<uint32_t>(q_step, (uint32_t)b->nq - q0);
        rc = launch_score(d->stream, h->tile_docs, ix->shard_ntiles, sa);
        if (rc != MSR_OK) return rc;
    }
    HIP_TRY(hipEventRecord(b->ev1, d->stream));
    MergeArgs ma;
    ma.lists = b->d_part;
    ma.list_stride = (uint64_t)b->nq * k;
    ma.n_lists = ix->shard_ntiles;
    ma.nq = (uint32_t)b->nq;
    ma.k = (uint32_t)k;
    ma.out_keys = b->d_keys;
    ma.out_ord = final_arrays ? b->d_ord : nullptr;
    ma.out_score_u32 = b->d_su32;
    ma.out_score = b->d_sf32;
    ma.out_n = b->d_n;
    rc = launch_merge(d->stream, ma);
    if (rc != MSR_OK) return rc;
    HIP_TRY(hipEventRecord(b->ev2, d->stream));
    b->last_k = k;
    b->timed = true;
    return MSR_OK;
}

int msr_batch_search(msr_batch* b, int k) {
    if (!b) {
        set_error("msr_batch_search: null batch");
        return MSR_E_INVAL;
    }
    if (k < 1 || k > b->kmax) {
        set_error("k=%d outside [1, kmax=%d]", k, b->kmax);
        return MSR_E_RANGE;
    }
    return batch_search_local(b, k, true);
}

int msr_batch_sync(msr_batch* b) {
    if (!b) {
        set_error("msr_batch_sync: null batch");
        return MSR_E_INVAL;
    }
    HIP_TRY(hipSetDevice(b->ix->dev->device));
    HIP_TRY(hipStreamSynchronize(b->ix->dev->stream));
    return MSR_OK;
}

int msr_batch_fetch(msr_batch* b, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    if (!b || b->last_k == 0) {
        set_error("msr_batch_fetch: no search has run on this batch");
        return MSR_E_INVAL;
    }
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    const size_t n = (size_t)b->nq * b->last_k;
    if (n) {
        if (out_doc_ord) HIP_TRY(hipMemcpy(out_doc_ord, b->d_ord, n * 4, hipMemcpyDeviceToHost));
        if (out_score) HIP_TRY(hipMemcpy(out_score, b->d_sf32, n * 4, hipMemcpyDeviceToHost));
        if (out_score_u32) HIP_TRY(hipMemcpy(out_score_u32, b->d_su32, n * 4, hipMemcpyDeviceToHost));
    }
    if (out_n && b->nq) HIP_TRY(hipMemcpy(out_n, b->d_n, (size_t)b->nq * 4, hipMemcpyDeviceToHost));
    return MSR_OK;
}

int msr_batch_kernel_ms(msr_batch* b, float* score_ms, float* merge_ms) {
    if (!b || !b->timed) {
        set_error("msr_batch_kernel_ms: no search has run on this batch");
        return MSR_E_INVAL;
    }
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    float a = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, b->ev0, b->ev1));
    HIP_TRY(hipEventElapsedTime(&c, b->ev1, b->ev2));
    if (score_ms) *score_ms = a;
    if (merge_ms) *merge_ms = c;
    return MSR_OK;
}

int msr_batch_timing_reset(msr_batch* b) {
    if (!b) {
        set_error("msr_batch_timing_reset: null batch");
        return MSR_E_INVAL;
    }
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    b->calls = 0;
    return MSR_OK;
}

int msr_batch_timing_sum(msr_batch* b, int* n_calls, float* score_ms, float* merge_ms) {
    if (!b) {
        set_error("msr_batch_timing_sum: null batch");
        return MSR_E_INVAL;
    }
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    double a = 0, c = 0;
    for (size_t i = 0; i < b->calls; ++i) {
        float x = 0, y = 0;
        HIP_TRY(hipEventElapsedTime(&x, b->events[i * 3], b->events[i * 3 + 1]));
        HIP_TRY(hipEventElapsedTime(&y, b->events[i * 3 + 1], b->events[i * 3 + 2]));
        a += x;
        c += y;
    }
    if (n_calls) *n_calls = (int)b->calls;
    if (score_ms) *score_ms = (float)a;
    if (merge_ms) *merge_ms = (float)c;
    return MSR_OK;
}

int msr_batch_debug_stamps(msr_batch* b, unsigned long long out[8]) {
    if (!b || !out) {
        set_error("msr_batch_debug_stamps: bad argument");
        return MSR_E_INVAL;
    }
    memset(out, 0, 8 * sizeof(unsigned long long));
    if (!b->d_stamps) return MSR_OK;
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    HIP_TRY(hipMemcpy(out, b->d_stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MSR_OK;
}

int msr_batch_algo_bytes(const msr_batch* b, int k, uint64_t* bytes, uint64_t* postings) {
    if (!b || k < 1) {
        set_error("msr_batch_algo_bytes: bad argument");
        return MSR_E_INVAL;
    }
    // bytes(q) = sum_t df(t)*(4+2) + |q|*12 + k*8
    if (bytes) *bytes = b->sum_df * 6 + b->nnz * 12 + (uint64_t)b->nq * k * 8;
    if (postings) *postings = b->sum_df;
    return MSR_OK;
}

void msr_batch_destroy(msr_batch* b) { batch_free(b); }

int msr_search_csr(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int k,
                   uint32_t flags, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    msr_batch* b = nullptr;
    int rc = msr_batch_create(ix, q_ptr, q_term, q_w, nq, k, flags, &b);
    if (rc != MSR_OK) return rc;
    rc = msr_batch_search(b, k);
    if (rc == MSR_OK) rc = msr_batch_fetch(b, out_doc_ord, out_score, out_score_u32, out_n);
    msr_batch_destroy(b);
    return rc;
}

// ------------------------------------------------------------------------------------------------ merge of host lists
int msr_merge_lists(msr_index* ix, int n_lists, int nq, int k, const uint32_t* doc_ord, const uint32_t* score_u32,
                    const int32_t* n, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    if (!ix || n_lists < 1 || nq < 0 || !doc_ord || !score_u32 || !n) {
        set_error("msr_merge_lists: bad argument");
        return MSR_E_INVAL;
    }
    if (!ix->dev) {
        set_error("index handle has no HIP device bound; there is no CPU merge path");
        return MSR_E_NODEVICE;
    }
    if (k < 1 || k > MSR_KMAX) {
        set_error("k must be in [1, %d]", MSR_KMAX);
        return MSR_E_RANGE;
    }
    DeviceIndex* d = ix->dev;
    HIP_TRY(hipSetDevice(d->device));
    const size_t per = (size_t)nq * k;
    std::vector<uint64_t> keys((size_t)n_lists * per, 0);
    for (int l = 0; l < n_lists; ++l)
        for (int q = 0; q < nq; ++q) {
            const int cnt = std::min(std::max(n[(size_t)l * nq + q], 0), k);
            for (int j = 0; j < cnt; ++j) {
                const size_t o = (size_t)l * per + (size_t)q * k + j;
                if (score_u32[o]) keys[o] = ((uint64_t)score_u32[o] << 32) | (uint64_t)(0xFFFFFFFFu - doc_ord[o]);
            }
        }
    uint64_t* d_lists = nullptr;
    uint32_t *d_ord = nullptr, *d_su = nullptr;
    float* d_sf = nullptr;
    int32_t* d_n = nullptr;
    const size_t perz = std::max<size_t>(per, 1);
    int rc = MSR_OK;
    bool ok = hipMalloc(&d_lists, std::max<size_t>(keys.size(), 1) * 8) == hipSuccess &&
              hipMalloc(&d_ord, perz * 4) == hipSuccess && hipMalloc(&d_su, perz * 4) == hipSuccess &&
              hipMalloc(&d_sf, perz * 4) == hipSuccess && hipMalloc(&d_n, std::max<size_t>(nq, 1) * 4) == hipSuccess;
    if (!ok) {
        set_error("hipMalloc failed in msr_merge_lists");
        rc = MSR_E_NOMEM;
    }
    if (rc == MSR_OK && !keys.empty() &&
        hipMemcpy(d_lists, keys.data(), keys.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("upload failed in msr_merge_lists");
        rc = MSR_E_HIP;
    }
    if (rc == MSR_OK) {
        MergeArgs ma;
        ma.lists = d_lists;
        ma.list_stride = per;
        ma.n_lists = (uint32_t)n_lists;
        ma.nq = (uint32_t)nq;
        ma.k = (uint32_t)k;
        ma.out_keys = nullptr;
        ma.out_ord = d_ord;
        ma.out_score_u32 = d_su;
        ma.out_score = d_sf;
        ma.out_n = d_n;
        rc = launch_merge(d->stream, ma);
    }
    if (rc == MSR_OK && hipStreamSynchronize(d->stream) != hipSuccess) {
        set_error("merge kernel failed");
        rc = MSR_E_HIP;
    }
    if (rc == MSR_OK && per) {
        bool c = (!out_doc_ord || hipMemcpy(out_doc_ord, d_ord, per * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
                 (!out_score || hipMemcpy(out_score, d_sf, per * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
                 (!out_score_u32 || hipMemcpy(out_score_u32, d_su, per * 4, hipMemcpyDeviceToHost) == hipSuccess) &&
                 (!out_n || hipMemcpy(out_n, d_n, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess);
        if (!c) {
            set_error("download failed in msr_merge_lists");
            rc = MSR_E_HIP;
        }
    }
    void* ptrs[] = {d_lists, d_ord, d_su, d_sf, d_n};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    return rc;
}

// ------------------------------------------------------------------------------------------------ RCCL exchange
int msr_comm_unique_id(char id[MSR_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= MSR_COMM_ID_BYTES, "id buffer too small");
    if (!id) {
        set_error("msr_comm_unique_id: null buffer");
        return MSR_E_INVAL;
    }
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) {
        set_error("ncclGetUniqueId failed: %s", ncclGetErrorString(r));
        return MSR_E_COMM;
    }
    memset(id, 0, MSR_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return MSR_OK;
}

int msr_comm_init(msr_index* ix, int n_ranks, int rank, const char id[MSR_COMM_ID_BYTES]) {
    if (!ix || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
        set_error("msr_comm_init: bad argument");
        return MSR_E_INVAL;
    }
    if (!ix->dev) {
        set_error("index handle has no HIP device bound");
        return MSR_E_NODEVICE;
    }
    DeviceIndex* d = ix->dev;
    HIP_TRY(hipSetDevice(d->device));
    if (d->comm) {
        ncclCommDestroy(d->comm);
        d->comm = nullptr;
    }
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&d->comm, n_ranks, u, rank);
    if (r != ncclSuccess) {
        d->comm = nullptr;
        set_error("ncclCommInitRank(%d of %d) failed: %s", rank, n_ranks, ncclGetErrorString(r));
        return MSR_E_COMM;
    }
    d->n_ranks = n_ranks;
    d->rank = rank;
    return MSR_OK;
}

int msr_comm_destroy(msr_index* ix) {
    if (!ix || !ix->dev) return MSR_OK;
    if (ix->dev->comm) {
        (void)hipSetDevice(ix->dev->device);
        ncclCommDestroy(ix->dev->comm);
        ix->dev->comm = nullptr;
    }
    ix->dev->n_ranks = 1;
    ix->dev->rank = 0;
    return MSR_OK;
}

// all-gather of every rank's [nq][k] keys (d_keys) + exact merge of the n_ranks lists into the result arrays
static int exchange_and_merge(msr_batch* b, int k) {
    DeviceIndex* d = b->ix->dev;
    const size_t per = std::max<size_t>((size_t)b->nq * b->kmax, 1);
    if (!b->d_gather) {
        if (hipMalloc(&b->d_gather, per * 8 * d->n_ranks) != hipSuccess) {
            set_error("hipMalloc of the all-gather buffer failed");
            return MSR_E_NOMEM;
        }
    }
    const size_t cnt = (size_t)b->nq * k;
    if (cnt) {
        ncclResult_t r = ncclAllGather(b->d_keys, b->d_gather, cnt, ncclUint64, d->comm, d->stream);
        if (r != ncclSuccess) {
            set_error("ncclAllGather failed: %s", ncclGetErrorString(r));
            return MSR_E_COMM;
        }
    }
    MergeArgs ma;
    ma.lists = b->d_gather;
    ma.list_stride = cnt;
    ma.n_lists = (uint32_t)d->n_ranks;
    ma.nq = (uint32_t)b->nq;
    ma.k = (uint32_t)k;
    ma.out_keys = nullptr;
    ma.out_ord = b->d_ord;
    ma.out_score_u32 = b->d_su32;
    ma.out_score = b->d_sf32;
    ma.out_n = b->d_n;
    int rc = launch_merge(d->stream, ma);
    if (rc != MSR_OK) return rc;
    HIP_TRY(hipEventRecord(b->ev2, d->stream));
    return MSR_OK;
}

static int check_sharded_call(msr_batch* b, int k, const char* who) {
    if (!b) {
        set_error("%s: null batch", who);
        return MSR_E_INVAL;
    }
    if (k < 1 || k > b->kmax) {
        set_error("k=%d outside [1, kmax=%d]", k, b->kmax);
        return MSR_E_RANGE;
    }
    if (!b->ix->dev->comm) {
        set_error("%s: call msr_comm_init first", who);
        return MSR_E_COMM;
    }
    return MSR_OK;
}

int msr_batch_search_sharded(msr_batch* b, int k) {
    int rc = check_sharded_call(b, k, "msr_batch_search_sharded");
    if (rc != MSR_OK) return rc;
    HIP_TRY(hipSetDevice(b->ix->dev->device));
    rc = batch_search_local(b, k, false);  // per-shard exact top-k keys in d_keys ([nq][k])
    if (rc != MSR_OK) return rc;
    return exchange_and_merge(b, k);
}

// ------------------------------------------------------------------------------------------------ term-range shards
// Partial sums are not mergeable by top-k alone, so the exact protocol moves the accumulators:
//   every rank scores its OWN TERM RANGE of every query against ALL docs and dumps the accumulator tiles,
//   ncclReduceScatter(sum) hands rank r the complete sums of doc range r, rank r selects its range's exact top-k,
//   and the per-range lists are all-gathered and merged as for doc-range shards.
struct TermShardPlan {
    uint32_t G, tpr, tile;
    uint64_t range_elems;  // tpr * tile accumulators per query and doc range
    uint32_t qt;           // queries per pass
};

static TermShardPlan term_plan(const msr_index* ix, int G, int nq) {
    TermShardPlan p;
    const IndexHeader* h = ix->host.h;
    p.G = (uint32_t)G;
    p.tile = h->tile_docs;
    p.tpr = (h->n_tiles + G - 1) / G;
    p.range_elems = (uint64_t)p.tpr * p.tile;
    const uint64_t per_query = p.range_elems * G * 4;                     // bytes of S per query
    const uint64_t budget = 4ull << 30;                                     // 4 GiB send buffer
    p.qt = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max(nq, 1), budget / std::max<uint64_t>(per_query, 1)));
    return p;
}

static int ensure_buf(void** p, size_t* have, size_t want_elems, size_t elem_bytes, const char* what) {
    if (*p && *have >= want_elems) return MSR_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    if (hipMalloc(p, std::max<size_t>(want_elems, 1) * elem_bytes) != hipSuccess) {
        set_error("hipMalloc of %zu bytes for %s failed", want_elems * elem_bytes, what);
        *have = 0;
        return MSR_E_NOMEM;
    }
    *have = want_elems;
    return MSR_OK;
}

static void fill_score_args(ScoreArgs& sa, msr_batch* b, int k) {
    msr_index* ix = b->ix;
    DeviceIndex* d = ix->dev;
    sa.seg_ptr = d->d_seg_ptr;
    sa.postings = d->d_postings;
    sa.q_ptr = b->d_qptr;
    sa.q_term = b->d_qterm;
    sa.q_w = b->d_qw;
    sa.dense = d->d_dense;
    sa.q_dense = b->d_qdense;
    sa.n_pairs = d->n_pairs;
    sa.part = b->d_part;
    sa.n_docs = ix->host.h->n_docs;
    sa.vec_base = d->vec_base;
    sa.n_terms = ix->host.h->n_terms;
    sa.tile0 = ix->shard_tile0;
    sa.nq = (uint32_t)b->nq;
    sa.q0 = 0;
    sa.qn = (uint32_t)b->nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.tpr = 1;
    sa.dump_add = 0;
    sa.dbg = 0;
    sa.stamps = nullptr;
}

int msr_batch_search_termshard(msr_batch* b, int k) {
    int rc = check_sharded_call(b, k, "msr_batch_search_termshard");
    if (rc != MSR_OK) return rc;
    msr_index* ix = b->ix;
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    if (ix->shard_ntiles != h->n_tiles) {
        set_error("term-sharded search needs a handle that holds every doc tile (open it with msr_index_open)");
        return MSR_E_INVAL;
    }
    if (b->term_nshards != d->n_ranks || b->term_shard != d->rank) {
        set_error("the batch was created for term shard %d/%d but the communicator is rank %d/%d", b->term_shard,
                  b->term_nshards, d->rank, d->n_ranks);
        return MSR_E_INVAL;
    }
    HIP_TRY(hipSetDevice(d->device));
    const TermShardPlan p = term_plan(ix, d->n_ranks, b->nq);
    rc = ensure_buf((void**)&b->d_S, &b->S_elems, (size_t)p.qt * p.range_elems * p.G, 4, "the reduce-scatter send buffer");
    if (rc == MSR_OK && p.G > 1)
        rc = ensure_buf((void**)&b->d_R, &b->R_elems, (size_t)p.qt * p.range_elems, 4, "the reduce-scatter receive buffer");
    if (rc == MSR_OK)
        rc = ensure_buf((void**)&b->d_tpart, &b->tpart_elems, (size_t)p.tpr * b->nq * b->kmax, 8, "the per-tile keys");
    if (rc != MSR_OK) return rc;
    if (b->calls >= 4096) b->calls = 0;
    while (b->events.size() < (b->calls + 1) * 3) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        b->events.push_back(e);
    }
    b->ev0 = b->events[b->calls * 3 + 0];
    b->ev1 = b->events[b->calls * 3 + 1];
    b->ev2 = b->events[b->calls * 3 + 2];
    b->calls++;
    HIP_TRY(hipEventRecord(b->ev0, d->stream));
    for (uint32_t q0 = 0; q0 < (uint32_t)b->nq; q0 += p.qt) {
        const uint32_t qn = std::min<uint32_t>(p.qt, (uint32_t)b->nq - q0);
        const size_t send = (size_t)qn * p.range_elems * p.G;
        HIP_TRY(hipMemsetAsync(b->d_S, 0, send * 4, d->stream));
        ScoreArgs sa;
        fill_score_args(sa, b, k);
        sa.q0 = q0;
        sa.qn = qn;
        sa.dump = b->d_S;
        sa.tpr = p.tpr;
        rc = launch_score(d->stream, h->tile_docs, h->n_tiles, sa, true);
        if (rc != MSR_OK) return rc;
        const uint32_t* reduced = b->d_S;
        if (p.G > 1) {
            ncclResult_t r = ncclReduceScatter(b->d_S, b->d_R, (size_t)qn * p.range_elems, ncclUint32, ncclSum, d->comm, d->stream);
            if (r != ncclSuccess) {
                set_error("ncclReduceScatter failed: %s", ncclGetErrorString(r));
                return MSR_E_COMM;
            }
            reduced = b->d_R;
        }
        SelectArgs se;
        se.src = reduced;
        se.part = b->d_tpart;
        se.n_docs = h->n_docs;
        se.n_tiles = h->n_tiles;
        se.tpr = p.tpr;
        se.rank = (uint32_t)d->rank;
        se.nq = (uint32_t)b->nq;
        se.q0 = q0;
        se.qn = qn;
        se.k = (uint32_t)k;
        rc = launch_select(d->stream, h->tile_docs, se);
        if (rc != MSR_OK) return rc;
    }
    HIP_TRY(hipEventRecord(b->ev1, d->stream));
    MergeArgs ma;
    ma.lists = b->d_tpart;
    ma.list_stride = (uint64_t)b->nq * k;
    ma.n_lists = p.tpr;
    ma.nq = (uint32_t)b->nq;
    ma.k = (uint32_t)k;
    ma.out_keys = b->d_keys;
    ma.out_ord = nullptr;
    ma.out_score_u32 = b->d_su32;
    ma.out_score = b->d_sf32;
    ma.out_n = b->d_n;
    rc = launch_merge(d->stream, ma);
    if (rc != MSR_OK) return rc;
    b->last_k = k;
    b->timed = true;
    return exchange_and_merge(b, k);
}

// The same protocol played on ONE GPU for `n_shards` logical term shards (tests, and a cross-check of the partition):
// the shards' dumps are summed in place (dump_add) instead of by ncclReduceScatter; every logical rank then selects
// its doc range and the n_shards lists are merged.
int msr_search_termshard_emulated(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                                  int k, uint32_t flags, int n_shards, uint32_t* out_doc_ord, float* out_score,
                                  uint32_t* out_score_u32, int32_t* out_n) {
    if (!ix || n_shards < 1 || n_shards > 64) {
        set_error("msr_search_termshard_emulated: bad argument");
        return MSR_E_INVAL;
    }
    if (!ix->dev) {
        set_error("index handle has no HIP device bound; there is no CPU scoring path");
        return MSR_E_NODEVICE;
    }
    if (ix->shard_ntiles != ix->host.h->n_tiles) {
        set_error("term-sharded search needs a handle that holds every doc tile");
        return MSR_E_INVAL;
    }
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    std::vector<msr_batch*> bs((size_t)n_shards, nullptr);
    uint32_t* d_S = nullptr;
    uint64_t *d_tpart = nullptr, *d_lists = nullptr;
    int rc = MSR_OK;
    auto cleanup = [&]() {
        for (msr_batch* x : bs) batch_free(x);
        if (d_S) (void)hipFree(d_S);
        if (d_tpart) (void)hipFree(d_tpart);
        if (d_lists) (void)hipFree(d_lists);
    };
    for (int g = 0; g < n_shards && rc == MSR_OK; ++g)
        rc = batch_create_impl(ix, q_ptr, q_term, q_w, nq, k, flags, g, n_shards, &bs[g]);
    if (rc != MSR_OK) {
        cleanup();
        return rc;
    }
    const TermShardPlan p = term_plan(ix, n_shards, nq);
    const size_t per = std::max<size_t>((size_t)nq * k, 1);
    if (hipSetDevice(d->device) != hipSuccess ||
        hipMalloc(&d_S, std::max<size_t>((size_t)p.qt * p.range_elems * p.G, 1) * 4) != hipSuccess ||
        hipMalloc(&d_tpart, std::max<size_t>((size_t)p.tpr * per, 1) * 8) != hipSuccess ||
        hipMalloc(&d_lists, per * 8 * n_shards) != hipSuccess) {
        set_error("hipMalloc failed in msr_search_termshard_emulated");
        cleanup();
        return MSR_E_NOMEM;
    }
    // pass 1..: accumulators of every query tile, summed over the logical term shards; selection per logical rank
    // writes into that rank's slice of d_lists after a per-rank merge of its tpr tile lists.
    std::vector<uint64_t*> rank_part((size_t)n_shards, nullptr);
    for (int r = 0; r < n_shards && rc == MSR_OK; ++r)
        if (hipMalloc(&rank_part[r], std::max<size_t>((size_t)p.tpr * per, 1) * 8) != hipSuccess) {
            set_error("hipMalloc failed in msr_search_termshard_emulated");
            rc = MSR_E_NOMEM;
        }
    for (uint32_t q0 = 0; q0 < (uint32_t)nq && rc == MSR_OK; q0 += p.qt) {
        const uint32_t qn = std::min<uint32_t>(p.qt, (uint32_t)nq - q0);
        if (hipMemsetAsync(d_S, 0, (size_t)qn * p.range_elems * p.G * 4, d->stream) != hipSuccess) {
            set_error("hipMemsetAsync failed");
            rc = MSR_E_HIP;
            break;
        }
        for (int g = 0; g < n_shards && rc == MSR_OK; ++g) {
            ScoreArgs sa;
            fill_score_args(sa, bs[g], k);
            sa.q0 = q0;
            sa.qn = qn;
            sa.dump = d_S;
            sa.tpr = p.tpr;
            sa.dump_add = 1;
            rc = launch_score(d->stream, h->tile_docs, h->n_tiles, sa, true);
        }
        for (int r = 0; r < n_shards && rc == MSR_OK; ++r) {
            SelectArgs se;
            se.src = d_S + (size_t)r * qn * p.range_elems;
            se.part = rank_part[r];
            se.n_docs = h->n_docs;
            se.n_tiles = h->n_tiles;
            se.tpr = p.tpr;
            se.rank = (uint32_t)r;
            se.nq = (uint32_t)nq;
            se.q0 = q0;
            se.qn = qn;
            se.k = (uint32_t)k;
            rc = launch_select(d->stream, h->tile_docs, se);
        }
    }
    for (int r = 0; r < n_shards && rc == MSR_OK; ++r) {
        MergeArgs ma;
        ma.lists = rank_part[r];
        ma.list_stride = (uint64_t)nq * k;
        ma.n_lists = p.tpr;
        ma.nq = (uint32_t)nq;
        ma.k = (uint32_t)k;
        ma.out_keys = d_lists + (size_t)r * nq * k;
        ma.out_ord = nullptr;
        ma.out_score_u32 = nullptr;
        ma.out_score = nullptr;
        ma.out_n = nullptr;
        rc = launch_merge(d->stream, ma);
    }
    if (rc == MSR_OK) {
        msr_batch* b0 = bs[0];
        MergeArgs ma;
        ma.lists = d_lists;
        ma.list_stride = (uint64_t)nq * k;
        ma.n_lists = (uint32_t)n_shards;
        ma.nq = (uint32_t)nq;
        ma.k = (uint32_t)k;
        ma.out_keys = nullptr;
        ma.out_ord = b0->d_ord;
        ma.out_score_u32 = b0->d_su32;
        ma.out_score = b0->d_sf32;
        ma.out_n = b0->d_n;
        rc = launch_merge(d->stream, ma);
        if (rc == MSR_OK) {
            b0->last_k = k;
            rc = msr_batch_fetch(b0, out_doc_ord, out_score, out_score_u32, out_n);
        }
    }
    (void)hipStreamSynchronize(d->stream);
    for (uint64_t* x : rank_part)
        if (x) (void)hipFree(x);
    cleanup();
    return rc;
}

}  // extern "C"

// ================================================================================================ dense (hybrid path)
// Flat inner-product search over fp16 passage vectors: the dense half of the reference's hybrid search
// (tevatron FaissFlatSearcher / faiss IndexFlatIP, fp16 storage on GPU: src/search.py:232-237,254-270; queries
// normalised at src/search.py:342, corpus at src/encode.py:301). Scores C[q][d] = sum_k Q[q][k] * P[d][k] on MFMA
// (v_mfma_f32_32x32x16_f16, f32 accumulate), written as order-preserving u32 keys into the accumulator layout of
// select_tiles, so that top-`depth` selection and the tile merge are the SAME kernels as on the sparse path.
namespace msr {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t f32_to_key(float f) {  // monotone: a < b  <=>  key(a) < key(b); never 0 for finite f
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// One workgroup = 4 waves = a 128 (queries) x 128 (docs) block, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32.
// K runs in steps of 32 through a double-buffered LDS stage: 128 rows x 32 halves per operand, row stride 80 B
// (5 sixteen-byte slots: 5r mod 16 is a bijection, so the 16-lane groups of ds_read_b128 hit 16 distinct slots).
// The next K-step's global loads (2 x 16 B per operand per thread) are in flight while the current step's 8 MFMAs run.
// Fragment map of v_mfma_f32_32x32x16_f16: lane (r = l & 31, h = l >> 5) holds elements k = 8h .. 8h+7 of row r of A
// and of column r of B (= row r of P). Q has Mpad rows, P has Npad rows (multiples of 128, zero padded), H % 32 == 0.
constexpr int kGemmRowB = 80;                    // LDS row stride in bytes (64 B of data + 16 B pad)
constexpr int kGemmTileB = 128 * kGemmRowB;      // one operand stage

__global__ __launch_bounds__(256) void dense_scores(const _Float16* __restrict__ Q, const _Float16* __restrict__ P,
                                                    uint32_t* __restrict__ out, uint32_t M, uint32_t N, uint32_t H,
                                                    uint64_t ld) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2][2][kGemmTileB];  // [buffer][A|B]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const uint32_t q_blk = blockIdx.x * 128, d_blk = blockIdx.y * 128;
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
                if (q < M) out[(uint64_t)q * ld + d] = d < N ? f32_to_key(acc[i][j][e]) : 0u;
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
    if (dx->d_P) (void)hipFree(dx->d_P);
    if (dx->stream) (void)hipStreamDestroy(dx->stream);
    delete dx;
}

// to_device = true: out_* are DEVICE buffers ([nq][k] / [nq]) filled on dx->stream (hybrid path); else host buffers.
static int dense_search_impl(msr_dense* dx, const uint16_t* q_fp16, int nq, int k, uint32_t* out_idx, uint32_t* out_key,
                             int32_t* out_n, float* gemm_ms, float* select_ms, bool to_device) {
    if (!dx || nq < 0 || (nq && !q_fp16) || !out_idx || !out_key || !out_n) {
        set_error("msr_dense_search: bad argument");
        return MSR_E_INVAL;
    }
    if (k < 1 || k > MSR_KMAX) {
        set_error("k must be in [1, %d] (got %d)", MSR_KMAX, k);
        return MSR_E_RANGE;
    }
    HIP_TRY(hipSetDevice(dx->device));
    const uint32_t QT = 8192;  // queries per pass: scores buffer QT x n_pad u32
    const uint32_t qt = (uint32_t)std::min<uint32_t>(QT, std::max(nq, 1));
    const uint32_t qt_pad = (qt + 127) / 128 * 128;
    _Float16* d_Q = nullptr;
    uint32_t* d_S = nullptr;
    uint64_t *d_part = nullptr;
    uint32_t *d_ord = nullptr, *d_su = nullptr;
    float* d_sf = nullptr;
    int32_t* d_n = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    int rc = MSR_OK;
    const size_t perq = std::max<size_t>((size_t)qt * k, 1);
    bool ok = hipMalloc(&d_Q, (size_t)qt_pad * dx->h * 2) == hipSuccess &&
              hipMalloc(&d_S, (size_t)qt_pad * dx->n_pad * 4) == hipSuccess &&
              hipMalloc(&d_part, (size_t)dx->n_tiles * perq * 8) == hipSuccess && hipMalloc(&d_ord, perq * 4) == hipSuccess &&
              hipMalloc(&d_su, perq * 4) == hipSuccess && hipMalloc(&d_sf, perq * 4) == hipSuccess &&
              hipMalloc(&d_n, (size_t)qt * 4) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
              hipEventCreate(&e1) == hipSuccess && hipEventCreate(&e2) == hipSuccess;
    if (!ok) {
        set_error("hipMalloc failed in msr_dense_search (%u queries per pass x %llu docs)", qt, (unsigned long long)dx->n_pad);
        rc = MSR_E_NOMEM;
    }
    double t_gemm = 0, t_sel = 0;
    for (int q0 = 0; q0 < nq && rc == MSR_OK; q0 += (int)qt) {
        const uint32_t qn = (uint32_t)std::min<int>((int)qt, nq - q0);
        const uint32_t qn_pad = (qn + 127) / 128 * 128;
        bool c = hipMemsetAsync(d_Q, 0, (size_t)qn_pad * dx->h * 2, dx->stream) == hipSuccess &&
                 hipMemcpyAsync(d_Q, q_fp16 + (size_t)q0 * dx->h, (size_t)qn * dx->h * 2, hipMemcpyHostToDevice,
                                dx->stream) == hipSuccess &&
                 hipEventRecord(e0, dx->stream) == hipSuccess;
        if (!c) {
            set_error("query upload failed in msr_dense_search");
            rc = MSR_E_HIP;
            break;
        }
        hipLaunchKernelGGL(dense_scores, dim3(qn_pad / 128, (uint32_t)(dx->n_pad / 128)), dim3(256), 0, dx->stream, d_Q,
                           dx->d_P, d_S, qn, (uint32_t)dx->n, dx->h, dx->n_pad);
        (void)hipEventRecord(e1, dx->stream);
        SelectArgs se;
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
    void* ptrs[] = {d_Q, d_S, d_part, d_ord, d_su, d_sf, d_n};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    return rc;
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

__device__ __forceinline__ float key_to_f32(uint32_t key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}

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
    __shared__ uint32_t ns_sh;
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
        ns_sh = 0;
    }
    __syncthreads();
    for (uint32_t j = tid; j < a.depth; j += NT)
        if (sk[j]) atomicMax(&ns_sh, j + 1);  // sparse hit count = index after the last non-empty slot
    __syncthreads();
    if (tid == 0) {
        // lists are best-first: max = first entry, min = last non-empty entry
        const uint32_t ns = ns_sh;
        const float smax = ns ? (float)(uint32_t)(sk[0] >> 32) : 0.f, smin = ns ? (float)(uint32_t)(sk[ns - 1] >> 32) : 0.f;
        const float dmax = dn > 0 ? key_to_f32(dk[0]) : 0.f, dmin = dn > 0 ? key_to_f32(dk[dn - 1]) : 0.f;
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
                                     a.part + ((uint64_t)tile * a.nq + q) * a.k, [](int) {});
}

}  // namespace msr

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
    if (rc == MSR_OK) rc = batch_search_local(b, depth, false);  // sparse top-depth keys -> b->d_keys
    if (rc == MSR_OK) rc = dense_search_impl(dx, q_fp16, nq, depth, d_didx, d_dkey, d_dn, &t_gemm, &t_sel, true);
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
