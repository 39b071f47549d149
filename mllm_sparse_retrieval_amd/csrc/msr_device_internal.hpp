// Shared by the two HIP translation units (msr_device.hip, msr_hybrid.hip): device-side state of an index, kernel
// argument blocks, the resident query batch, and the few host functions that cross the unit boundary.
#pragma once

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
constexpr uint32_t kMaxGridY = 65535;  // HIP grid limit in y (tiles per launch)
constexpr uint64_t kMinStagedPairs = 4096;  // ... and only launches with at least this many (tile, query) pairs are staged
constexpr uint32_t kStage1Fraction = 16;  // staged search: 1/16 of the tiles (at least one) set the thresholds
constexpr size_t kZeroCopyBytes = 64u << 10;  // query block / result block of a small call (mapped host memory)
constexpr uint32_t kSmallMaxPairs = 1024;     // a small call launches at most this many (tile, query) workgroups
constexpr int kChunkVecs = 64;   // one chunk = one wave-wide uint4 load = 256 postings = 1 KiB
static_assert(kCandCap >= MSR_KMAX, "candidate buffer must hold k keys");

// Caching device allocator of one index handle. msr_search_csr at the reference's call shape (4 queries per call,
// scripts/search_sparse.sh:16) would otherwise spend more time in hipMalloc / hipFree than in its kernels.
struct DevicePool {
    static constexpr size_t kMaxCachedBlock = 64ull << 20, kMaxCachedTotal = 512ull << 20;
    std::vector<std::pair<size_t, void*>> free_;  // (bytes, ptr)
    size_t cached = 0;
    static size_t round_up(size_t b) {
        size_t r = 256;
        while (r < b) r <<= 1;
        return r;
    }
    void* alloc(size_t bytes) {
        const size_t want = round_up(std::max<size_t>(bytes, 1));
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].first == want) {
                void* p = free_[i].second;
                free_[i] = free_.back();
                free_.pop_back();
                cached -= want;
                return p;
            }
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        return p;
    }
    void release(void* p, size_t bytes) {
        if (!p) return;
        const size_t have = round_up(std::max<size_t>(bytes, 1));
        if (have > kMaxCachedBlock || cached + have > kMaxCachedTotal) {
            (void)hipFree(p);
            return;
        }
        free_.emplace_back(have, p);
        cached += have;
    }
    void purge() {
        for (auto& e : free_) (void)hipFree(e.second);
        free_.clear();
        cached = 0;
    }
};

struct DeviceIndex {
    int device = -1;
    hipStream_t stream = nullptr;
    DevicePool pool;
    std::vector<hipEvent_t> spare_events;  // recycled by the batches of this handle
    void* h_stage = nullptr;               // pinned staging for small uploads / downloads
    size_t h_stage_bytes = 0;
    hipEvent_t stage_ev = nullptr;         // recorded after an async upload out of h_stage ...
    bool stage_pending = false;            // ... and waited for before the buffer is written or freed again
    // small calls (the reference's 4 queries per batch_search): the kernels read the queries from, and write the results
    // to, mapped pinned host memory — no upload and no download on the stream (each costs it ~3.5 us, scripts/latency_lab.hip)
    void* h_zc = nullptr;                  // hipHostMalloc(mapped), 2 x kZeroCopyBytes: queries | results
    void* d_zc = nullptr;                  // its device view
    uint32_t* d_seg_ptr = nullptr;   // [shard_ntiles][n_terms+1] absolute vec index
    uint32_t* d_postings = nullptr;  // the shard's vecs; vec v of the index lives at d_postings + (v - vec_base)*4
    uint32_t* d_dense = nullptr;     // [shard_ntiles][n_pairs][tile_docs] dense head of the shard's tiles
    uint32_t n_pairs = 0;            // dense-head pairs resident per tile (all of them, or those with an owned term)
    std::vector<int32_t> pair_local; // index pair p -> resident pair (or -1); identity unless the handle is a term shard
    uint32_t seg_terms = 0;          // terms per row of d_seg_ptr (n_terms, or term_hi - term_lo on a term shard)
    uint32_t term_base = 0;          // query term ids are rebased by this before upload (term_lo on a term shard)
    uint64_t resident_bytes = 0;     // d_seg_ptr + d_postings + d_dense
    uint32_t vec_base = 0;
    uint64_t shard_vecs = 0;
    std::vector<uint32_t> df_shard;  // postings of each term inside this shard (for algorithmic bytes)
    bool df_shard_ready = false;
    // exchange
    ncclComm_t comm = nullptr;
    int n_ranks = 1;
    int rank = 0;
};

// ------------------------------------------------------------------------------------------------ kernel arguments
struct ScoreArgs {
    const uint32_t* seg_ptr;   // [ntiles][n_terms+1]
    const uint32_t* postings;  // shard base
    const uint4* q_meta;       // [nq] {first term, end term, bit mask of the query's dense-head pairs, 0}
    const uint32_t* q_term;
    const uint32_t* q_w;
    const uint32_t* dense;     // [ntiles][n_pairs][TILE_DOCS] dense head (weights of term 2p+1 << 16 | term 2p)
    const uint32_t* q_dense;   // [nq][n_pairs] packed query weights of the dense-head terms (0 = absent)
    uint32_t n_pairs;
    uint64_t* part;            // [ntiles][nq][k] keys
    const uint64_t* theta;     // [nq][k] best keys of an earlier launch's tiles (staged search), or null
    uint32_t unsorted;         // 1: the tile's k keys may be emitted in any order (single-tile hybrid lists)
    uint64_t n_docs;           // whole index
    uint32_t vec_base;
    uint32_t n_terms;
    uint32_t tile0;            // first (global) tile of the shard
    uint32_t tl0;              // first shard-local tile of this launch (grid y is limited to 65535 tiles)
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
    uint32_t light;            // 1: every query of the batch holds <= 64 sparse terms (score_tiles<..., LIGHT>)
    uint32_t dbg;              // MSR_DEBUG_FLAGS (timing ablations only; results are wrong when bits 0-2 are set)
    unsigned long long* stamps;  // [8] summed s_memtime deltas of wave 0 per phase (dbg bit 3), else null
};

struct SelectArgs {
    const uint32_t* src;
    uint64_t* part;   // [tpr][nq][k]
    uint64_t n_docs;
    uint32_t n_tiles; // tiles of the whole index
    uint32_t tpr;
    uint32_t rank;
    uint32_t nq, q0, qn, k;
    uint32_t unsorted;  // as ScoreArgs::unsorted
};

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

// defined in msr_device.hip
int launch_select(hipStream_t st, uint32_t tile_docs, const SelectArgs& a);
int launch_merge(hipStream_t st, const MergeArgs& a);

}  // namespace msr

struct msr_batch {
    msr_index* ix = nullptr;
    bool unsorted_ok = false;  // hybrid path: the consumer of d_keys treats each query's list as a set
    bool zero_copy = false;    // small call: d_qptr .. d_qdense and d_ord .. d_n are device views of the handle's mapped host block
    bool untimed = false;      // no HIP events around the kernels (msr_search_csr's own batches)
    int nq = 0;
    int kmax = 0;
    int last_k = 0;
    uint64_t nnz = 0;             // kept query entries
    uint64_t sum_df = 0;          // sum over kept entries of df_shard(term)
    uint64_t sum_df_dense = 0;    // ... of which the dense-head terms' (scored doc-major, no inverted list)
    uint32_t max_sparse_terms = 0; // most sparse (non dense-head) terms in one query: <= 64 selects the light kernel
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
    std::vector<std::pair<void*, size_t>> pooled;  // blocks taken from the index's DevicePool (the d_* above point into them)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;  // the current call's events (borrowed from `events`)
    std::vector<hipEvent_t> events;  // 3 per recorded search call since the last timing reset
    size_t calls = 0;                // recorded calls
    bool timed = false;
};

// defined in msr_device.hip
void batch_free(msr_batch* b);          // release + unregister from the index + delete
void batch_release_device(msr_batch* b); // device buffers and events only (the index is being closed)
extern "C" int batch_search_local(msr_batch* b, int k, bool final_arrays);  // internal, not part of include/msr.h
