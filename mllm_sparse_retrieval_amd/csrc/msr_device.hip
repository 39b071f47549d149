// Host side of the HIP search path for gfx950 (MI355X): kernel launchers, device residency of the tile-major inverted
// index, resident query batches, the host-list merge, the RCCL exchange for doc-range shards and the term-range
// protocol. The kernels themselves live in msr_kernels.hpp / msr_select.hpp.
//
// Replaces what runs below `LuceneImpactSearcher.batch_search(queries, qids, k, threads)` in the reference
// (call site src/search.py:86-87; Lucene impact scoring):
//     score(d) = sum over query terms t of  q_w(t) * tf(t, d),   top-k of the docs with score > 0,
//     ties broken by external doc id ascending (= lower ordinal).
#include <dlfcn.h>

#include <chrono>

#include "msr_kernels.hpp"

namespace msr {

// ------------------------------------------------------------------------------------------------ launch
// scores the shard-local tiles [t_begin, t_end) for queries [a0.q0, a0.q0 + a0.qn)
static int launch_score(hipStream_t st, uint32_t tile_docs, uint32_t t_begin, uint32_t t_end, const ScoreArgs& a0,
                        bool dump = false) {
    if (t_end <= t_begin || a0.qn == 0) return MSR_OK;
    // grid = (queries, tiles): x runs fastest in dispatch order, so the workgroups in flight share a tile
    for (uint32_t tl0 = t_begin; tl0 < t_end; tl0 += kMaxGridY) {
        ScoreArgs a = a0;
        a.tl0 = tl0;
        const dim3 grid(a.qn, std::min<uint32_t>(kMaxGridY, t_end - tl0));
        switch (tile_docs) {
#define MSR_LAUNCH(T, N, UU, W, WR)                                                                \
    if (dump)                                                                                      \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 512, false, 1>), grid, dim3(N), 0, st, a);   \
    else if (a.dbg == 8u && a.k <= 512) /* stamps only: same shape as the production instance */  \
        hipLaunchKernelGGL((score_tiles<T, N, UU, W, 512, true>), grid, dim3(N), 0, st, a);        \
    else if (a.dbg)                                                                                \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 1024, true>), grid, dim3(N), 0, st, a);      \
    else if (a.k <= 512 && a.light && (T == 8192 || T == 4096))                                    \
        hipLaunchKernelGGL((score_tiles<T, N, UU, W, 512, false, 0, (T == 8192 || T == 4096)>), grid, dim3(N), 0, st, a); \
    else if (a.k <= 512)                                                                           \
        hipLaunchKernelGGL((score_tiles<T, N, UU, W, 512, false>), grid, dim3(N), 0, st, a);       \
    else                                                                                           \
        hipLaunchKernelGGL((score_tiles<T, N, UU, WR, 1024, false>), grid, dim3(N), 0, st, a);     \
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
    }
    return MSR_OK;
}

int launch_select(hipStream_t st, uint32_t tile_docs, const SelectArgs& a) {
    if (a.tpr == 0 || a.qn == 0) return MSR_OK;
    if (a.tpr > kMaxGridY) {
        set_error("select_tiles: %u tiles per rank exceed the grid limit of %u", a.tpr, kMaxGridY);
        return MSR_E_RANGE;
    }
    const dim3 grid(a.qn, a.tpr);  // x = query, y = tile of the rank's range
    switch (tile_docs) {
        case 32768: hipLaunchKernelGGL((select_tiles<32768, 1024, 1024>), grid, dim3(1024), 0, st, a); break;
        case 16384: hipLaunchKernelGGL((select_tiles<16384, 512, 1024>), grid, dim3(512), 0, st, a); break;
        case 12288: hipLaunchKernelGGL((select_tiles<12288, 512, 1024>), grid, dim3(512), 0, st, a); break;
        case 8192: hipLaunchKernelGGL((select_tiles<8192, 512, 1024>), grid, dim3(512), 0, st, a); break;
        case 4096: hipLaunchKernelGGL((select_tiles<4096, 256, 1024>), grid, dim3(256), 0, st, a); break;
        default:
            set_error("no kernel instance for tile_docs=%u", tile_docs);
            return MSR_E_RANGE;
    }
    HIP_TRY(hipGetLastError());
    return MSR_OK;
}

int launch_merge(hipStream_t st, const MergeArgs& a) {
    if (a.nq == 0) return MSR_OK;
    if ((uint64_t)a.n_lists * a.k <= 64 && a.k <= 64)
        hipLaunchKernelGGL(merge_small, dim3((a.nq + 3) / 4), dim3(256), 0, st, a);  // one wave per query
    else
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
    // (optional: without it every call takes the ordinary path)
    if (hipHostMalloc(&d->h_zc, 2 * kZeroCopyBytes, hipHostMallocMapped) == hipSuccess) {
        if (hipHostGetDevicePointer(&d->d_zc, d->h_zc, 0) != hipSuccess) d->d_zc = nullptr;
    } else {
        d->h_zc = nullptr;
    }
    const uint64_t stride = (uint64_t)h->n_terms + 1;
    const uint32_t t0 = ix->shard_tile0, nt = ix->shard_ntiles;
    const bool by_terms = ix->term_nshards > 0;
    const uint32_t lo = by_terms ? ix->term_lo : 0u, hi = by_terms ? ix->term_hi : h->n_terms;
    d->seg_terms = hi - lo;
    d->term_base = lo;
    // dense-head pairs this handle keeps: all of them, or (term shard) those holding at least one owned term
    d->pair_local.assign(h->n_dense / 2, -1);
    d->n_pairs = 0;
    for (uint32_t p = 0; p < h->n_dense / 2; ++p) {
        bool keep = !by_terms;
        for (uint32_t s2 = 2 * p; s2 < 2 * p + 2 && !keep; ++s2) {
            const uint32_t term = ix->host.dense_terms[s2];
            keep = term != 0xFFFFFFFFu && term >= lo && term < hi;
        }
        if (keep) d->pair_local[p] = (int32_t)d->n_pairs++;
    }
    if (nt) {
        const uint32_t* sp = ix->host.seg_ptr + (uint64_t)t0 * stride;
        // the kernels trust seg_ptr: a non-monotone or out-of-range table (corrupt file) must not reach the GPU
        {
            uint32_t prev = sp[0];
            bool bad = false;
            for (uint64_t i = 1; i < (uint64_t)nt * stride && !bad; ++i) {
                bad = sp[i] < prev;
                prev = sp[i];
            }
            if (bad || (uint64_t)prev > h->n_vecs) {
                set_error("index file is corrupt: the segment table of tiles [%u, %u) is not monotone / exceeds the postings",
                          t0, t0 + nt);
                return fail(MSR_E_FORMAT);
            }
            // score_tiles addresses a tile's postings with 32-bit byte offsets from the tile's first vec
            for (uint32_t t = 0; t < nt; ++t) {
                const uint64_t span = (uint64_t)sp[(uint64_t)t * stride + h->n_terms] - sp[(uint64_t)t * stride];
                if (span >= (1ull << 28)) {
                    set_error("tile %u holds %llu posting vecs (>= 2^28): build the index with a smaller tile_docs",
                              t0 + t, (unsigned long long)span);
                    return fail(MSR_E_RANGE);
                }
            }
        }
        const size_t dense_row = (size_t)h->tile_docs * 4;
        const size_t dense_bytes = std::max<size_t>((size_t)nt * d->n_pairs * dense_row, 16);
        if (hipMalloc(&d->d_dense, dense_bytes) != hipSuccess) {
            set_error("hipMalloc of %zu bytes for the dense head failed", dense_bytes);
            return fail(MSR_E_NOMEM);
        }
        const uint32_t np_all = h->n_dense / 2;
        if (d->n_pairs == np_all) {
            if (np_all && hipMemcpy(d->d_dense, ix->host.dense + (uint64_t)t0 * np_all * h->tile_docs,
                                    (size_t)nt * np_all * dense_row, hipMemcpyHostToDevice) != hipSuccess) {
                set_error("upload of the dense head failed");
                return fail(MSR_E_HIP);
            }
        } else {
            for (uint32_t t = 0; t < nt; ++t)
                for (uint32_t p = 0; p < np_all; ++p) {
                    if (d->pair_local[p] < 0) continue;
                    if (hipMemcpy(reinterpret_cast<char*>(d->d_dense) + ((size_t)t * d->n_pairs + (size_t)d->pair_local[p]) * dense_row,
                                  ix->host.dense + ((uint64_t)(t0 + t) * np_all + p) * h->tile_docs, dense_row,
                                  hipMemcpyHostToDevice) != hipSuccess) {
                        set_error("upload of the dense head failed");
                        return fail(MSR_E_HIP);
                    }
                }
        }
        if (!by_terms) {
            d->vec_base = sp[0];
            d->shard_vecs = (uint64_t)sp[(uint64_t)(nt - 1) * stride + h->n_terms] - d->vec_base;
            const size_t seg_bytes = (size_t)nt * stride * 4;
            // the kernel's unconditional loads may read the first 64 vecs of the shard even when it holds fewer
            const size_t post_bytes = std::max<size_t>((size_t)d->shard_vecs * 16, 64 * 16);
            if (hipMalloc(&d->d_seg_ptr, seg_bytes) != hipSuccess || hipMalloc(&d->d_postings, post_bytes) != hipSuccess) {
                set_error("hipMalloc of %zu + %zu bytes for the index shard failed", seg_bytes, post_bytes);
                return fail(MSR_E_NOMEM);
            }
            if (hipMemcpy(d->d_seg_ptr, sp, seg_bytes, hipMemcpyHostToDevice) != hipSuccess ||
                (d->shard_vecs && hipMemcpy(d->d_postings, ix->host.postings + (uint64_t)d->vec_base * 4,
                                            (size_t)d->shard_vecs * 16, hipMemcpyHostToDevice) != hipSuccess)) {
                set_error("upload of the index shard failed");
                return fail(MSR_E_HIP);
            }
            d->resident_bytes = seg_bytes + (size_t)d->shard_vecs * 16 + (size_t)nt * d->n_pairs * dense_row;
        } else {
            // Term-range shard: segments are term-ordered inside a tile, so the owned terms of tile t are ONE slice
            // [sp[t][lo], sp[t][hi]) of its postings. The slices are packed back to back, and the segment table keeps
            // only the owned columns, rebased to the packed positions: score_tiles runs unchanged on (term id - lo).
            const uint64_t cols = (uint64_t)d->seg_terms + 1;
            std::vector<uint32_t> seg((size_t)nt * cols);
            uint64_t total = 0;
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t* row = sp + (uint64_t)t * stride;
                const uint32_t a = row[lo];
                for (uint64_t j = 0; j < cols; ++j) seg[(size_t)t * cols + j] = (uint32_t)(total + (row[lo + j] - a));
                total += (uint64_t)row[hi] - a;
                if (total > 0xFFFFFFFFull) {
                    set_error("term shard holds more than 2^32 posting vecs");
                    return fail(MSR_E_RANGE);
                }
            }
            d->vec_base = 0;
            d->shard_vecs = total;
            const size_t seg_bytes = seg.size() * 4, post_bytes = std::max<size_t>((size_t)total * 16, 64 * 16);
            if (hipMalloc(&d->d_seg_ptr, seg_bytes) != hipSuccess || hipMalloc(&d->d_postings, post_bytes) != hipSuccess) {
                set_error("hipMalloc of %zu + %zu bytes for the term shard failed", seg_bytes, post_bytes);
                return fail(MSR_E_NOMEM);
            }
            bool ok = hipMemcpy(d->d_seg_ptr, seg.data(), seg_bytes, hipMemcpyHostToDevice) == hipSuccess;
            for (uint32_t t = 0; t < nt && ok; ++t) {
                const uint32_t* row = sp + (uint64_t)t * stride;
                const uint64_t n = (uint64_t)row[hi] - row[lo];
                if (n)
                    ok = hipMemcpy(d->d_postings + (uint64_t)seg[(size_t)t * cols] * 4,
                                   ix->host.postings + (uint64_t)row[lo] * 4, (size_t)n * 16, hipMemcpyHostToDevice) == hipSuccess;
            }
            if (!ok) {
                set_error("upload of the term shard failed");
                return fail(MSR_E_HIP);
            }
            d->resident_bytes = seg_bytes + (size_t)total * 16 + (size_t)nt * d->n_pairs * dense_row;
        }
    }
    return MSR_OK;
}

uint64_t device_resident_bytes(const msr_index* ix) { return ix->dev ? ix->dev->resident_bytes : 0; }

void device_detach(msr_index* ix) {
    DeviceIndex* d = ix->dev;
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);  // nothing may still read the staging buffer or the index
    // batches that outlive the handle: their device buffers go now (into the pool that is purged below), the host
    // objects stay with their owners and refuse further use
    for (msr_batch* b : ix->live_batches) {
        batch_release_device(b);
        b->ix = nullptr;
    }
    ix->live_batches.clear();
    if (d->comm) ncclCommDestroy(d->comm);
    d->pool.purge();
    for (hipEvent_t e : d->spare_events) (void)hipEventDestroy(e);
    if (d->stage_ev) (void)hipEventDestroy(d->stage_ev);
    if (d->h_stage) (void)hipHostFree(d->h_stage);
    if (d->h_zc) (void)hipHostFree(d->h_zc);
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



// Device buffers and events of a batch go back to the handle (pool, spare events). Needs the index alive: once the
// index is closed the batch is detached (b->ix == nullptr) and holds nothing on the device any more.
void batch_release_device(msr_batch* b) {
    if (!b || !b->ix || !b->ix->dev) return;
    DeviceIndex* d = b->ix->dev;
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);  // nothing in flight may still use the buffers
    // d_qptr .. d_n live inside pooled blocks; the rest are plain allocations
    void** ptrs[] = {(void**)&b->d_gather, (void**)&b->d_stamps, (void**)&b->d_S, (void**)&b->d_R, (void**)&b->d_tpart};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (auto& blk : b->pooled) d->pool.release(blk.first, blk.second);
    b->pooled.clear();
    b->d_qptr = b->d_qterm = b->d_qw = b->d_qdense = b->d_ord = b->d_su32 = nullptr;
    b->d_sf32 = nullptr;
    b->d_n = nullptr;
    b->d_part = b->d_keys = nullptr;
    for (hipEvent_t e : b->events) {
        if (!e) continue;
        if (d->spare_events.size() < 64)
            d->spare_events.push_back(e);
        else
            (void)hipEventDestroy(e);
    }
    b->events.clear();
    b->ev0 = b->ev1 = b->ev2 = nullptr;
    b->calls = 0;
    b->timed = false;
    b->last_k = 0;
}

void batch_free(msr_batch* b) {
    if (!b) return;
    if (b->ix) {
        batch_release_device(b);
        auto& lb = b->ix->live_batches;
        lb.erase(std::remove(lb.begin(), lb.end(), b), lb.end());
    }
    delete b;
}

static int batch_alive(const msr_batch* b, const char* who) {
    if (!b) {
        set_error("%s: null batch", who);
        return MSR_E_INVAL;
    }
    if (!b->ix || !b->ix->dev) {
        set_error("%s: the batch's index has been closed", who);
        return MSR_E_INVAL;
    }
    return MSR_OK;
}

extern "C" {

}  // extern "C"

// small = msr_search_csr's own batch: no HIP events, and — when the result block fits — results written straight to
// mapped host memory
static int batch_create_impl(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq,
                             int kmax, uint32_t flags, int shard, int n_shards, msr_batch** out, bool small = false) {
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
    if (ix->term_nshards > 0 && (n_shards != ix->term_nshards || shard != ix->term_shard)) {
        set_error("this handle holds term shard %d of %d only: create its batches with msr_batch_create_termshard(..., %d, %d)",
                  ix->term_shard, ix->term_nshards, ix->term_shard, ix->term_nshards);
        return MSR_E_INVAL;
    }
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
    // per query {first term, end term, bit mask of its dense-head pairs, 0}: ONE scalar load in score_tiles
    std::vector<uint32_t> qptr((size_t)nq * 4 + 4, 0), qterm, qw;
    const int64_t total_in = q_ptr[nq];
    if (total_in < 0 || (total_in && (!q_term || !q_w))) {
        set_error("msr_batch_create: bad CSR arrays");
        return MSR_E_INVAL;
    }
    const uint32_t n_pairs = d->n_pairs;  // resident pairs (a term shard keeps only the pairs with an owned term)
    std::vector<uint32_t> qdense((size_t)nq * n_pairs, 0u);  // packed 16-bit query weights of the dense-head terms
    // Big batches (the 155 070-query Flickr batch spends more time here than on the GPU) are normalised by several
    // threads, each over a contiguous query range into its own term/weight arrays; the ranges are then laid end to
    // end, so the result is the serial one.  The error reported is the one of the LOWEST failing query, as in a
    // serial walk.
    struct Part {
        int q0 = 0, q1 = 0;
        std::vector<uint32_t> term, w;
        uint64_t n_kept = 0, sum_df = 0, sum_df_dense = 0, max_sparse = 0;
        int rc = MSR_OK;
        char err[192] = {0};
    };
    // (by entries, not by queries: 5 000 hybrid queries of 120 terms are as much work as 60 000 captions)
    const int n_parts = (nq >= 64 && total_in >= (1 << 16)) ? std::min(std::min(clamp_threads(0), 16), nq / 16) : 1;
    std::vector<Part> parts((size_t)n_parts);
    auto normalise = [&](int pi) {
        Part& pt = parts[(size_t)pi];
        pt.q0 = (int)((int64_t)nq * pi / n_parts);
        pt.q1 = (int)((int64_t)nq * (pi + 1) / n_parts);
        auto bad = [&](int rc, const char* fmt, auto... a) {
            snprintf(pt.err, sizeof(pt.err), fmt, a...);
            pt.rc = rc;
        };
        std::vector<uint32_t> dsum(h->n_dense);
        if (pt.q1 > pt.q0 && q_ptr[pt.q0] >= 0 && q_ptr[pt.q1] >= q_ptr[pt.q0] && q_ptr[pt.q1] <= total_in) {
            pt.term.reserve((size_t)(q_ptr[pt.q1] - q_ptr[pt.q0]));
            pt.w.reserve((size_t)(q_ptr[pt.q1] - q_ptr[pt.q0]));
        }
        for (int i = pt.q0; i < pt.q1; ++i) {
            if (q_ptr[i] < 0 || q_ptr[i + 1] < q_ptr[i] || q_ptr[i + 1] > total_in)
                return bad(MSR_E_INVAL, "q_ptr is not monotone at query %d", i);
            uint64_t bound = 0;
            const size_t first = pt.term.size();
            std::fill(dsum.begin(), dsum.end(), 0u);
            for (int64_t e = q_ptr[i]; e < q_ptr[i + 1]; ++e) {
                const int32_t t = q_term[e];
                const int32_t w = q_w[e];
                if (t < 0 || w <= 0) continue;
                if ((uint32_t)t >= h->n_terms)
                    return bad(MSR_E_RANGE, "query %d: term id %d is outside the dictionary (%u terms)", i, t, h->n_terms);
                if (w > 0xFFFFFF)  // the kernel multiplies with v_mul_u32_u24
                    return bad(MSR_E_RANGE, "query %d: weight %d of term %d exceeds the supported maximum 16777215", i, w, t);
                if ((flags & MSR_F_DROP_DF_EQ_N) && ix->host.df[t] == h->n_docs) continue;
                if (ix->host.df[t] == 0) continue;
                bound += (uint64_t)w * ix->host.maxw[t];  // the bound covers the WHOLE query, whoever owns the term
                if ((uint32_t)t < term_lo || (uint32_t)t >= term_hi) continue;
                pt.sum_df += d->df_shard[t];
                ++pt.n_kept;
                const int ds = ix->host.dense_slot[t];
                if (ds >= 0) pt.sum_df_dense += d->df_shard[t];
                if (ds >= 0) {  // dense-head term: repeated entries add up; v_dot2_u32_u16 takes 16-bit weights
                    if ((uint64_t)dsum[ds] + (uint64_t)w > 0xFFFFull)
                        return bad(MSR_E_RANGE, "query %d: weight of dense-head term %d exceeds the supported maximum 65535", i, t);
                    dsum[ds] += (uint32_t)w;
                } else {
                    pt.term.push_back((uint32_t)t - d->term_base);  // column of the (possibly term-sharded) segment table
                    pt.w.push_back((uint32_t)w);
                }
            }
            uint32_t pmask = 0;
            for (uint32_t s2 = 0; s2 < h->n_dense; ++s2) {
                if (!dsum[s2]) continue;
                const int32_t lp = d->pair_local[s2 >> 1];  // >= 0: an owned term's pair is resident by construction
                qdense[(size_t)i * n_pairs + (uint32_t)lp] |= dsum[s2] << (16 * (s2 & 1));
                pmask |= 1u << lp;
            }
            if (bound > 0xFFFFFFFFull)
                return bad(MSR_E_OVERFLOW, "query %d: worst-case score %llu exceeds the exact u32 range", i, (unsigned long long)bound);
            pt.max_sparse = std::max<uint64_t>(pt.max_sparse, pt.term.size() - first);
            qptr[(size_t)i * 4 + 1] = (uint32_t)(pt.term.size() - first);  // the COUNT for now; made an end offset below
            qptr[(size_t)i * 4 + 2] = pmask;
        }
    };
    parallel_run(n_parts, normalise);
    uint64_t n_kept = 0;
    uint64_t sum_df = 0, sum_df_dense = 0;
    uint64_t max_sparse = 0;  // most sparse (non dense-head) terms in one query
    uint64_t n_sparse = 0;
    for (const Part& pt : parts) {  // parts are in query order: the first failing part holds the lowest failing query
        if (pt.rc != MSR_OK) {
            set_error("%s", pt.err);
            return pt.rc;
        }
        n_kept += pt.n_kept;
        sum_df += pt.sum_df;
        sum_df_dense += pt.sum_df_dense;
        max_sparse = std::max(max_sparse, pt.max_sparse);
        n_sparse += pt.term.size();
    }
    if (n_sparse > 0xFFFFFFF0ull) {
        set_error("query batch too large");
        return MSR_E_RANGE;
    }
    qterm.resize((size_t)n_sparse);
    qw.resize((size_t)n_sparse);
    std::vector<uint64_t> part_base((size_t)n_parts + 1, 0);
    for (int pi = 0; pi < n_parts; ++pi) part_base[(size_t)pi + 1] = part_base[(size_t)pi] + parts[(size_t)pi].term.size();
    parallel_run(n_parts, [&](int pi) {
        const Part& pt = parts[(size_t)pi];
        uint32_t at = (uint32_t)part_base[(size_t)pi];
        if (!pt.term.empty()) {
            memcpy(qterm.data() + at, pt.term.data(), pt.term.size() * 4);
            memcpy(qw.data() + at, pt.w.data(), pt.w.size() * 4);
        }
        for (int i = pt.q0; i < pt.q1; ++i) {
            qptr[(size_t)i * 4] = at;
            at += qptr[(size_t)i * 4 + 1];
            qptr[(size_t)i * 4 + 1] = at;
        }
    });
    qptr[(size_t)nq * 4] = (uint32_t)n_sparse;  // the sentinel row's first term

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
    b->sum_df_dense = sum_df_dense;
    b->max_sparse_terms = (uint32_t)std::min<uint64_t>(max_sparse, 0xFFFFFFFFull);
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
    // one input block (q_ptr | q_term | q_w | q_dense) and one output block (ord | u32 scores | f32 scores | n), both
    // from the handle's caching pool; small batches travel through pinned staging in ONE copy each way
    auto al = [](size_t n) { return (n + 63) / 64 * 64; };  // sub-buffers start on 256-byte boundaries (in u32 units)
    const size_t o_ptr = 0, o_term = o_ptr + al(qptr.size()), o_w = o_term + al(qterm.size()),
                 o_dense = o_w + al(qw.size()), in_words = o_dense + al(qdense.size());
    const size_t o_ord = 0, o_su = o_ord + al(nqk), o_sf = o_su + al(nqk), o_n = o_sf + al(nqk), out_words = o_n + al((size_t)nq);
    auto take = [&](size_t bytes) -> void* {
        void* p = d->pool.alloc(bytes);
        if (p) b->pooled.emplace_back(p, bytes);
        return p;
    };
    // small call: queries and results live in the handle's mapped host block (first half in, second half out)
    b->untimed = small;
    b->zero_copy = small && d->d_zc && in_words * 4 <= kZeroCopyBytes && out_words * 4 <= kZeroCopyBytes &&
                   ntiles * (size_t)std::max(nq, 1) <= kSmallMaxPairs && n_shards == 0;
    uint32_t* d_in = b->zero_copy ? (uint32_t*)d->d_zc : (uint32_t*)take(std::max<size_t>(in_words, 1) * 4);
    uint32_t* d_out = b->zero_copy ? (uint32_t*)d->d_zc + kZeroCopyBytes / 4 : (uint32_t*)take(std::max<size_t>(out_words, 1) * 4);
    b->d_part = (uint64_t*)take(ntiles * nqk * 8);
    b->d_keys = (uint64_t*)take(nqk * 8);
    if (!d_in || !d_out || !b->d_part || !b->d_keys) {
        set_error("device allocation for the query batch failed (%d queries, kmax %d, %zu tiles)", nq, kmax, ntiles);
        return fail(MSR_E_NOMEM);
    }
    b->d_qptr = d_in + o_ptr;
    b->d_qterm = d_in + o_term;
    b->d_qw = d_in + o_w;
    b->d_qdense = d_in + o_dense;
    b->d_ord = d_out + o_ord;
    b->d_su32 = d_out + o_su;
    b->d_sf32 = reinterpret_cast<float*>(d_out + o_sf);
    b->d_n = reinterpret_cast<int32_t*>(d_out + o_n);
    if (b->zero_copy) {
        // the kernels read the few hundred bytes of a small batch over PCIe themselves: no staging copy, no upload on
        // the stream (an upload costs the stream ~3.5 us, three dependent reads from host memory ~2.6: scripts/latency_lab.hip)
        uint32_t* hs = (uint32_t*)d->h_zc;
        memcpy(hs + o_ptr, qptr.data(), qptr.size() * 4);
        if (!qterm.empty()) {
            memcpy(hs + o_term, qterm.data(), qterm.size() * 4);
            memcpy(hs + o_w, qw.data(), qw.size() * 4);
        }
        if (!qdense.empty()) memcpy(hs + o_dense, qdense.data(), qdense.size() * 4);
    } else {
        const size_t in_bytes = in_words * 4;
        bool ok = true;
        if (in_bytes <= (8u << 20)) {
            if (d->stage_pending) {  // an earlier batch's upload may still be reading the staging buffer
                (void)hipEventSynchronize(d->stage_ev);
                d->stage_pending = false;
            }
            if (d->h_stage_bytes < in_bytes) {
                if (d->h_stage) (void)hipHostFree(d->h_stage);
                d->h_stage = nullptr;
                d->h_stage_bytes = 0;
                const size_t want = std::max<size_t>(in_bytes, 1u << 20);
                if (hipHostMalloc(&d->h_stage, want, hipHostMallocDefault) == hipSuccess) d->h_stage_bytes = want;
            }
        }
        if (d->h_stage && d->h_stage_bytes >= in_bytes && in_bytes <= (8u << 20)) {
            uint32_t* hs = (uint32_t*)d->h_stage;
            memcpy(hs + o_ptr, qptr.data(), qptr.size() * 4);
            if (!qterm.empty()) {
                memcpy(hs + o_term, qterm.data(), qterm.size() * 4);
                memcpy(hs + o_w, qw.data(), qw.size() * 4);
            }
            if (!qdense.empty()) memcpy(hs + o_dense, qdense.data(), qdense.size() * 4);
            // asynchronous: the kernels follow on the same stream; the event guards the staging buffer's next use
            if (!d->stage_ev && hipEventCreateWithFlags(&d->stage_ev, hipEventDisableTiming) != hipSuccess) d->stage_ev = nullptr;
            ok = hipMemcpyAsync(d_in, hs, in_bytes, hipMemcpyHostToDevice, d->stream) == hipSuccess;
            if (ok && d->stage_ev && hipEventRecord(d->stage_ev, d->stream) == hipSuccess)
                d->stage_pending = true;
            else
                ok = ok && hipStreamSynchronize(d->stream) == hipSuccess;
        } else {
            ok = hipMemcpy(b->d_qptr, qptr.data(), qptr.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                 (qterm.empty() || (hipMemcpy(b->d_qterm, qterm.data(), qterm.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                                    hipMemcpy(b->d_qw, qw.data(), qw.size() * 4, hipMemcpyHostToDevice) == hipSuccess)) &&
                 (qdense.empty() || hipMemcpy(b->d_qdense, qdense.data(), qdense.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
        }
        if (!ok) {
            set_error("upload of the query batch failed");
            return fail(MSR_E_HIP);
        }
    }
    ix->live_batches.push_back(b);
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

int batch_search_local(msr_batch* b, int k, bool final_arrays) {
    msr_index* ix = b->ix;
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    HIP_TRY(hipSetDevice(d->device));
    // every call gets its own event triple so that a whole timed region can be summed afterwards
    if (b->calls >= 4096) b->calls = 0;  // bounded: callers that never reset keep only the recent calls
    while (!b->untimed && b->events.size() < (b->calls + 1) * 3) {
        hipEvent_t e = nullptr;
        if (!d->spare_events.empty()) {
            e = d->spare_events.back();
            d->spare_events.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&e));
        }
        b->events.push_back(e);
    }
    if (!b->untimed) {
        b->ev0 = b->events[b->calls * 3 + 0];
        b->ev1 = b->events[b->calls * 3 + 1];
        b->ev2 = b->events[b->calls * 3 + 2];
        b->calls++;
        HIP_TRY(hipEventRecord(b->ev0, d->stream));
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
    sa.part = b->d_part;
    sa.n_docs = h->n_docs;
    sa.vec_base = d->vec_base;
    sa.n_terms = d->seg_terms;
    sa.tile0 = ix->shard_tile0;
    sa.nq = (uint32_t)b->nq;
    sa.q0 = 0;
    sa.qn = (uint32_t)b->nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.theta = nullptr;
    sa.tl0 = 0;
    sa.unsorted = 0;
    sa.tpr = 1;
    sa.dump_add = 0;
    {
        static const bool no_light = getenv("MSR_NO_LIGHT") != nullptr;  // diagnostic: the general accumulation only
        sa.light = (!no_light && b->max_sparse_terms <= 64) ? 1u : 0u;
    }
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
    // Staged search: the first `t1` tiles are scored with the full per-tile selection; their merged k-th best key
    // per query is a lower bound on the final k-th key, so the remaining tiles only have to report keys at or above
    // it (theta_select: one pass, usually no survivor). Exact: the final merge sees every key that can be in the
    // top-k. (A launch holds at most 65535 tiles in y; x = queries.)
    int rc = MSR_OK;
    const uint32_t nt = ix->shard_ntiles;
    static const int stage_env = getenv("MSR_STAGE1_TILES") ? atoi(getenv("MSR_STAGE1_TILES")) : -1;
    uint32_t t1 = nt;
    // (small launches — fewer (tile, query) pairs than one round of resident workgroups — stay in ONE launch: two
    // more kernels in the chain would cost the call more latency than the cheaper selection saves)
    if (nt >= 2 && !sa.dbg && stage_env != 0 && (stage_env > 0 || (uint64_t)nt * (uint64_t)b->nq >= kMinStagedPairs))
        t1 = stage_env > 0 ? std::min<uint32_t>(nt, (uint32_t)stage_env) : std::max<uint32_t>(1u, nt / kStage1Fraction);
    sa.theta = nullptr;
    sa.unsorted = (b->unsorted_ok && nt == 1) ? 1u : 0u;  // one tile: its list IS the query's list
    rc = launch_score(d->stream, h->tile_docs, 0, t1, sa);
    if (rc != MSR_OK) return rc;
    if (t1 < nt) {
        MergeArgs m1;
        m1.lists = b->d_part;
        m1.list_stride = (uint64_t)b->nq * k;
        m1.n_lists = t1;
        m1.nq = (uint32_t)b->nq;
        m1.k = (uint32_t)k;
        m1.out_keys = b->d_keys;
        m1.out_ord = nullptr;
        m1.out_score_u32 = nullptr;
        m1.out_score = nullptr;
        m1.out_n = nullptr;
        rc = launch_merge(d->stream, m1);
        if (rc != MSR_OK) return rc;
        sa.theta = b->d_keys;
        rc = launch_score(d->stream, h->tile_docs, t1, nt, sa);
        if (rc != MSR_OK) return rc;
    }
    if (!b->untimed) HIP_TRY(hipEventRecord(b->ev1, d->stream));
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
    if (!b->untimed) HIP_TRY(hipEventRecord(b->ev2, d->stream));
    b->last_k = k;
    b->timed = !b->untimed;
    return MSR_OK;
}

int msr_batch_search(msr_batch* b, int k) {
    if (int rc = batch_alive(b, "msr_batch_search")) return rc;
    if (b->ix->term_nshards > 0) {
        set_error("msr_batch_search: this handle holds one term shard (partial sums only); use msr_batch_search_termshard");
        return MSR_E_INVAL;
    }
    if (k < 1 || k > b->kmax) {
        set_error("k=%d outside [1, kmax=%d]", k, b->kmax);
        return MSR_E_RANGE;
    }
    return batch_search_local(b, k, true);
}

int msr_batch_sync(msr_batch* b) {
    if (int rc = batch_alive(b, "msr_batch_sync")) return rc;
    HIP_TRY(hipSetDevice(b->ix->dev->device));
    HIP_TRY(hipStreamSynchronize(b->ix->dev->stream));
    return MSR_OK;
}

int msr_batch_fetch(msr_batch* b, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    if (int rc0 = batch_alive(b, "msr_batch_fetch")) return rc0;
    if (b->last_k == 0) {
        set_error("msr_batch_fetch: no search has run on this batch");
        return MSR_E_INVAL;
    }
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    const size_t n = (size_t)b->nq * b->last_k;
    if (b->zero_copy) {  // the kernels wrote into mapped host memory: nothing to download
        const uint32_t* hs = (const uint32_t*)b->ix->dev->h_zc + kZeroCopyBytes / 4;
        if (n && out_doc_ord) memcpy(out_doc_ord, hs, n * 4);
        if (n && out_score_u32) memcpy(out_score_u32, hs + (b->d_su32 - b->d_ord), n * 4);
        if (n && out_score) memcpy(out_score, hs + ((const uint32_t*)b->d_sf32 - b->d_ord), n * 4);
        if (out_n && b->nq) memcpy(out_n, hs + ((const uint32_t*)b->d_n - b->d_ord), (size_t)b->nq * 4);
        return MSR_OK;
    }
    {
        // small results: the whole output block in one copy through the pinned staging buffer
        DeviceIndex* d = b->ix->dev;
        const size_t span = (size_t)((const uint32_t*)b->d_n - b->d_ord) + (size_t)b->nq;  // words from d_ord to the end of d_n
        if (n && d->h_stage && span * 4 <= d->h_stage_bytes) {
            HIP_TRY(hipMemcpyAsync(d->h_stage, b->d_ord, span * 4, hipMemcpyDeviceToHost, d->stream));
            HIP_TRY(hipStreamSynchronize(d->stream));
            d->stage_pending = false;  // (everything queued on the stream, uploads included, is done)
            const uint32_t* hs = (const uint32_t*)d->h_stage;
            if (out_doc_ord) memcpy(out_doc_ord, hs, n * 4);
            if (out_score_u32) memcpy(out_score_u32, hs + (b->d_su32 - b->d_ord), n * 4);
            if (out_score) memcpy(out_score, hs + ((const uint32_t*)b->d_sf32 - b->d_ord), n * 4);
            if (out_n) memcpy(out_n, hs + ((const uint32_t*)b->d_n - b->d_ord), (size_t)b->nq * 4);
            return MSR_OK;
        }
    }
    if (n) {
        if (out_doc_ord) HIP_TRY(hipMemcpy(out_doc_ord, b->d_ord, n * 4, hipMemcpyDeviceToHost));
        if (out_score) HIP_TRY(hipMemcpy(out_score, b->d_sf32, n * 4, hipMemcpyDeviceToHost));
        if (out_score_u32) HIP_TRY(hipMemcpy(out_score_u32, b->d_su32, n * 4, hipMemcpyDeviceToHost));
    }
    if (out_n && b->nq) HIP_TRY(hipMemcpy(out_n, b->d_n, (size_t)b->nq * 4, hipMemcpyDeviceToHost));
    return MSR_OK;
}

int msr_batch_kernel_ms(msr_batch* b, float* score_ms, float* merge_ms) {
    if (int rc0 = batch_alive(b, "msr_batch_kernel_ms")) return rc0;
    if (!b->timed) {
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
    int rc = msr_batch_sync(b);
    if (rc != MSR_OK) return rc;
    b->calls = 0;
    return MSR_OK;
}

int msr_batch_timing_sum(msr_batch* b, int* n_calls, float* score_ms, float* merge_ms) {
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
    if (!b->ix || !b->d_stamps) return MSR_OK;
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

int msr_batch_work(const msr_batch* b, uint64_t out[6]) {
    if (!b || !out || !b->ix) {
        set_error("msr_batch_work: bad argument");
        return MSR_E_INVAL;
    }
    const msr_index* ix = b->ix;
    const IndexHeader* h = ix->host.h;
    // threads of the scoring instance (4 accumulators per thread and round)
    const uint64_t nt = h->tile_docs == 4096 ? 256 : (h->tile_docs == 32768 ? 1024 : 512);
    uint64_t acc_words = 0;  // accumulators one query initialises (and selects over) across the shard's tiles
    for (uint32_t t = ix->shard_tile0; t < ix->shard_tile0 + ix->shard_ntiles; ++t) {
        const uint64_t docs = std::min<uint64_t>(h->tile_docs, h->n_docs - (uint64_t)t * h->tile_docs);
        acc_words += (docs + 4 * nt - 1) / (4 * nt) * (4 * nt);
    }
    out[0] = b->sum_df - b->sum_df_dense;                 // postings walked through the inverted lists (one ds_add_u32 each)
    out[1] = b->sum_df_dense;                             // postings scored out of the dense head (half a v_dot2_u32_u16 each)
    out[2] = (uint64_t)ix->shard_ntiles * (uint64_t)b->nq;  // (tile, query) workgroups
    out[3] = acc_words * 4 * (uint64_t)b->nq;             // LDS bytes written to initialise the accumulator tiles
    out[4] = 2 * acc_words * 4 * (uint64_t)b->nq;         // LDS bytes read back by the selection (maxima pass + candidate pass)
    out[5] = b->nnz;                                      // kept query entries
    return MSR_OK;
}

void msr_batch_destroy(msr_batch* b) { batch_free(b); }

// host-side laps of the calling thread's last msr_search_csr (msr_search_laps): where a small call's time goes
static thread_local double g_laps[8] = {0, 0, 0, 0, 0, 0, 0, 0};

int msr_search_csr(msr_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int k,
                   uint32_t flags, uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = clk::now();
    msr_batch* b = nullptr;
    static const bool timed_calls = getenv("MSR_TIME_SEARCH_CALLS") != nullptr;  // diagnostic: HIP events around the kernels
    int rc = batch_create_impl(ix, q_ptr, q_term, q_w, nq, k, flags, 0, 0, &b, !timed_calls);
    if (rc != MSR_OK) return rc;
    const auto t1 = clk::now();
    rc = msr_batch_search(b, k);
    const auto t2 = clk::now();
    if (rc == MSR_OK) rc = msr_batch_sync(b);
    const auto t3 = clk::now();
    if (rc == MSR_OK) rc = msr_batch_fetch(b, out_doc_ord, out_score, out_score_u32, out_n);
    const auto t4 = clk::now();
    float score_ms = 0, merge_ms = 0;
    if (rc == MSR_OK && b->timed) (void)msr_batch_kernel_ms(b, &score_ms, &merge_ms);
    msr_batch_destroy(b);
    const auto t5 = clk::now();
    g_laps[0] = us(t0, t1);  // query normalisation + pooled buffers + staging + upload enqueued
    g_laps[1] = us(t1, t2);  // kernels enqueued
    g_laps[2] = us(t2, t3);  // wait for the stream: upload + kernels (+ launch latency)
    g_laps[3] = us(t3, t4);  // download through pinned staging + copy out
    g_laps[4] = us(t4, t5);  // events read, buffers back to the pool
    g_laps[5] = us(t0, t5);  // the whole call
    g_laps[6] = 1e3 * score_ms;  // HIP-event spans on the stream: scoring kernel(s) ...
    g_laps[7] = 1e3 * merge_ms;  // ... and the merge
    return rc;
}

int msr_search_laps(double out_us[8]) {
    if (!out_us) {
        set_error("msr_search_laps: null output");
        return MSR_E_INVAL;
    }
    memcpy(out_us, g_laps, sizeof(g_laps));
    return MSR_OK;
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
    if (int rc = batch_alive(b, who)) return rc;
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
    if (b->ix->term_nshards > 0) {
        set_error("msr_batch_search_sharded merges DOC-range shards; this handle is a term shard (msr_batch_search_termshard)");
        return MSR_E_INVAL;
    }
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
    sa.q_meta = reinterpret_cast<const uint4*>(b->d_qptr);
    sa.q_term = b->d_qterm;
    sa.q_w = b->d_qw;
    sa.dense = d->d_dense;
    sa.q_dense = b->d_qdense;
    sa.n_pairs = d->n_pairs;
    sa.part = b->d_part;
    sa.n_docs = ix->host.h->n_docs;
    sa.vec_base = d->vec_base;
    sa.n_terms = d->seg_terms;
    sa.tile0 = ix->shard_tile0;
    sa.nq = (uint32_t)b->nq;
    sa.q0 = 0;
    sa.qn = (uint32_t)b->nq;
    sa.k = (uint32_t)k;
    sa.dump = nullptr;
    sa.theta = nullptr;
    sa.tl0 = 0;
    sa.unsorted = 0;
    sa.tpr = 1;
    sa.dump_add = 0;
    sa.dbg = 0;
    sa.stamps = nullptr;
    sa.light = 0;
}

int msr_batch_search_termshard(msr_batch* b, int k) {
    int rc = check_sharded_call(b, k, "msr_batch_search_termshard");
    if (rc != MSR_OK) return rc;
    msr_index* ix = b->ix;
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    if (ix->shard_ntiles != h->n_tiles) {
        set_error("term-sharded search needs a handle that holds every doc tile (msr_index_open or msr_index_open_termshard)");
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
        rc = launch_score(d->stream, h->tile_docs, 0, h->n_tiles, sa, true);
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
        se.unsorted = 0;
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
// its doc range and the n_shards lists are merged. hs[g] scores term shard g: either one handle that holds every term
// (the same pointer n_shards times) or n_shards handles opened by msr_index_open_termshard on the same device.
static int termshard_emulated_impl(msr_index* const* hs, int n_shards, const int64_t* q_ptr, const int32_t* q_term,
                                   const int32_t* q_w, int nq, int k, uint32_t flags, uint32_t* out_doc_ord,
                                   float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    msr_index* ix = hs[0];
    DeviceIndex* d = ix->dev;
    const IndexHeader* h = ix->host.h;
    std::vector<msr_batch*> bs((size_t)n_shards, nullptr);
    uint32_t* d_S = nullptr;
    uint64_t *d_tpart = nullptr, *d_lists = nullptr;
    int rc = MSR_OK;
    std::vector<uint64_t*> rank_part((size_t)n_shards, nullptr);
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(d->stream);
        for (msr_batch* x : bs) batch_free(x);
        for (uint64_t* x : rank_part)
            if (x) (void)hipFree(x);
        if (d_S) (void)hipFree(d_S);
        if (d_tpart) (void)hipFree(d_tpart);
        if (d_lists) (void)hipFree(d_lists);
    };
    for (int g = 0; g < n_shards && rc == MSR_OK; ++g) {
        rc = batch_create_impl(hs[g], q_ptr, q_term, q_w, nq, k, flags, g, n_shards, &bs[g]);
        // the batch's upload runs on ITS handle's stream; everything below runs on shard 0's stream
        if (rc == MSR_OK && hs[g] != ix && hipStreamSynchronize(hs[g]->dev->stream) != hipSuccess) {
            set_error("hipStreamSynchronize failed");
            rc = MSR_E_HIP;
        }
    }
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
    for (int r = 0; r < n_shards && rc == MSR_OK; ++r)
        if (hipMalloc(&rank_part[r], std::max<size_t>((size_t)p.tpr * per, 1) * 8) != hipSuccess) {
            set_error("hipMalloc failed in msr_search_termshard_emulated");
            rc = MSR_E_NOMEM;
        }
    // diagnostic (MSR_DEBUG_TERMSHARD): HIP-event time of every shard's dump and every logical rank's selection — the
    // per-rank compute of the real protocol, which an 8-GPU run adds the reduce-scatter to (scripts/gpu_c4_termshard_probe.py)
    static const bool dbg_ts = getenv("MSR_DEBUG_TERMSHARD") != nullptr;
    std::vector<hipEvent_t> tev;
    auto mark = [&]() {
        if (!dbg_ts) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) == hipSuccess) (void)hipEventRecord(e, d->stream);
        tev.push_back(e);
    };
    int n_passes = 0;
    for (uint32_t q0 = 0; q0 < (uint32_t)nq && rc == MSR_OK; q0 += p.qt) {
        const uint32_t qn = std::min<uint32_t>(p.qt, (uint32_t)nq - q0);
        ++n_passes;
        if (hipMemsetAsync(d_S, 0, (size_t)qn * p.range_elems * p.G * 4, d->stream) != hipSuccess) {
            set_error("hipMemsetAsync failed");
            rc = MSR_E_HIP;
            break;
        }
        mark();
        for (int g = 0; g < n_shards && rc == MSR_OK; ++g) {
            ScoreArgs sa;
            fill_score_args(sa, bs[g], k);
            sa.q0 = q0;
            sa.qn = qn;
            sa.dump = d_S;
            sa.tpr = p.tpr;
            sa.dump_add = 1;
            rc = launch_score(d->stream, h->tile_docs, 0, h->n_tiles, sa, true);
            mark();
        }
        for (int r = 0; r < n_shards && rc == MSR_OK; ++r) {
            SelectArgs se;
            se.unsorted = 0;
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
            mark();
        }
    }
    if (dbg_ts && rc == MSR_OK && hipStreamSynchronize(d->stream) == hipSuccess) {
        // per pass: 1 + n_shards + n_shards marks
        std::vector<double> dump_ms((size_t)n_shards, 0.0), sel_ms((size_t)n_shards, 0.0);
        const size_t per = 1 + 2 * (size_t)n_shards;
        for (int ps = 0; ps < n_passes; ++ps)
            for (int i = 0; i < 2 * n_shards; ++i) {
                hipEvent_t a = tev[(size_t)ps * per + (size_t)i], b2 = tev[(size_t)ps * per + (size_t)i + 1];
                float ms = 0;
                if (a && b2) (void)hipEventElapsedTime(&ms, a, b2);
                (i < n_shards ? dump_ms[(size_t)i] : sel_ms[(size_t)(i - n_shards)]) += ms;
            }
        fprintf(stderr, "[msr] term shards G=%d, %d queries in %d passes of <= %u (send buffer %.2f GB per pass): dump ms per shard [",
                n_shards, nq, n_passes, p.qt, (double)p.qt * (double)p.range_elems * p.G * 4 / 1e9);
        for (int g = 0; g < n_shards; ++g) fprintf(stderr, "%s%.2f", g ? ", " : "", dump_ms[(size_t)g]);
        fprintf(stderr, "] (on this one GPU a dump ADDS to the shared buffer: read + write), select ms per doc range [");
        for (int r = 0; r < n_shards; ++r) fprintf(stderr, "%s%.2f", r ? ", " : "", sel_ms[(size_t)r]);
        fprintf(stderr, "]\n");
    }
    for (hipEvent_t e : tev)
        if (e) (void)hipEventDestroy(e);
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
    cleanup();
    return rc;
}

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
    if (ix->shard_ntiles != ix->host.h->n_tiles || ix->term_nshards > 0) {
        set_error("term-sharded search needs a handle that holds every doc tile and every term (or one handle per shard: "
                  "msr_search_termshard_emulated_handles)");
        return MSR_E_INVAL;
    }
    std::vector<msr_index*> hs((size_t)n_shards, ix);
    return termshard_emulated_impl(hs.data(), n_shards, q_ptr, q_term, q_w, nq, k, flags, out_doc_ord, out_score,
                                   out_score_u32, out_n);
}

int msr_search_termshard_emulated_handles(msr_index* const* shards, int n_shards, const int64_t* q_ptr,
                                          const int32_t* q_term, const int32_t* q_w, int nq, int k, uint32_t flags,
                                          uint32_t* out_doc_ord, float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    if (!shards || n_shards < 1 || n_shards > 64) {
        set_error("msr_search_termshard_emulated_handles: bad argument");
        return MSR_E_INVAL;
    }
    for (int g = 0; g < n_shards; ++g) {
        const msr_index* x = shards[g];
        if (!x || !x->dev) {
            set_error("shard handle %d is null or has no HIP device bound; there is no CPU scoring path", g);
            return x ? MSR_E_NODEVICE : MSR_E_INVAL;
        }
        if (x->term_nshards != n_shards || x->term_shard != g || x->dev->device != shards[0]->dev->device ||
            x->host.h->n_docs != shards[0]->host.h->n_docs || x->host.h->n_vecs != shards[0]->host.h->n_vecs) {
            set_error("shard handle %d must be term shard %d of %d of the same index on the same device", g, g, n_shards);
            return MSR_E_INVAL;
        }
    }
    return termshard_emulated_impl(shards, n_shards, q_ptr, q_term, q_w, nq, k, flags, out_doc_ord, out_score,
                                   out_score_u32, out_n);
}

// ------------------------------------------------------------------------------------------------ runtime facts
int msr_comm_info(const msr_index* ix, int* n_ranks, int* rank, int* device) {
    if (!ix || !ix->dev || !ix->dev->comm) {
        set_error("msr_comm_info: no communicator on this handle (call msr_comm_init first)");
        return MSR_E_COMM;
    }
    int c = 0, r = 0, dv = 0;
    ncclResult_t e = ncclCommCount(ix->dev->comm, &c);
    if (e == ncclSuccess) e = ncclCommUserRank(ix->dev->comm, &r);
    if (e == ncclSuccess) e = ncclCommCuDevice(ix->dev->comm, &dv);
    if (e != ncclSuccess) {
        set_error("ncclCommCount / ncclCommUserRank failed: %s", ncclGetErrorString(e));
        return MSR_E_COMM;
    }
    if (n_ranks) *n_ranks = c;
    if (rank) *rank = r;
    if (device) *device = dv;
    return MSR_OK;
}

int msr_runtime_info(char* buf, int cap) {
    if (!buf || cap < 1) {
        set_error("msr_runtime_info: bad argument");
        return MSR_E_INVAL;
    }
    int hip_rt = 0, rccl_v = 0;
    (void)hipRuntimeGetVersion(&hip_rt);
    (void)ncclGetVersion(&rccl_v);
    Dl_info rccl_so, hip_so;
    const char* rccl_path = dladdr(reinterpret_cast<void*>(&ncclGetVersion), &rccl_so) && rccl_so.dli_fname ? rccl_so.dli_fname : "?";
    const char* hip_path = dladdr(reinterpret_cast<void*>(&hipRuntimeGetVersion), &hip_so) && hip_so.dli_fname ? hip_so.dli_fname : "?";
    snprintf(buf, (size_t)cap, "hip_runtime=%d hip_lib=%s rccl=%d rccl_header=%d rccl_lib=%s", hip_rt, hip_path, rccl_v,
             NCCL_VERSION_CODE, rccl_path);
    return MSR_OK;
}

int msr_device_sync(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return MSR_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ pipe peaks
// The instruction rates the scorer's useful work is priced against (bench.py: roofline.useful), MEASURED on the device
// the benchmark runs on: each kernel issues nothing but the instruction in question (plus its loop), at the production
// kernel's shape (512 threads, 32 KiB of LDS per workgroup: four workgroups = 32 waves per CU).
namespace msr {

// ds_add_u32 without return, 64 lanes on 64 consecutive words (two per bank: the layout the indexer arranges postings
// for); sixteen adds per iteration at immediate offsets from one address register
__global__ __launch_bounds__(512, 8) void peak_ds_add(uint32_t* out, int iters) {
    __shared__ uint32_t acc[8192];
    const uint32_t tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 512) acc[i] = 0;
    __syncthreads();
    uint32_t* const mine = acc + tid;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) atomicAdd(mine + 512 * e, (uint32_t)i);
    }
    __syncthreads();
    if (acc[tid] == 0xFFFFFFFFu) out[blockIdx.x] = acc[tid];
}

// v_dot2_u32_u16: eight independent accumulator chains per lane
__global__ __launch_bounds__(512, 8) void peak_dot2(uint32_t* out, int iters, uint32_t seed) {
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    uint32_t a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = threadIdx.x + e;
    const us2 q = __builtin_bit_cast(us2, seed);
    uint32_t x = seed ^ threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const us2 w = __builtin_bit_cast(us2, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = __builtin_amdgcn_udot2(w, q, a[e], false);
        x += 0x00010001u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) s ^= a[e];
    if (s == 0xDEADBEEFu) out[blockIdx.x] = s;
}

// the accumulator traffic of one (tile, query): one ds_write_b128 (init) and two ds_read_b128 (selection passes) per vec,
// thread `tid` on vec r * 512 + tid like the kernels (conflict-free); inline asm so that exactly these are issued
__global__ __launch_bounds__(512, 8) void peak_lds_rw(uint32_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint4 acc4[2048];
    const uint32_t tid = threadIdx.x;
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(acc4 + tid);
    u32x4 v = {tid, 1u, 2u, 3u};
    u32x4 s = {0u, 0u, 0u, 0u};
    for (int i = 0; i < iters; ++i) {
        u32x4 x[8];
        asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:8192\n\tds_write_b128 %0, %1 offset:16384\n\t"
                     "ds_write_b128 %0, %1 offset:24576" ::"v"(addr), "v"(v) : "memory");
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:8192\n\tds_read_b128 %2, %8 offset:16384\n\t"
                     "ds_read_b128 %3, %8 offset:24576\n\tds_read_b128 %4, %8\n\tds_read_b128 %5, %8 offset:8192\n\t"
                     "ds_read_b128 %6, %8 offset:16384\n\tds_read_b128 %7, %8 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                     : "v"(addr)
                     : "memory");
        s ^= x[0] ^ x[1] ^ x[2] ^ x[3] ^ x[4] ^ x[5] ^ x[6] ^ x[7];
        v.x += s.x;
    }
    if ((s.x ^ s.y ^ s.z ^ s.w) == 0xDEADBEEFu) out[blockIdx.x] = s.x;
}

}  // namespace msr

extern "C" {

int msr_device_peak_rates(int device, double out[4]) {
    if (!out) {
        set_error("msr_device_peak_rates: null output");
        return MSR_E_INVAL;
    }
    int n_dev = 0;
    if (device < 0 || hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) {
        set_error("no usable HIP device %d", device);
        return MSR_E_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const uint32_t n_wg = (uint32_t)std::max(prop.multiProcessorCount, 1) * 4u * 4u;  // four rounds of four workgroups per CU
    uint32_t* d_out = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = MSR_OK;
    bool ok = hipMalloc(&d_out, (size_t)n_wg * 4) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    auto timed = [&](auto launch) -> double {  // best of five, seconds
        double best = 1e30;
        for (int rep = 0; rep < 6 && ok; ++rep) {
            ok = hipEventRecord(e0, st) == hipSuccess;
            launch();
            float ms = 0;
            ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, st) == hipSuccess &&
                 hipStreamSynchronize(st) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
            if (rep > 0 && ms > 0) best = std::min(best, (double)ms * 1e-3);
        }
        return best;
    };
    const double lanes = (double)n_wg * 512.0;
    const int it_add = 2000, it_dot = 20000, it_lds = 4000;
    double t;
    t = ok ? timed([&] { hipLaunchKernelGGL(peak_ds_add, dim3(n_wg), dim3(512), 0, st, d_out, it_add); }) : 0;
    out[0] = ok ? lanes * it_add * 16.0 / t : 0;   // ds_add_u32 lane operations per second = inverted-list postings / s
    t = ok ? timed([&] { hipLaunchKernelGGL(peak_dot2, dim3(n_wg), dim3(512), 0, st, d_out, it_dot, 0x00030002u); }) : 0;
    out[1] = ok ? lanes * it_dot * 8.0 / t : 0;    // v_dot2_u32_u16 lane operations per second (two dense-head postings each)
    t = ok ? timed([&] { hipLaunchKernelGGL(peak_lds_rw, dim3(n_wg), dim3(512), 0, st, d_out, it_lds); }) : 0;
    out[2] = ok ? lanes * it_lds * 12.0 * 16.0 / t : 0;  // LDS bytes per second at the 1 write : 2 reads mix of an accumulator tile
    out[3] = (double)prop.multiProcessorCount;
    if (!ok) {
        set_error("peak-rate measurement failed: %s", hipGetErrorString(hipGetLastError()));
        rc = MSR_E_HIP;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int msr_device_copy_gbs(int device, uint64_t bytes, int reps, double* gbs) {
    if (!gbs || bytes == 0 || reps < 1) {
        set_error("msr_device_copy_gbs: bad argument");
        return MSR_E_INVAL;
    }
    HIP_TRY(hipSetDevice(device));
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    int rc = MSR_OK;
    float ms = 0;
    bool ok = hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes) == hipSuccess &&
              hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&e0) == hipSuccess &&
              hipEventCreate(&e1) == hipSuccess && hipMemsetAsync(a, 0, bytes, st) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) ok = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipEventRecord(e0, st) == hipSuccess;
    for (int i = 0; i < reps && ok; ++i) ok = hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (!ok || ms <= 0) {
        set_error("device copy measurement failed: %s", hipGetErrorString(hipGetLastError()));
        rc = MSR_E_HIP;
    } else {
        *gbs = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;  // read + write
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}

}  // extern "C"
