// Host-side index code: jsonl reader, tile-major inverted index builder, index file mmap, dictionary lookup.
//
// Replaces the reference's offline step scripts/sparse_index.sh:12-18
//   python -m pyserini.index.lucene --collection JsonVectorCollection --impact --pretokenized
// over the corpus jsonl written by src/encode.py:351-359,426
//   {"id": "<str>", "content": "", "vector": {"<token>": <int>, ...}}
//
// Semantics kept (declared contract §8c):
//   - the term frequency of a token in a doc is its integer weight; entries with weight <= 0 are absent;
//   - a JSON key that repeats inside one "vector" object: the last value wins (JSON object semantics);
//   - a key containing whitespace is split on whitespace (whitespace analyzer) and every piece receives the
//     weight; pieces that coincide add up;
//   - no length normalisation, no stemming, no stop words.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <dirent.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include "msr_internal.h"

namespace msr {

// ----------------------------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

int clamp_threads(int threads) {
    if (threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        threads = hc ? (int)hc : 1;
    }
    return std::min(threads, 256);
}

void parallel_run(int n_threads, const std::function<void(int)>& fn) {
    if (n_threads <= 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> ts;
    ts.reserve(n_threads);
    for (int t = 0; t < n_threads; ++t) ts.emplace_back(fn, t);
    for (auto& t : ts) t.join();
}

// ----------------------------------------------------------------------------------------------- mmap
int HostIndex::open(const char* path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        set_error("cannot open index file '%s': %s", path, strerror(errno));
        return MSR_E_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(IndexHeader)) {
        set_error("index file '%s' is too small to hold a header", path);
        close();
        return MSR_E_FORMAT;
    }
    bytes = (size_t)st.st_size;
    void* p = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    if (p == MAP_FAILED) {
        set_error("mmap of '%s' failed: %s", path, strerror(errno));
        base = nullptr;
        close();
        return MSR_E_IO;
    }
    base = (const uint8_t*)p;
    h = (const IndexHeader*)base;
    if (memcmp(h->magic, "MSRIDX01", 8) != 0 || h->version != kIndexVersion) {
        set_error("'%s' is not an MSRIDX01 v%u index", path, kIndexVersion);
        close();
        return MSR_E_FORMAT;
    }
    if (h->file_size != bytes) {
        set_error("index '%s' is truncated: header says %llu bytes, file has %zu", path,
                  (unsigned long long)h->file_size, bytes);
        close();
        return MSR_E_FORMAT;
    }
    if (h->n_dense > kMaxDense || h->n_dense % 2 != 0) {
        set_error("index '%s': bad dense head size %u", path, h->n_dense);
        close();
        return MSR_E_FORMAT;
    }
    const uint64_t want[SEC_COUNT] = {
        (uint64_t)(h->n_terms + 1ull) * 8, h->size[SEC_TERM_STR], (uint64_t)h->n_terms * 4, (uint64_t)h->n_terms * 4,
        (uint64_t)h->n_terms * 4,          (h->n_docs + 1) * 8,   h->size[SEC_DOC_STR],     (uint64_t)h->n_tiles * (h->n_terms + 1ull) * 4,
        h->n_vecs * 16,                    (uint64_t)h->n_dense * 4,
        (uint64_t)h->n_tiles * (h->n_dense / 2) * h->tile_docs * 4};
    for (int s = 0; s < SEC_COUNT; ++s) {
        if (h->size[s] != want[s] || h->off[s] % 16 != 0 || h->off[s] + h->size[s] > bytes) {
            set_error("index '%s': section %d has inconsistent offset/size", path, s);
            close();
            return MSR_E_FORMAT;
        }
    }
    if (h->tile_docs == 0 || h->tile_docs > 65536 || h->n_tiles != (h->n_docs + h->tile_docs - 1) / h->tile_docs) {
        set_error("index '%s': bad tiling (%u docs/tile, %u tiles, %llu docs)", path, h->tile_docs, h->n_tiles,
                  (unsigned long long)h->n_docs);
        close();
        return MSR_E_FORMAT;
    }
    term_off = (const uint64_t*)(base + h->off[SEC_TERM_OFF]);
    term_str = (const char*)(base + h->off[SEC_TERM_STR]);
    term_sorted = (const uint32_t*)(base + h->off[SEC_TERM_SORTED]);
    df = (const uint32_t*)(base + h->off[SEC_DF]);
    maxw = (const uint32_t*)(base + h->off[SEC_MAXW]);
    doc_off = (const uint64_t*)(base + h->off[SEC_DOC_OFF]);
    doc_str = (const char*)(base + h->off[SEC_DOC_STR]);
    seg_ptr = (const uint32_t*)(base + h->off[SEC_SEG_PTR]);
    postings = (const uint32_t*)(base + h->off[SEC_POSTINGS]);
    dense_terms = (const uint32_t*)(base + h->off[SEC_DENSE_TERMS]);
    dense = (const uint32_t*)(base + h->off[SEC_DENSE]);
    dense_slot.assign(h->n_terms, (int8_t)-1);
    for (uint32_t s = 0; s < h->n_dense; ++s) {
        if (dense_terms[s] == 0xFFFFFFFFu) continue;  // padding slot of an odd-sized head
        if (dense_terms[s] >= h->n_terms) {
            set_error("index '%s': dense slot %u names term %u", path, s, dense_terms[s]);
            close();
            return MSR_E_FORMAT;
        }
        dense_slot[dense_terms[s]] = (int8_t)s;
    }
    return MSR_OK;
}

void HostIndex::close() {
    if (base) munmap((void*)base, bytes);
    if (fd >= 0) ::close(fd);
    base = nullptr;
    fd = -1;
    h = nullptr;
}

int32_t HostIndex::lookup(const char* tok) const {
    // binary search over term ids ordered by their strings (bytewise, like strcmp on unsigned chars)
    uint32_t lo = 0, hi = h->n_terms;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        uint32_t tid = term_sorted[mid];
        int c = strcmp(term_str + term_off[tid], tok);
        if (c == 0) return (int32_t)tid;
        if (c < 0)
            lo = mid + 1;
        else
            hi = mid;
    }
    return -1;
}

// ----------------------------------------------------------------------------------------------- builder
BuildOptions& build_options() {
    static BuildOptions o;
    return o;
}

namespace {

struct FileWriter {
    FILE* f = nullptr;
    uint64_t pos = 0;
    bool ok = true;
    bool open(const char* path) {
        f = fopen(path, "wb");
        return f != nullptr;
    }
    void write(const void* p, uint64_t n) {
        if (!ok || n == 0) return;
        if (fwrite(p, 1, n, f) != n) ok = false;
        pos += n;
    }
    void pad16() {
        static const char z[16] = {0};
        uint64_t r = pos % 16;
        if (r) write(z, 16 - r);
    }
    bool close() {
        if (f && fclose(f) != 0) ok = false;
        f = nullptr;
        return ok;
    }
};

}  // namespace

int build_from_csr(const char* out_path, uint64_t n_docs, uint32_t n_terms, const uint64_t* doc_ptr,
                   const uint32_t* term_id, const uint32_t* weight, const char* const* doc_ids,
                   const char* const* term_strs, int threads, uint32_t tile_docs) {
    if (!out_path || !doc_ptr || (doc_ptr[n_docs] && (!term_id || !weight))) {
        set_error("build_from_csr: null argument");
        return MSR_E_INVAL;
    }
    if (tile_docs == 0) tile_docs = kDefaultTileDocs;
    if (tile_docs > 65536 || tile_docs % 1024 != 0) {
        set_error("tile_docs must be a multiple of 1024 and at most 65536 (got %u)", tile_docs);
        return MSR_E_RANGE;
    }
    if (n_docs >= (1ull << 32)) {
        set_error("at most 2^32-1 docs are supported (got %llu)", (unsigned long long)n_docs);
        return MSR_E_RANGE;
    }
    threads = clamp_threads(threads);
    const uint64_t nnz_in = doc_ptr[n_docs];
    for (uint64_t d = 0; d < n_docs; ++d)
        if (doc_ptr[d + 1] < doc_ptr[d]) {
            set_error("doc_ptr is not monotone at row %llu", (unsigned long long)d);
            return MSR_E_INVAL;
        }

    // ---- validate entries
    {
        std::atomic<int> bad{0};
        std::atomic<uint64_t> bad_at{0};
        parallel_run(threads, [&](int t) {
            uint64_t a = nnz_in * t / threads, b = nnz_in * (t + 1) / threads;
            for (uint64_t i = a; i < b; ++i) {
                if (term_id[i] >= n_terms) {
                    bad = 1;
                    bad_at = i;
                    return;
                }
                if (weight[i] > kMaxWeight) {
                    bad = 2;
                    bad_at = i;
                    return;
                }
            }
        });
        if (bad == 1) {
            set_error("term id %u at entry %llu is outside the dictionary (%u terms)", term_id[bad_at],
                      (unsigned long long)bad_at.load(), n_terms);
            return MSR_E_RANGE;
        }
        if (bad == 2) {
            set_error("weight %u at entry %llu exceeds the supported maximum %u", weight[bad_at],
                      (unsigned long long)bad_at.load(), kMaxWeight);
            return MSR_E_RANGE;
        }
    }

    // ---- doc ordinals: rank of the id string, bytewise ascending; equal ids keep input order (stable)
    std::vector<std::string> gen_ids;
    std::vector<const char*> idp(n_docs);
    if (doc_ids) {
        for (uint64_t d = 0; d < n_docs; ++d) {
            if (!doc_ids[d]) {
                set_error("doc_ids[%llu] is null", (unsigned long long)d);
                return MSR_E_INVAL;
            }
            idp[d] = doc_ids[d];
        }
    } else {
        gen_ids.resize(n_docs);
        for (uint64_t d = 0; d < n_docs; ++d) {
            gen_ids[d] = std::to_string(d);
            idp[d] = gen_ids[d].c_str();
        }
    }
    std::vector<uint32_t> row_of_ord(n_docs);
    std::iota(row_of_ord.begin(), row_of_ord.end(), 0u);
    // (build option "tie_order" = 1 keeps the INPUT order instead: a score tie then goes to the doc that was indexed
    // first — what Lucene's internal doc numbers give under a single indexing thread — so that an integrator can check
    // either tie rule against a real pyserini run; contract T1 is unpinned)
    if (!build_options().tie_input_order)
        std::stable_sort(row_of_ord.begin(), row_of_ord.end(),
                         [&](uint32_t a, uint32_t b) { return strcmp(idp[a], idp[b]) < 0; });

    const uint32_t n_tiles = (uint32_t)((n_docs + tile_docs - 1) / tile_docs);
    const uint64_t stride = (uint64_t)n_terms + 1;

    // ---- per doc: sort entries by term, merge duplicates by adding (tf accumulates), drop weight 0.
    // cnt[tile][term] = postings of the segment; done tile-parallel.
    std::vector<uint32_t> seg_ptr((uint64_t)n_tiles * stride + 1, 0);  // first holds counts, then offsets
    std::vector<uint32_t> df(n_terms, 0), maxw(n_terms, 0);
    std::atomic<int> err{0};
    std::atomic<uint32_t> next_tile{0};
    std::vector<std::vector<uint32_t>> df_local(threads), mw_local(threads);
    std::atomic<uint64_t> n_postings{0};
    std::atomic<uint32_t> gmaxw{0};

    auto gather_doc = [&](uint32_t row, std::vector<std::pair<uint32_t, uint32_t>>& buf) -> bool {
        buf.clear();
        for (uint64_t i = doc_ptr[row]; i < doc_ptr[row + 1]; ++i)
            if (weight[i] > 0) buf.emplace_back(term_id[i], weight[i]);
        std::sort(buf.begin(), buf.end());
        size_t o = 0;
        for (size_t i = 0; i < buf.size(); ++i) {
            if (o && buf[o - 1].first == buf[i].first) {
                uint64_t s = (uint64_t)buf[o - 1].second + buf[i].second;
                if (s > kMaxWeight) return false;
                buf[o - 1].second = (uint32_t)s;
            } else {
                buf[o++] = buf[i];
            }
        }
        buf.resize(o);
        return true;
    };

    parallel_run(threads, [&](int t) {
        df_local[t].assign(n_terms, 0);
        mw_local[t].assign(n_terms, 0);
        std::vector<std::pair<uint32_t, uint32_t>> buf;
        uint64_t np = 0;
        uint32_t mw = 0;
        for (;;) {
            uint32_t tile = next_tile.fetch_add(1);
            if (tile >= n_tiles || err) break;
            uint32_t* cnt = seg_ptr.data() + (uint64_t)tile * stride;
            uint64_t o0 = (uint64_t)tile * tile_docs, o1 = std::min<uint64_t>(o0 + tile_docs, n_docs);
            for (uint64_t o = o0; o < o1; ++o) {
                if (!gather_doc(row_of_ord[o], buf)) {
                    err = 1;
                    break;
                }
                for (auto& e : buf) {
                    cnt[e.first]++;
                    df_local[t][e.first]++;
                    if (e.second > mw_local[t][e.first]) mw_local[t][e.first] = e.second;
                    if (e.second > mw) mw = e.second;
                }
                np += buf.size();
            }
        }
        n_postings += np;
        uint32_t cur = gmaxw.load();
        while (mw > cur && !gmaxw.compare_exchange_weak(cur, mw)) {
        }
    });
    if (err) {
        set_error("a term repeated inside one doc adds up to a weight above %u", kMaxWeight);
        return MSR_E_RANGE;
    }
    for (int t = 0; t < threads; ++t)
        for (uint32_t v = 0; v < n_terms; ++v) {
            df[v] += df_local[t][v];
            maxw[v] = std::max(maxw[v], mw_local[t][v]);
        }
    df_local.clear();
    mw_local.clear();

    // ---- dense head: the (at most dense_max_terms) terms with df >= dense_min_density * n_docs, by df descending
    const BuildOptions& bo = build_options();
    std::vector<uint32_t> dense_terms;
    std::vector<int8_t> dense_slot(n_terms, (int8_t)-1);
    if (bo.dense_max_terms > 0 && n_docs > 0) {
        std::vector<uint32_t> cand;
        const double thr = bo.dense_min_density * (double)n_docs;
        for (uint32_t v = 0; v < n_terms; ++v)
            if ((double)df[v] >= thr && df[v] > 0) cand.push_back(v);
        std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return df[a] > df[b] || (df[a] == df[b] && a < b); });
        if (cand.size() > std::min<uint32_t>(bo.dense_max_terms, kMaxDense)) cand.resize(std::min<uint32_t>(bo.dense_max_terms, kMaxDense));
        dense_terms = cand;
        if (dense_terms.size() % 2) dense_terms.push_back(0xFFFFFFFFu);  // padding slot: weights stay 0
        for (size_t s2 = 0; s2 < dense_terms.size(); ++s2)
            if (dense_terms[s2] != 0xFFFFFFFFu) dense_slot[dense_terms[s2]] = (int8_t)s2;
        // dense terms have no inverted lists
        for (uint32_t tile = 0; tile < n_tiles; ++tile) {
            uint32_t* c = seg_ptr.data() + (uint64_t)tile * stride;
            for (uint32_t t2 : cand) c[t2] = 0;
        }
    }
    const uint32_t n_dense = (uint32_t)dense_terms.size();
    const uint32_t n_pairs = n_dense / 2;

    // ---- counts -> vec offsets (each segment padded to a multiple of 4 postings)
    uint64_t n_vecs = 0;
    for (uint32_t tile = 0; tile < n_tiles; ++tile) {
        uint32_t* p = seg_ptr.data() + (uint64_t)tile * stride;
        for (uint32_t v = 0; v < n_terms; ++v) {
            uint32_t c = p[v];
            p[v] = (uint32_t)n_vecs;
            n_vecs += (c + 3) / 4;
            if (n_vecs >= (1ull << 32)) {
                set_error("index exceeds 2^32 posting vectors (64 GiB)");
                return MSR_E_RANGE;
            }
        }
        p[n_terms] = (uint32_t)n_vecs;
    }
    seg_ptr.resize((uint64_t)n_tiles * stride);

    // ---- fill postings, tile-parallel. Inside a segment the postings are sorted by ordinal and then laid out
    // chunk-interleaved (msr_internal.h): so that one wave's uint4 load of a chunk hands lane l the postings
    // l, l+nv, l+2nv, l+3nv of that chunk — consecutive lanes then hit consecutive accumulators (LDS banks).
    std::vector<uint32_t> postings;
    try {
        postings.assign(n_vecs * 4, 0u);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory for %llu posting vectors", (unsigned long long)n_vecs);
        return MSR_E_NOMEM;
    }
    std::vector<uint32_t> dense;
    try {
        dense.assign((uint64_t)n_tiles * n_pairs * tile_docs, 0u);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory for the dense head");
        return MSR_E_NOMEM;
    }
    next_tile = 0;
    parallel_run(threads, [&](int) {
        std::vector<std::pair<uint32_t, uint32_t>> buf;
        std::vector<uint32_t> cursor(n_terms), seg_cnt(n_terms);
        for (;;) {
            uint32_t tile = next_tile.fetch_add(1);
            if (tile >= n_tiles) break;
            const uint32_t* p = seg_ptr.data() + (uint64_t)tile * stride;
            uint64_t o0 = (uint64_t)tile * tile_docs, o1 = std::min<uint64_t>(o0 + tile_docs, n_docs);
            std::fill(cursor.begin(), cursor.end(), 0u);
            // pass A: how many postings each segment holds (needed to place an entry inside its chunk)
            std::fill(seg_cnt.begin(), seg_cnt.end(), 0u);
            for (uint64_t o = o0; o < o1; ++o) {
                gather_doc(row_of_ord[o], buf);
                for (auto& e : buf)
                    if (dense_slot[e.first] < 0) seg_cnt[e.first]++;
            }
            uint32_t* dtile = dense.data() + (uint64_t)tile * n_pairs * tile_docs;
            // pass B: place. Entry i of a segment lives in chunk c = i / 256 at position j = i % 256; a chunk with
            // m entries spans nv = ceil(m/4) vecs and entry j sits at vec (j % nv), element (j / nv).
            for (uint64_t o = o0; o < o1; ++o) {
                gather_doc(row_of_ord[o], buf);
                const uint32_t local = (uint32_t)(o - o0);
                for (auto& e : buf) {
                    const int ds = dense_slot[e.first];
                    if (ds >= 0) {
                        dtile[(uint64_t)(ds >> 1) * tile_docs + local] |= e.second << (16 * (ds & 1));
                        continue;
                    }
                    const uint32_t i = cursor[e.first]++;
                    const uint32_t c = i / kChunkPostings, j = i % kChunkPostings;
                    const uint32_t m = std::min<uint32_t>(kChunkPostings, seg_cnt[e.first] - c * kChunkPostings);
                    const uint32_t nv = (m + 3) / 4;
                    const uint64_t vec = (uint64_t)p[e.first] + (uint64_t)c * (kChunkPostings / 4) + (j % nv);
                    postings[vec * 4 + j / nv] = (e.second << 16) | local;
                }
            }
            // pass C: bank-aware arrangement inside every chunk. One LDS atomic of the scoring kernel adds element e of
            // the 64 lanes' vecs; the hardware serves lanes 0-31 and 32-63 as two groups over 32 banks
            // (bank = accumulator index mod 32), one extra cycle per additional lane on a bank. A chunk therefore
            // consists of up to 8 groups (4 elements x 2 half-waves) and its postings are dealt to them so that a
            // group holds as few postings per bank as possible: one from every non-empty bank first (largest bank
            // first), repeats only when fewer than 32 banks are left. Padding slots get weight 0 and a free bank.
            // The order of postings inside a chunk carries no meaning for the kernel.
            {
                std::vector<uint32_t> bucket[32];
                std::vector<uint32_t> chunk;
                for (uint32_t v = 0; v < n_terms; ++v) {
                    const uint32_t cntv = seg_cnt[v];
                    for (uint32_t c0 = 0; c0 < cntv; c0 += kChunkPostings) {
                        const uint32_t m = std::min<uint32_t>(kChunkPostings, cntv - c0);
                        const uint32_t nv = (m + 3) / 4;
                        uint32_t* base = postings.data() + ((uint64_t)p[v] + (uint64_t)(c0 / kChunkPostings) * (kChunkPostings / 4)) * 4;
                        for (auto& b : bucket) b.clear();
                        for (uint32_t j = 0; j < m; ++j) {
                            const uint32_t x = base[(j % nv) * 4 + j / nv];
                            bucket[x & 31u].push_back(x);
                        }
                        for (uint32_t e = 0; e < 4; ++e)
                            for (uint32_t half = 0; half < 2; ++half) {
                                const uint32_t l0 = half * 32;
                                if (l0 >= nv) continue;
                                const uint32_t cap = std::min<uint32_t>(32, nv - l0);
                                chunk.clear();
                                uint32_t used = 0;  // banks already taken in this group
                                while (chunk.size() < cap) {
                                    // one pass over the banks, fullest first, taking one posting from each
                                    uint32_t order[32];
                                    for (uint32_t b = 0; b < 32; ++b) order[b] = b;
                                    std::sort(order, order + 32, [&](uint32_t a2, uint32_t b2) {
                                        return bucket[a2].size() > bucket[b2].size() || (bucket[a2].size() == bucket[b2].size() && a2 < b2);
                                    });
                                    bool any = false;
                                    for (uint32_t oi = 0; oi < 32 && chunk.size() < cap; ++oi) {
                                        auto& b = bucket[order[oi]];
                                        if (b.empty()) break;
                                        chunk.push_back(b.back());
                                        b.pop_back();
                                        used |= 1u << order[oi];
                                        any = true;
                                    }
                                    if (!any) break;  // postings exhausted: the rest of the group is padding
                                }
                                for (uint32_t i = 0; i < cap; ++i) {
                                    uint32_t x;
                                    if (i < chunk.size()) {
                                        x = chunk[i];
                                    } else {
                                        uint32_t b = 0;
                                        while (b < 31 && (used >> b & 1u)) ++b;
                                        used |= 1u << b;
                                        x = b;  // weight 0, accumulator b
                                    }
                                    base[(l0 + i) * 4 + e] = x;
                                }
                            }
                    }
                }
            }
        }
    });

    // ---- dictionary strings
    std::vector<std::string> gen_terms;
    std::vector<const char*> tp(n_terms);
    if (term_strs) {
        for (uint32_t v = 0; v < n_terms; ++v) {
            if (!term_strs[v]) {
                set_error("term_strs[%u] is null", v);
                return MSR_E_INVAL;
            }
            tp[v] = term_strs[v];
        }
    } else {
        gen_terms.resize(n_terms);
        for (uint32_t v = 0; v < n_terms; ++v) {
            gen_terms[v] = std::to_string(v);
            tp[v] = gen_terms[v].c_str();
        }
    }
    std::vector<uint32_t> term_sorted(n_terms);
    std::iota(term_sorted.begin(), term_sorted.end(), 0u);
    std::sort(term_sorted.begin(), term_sorted.end(), [&](uint32_t a, uint32_t b) {
        int c = strcmp(tp[a], tp[b]);
        return c < 0 || (c == 0 && a < b);
    });
    for (uint32_t i = 1; i < n_terms; ++i)
        if (strcmp(tp[term_sorted[i - 1]], tp[term_sorted[i]]) == 0) {
            set_error("dictionary holds the term '%s' twice", tp[term_sorted[i]]);
            return MSR_E_INVAL;
        }
    std::vector<uint64_t> term_off(n_terms + 1ull, 0), doc_off(n_docs + 1, 0);
    for (uint32_t v = 0; v < n_terms; ++v) term_off[v + 1] = term_off[v] + strlen(tp[v]) + 1;
    for (uint64_t o = 0; o < n_docs; ++o) doc_off[o + 1] = doc_off[o] + strlen(idp[row_of_ord[o]]) + 1;

    // ---- write the file
    IndexHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "MSRIDX01", 8);
    h.version = kIndexVersion;
    h.tile_docs = tile_docs;
    h.n_docs = n_docs;
    h.n_postings = n_postings;
    h.n_vecs = n_vecs;
    h.n_terms = n_terms;
    h.n_tiles = n_tiles;
    h.max_weight = gmaxw;
    h.n_dense = n_dense;
    FileWriter w;
    if (!w.open(out_path)) {
        set_error("cannot create index file '%s': %s", out_path, strerror(errno));
        return MSR_E_IO;
    }
    w.write(&h, sizeof(h));
    w.pad16();
    auto sec = [&](int s, const void* p, uint64_t n) {
        h.off[s] = w.pos;
        h.size[s] = n;
        w.write(p, n);
        w.pad16();
    };
    sec(SEC_TERM_OFF, term_off.data(), term_off.size() * 8);
    h.off[SEC_TERM_STR] = w.pos;
    for (uint32_t v = 0; v < n_terms; ++v) w.write(tp[v], strlen(tp[v]) + 1);
    h.size[SEC_TERM_STR] = w.pos - h.off[SEC_TERM_STR];
    w.pad16();
    sec(SEC_TERM_SORTED, term_sorted.data(), term_sorted.size() * 4);
    sec(SEC_DF, df.data(), df.size() * 4);
    sec(SEC_MAXW, maxw.data(), maxw.size() * 4);
    sec(SEC_DOC_OFF, doc_off.data(), doc_off.size() * 8);
    h.off[SEC_DOC_STR] = w.pos;
    for (uint64_t o = 0; o < n_docs; ++o) w.write(idp[row_of_ord[o]], strlen(idp[row_of_ord[o]]) + 1);
    h.size[SEC_DOC_STR] = w.pos - h.off[SEC_DOC_STR];
    w.pad16();
    sec(SEC_SEG_PTR, seg_ptr.data(), seg_ptr.size() * 4);
    sec(SEC_POSTINGS, postings.data(), postings.size() * 4);
    sec(SEC_DENSE_TERMS, dense_terms.data(), dense_terms.size() * 4);
    sec(SEC_DENSE, dense.data(), dense.size() * 4);
    h.file_size = w.pos;
    if (!w.ok || fseek(w.f, 0, SEEK_SET) != 0) {
        w.close();
        set_error("write to '%s' failed", out_path);
        return MSR_E_IO;
    }
    w.pos = 0;
    w.write(&h, sizeof(h));
    if (!w.close()) {
        set_error("write to '%s' failed", out_path);
        return MSR_E_IO;
    }
    return MSR_OK;
}

// ----------------------------------------------------------------------------------------------- jsonl
namespace {

struct ParsedDoc {
    std::string id;
    std::vector<std::pair<uint32_t, uint32_t>> ent;  // (thread-local term id, weight), one per whitespace piece
};

struct LocalDict {
    std::unordered_map<std::string, uint32_t> map;
    std::vector<const std::string*> strs;
    uint32_t id_of(const std::string& s) {
        auto it = map.find(s);
        if (it != map.end()) return it->second;
        uint32_t id = (uint32_t)strs.size();
        auto ins = map.emplace(s, id);
        strs.push_back(&ins.first->first);
        return id;
    }
};

// A small strict JSON reader over one line.
struct JsonLine {
    const char* p;
    const char* end;
    const char* err = nullptr;

    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
    }
    bool fail(const char* m) {
        if (!err) err = m;
        return false;
    }
    static void put_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80)
            out.push_back((char)cp);
        else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    bool hex4(uint32_t& v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9')
                v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f')
                v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                v |= (uint32_t)(c - 'A' + 10);
            else
                return fail("bad hex digit in \\u escape");
        }
        return true;
    }
    bool string(std::string* out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        if (out) out->clear();
        while (p < end) {
            unsigned char c = (unsigned char)*p++;
            if (c == '"') return true;
            if (c == '\\') {
                if (p >= end) break;
                char e = *p++;
                uint32_t cp;
                switch (e) {
                    case '"': cp = '"'; break;
                    case '\\': cp = '\\'; break;
                    case '/': cp = '/'; break;
                    case 'b': cp = '\b'; break;
                    case 'f': cp = '\f'; break;
                    case 'n': cp = '\n'; break;
                    case 'r': cp = '\r'; break;
                    case 't': cp = '\t'; break;
                    case 'u': {
                        if (!hex4(cp)) return false;
                        if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                            const char* save = p;
                            p += 2;
                            uint32_t lo;
                            if (!hex4(lo)) return false;
                            if (lo >= 0xDC00 && lo < 0xE000)
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            else
                                p = save;  // lone high surrogate: emitted as its own 3-byte sequence
                        }
                        break;
                    }
                    default: return fail("bad escape in string");
                }
                if (out) put_utf8(*out, cp);
            } else {
                if (c < 0x20) return fail("control character in string");
                if (out) out->push_back((char)c);
            }
        }
        return fail("unterminated string");
    }
    // number -> truncated-toward-zero integer (Jackson asInt semantics for the reference's int weights)
    bool number(double& v, const char** b, const char** e) {
        const char* s = p;
        if (p < end && (*p == '-' || *p == '+')) ++p;
        bool digits = false;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
            if (*p >= '0' && *p <= '9') digits = true;
            ++p;
        }
        if (!digits) return fail("expected number");
        std::string tmp(s, p);
        char* ep = nullptr;
        v = strtod(tmp.c_str(), &ep);
        if (!ep || *ep) return fail("malformed number");
        if (b) *b = s;
        if (e) *e = p;
        return true;
    }
    bool literal(const char* lit) {
        size_t n = strlen(lit);
        if ((size_t)(end - p) < n || memcmp(p, lit, n) != 0) return fail("bad literal");
        p += n;
        return true;
    }
    bool skip_value(int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end of line");
        char c = *p;
        if (c == '"') return string(nullptr);
        if (c == '{') {
            ++p;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return true;
            }
            for (;;) {
                ws();
                if (!string(nullptr)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                if (!skip_value(depth + 1)) return false;
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return true;
            }
            for (;;) {
                if (!skip_value(depth + 1)) return false;
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (c == 't') return literal("true");
        if (c == 'f') return literal("false");
        if (c == 'n') return literal("null");
        double v;
        return number(v, nullptr, nullptr);
    }
};

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 0x09 && c <= 0x0D) || c == 0x1C || c == 0x1D || c == 0x1E || c == 0x1F; }

// Parse one corpus line. Returns false with *msg set on malformed input.
bool parse_corpus_line(const char* b, const char* e, LocalDict& dict, ParsedDoc& doc, const char** msg, int* rc) {
    JsonLine js{b, e};
    *rc = MSR_E_FORMAT;
    js.ws();
    if (js.p >= js.end || *js.p != '{') {
        *msg = "line is not a JSON object";
        return false;
    }
    ++js.p;
    bool have_id = false, have_vec = false;
    std::string key, tok;
    // last occurrence of a key wins, as in a JSON object
    std::vector<std::pair<std::string, int64_t>> vec;
    js.ws();
    if (js.p < js.end && *js.p == '}') {
        ++js.p;
    } else {
        for (;;) {
            js.ws();
            if (!js.string(&key)) break;
            js.ws();
            if (js.p >= js.end || *js.p != ':') {
                js.fail("expected ':'");
                break;
            }
            ++js.p;
            js.ws();
            if (key == "id") {
                if (js.p < js.end && *js.p == '"') {
                    if (!js.string(&doc.id)) break;
                } else {
                    double v;
                    const char *nb, *ne;
                    if (!js.number(v, &nb, &ne)) break;
                    doc.id.assign(nb, ne);
                }
                have_id = true;
            } else if (key == "vector") {
                vec.clear();
                if (js.p >= js.end || *js.p != '{') {
                    js.fail("\"vector\" is not an object");
                    break;
                }
                ++js.p;
                js.ws();
                if (js.p < js.end && *js.p == '}') {
                    ++js.p;
                } else {
                    bool bad = false;
                    for (;;) {
                        js.ws();
                        if (!js.string(&tok)) {
                            bad = true;
                            break;
                        }
                        js.ws();
                        if (js.p >= js.end || *js.p != ':') {
                            js.fail("expected ':'");
                            bad = true;
                            break;
                        }
                        ++js.p;
                        js.ws();
                        double v;
                        if (!js.number(v, nullptr, nullptr)) {
                            bad = true;
                            break;
                        }
                        if (!(v > -9.2e18 && v < 9.2e18)) {
                            js.fail("weight out of range");
                            bad = true;
                            break;
                        }
                        vec.emplace_back(tok, (int64_t)v);  // truncation toward zero
                        js.ws();
                        if (js.p < js.end && *js.p == ',') {
                            ++js.p;
                            continue;
                        }
                        if (js.p < js.end && *js.p == '}') {
                            ++js.p;
                            break;
                        }
                        js.fail("expected ',' or '}'");
                        bad = true;
                        break;
                    }
                    if (bad) break;
                }
                have_vec = true;
            } else {
                if (!js.skip_value()) break;
            }
            js.ws();
            if (js.p < js.end && *js.p == ',') {
                ++js.p;
                continue;
            }
            if (js.p < js.end && *js.p == '}') {
                ++js.p;
                break;
            }
            js.fail("expected ',' or '}'");
            break;
        }
    }
    if (!js.err) {
        js.ws();
        if (js.p != js.end) js.fail("trailing characters after the JSON object");
    }
    if (js.err) {
        *msg = js.err;
        return false;
    }
    if (!have_id) {
        *msg = "missing \"id\"";
        return false;
    }
    if (!have_vec) {
        *msg = "missing \"vector\"";
        return false;
    }
    // duplicate keys: last wins. Walk backwards and keep the first sighting of every key.
    doc.ent.clear();
    {
        std::unordered_map<std::string, char> seen;
        std::vector<std::pair<const std::string*, int64_t>> kept;
        for (size_t i = vec.size(); i-- > 0;) {
            if (seen.emplace(vec[i].first, 1).second) kept.emplace_back(&vec[i].first, vec[i].second);
        }
        for (size_t i = kept.size(); i-- > 0;) {
            int64_t wv = kept[i].second;
            if (wv <= 0) continue;
            if (wv > (int64_t)kMaxWeight) {
                *msg = "weight above 65535 is not supported";
                *rc = MSR_E_RANGE;
                return false;
            }
            const std::string& k = *kept[i].first;
            size_t a = 0, n = k.size();
            while (a < n) {
                while (a < n && is_space((unsigned char)k[a])) ++a;
                size_t z = a;
                while (z < n && !is_space((unsigned char)k[z])) ++z;
                if (z > a) {
                    tok.assign(k, a, z - a);
                    if (memchr(tok.data(), 0, tok.size())) {
                        *msg = "token contains a NUL byte";
                        return false;
                    }
                    doc.ent.emplace_back(dict.id_of(tok), (uint32_t)wv);
                }
                a = z;
            }
        }
    }
    if (memchr(doc.id.data(), 0, doc.id.size())) {
        *msg = "doc id contains a NUL byte";
        return false;
    }
    return true;
}

struct MappedFile {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const std::string& path) {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const char*)m;
        return true;
    }
    ~MappedFile() {
        if (p) munmap((void*)p, n);
        if (fd >= 0) ::close(fd);
    }
};

bool ends_with(const std::string& s, const char* suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

static int build_from_jsonl_dir(const char* dir, const char* out_path, int threads, uint32_t tile_docs) {
    threads = clamp_threads(threads);
    std::vector<std::string> files;
    {
        DIR* d = opendir(dir);
        if (!d) {
            set_error("cannot open directory '%s': %s", dir, strerror(errno));
            return MSR_E_IO;
        }
        while (dirent* e = readdir(d)) {
            std::string name = e->d_name;
            if (!ends_with(name, ".jsonl") && !ends_with(name, ".json")) continue;
            std::string full = std::string(dir) + "/" + name;
            struct stat st;
            if (stat(full.c_str(), &st) == 0 && S_ISREG(st.st_mode)) files.push_back(full);
        }
        closedir(d);
    }
    std::sort(files.begin(), files.end());
    if (files.empty()) {
        set_error("no *.jsonl / *.json file in '%s'", dir);
        return MSR_E_IO;
    }

    // Each thread parses a byte range (aligned to line starts) of each file; docs keep file order.
    struct Chunk {
        std::vector<ParsedDoc> docs;
    };
    std::vector<LocalDict> dicts(threads);
    std::vector<std::vector<Chunk>> chunks(files.size(), std::vector<Chunk>(threads));
    std::atomic<int> err{0};
    std::string err_msg;
    std::mutex* mu = new std::mutex;
    std::unique_ptr<std::mutex> mu_guard(mu);

    for (size_t fi = 0; fi < files.size() && !err; ++fi) {
        MappedFile mf;
        if (!mf.open(files[fi])) {
            set_error("cannot read '%s': %s", files[fi].c_str(), strerror(errno));
            return MSR_E_IO;
        }
        const char* base = mf.p;
        const size_t n = mf.n;
        // line-aligned split points
        std::vector<size_t> cut(threads + 1);
        cut[0] = 0;
        cut[threads] = n;
        for (int t = 1; t < threads; ++t) {
            size_t c = n * t / threads;
            if (c < cut[t - 1]) c = cut[t - 1];
            while (c < n && c > 0 && base[c - 1] != '\n') ++c;
            cut[t] = c;
        }
        parallel_run(threads, [&](int t) {
            size_t a = cut[t], b = cut[t + 1];
            // line number of `a` (for messages) is computed lazily on error
            ParsedDoc doc;
            while (a < b && !err) {
                const char* nl = (const char*)memchr(base + a, '\n', b - a);
                size_t e = nl ? (size_t)(nl - base) : b;
                const char* lb = base + a;
                const char* le = base + e;
                bool blank = true;
                for (const char* q = lb; q < le; ++q)
                    if (!is_space((unsigned char)*q)) {
                        blank = false;
                        break;
                    }
                if (!blank) {
                    const char* msg = nullptr;
                    int rc = MSR_E_FORMAT;
                    if (!parse_corpus_line(lb, le, dicts[t], doc, &msg, &rc)) {
                        size_t line = 1;
                        for (size_t i = 0; i < a; ++i) line += base[i] == '\n';
                        std::lock_guard<std::mutex> g(*mu);
                        if (!err) {
                            err = rc;
                            char tmp[900];
                            snprintf(tmp, sizeof(tmp), "%s:%zu: %s", files[fi].c_str(), line, msg ? msg : "parse error");
                            err_msg = tmp;
                        }
                        return;
                    }
                    chunks[fi][t].docs.push_back(doc);
                }
                a = e + 1;
            }
        });
    }
    if (err) {
        set_error("%s", err_msg.c_str());
        return err;
    }

    // ---- global dictionary: sorted unique strings; ids are ranks
    std::vector<const std::string*> all;
    for (auto& d : dicts) all.insert(all.end(), d.strs.begin(), d.strs.end());
    std::sort(all.begin(), all.end(), [](const std::string* a, const std::string* b) { return *a < *b; });
    all.erase(std::unique(all.begin(), all.end(), [](const std::string* a, const std::string* b) { return *a == *b; }),
              all.end());
    if (all.size() >= (1ull << 31)) {
        set_error("dictionary too large");
        return MSR_E_RANGE;
    }
    const uint32_t n_terms = (uint32_t)all.size();
    std::vector<std::vector<uint32_t>> remap(threads);
    parallel_run(threads, [&](int t) {
        remap[t].resize(dicts[t].strs.size());
        for (size_t i = 0; i < dicts[t].strs.size(); ++i) {
            auto it = std::lower_bound(all.begin(), all.end(), dicts[t].strs[i],
                                       [](const std::string* a, const std::string* b) { return *a < *b; });
            remap[t][i] = (uint32_t)(it - all.begin());
        }
    });

    // ---- CSR in file order
    uint64_t n_docs = 0, nnz = 0;
    for (auto& f : chunks)
        for (auto& c : f) {
            n_docs += c.docs.size();
            for (auto& d : c.docs) nnz += d.ent.size();
        }
    std::vector<uint64_t> doc_ptr;
    std::vector<uint32_t> tid, wv;
    std::vector<const char*> ids;
    doc_ptr.reserve(n_docs + 1);
    tid.reserve(nnz);
    wv.reserve(nnz);
    ids.reserve(n_docs);
    doc_ptr.push_back(0);
    for (auto& f : chunks)
        for (int t = 0; t < threads; ++t)
            for (auto& d : f[t].docs) {
                for (auto& e : d.ent) {
                    tid.push_back(remap[t][e.first]);
                    wv.push_back(e.second);
                }
                doc_ptr.push_back(tid.size());
                ids.push_back(d.id.c_str());
            }
    std::vector<const char*> tstr(n_terms);
    for (uint32_t v = 0; v < n_terms; ++v) tstr[v] = all[v]->c_str();
    return build_from_csr(out_path, n_docs, n_terms, doc_ptr.data(), tid.data(), wv.data(), ids.data(), tstr.data(),
                          threads, tile_docs);
}

}  // namespace msr

// ================================================================================================ C-ABI (host part)
using namespace msr;

extern "C" {

const char* msr_last_error(void) { return msr::last_error(); }
const char* msr_version(void) { return "mllm_sparse_retrieval_amd 0.1 (gfx950)"; }

int msr_set_build_option(const char* key, double value) {
    if (!key) {
        set_error("msr_set_build_option: null key");
        return MSR_E_INVAL;
    }
    BuildOptions& o = build_options();
    if (strcmp(key, "dense_min_density") == 0 && value >= 0) {
        o.dense_min_density = value;
    } else if (strcmp(key, "dense_max_terms") == 0 && value >= 0 && value <= kMaxDense) {
        o.dense_max_terms = (uint32_t)value;
    } else if (strcmp(key, "tie_order") == 0 && (value == 0 || value == 1)) {
        o.tie_input_order = value == 1;
    } else {
        set_error("unknown build option '%s' or value %g out of range", key, value);
        return MSR_E_INVAL;
    }
    return MSR_OK;
}

int msr_index_build(const char* jsonl_dir, const char* out_path, int threads, uint32_t tile_docs) {
    if (!jsonl_dir || !out_path) {
        set_error("msr_index_build: null path");
        return MSR_E_INVAL;
    }
    try {
        return build_from_jsonl_dir(jsonl_dir, out_path, threads, tile_docs);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while building the index");
        return MSR_E_NOMEM;
    } catch (const std::exception& e) {
        set_error("index build failed: %s", e.what());
        return MSR_E_INVAL;
    }
}

int msr_index_build_csr(const char* out_path, uint64_t n_docs, uint32_t n_terms, const uint64_t* doc_ptr,
                        const uint32_t* term_id, const uint32_t* weight, const char* const* doc_ids,
                        const char* const* term_strs, int threads, uint32_t tile_docs) {
    try {
        return build_from_csr(out_path, n_docs, n_terms, doc_ptr, term_id, weight, doc_ids, term_strs, threads,
                              tile_docs);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while building the index");
        return MSR_E_NOMEM;
    } catch (const std::exception& e) {
        set_error("index build failed: %s", e.what());
        return MSR_E_INVAL;
    }
}

}  // extern "C"

namespace msr {
void term_bounds(const HostIndex& hx, int G, std::vector<uint32_t>& bounds) {
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
}  // namespace msr

// ----------------------------------------------------------------------------------------------- f32 -> fp16 (host)
// Round to nearest even, as IEEE / numpy / torch .half(): F16C (vcvtps2ph) where the CPU has it, else bit arithmetic.
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("f16c,avx"))) static void f32_to_f16_f16c(const float* src, uint16_t* dst, uint64_t n) {
    uint64_t i = 0;
    for (; i + 8 <= n; i += 8)
        _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i),
                         _mm256_cvtps_ph(_mm256_loadu_ps(src + i), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
    for (; i < n; ++i) {
        const __m128i h = _mm_cvtps_ph(_mm_set_ss(src[i]), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        dst[i] = (uint16_t)_mm_extract_epi16(h, 0);
    }
}
#endif
static uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x0200u | ((x >> 13) & 0x03FFu) : 0u));  // inf / nan
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to >= 65520: overflow to inf
    if (x < 0x33000001u) return (uint16_t)sign;               // <= 2^-25: rounds to zero (ties to even)
    uint32_t mant = (x & 0x007FFFFFu) | 0x00800000u;
    const int exp = (int)(x >> 23) - 127;
    int shift = 13;
    uint32_t he = 0;
    if (exp < -14) shift += -14 - exp;  // subnormal half: more bits leave
    else he = (uint32_t)(exp + 15) << 10;
    const uint32_t rest = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t hm = mant >> shift;
    if (rest > half || (rest == half && (hm & 1u))) ++hm;
    // (normal: the implicit bit sits at 0x400 and adds one to the exponent field on top of he - 0x400)
    return (uint16_t)(sign | (exp < -14 ? hm : (he - 0x400u + hm)));
}

extern "C" int msr_f32_to_f16(const float* src, uint16_t* dst, uint64_t n, int threads) {
    if ((!src || !dst) && n) {
        msr::set_error("msr_f32_to_f16: null argument");
        return MSR_E_INVAL;
    }
    bool f16c = false;
#if defined(__x86_64__)
    f16c = __builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx") && !getenv("MSR_NO_F16C");
#endif
    int nt = msr::clamp_threads(threads);
    nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min(nt, 32), n >> 16));  // >= 64 Ki values per thread
    msr::parallel_run(nt, [&](int t) {
        const uint64_t a = n * (uint64_t)t / (uint64_t)nt / 8 * 8, b = t + 1 == nt ? n : n * (uint64_t)(t + 1) / (uint64_t)nt / 8 * 8;
#if defined(__x86_64__)
        if (f16c) {
            f32_to_f16_f16c(src + a, dst + a, b - a);
            return;
        }
#endif
        for (uint64_t i = a; i < b; ++i) dst[i] = f32_to_f16_bits(src[i]);
    });
    return MSR_OK;
}

extern "C" {

// by_terms = false: doc-range shard (contiguous tile range); true: term-range shard (every tile, own term range)
static int open_common(const char* path, int device, int shard, int n_shards, msr_index** out, bool by_terms = false) {
    if (!path || !out) {
        set_error("msr_index_open: null argument");
        return MSR_E_INVAL;
    }
    *out = nullptr;
    if (n_shards < 1 || shard < 0 || shard >= n_shards) {
        set_error("bad shard %d of %d", shard, n_shards);
        return MSR_E_INVAL;
    }
    msr_index* ix = new (std::nothrow) msr_index;
    if (!ix) {
        set_error("out of host memory");
        return MSR_E_NOMEM;
    }
    int rc = ix->host.open(path);
    if (rc != MSR_OK) {
        delete ix;
        return rc;
    }
    const uint32_t T = ix->host.h->n_tiles;
    ix->term_hi = ix->host.h->n_terms;
    if (by_terms) {
        std::vector<uint32_t> tb;
        term_bounds(ix->host, n_shards, tb);
        ix->term_shard = shard;
        ix->term_nshards = n_shards;
        ix->term_lo = tb[(size_t)shard];
        ix->term_hi = tb[(size_t)shard + 1];
        ix->shard_tile0 = 0;
        ix->shard_ntiles = T;
    } else {
        // doc-range shard = contiguous tile range, balanced by tile count
        uint32_t t0 = (uint32_t)((uint64_t)T * shard / n_shards), t1 = (uint32_t)((uint64_t)T * (shard + 1) / n_shards);
        ix->shard_tile0 = t0;
        ix->shard_ntiles = t1 - t0;
    }
    if (device >= 0) {
        rc = device_attach(ix, device);
        if (rc != MSR_OK) {
            ix->host.close();
            delete ix;
            return rc;
        }
    }
    *out = ix;
    return MSR_OK;
}

int msr_index_open(const char* path, int device, msr_index** out) { return open_common(path, device, 0, 1, out); }

int msr_index_open_shard(const char* path, int device, int shard, int n_shards, msr_index** out) {
    return open_common(path, device, shard, n_shards, out);
}

int msr_index_open_termshard(const char* path, int device, int shard, int n_shards, msr_index** out) {
    return open_common(path, device, shard, n_shards, out, true);
}

void msr_index_close(msr_index* ix) {
    if (!ix) return;
    if (ix->dev) device_detach(ix);
    ix->host.close();
    delete ix;
}

int msr_index_info(const msr_index* ix, msr_info* info) {
    if (!ix || !info) {
        set_error("msr_index_info: null argument");
        return MSR_E_INVAL;
    }
    const IndexHeader* h = ix->host.h;
    memset(info, 0, sizeof(*info));
    info->n_docs = h->n_docs;
    info->n_postings = h->n_postings;
    info->n_vecs = h->n_vecs;
    info->n_terms = h->n_terms;
    info->tile_docs = h->tile_docs;
    info->n_tiles = h->n_tiles;
    info->max_weight = h->max_weight;
    info->shard_tile0 = ix->shard_tile0;
    info->shard_ntiles = ix->shard_ntiles;
    info->device = ix->dev ? ix->device : -1;
    info->n_dense = h->n_dense;
    info->term_lo = ix->term_lo;
    info->term_hi = ix->term_hi;
    info->resident_bytes = ix->dev ? device_resident_bytes(ix) : 0;
    return MSR_OK;
}

int msr_term_lookup(const msr_index* ix, const char* const* toks, int n, int32_t* term_ids) {
    if (!ix || n < 0 || (n && (!toks || !term_ids))) {
        set_error("msr_term_lookup: bad argument");
        return MSR_E_INVAL;
    }
    for (int i = 0; i < n; ++i) term_ids[i] = toks[i] ? ix->host.lookup(toks[i]) : -1;
    return MSR_OK;
}

int msr_term_df(const msr_index* ix, const int32_t* term_ids, int n, uint32_t* df) {
    if (!ix || n < 0 || (n && (!term_ids || !df))) {
        set_error("msr_term_df: bad argument");
        return MSR_E_INVAL;
    }
    for (int i = 0; i < n; ++i) {
        int32_t t = term_ids[i];
        df[i] = (t >= 0 && (uint32_t)t < ix->host.h->n_terms) ? ix->host.df[t] : 0;
    }
    return MSR_OK;
}

int msr_term_str(const msr_index* ix, uint32_t term_id, const char** s) {
    if (!ix || !s || term_id >= ix->host.h->n_terms) {
        set_error("msr_term_str: bad argument");
        return MSR_E_INVAL;
    }
    *s = ix->host.term_str + ix->host.term_off[term_id];
    return MSR_OK;
}

// Query strings as the reference builds them (every token repeated `weight` times, src/search.py:419-422) -> CSR of
// (term id, count): whitespace split, token-frequency count, dictionary lookup (out-of-vocabulary tokens dropped), all
// in one pass on the host. This is the half of pyserini's batch_search that runs before Lucene.
static void encode_query_range(const msr_index* ix, const char* const* queries, int q0, int q1, int64_t* counts,
                               std::vector<int32_t>& q_term, std::vector<int32_t>& q_w) {
    std::unordered_map<std::string, int32_t> slot;  // token -> index into q_term/q_w of the current query
    std::string tok;
    for (int i = q0; i < q1; ++i) {
        slot.clear();
        const size_t first = q_term.size();
        const char* p = queries[i] ? queries[i] : "";
        // The reference writes a token `weight` times IN A ROW (src/search.py:419-422): a token equal to its
        // predecessor bumps the same slot without touching the hash map.
        const char* prev_b = nullptr;
        size_t prev_len = 0;
        int32_t prev_slot = -1;
        while (*p) {
            while (*p && is_space((unsigned char)*p)) ++p;
            const char* b = p;
            while (*p && !is_space((unsigned char)*p)) ++p;
            if (p == b) break;
            const size_t len = (size_t)(p - b);
            if (prev_b && len == prev_len && memcmp(b, prev_b, len) == 0) {
                if (prev_slot >= 0 && q_w[(size_t)prev_slot] < INT32_MAX) q_w[(size_t)prev_slot]++;
                continue;
            }
            prev_b = b;
            prev_len = len;
            tok.assign(b, p);
            auto it = slot.find(tok);
            if (it != slot.end()) {
                prev_slot = it->second;
                if (it->second >= 0 && q_w[(size_t)it->second] < INT32_MAX) q_w[(size_t)it->second]++;
                continue;
            }
            const int32_t tid = ix->host.lookup(tok.c_str());
            if (tid < 0) {
                slot.emplace(tok, -1);  // out of vocabulary: remember, so that repeats cost one hash probe
                prev_slot = -1;
                continue;
            }
            prev_slot = (int32_t)q_term.size();
            slot.emplace(tok, prev_slot);
            q_term.push_back(tid);
            q_w.push_back(1);
        }
        counts[i] = (int64_t)(q_term.size() - first);
    }
}

// Big batches (a whole query file: 5 000 strings of ~18 000 tokens each at the reference's shapes) are encoded by several
// threads over contiguous query ranges and laid end to end: the result is the serial one.
static void encode_query_strings(const msr_index* ix, const char* const* queries, int nq, std::vector<int64_t>& q_ptr,
                                 std::vector<int32_t>& q_term, std::vector<int32_t>& q_w) {
    q_ptr.assign((size_t)nq + 1, 0);
    q_term.clear();
    q_w.clear();
    const int n_parts = nq >= 64 ? std::min(clamp_threads(0), std::min(16, nq / 16)) : 1;
    if (n_parts <= 1) {
        encode_query_range(ix, queries, 0, nq, q_ptr.data() + 1, q_term, q_w);
    } else {
        std::vector<std::vector<int32_t>> pt((size_t)n_parts), pw((size_t)n_parts);
        parallel_run(n_parts, [&](int pi) {
            const int a = (int)((int64_t)nq * pi / n_parts), b = (int)((int64_t)nq * (pi + 1) / n_parts);
            encode_query_range(ix, queries, a, b, q_ptr.data() + 1, pt[(size_t)pi], pw[(size_t)pi]);
        });
        for (int pi = 0; pi < n_parts; ++pi) {
            q_term.insert(q_term.end(), pt[(size_t)pi].begin(), pt[(size_t)pi].end());
            q_w.insert(q_w.end(), pw[(size_t)pi].begin(), pw[(size_t)pi].end());
        }
    }
    for (int i = 0; i < nq; ++i) q_ptr[(size_t)i + 1] += q_ptr[(size_t)i];  // counts -> end offsets
}

int msr_encode_queries(const msr_index* ix, const char* const* queries, int nq, int64_t* q_ptr, int32_t* q_term,
                       int32_t* q_w, int64_t cap, int64_t* n_entries) {
    if (!ix || nq < 0 || (nq && !queries) || !q_ptr || !n_entries || cap < 0 || (cap && (!q_term || !q_w))) {
        set_error("msr_encode_queries: bad argument");
        return MSR_E_INVAL;
    }
    try {
        std::vector<int64_t> p;
        std::vector<int32_t> t, w;
        encode_query_strings(ix, queries, nq, p, t, w);
        memcpy(q_ptr, p.data(), ((size_t)nq + 1) * sizeof(int64_t));
        *n_entries = (int64_t)t.size();
        if ((int64_t)t.size() <= cap && !t.empty()) {
            memcpy(q_term, t.data(), t.size() * sizeof(int32_t));
            memcpy(q_w, w.data(), w.size() * sizeof(int32_t));
        }
        return MSR_OK;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while encoding the queries");
        return MSR_E_NOMEM;
    }
}

int msr_search_text(msr_index* ix, const char* const* queries, int nq, int k, uint32_t flags, uint32_t* out_doc_ord,
                    float* out_score, uint32_t* out_score_u32, int32_t* out_n) {
    if (!ix || nq < 0 || (nq && !queries)) {
        set_error("msr_search_text: bad argument");
        return MSR_E_INVAL;
    }
    try {
        std::vector<int64_t> q_ptr;
        std::vector<int32_t> q_term, q_w;
        encode_query_strings(ix, queries, nq, q_ptr, q_term, q_w);
        return msr_search_csr(ix, q_ptr.data(), q_term.data(), q_w.data(), nq, k, flags, out_doc_ord, out_score, out_score_u32,
                              out_n);
    } catch (const std::bad_alloc&) {
        set_error("out of host memory while encoding the queries");
        return MSR_E_NOMEM;
    }
}

int msr_docid_str(const msr_index* ix, uint32_t ord, const char** s) {
    if (!ix || !s || ord >= ix->host.h->n_docs) {
        set_error("msr_docid_str: ordinal %u out of range", ord);
        return MSR_E_INVAL;
    }
    *s = ix->host.doc_str + ix->host.doc_off[ord];
    return MSR_OK;
}

}  // extern "C"
