// Internal declarations shared by the host-side indexer and the HIP search path. Not part of the C-ABI.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msr.h"

namespace msr {

// ---------------------------------------------------------------------------------------------
// On-disk / in-HBM index layout ("tile-major" inverted index)
//
//   docs are numbered by ORDINAL = rank of the external doc-id string in bytewise ascending order
//   (tie rule T1 for free), and cut into tiles of `tile_docs` consecutive ordinals. For every
//   (tile, term) pair the postings of that term that fall into that tile form one SEGMENT:
//
//     posting  = u32   (weight << 16) | (ordinal - tile*tile_docs)      weight in [1, 65535]; weight 0 = padding
//     segment  = postings sorted by ordinal, zero-padded to a multiple of 4 (one 16-byte "vec"), stored
//                chunk-interleaved: a chunk is 256 consecutive postings (64 vecs, one wave-wide uint4 load);
//                inside a chunk of m postings spanning nv = ceil(m/4) vecs, posting j sits at vec j % nv,
//                element j / nv — lane l of a wave therefore receives postings l, l+nv, l+2nv, l+3nv, and the
//                64 lanes of one LDS atomic touch (nearly) consecutive accumulators instead of a stride of 4
//     seg_ptr  = u32[n_tiles][n_terms+1]  first vec of each segment (absolute vec index)
//
//   Segments of one tile are contiguous (term-major inside the tile), so one tile's slice of the
//   index is a contiguous byte range: that is what a doc-range shard uploads, and what the
//   workgroups scoring that tile keep hot in their XCD's L2.
//
//   DENSE HEAD. Learned-sparse vocabularies have a head of terms that occur in a large share of the docs
//   (Zipf(0.8): the top 16 terms carry 2/3 of the postings a query touches). For such terms an inverted
//   list is the wrong shape: every posting costs an LDS atomic. The `n_dense` (even, <= 32) terms with
//   df >= dense_min_density * n_docs are therefore stored doc-major per tile instead,
//
//     dense    = u32[n_tiles][n_dense/2][tile_docs]   (weight of term 2p+1) << 16 | (weight of term 2p), 0 = absent
//
//   have EMPTY segments, and are scored by the owning thread of each accumulator with v_dot2_u32_u16
//   (two postings per VALU op, no atomics), which also replaces the zeroing pass of the accumulators.
// ---------------------------------------------------------------------------------------------

enum Section : int {
    SEC_TERM_OFF = 0,   // u64[n_terms+1]   byte offsets into SEC_TERM_STR (NUL-terminated strings)
    SEC_TERM_STR,       // bytes
    SEC_TERM_SORTED,    // u32[n_terms]     term ids in bytewise ascending order of their strings
    SEC_DF,             // u32[n_terms]     document frequency
    SEC_MAXW,           // u32[n_terms]     largest stored weight of the term
    SEC_DOC_OFF,        // u64[n_docs+1]    byte offsets into SEC_DOC_STR (NUL-terminated), by ordinal
    SEC_DOC_STR,        // bytes
    SEC_SEG_PTR,        // u32[n_tiles*(n_terms+1)]
    SEC_POSTINGS,       // u32[n_vecs*4]
    SEC_DENSE_TERMS,    // u32[n_dense]     term id of every dense slot (slot s = bits 16*(s&1) of pair s/2)
    SEC_DENSE,          // u32[n_tiles*(n_dense/2)*tile_docs]
    SEC_COUNT
};

struct IndexHeader {
    char magic[8];  // "MSRIDX01"
    uint32_t version;
    uint32_t tile_docs;
    uint64_t n_docs;
    uint64_t n_postings;
    uint64_t n_vecs;
    uint32_t n_terms;
    uint32_t n_tiles;
    uint32_t max_weight;
    uint32_t n_dense;          // dense-head terms (even)
    uint64_t off[SEC_COUNT];   // byte offset of each section in the file
    uint64_t size[SEC_COUNT];  // byte size of each section
    uint64_t file_size;
    uint64_t reserved[3];
};

constexpr uint32_t kIndexVersion = 2;
constexpr uint32_t kMaxDense = 32;
constexpr uint32_t kDefaultTileDocs = 8192;
constexpr uint32_t kMaxWeight = 65535;
constexpr uint32_t kChunkPostings = 256;  // postings per chunk (64 lanes x uint4)

// thread-local error message (msr_last_error)
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

// run fn(thread_idx) on n_threads threads (n_threads <= 1 runs inline)
void parallel_run(int n_threads, const std::function<void(int)>& fn);
int clamp_threads(int threads);

// Host side of an opened index (mmap of the file). The device side lives in msr_device.hip.
struct HostIndex {
    int fd = -1;
    const uint8_t* base = nullptr;
    size_t bytes = 0;
    const IndexHeader* h = nullptr;
    const uint64_t* term_off = nullptr;
    const char* term_str = nullptr;
    const uint32_t* term_sorted = nullptr;
    const uint32_t* df = nullptr;
    const uint32_t* maxw = nullptr;
    const uint64_t* doc_off = nullptr;
    const char* doc_str = nullptr;
    const uint32_t* seg_ptr = nullptr;
    const uint32_t* postings = nullptr;
    const uint32_t* dense_terms = nullptr;
    const uint32_t* dense = nullptr;
    std::vector<int8_t> dense_slot;  // term id -> dense slot, -1 = sparse term

    int open(const char* path);  // MSR_OK or error (message set)
    void close();
    int32_t lookup(const char* tok) const;  // -1 if absent
};

struct DeviceIndex;  // defined in msr_device.hip

}  // namespace msr

struct msr_batch;

// The opaque C handle: host mmap + (optional) device residency.
struct msr_index {
    msr::HostIndex host;
    msr::DeviceIndex* dev = nullptr;  // null when opened with device < 0
    int device = -1;
    uint32_t shard_tile0 = 0;
    uint32_t shard_ntiles = 0;
    // term-range shard (msr_index_open_termshard): only the segments of terms [term_lo, term_hi) and the dense-head
    // pairs that hold one of them are resident; term_nshards = 0 for handles that hold every term
    int term_shard = -1;
    int term_nshards = 0;
    uint32_t term_lo = 0;
    uint32_t term_hi = 0;  // set to n_terms at open
    // batches created on this handle and not destroyed yet: msr_index_close releases their device buffers and
    // detaches them (a later msr_batch_destroy only frees the host object), so the order of the two calls is free
    std::vector<msr_batch*> live_batches;
};

namespace msr {

// implemented in msr_device.hip
int device_attach(msr_index* ix, int device);  // upload the shard [shard_tile0, shard_tile0+shard_ntiles)
void device_detach(msr_index* ix);
uint64_t device_resident_bytes(const msr_index* ix);  // index bytes held in HBM by this handle

// Term ownership for term-range sharding: G contiguous term-id ranges balanced by postings (cumulative df), the same
// on every rank because it depends only on the index. bounds has G+1 entries. (msr_index.cpp)
void term_bounds(const HostIndex& hx, int G, std::vector<uint32_t>& bounds);

// build options (msr_set_build_option)
struct BuildOptions {
    double dense_min_density = 0.4;  // a term is stored in the dense head when df >= this * n_docs
    uint32_t dense_max_terms = 16;   // at most this many (<= kMaxDense); 0 disables the dense head
    bool tie_input_order = false;    // doc ordinals = input order instead of doc-id string order (tie rule switch)
};
BuildOptions& build_options();

int build_from_csr(const char* out_path, uint64_t n_docs, uint32_t n_terms, const uint64_t* doc_ptr,
                   const uint32_t* term_id, const uint32_t* weight, const char* const* doc_ids,
                   const char* const* term_strs, int threads, uint32_t tile_docs);

}  // namespace msr
