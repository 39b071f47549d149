// Accumulation phase of the search-path kernels, shared by score_tiles (msr_kernels.hpp) and hybrid_tiles
// (msr_hybrid.hip): after accumulate_tile() the workgroup's LDS tile holds the exact integer scores of ONE query against
// ONE doc tile,  acc[d] = sum over the query's terms t of  q_w(t) * tf(t, tile doc d)   (Lucene impact scoring below
// LuceneImpactSearcher.batch_search, src/search.py:86-87).
//     - dense-head terms: doc-major rows scored by the accumulator's owner with v_dot2_u32_u16 (this is also the
//       accumulator init),
//     - the query's other (term, tile) segments are cut into 1-KiB chunks (64 lanes x 16 B = 256 postings); waves take
//       chunks round-robin, resolve 64 chunks lane-parallel, then walk them with v_readlane broadcasts: one 16-byte
//       buffer load (chunk end in the resource's size word) and four ds_add_u32 per lane and chunk, two register
//       banks in flight.
#pragma once

#include "msr_select.hpp"

namespace msr {

// <docs per tile, threads, 1-KiB chunk loads per register bank, diagnostic build>; `lds` is the TileLds carve of the
// calling kernel (accumulators first, then the staging / select union). Ends with a workgroup barrier: every
// accumulator is final, the staging view of the union is dead.
template <int TILE_DOCS, int NT, int U, bool DBG, class Stamp>
__device__ __forceinline__ void accumulate_tile(const ScoreArgs& a, const uint32_t q, const uint32_t tile_l,
                                                const int rounds, uint8_t* const lds, SelectScratch& ss, Stamp stamp,
                                                const uint32_t tid) {
    constexpr int NW = NT / 64;
    static_assert(TILE_DOCS % (4 * NT) == 0, "tile must be a multiple of 4*NT");
    static_assert(NT >= kQtBlock, "the staging scan uses the first 256 threads");
    using L = TileLds<TILE_DOCS, NT, 512>;  // (kAcc does not depend on the candidate capacity)
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    uint8_t* const un = lds + L::kAcc;
    // streaming-phase view of the union
    uint32_t* const seg_start = reinterpret_cast<uint32_t*>(un);
    uint32_t* const seg_len = seg_start + kQtBlock;
    uint32_t* const seg_w = seg_len + kQtBlock;
    uint32_t* const pref = seg_w + kQtBlock;
    uint32_t* const wsum = pref + kQtBlock + 4;
    const uint32_t lane = tid & 63;
    const uint32_t lane16 = lane * 16u;         // byte offset of the lane's vec inside a chunk
    const uint32_t wave = rfl(tid >> 6);        // provably wave-uniform for the compiler
    uint4* const a4 = reinterpret_cast<uint4*>(acc);

    const uint4 meta = a.q_meta[q];  // one scalar load: term range + which dense-head pairs the query holds
    const uint32_t qb = meta.x, qe = meta.y;
    const uint32_t* seg_row = a.seg_ptr + (uint64_t)tile_l * (a.n_terms + 1);
    // Postings and dense rows are read with buffer loads: (tile base + size in SGPRs) + (32-bit byte offset per
    // lane). A tile's segments span less than 4 GiB (checked when the index is attached); there is no 64-bit address
    // arithmetic per chunk (the scalar ALU is shared by the CU's 32 waves), no zero-extended offset pairs in VGPRs,
    // and a read past the tile's postings returns 0 instead of faulting.
    const uint32_t tile_first = seg_row[0];
    const char* const post_base = reinterpret_cast<const char*>(a.postings) + (uint64_t)(tile_first - a.vec_base) * 16u;

    // ---- first round's (term -> segment) lookups: two dependent global loads. The first (terms, weights) goes out
    // now; the second (segment pointers) needs the terms, so it is issued AFTER the first dense rows have been
    // requested — otherwise the wait for the terms would hold the rows back by a round trip.
    uint32_t pre_t = 0, pre_w = 0, pre_s0 = 0, pre_s1 = 0;
    const bool has_lk = tid < min((uint32_t)kQtBlock, qe - qb);
    if (has_lk) {
        pre_t = a.q_term[qb + tid];
        pre_w = a.q_w[qb + tid];
    }
    auto lookup2 = [&]() {
        if (has_lk) {
            pre_s0 = seg_row[pre_t];
            pre_s1 = seg_row[pre_t + 1];
        }
    };

    // ---- initialise the accumulators: zero, or — when the query holds dense-head terms — their whole contribution.
    // Thread `tid` owns vecs r*NT + tid (4 consecutive docs each); the dense head is doc-major, one dword per doc and
    // term pair, so the owner scores two postings per v_dot2_u32_u16 and stores the sums with a plain ds_write_b128:
    // no atomics, and no separate zeroing pass. Term pairs the query does not hold are skipped (wave-uniform bit
    // mask); the rows of the next pair are in flight while the current pair is accumulated (two register banks).
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        constexpr int RG = 4;  // rounds per register group
        const uint32_t qv = lane < a.n_pairs ? a.q_dense[(uint64_t)q * a.n_pairs + lane] : 0u;
        // (the mask comes with the scalar load above, so the first rows are requested without waiting for qv)
        const uint32_t pmask = (DBG && (a.dbg & 16u)) ? 0u : meta.z;
        const __amdgpu_buffer_rsrc_t rs_dense =
            make_rsrc(reinterpret_cast<const char*>(a.dense) + (uint64_t)tile_l * a.n_pairs * (TILE_DOCS * 4u),
                      a.n_pairs * (uint32_t)(TILE_DOCS * 4));
        for (int r0 = 0; r0 < rounds; r0 += RG) {
            uint4 sacc[RG];
#pragma unroll
            for (int i = 0; i < RG; ++i) sacc[i] = make_uint4(0, 0, 0, 0);
            if (pmask) {
                uint32_t voff[RG];  // byte offset of this thread's vec in a pair's row, per round of the group
#pragma unroll
                for (int i = 0; i < RG; ++i)  // rows past `rounds` re-read the last real round (result unused)
                    voff[i] = ((uint32_t)min(r0 + i, rounds - 1) * NT + tid) * 16u;
                auto load_rows = [&](uint4 (&x)[RG], uint32_t p) {
#pragma unroll
                    for (int i = 0; i < RG; ++i) x[i] = buf_load16(rs_dense, voff[i], p * (uint32_t)(TILE_DOCS * 4));
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
                uint32_t m = pmask;
                uint4 xa[RG], xb[RG];
                uint32_t pa = (uint32_t)__builtin_ctz(m), pb = 0;
                m &= m - 1;
                load_rows(xa, pa);
                if (r0 == 0) lookup2();
                for (;;) {
                    const bool more_b = m != 0;
                    if (more_b) {
                        pb = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        load_rows(xb, pb);
                    }
                    add_rows(xa, rdl(qv, pa));
                    if (!more_b) break;
                    const bool more_a = m != 0;
                    if (more_a) {
                        pa = (uint32_t)__builtin_ctz(m);
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
        if (!pmask) lookup2();  // (no dense rows were requested)
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
                seg_start[tid] = s0 - tile_first;  // vec index inside the tile
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
            // (written without a loop: hipcc vectorises `for (w < wave) off += wsum[w]` into a 32-wide LDS sweep)
            const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
            const uint32_t off = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
            pref[tid] = nch + off;
            if (tid == kQtBlock - 1) pref[kQtBlock] = w0 + w1 + w2 + w3;
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
            // Branch-free lower-bound search, the same (wave-uniform) number of steps in every lane: largest lo with
            // pref[lo] <= my_c. Entries past `cnt` hold the round's total (> every chunk index), and lanes past c_end
            // search for a chunk that does not exist — their reads stay inside pref[] and their m_n is forced to 0.
            const uint32_t my_c = wave + my_i * NW;
            uint32_t lo = 0;
            // first step = largest power of two below cnt (lo + step >= cnt can never be taken); none when cnt == 1
            for (uint32_t step = cnt > 1 ? 1u << (31 - __clz((int)(cnt - 1))) : 0u; step > 0; step >>= 1) {
                const uint32_t mid = lo + step;
                lo = pref[mid] <= my_c ? mid : lo;
            }
            // (byte units: the loads below take them as scalar offset / clamp without further arithmetic)
            const uint32_t voff = (my_c - pref[lo]) * kChunkVecs;
            const uint32_t m_b16 = my_i < c_end ? (seg_start[lo] + voff) << 4 : 0u;  // idle slots: the tile's first vec
            const uint32_t m_n16 = my_i < c_end ? min((uint32_t)kChunkVecs, seg_len[lo] - voff) << 4 : 0u;
            const uint32_t m_w = seg_w[lo];
            stamp(7);  // lane-parallel chunk resolution (binary search)
            if (DBG && (a.dbg & 256u)) break;  // ablation: stop after the first chunk resolution
            const uint32_t nchunk = min(64u, c_end - cb);
            // Software pipeline, two register banks of U chunks: the next bank's 1-KiB loads are in flight while
            // the current bank's LDS atomics issue. Loads are unconditional (lanes past a chunk's end, and chunk
            // slots past nchunk, re-read the chunk's / the shard's first vec) so that the compiler can count them
            // with s_waitcnt vmcnt(N) instead of draining to vmcnt(0); only the atomics are predicated.
            // (a chunk's byte length stays in an SGPR from its load to its atomics: one v_readlane less per chunk)
            auto load_bank = [&](uint4 (&v)[U], uint32_t (&sn16)[U], uint32_t u0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t idx = u0 + u;  // < 64: u0 + U <= nchunk rounded up to a multiple of U
                    const uint32_t b16 = rdl(m_b16, idx);
                    const uint32_t n16 = rdl(m_n16, idx);
                    sn16[u] = n16;
                    if (DBG && (a.dbg & 2u)) {  // ablation: no global loads, synthetic postings
                        const uint32_t hsh = ((cb + idx) * 64u + lane) * 2654435761u;
                        v[u] = make_uint4((1u << 16) | (hsh >> 17), (1u << 16) | ((hsh * 31u) >> 17),
                                          (1u << 16) | ((hsh * 131u) >> 17), (1u << 16) | ((hsh * 1031u) >> 17));
                    } else {
                        // The chunk's END offset goes into the resource's size word: the buffer range check compares
                        // scalar + lane offset with it, so lanes past the chunk's end — and every lane of an idle
                        // slot — get zeros without a memory request, with no clamp instruction; the lane offset is
                        // the loop-invariant lane16.
                        v[u] = buf_load16(make_rsrc(post_base, b16 + n16), lane16, b16);
                    }
                }
            };
            auto add_bank = [&](const uint4 (&v)[U], const uint32_t (&sn16)[U], uint32_t u0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t idx = u0 + u;
                    const uint32_t n16 = sn16[u];  // 0 in the slots past nchunk
                    const uint32_t w = rdl(m_w, idx);
                    // lanes past the chunk's last vec must not touch LDS (64 lanes adding to one accumulator would
                    // serialise); padding INSIDE a vec has weight 0 and a lane-distinct ordinal
                    if (lane16 < n16) {
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
            uint32_t na[U], nb[U];
            load_bank(va, na, 0);
            for (uint32_t u0 = 0; u0 < nchunk; u0 += 2 * U) {
                const bool more = u0 + U < nchunk;  // wave-uniform
                if (more) load_bank(vb, nb, u0 + U);
                add_bank(va, na, u0);
                if (more) {
                    if (u0 + 2 * U < nchunk) load_bank(va, na, u0 + 2 * U);
                    add_bank(vb, nb, u0 + U);
                }
            }
        }
    }
    stamp(2);  // wave 0's own streaming
    __syncthreads();  // accumulation complete; the staging view of the union is dead from here on
}


// ------------------------------------------------------------------------------------------------ light queries
// The same accumulation for batches whose queries hold at most 64 sparse (non dense-head) terms — the shape of the
// headline workload (Flickr30K captions: 8-15 terms, a handful of one-chunk segments per tile). The general path above
// spends most of such a workgroup's life on machinery sized for 256-term rounds: segment table in LDS, a cross-wave
// scan, two barriers, a 64-lane binary search per wave — and only then requests the first postings, whose L2 round trip
// is fully exposed. Here EVERY wave keeps the query's segment table in registers (lane j = term j: redundant loads,
// no LDS, no barrier), scans the chunk counts with one DPP scan, resolves its own chunks (chunk c -> wave c % NW) with
// a ballot + four v_readlane each, and has its first UL chunk loads IN FLIGHT before the accumulators are initialised,
// so the posting latency hides behind the dense-head / zeroing phase. Two barriers in all (accumulators initialised;
// accumulation complete).
template <int TILE_DOCS, int NT, class Stamp>
__device__ __forceinline__ void accumulate_tile_light(const ScoreArgs& a, const uint32_t q, const uint32_t tile_l,
                                                      const int rounds, uint8_t* const lds, SelectScratch& ss,
                                                      Stamp stamp, const uint32_t tid) {
    constexpr int NW = NT / 64;
    constexpr int UL = 4;  // chunk loads per wave and batch
    static_assert(TILE_DOCS % (4 * NT) == 0, "tile must be a multiple of 4*NT");
    uint32_t* const acc = reinterpret_cast<uint32_t*>(lds);
    const uint32_t lane = tid & 63;
    const uint32_t lane16 = lane * 16u;
    const uint32_t wave = rfl(tid >> 6);
    uint4* const a4 = reinterpret_cast<uint4*>(acc);

    const uint4 meta = a.q_meta[q];  // {first term, end term, dense-head pair mask}
    const uint32_t qb = meta.x, cnt = min(meta.y - meta.x, 64u);  // (the host picks this path only when every query fits)
    const uint32_t* seg_row = a.seg_ptr + (uint64_t)tile_l * (a.n_terms + 1);
    const uint32_t tile_first = seg_row[0];
    const char* const post_base = reinterpret_cast<const char*>(a.postings) + (uint64_t)(tile_first - a.vec_base) * 16u;
    const bool has = lane < cnt;
    uint32_t s_t = 0, s_w = 0;
    if (has) {
        s_t = a.q_term[qb + lane];
        s_w = a.q_w[qb + lane];
    }
    // chunk data of this wave's first UL chunks: requested inside the init phase, consumed after it
    uint4 v[UL];
    uint32_t n16[UL], wq[UL];
    uint32_t s_start = 0, s_len = 0, pref = 0, count_w = 0;
    auto resolve_and_load = [&](uint32_t u0) {  // chunks u0 .. u0+UL-1 of this wave (chunk index c = wave + u * NW)
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            n16[u] = 0;
            v[u] = make_uint4(0, 0, 0, 0);
            if (u0 + (uint32_t)u < count_w) {  // wave-uniform: most waves of a light query hold one or two chunks
                const uint32_t c = wave + (u0 + (uint32_t)u) * NW;
                // pref is non-decreasing over the lanes: the lanes with pref <= c are a prefix, its last lane owns chunk c
                const uint32_t lo = (uint32_t)__popcll(__ballot(has && pref <= c)) - 1u;
                const uint32_t voff = (c - rdl(pref, lo)) << 6;  // vecs into the segment
                const uint32_t b16 = (rdl(s_start, lo) + voff) << 4;
                n16[u] = min(64u, rdl(s_len, lo) - voff) << 4;
                wq[u] = rdl(s_w, lo);
                // (the chunk's END in the resource's size word: lanes past it read zeros without a request)
                v[u] = buf_load16(make_rsrc(post_base, b16 + n16[u]), lane16, b16);
            }
        }
    };
    auto add_loaded = [&]() {
#pragma unroll
        for (int u = 0; u < UL; ++u)
            if (n16[u] != 0 && lane16 < n16[u]) {  // (n16 is wave-uniform: an idle slot costs one scalar compare)
                const uint32_t p[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&acc[p[e] & 0xFFFFu], __umul24(p[e] >> 16, wq[u]));
            }
    };
    auto prefetch = [&]() {  // segment table, chunk scan, first chunk loads: all per wave, in registers
        if (has) {
            const uint32_t s0 = seg_row[s_t], s1 = seg_row[s_t + 1];
            s_start = s0 - tile_first;
            s_len = s1 - s0;
        }
        const uint32_t nch = (s_len + (uint32_t)kChunkVecs - 1) / (uint32_t)kChunkVecs;
        const uint32_t inc = wave_inclusive_scan_u32(nch);
        pref = inc - nch;
        const uint32_t total = rdl(inc, 63);
        count_w = total > wave ? (total - wave + NW - 1) / NW : 0u;
        resolve_and_load(0);
    };

    // ---- accumulator init (dense-head rows scored by the accumulator's owner, or zero): as in accumulate_tile, with
    // the sparse prefetch issued right after the first rows have been requested
    {
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        constexpr int RG = 4;
        const uint32_t qv = lane < a.n_pairs ? a.q_dense[(uint64_t)q * a.n_pairs + lane] : 0u;
        const uint32_t pmask = meta.z;
        const __amdgpu_buffer_rsrc_t rs_dense =
            make_rsrc(reinterpret_cast<const char*>(a.dense) + (uint64_t)tile_l * a.n_pairs * (TILE_DOCS * 4u),
                      a.n_pairs * (uint32_t)(TILE_DOCS * 4));
        for (int r0 = 0; r0 < rounds; r0 += RG) {
            uint4 sacc[RG];
#pragma unroll
            for (int i = 0; i < RG; ++i) sacc[i] = make_uint4(0, 0, 0, 0);
            if (pmask) {
                uint32_t voff[RG];
#pragma unroll
                for (int i = 0; i < RG; ++i) voff[i] = ((uint32_t)min(r0 + i, rounds - 1) * NT + tid) * 16u;
                auto load_rows = [&](uint4 (&x)[RG], uint32_t p) {
#pragma unroll
                    for (int i = 0; i < RG; ++i) x[i] = buf_load16(rs_dense, voff[i], p * (uint32_t)(TILE_DOCS * 4));
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
                uint32_t m = pmask;
                uint4 xa[RG];
                for (;;) {  // one pair at a time (light queries hold one or two): the register banks go to the chunks
                    const uint32_t pa = (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    load_rows(xa, pa);
                    if (r0 == 0 && pa == (uint32_t)__builtin_ctz(pmask)) prefetch();
                    add_rows(xa, rdl(qv, pa));
                    if (!m) break;
                }
            }
#pragma unroll
            for (int i = 0; i < RG; ++i)
                if (r0 + i < rounds) a4[(r0 + i) * NT + tid] = sacc[i];
        }
        if (!pmask) prefetch();
    }
    if (tid < 64) ss.cnt[tid] = 0;
    if (tid == 0) {
        ss.n_cand = 0;
        ss.tau0 = 1;
        ss.smax = 0;
    }
    stamp(0);
    __syncthreads();  // every accumulator is initialised
    add_loaded();
    for (uint32_t u0 = UL; u0 < count_w; u0 += UL) {  // (rare on this path: more than UL chunks for one wave)
        resolve_and_load(u0);
        add_loaded();
    }
    stamp(2);
    __syncthreads();  // accumulation complete
}

}  // namespace msr
