// Exact "k best of a tile" THRESHOLD for elements held in registers (hybrid_tiles, msr_hybrid.hip).
//
// Every thread owns E = TILE_DOCS / NT u32 keys (higher is better, 0 = absent); element j of thread `tid` is tile doc
// local(j) = 4 * ((j / 4) * NT + tid) + (j % 4) — the accumulator tile's ownership map. Ties in the key go to the lower
// doc, so the order is that of the unique composite  c = key << 13 | (TILE_DOCS - 1 - local).  hist_threshold returns
// the composite T of the k-th best element: an element belongs to the k best  <=>  key != 0 and c >= T  (T = 1 when
// at most k elements are present: all of them belong).
//
// Method: one histogram pass instead of per-bit bisection or byte-wise radix passes. The elements are binned LINEARLY
// between the smallest and the largest present value (1024 bins; the position function pos0(key, smallest key) of the first level is
// the caller's — for float scores it must be linear in the VALUE, not in the bit pattern, or everything lands in the few
// bins of the top exponents), a suffix scan of the bins finds the bin that holds the k-th best, and only that bin's
// elements (about n / 1024 x a few) are ranked exactly. A bin that still holds more than kHistCand elements (mass
// ties) is split again, linearly in the composite, until it does not: at most 5 levels, one in practice.
#pragma once

// (included by msr_select.hpp, after its cross-lane helpers and before tile_select, which uses it for large k)
namespace msr {

constexpr int kHistBins = 1024;
constexpr int kHistCand = 512;  // candidates ranked exactly (u64 keys in LDS)

struct HistScratch {
    uint32_t wa[16], wb[16], wc[16];  // per-wave partials (max / min / count, bin totals)
    uint64_t w64a[16], w64b[16];      // per-wave composite min / max
    uint32_t bin, above, cnt, n_cand;
    uint64_t T;
};

struct HistResult {
    uint64_t T;       // threshold composite (1: every present element belongs)
    uint32_t n;       // present elements
    uint32_t kmax;    // largest present key (0 when n == 0)
    uint32_t kmin;    // smallest present key
};

// get4(r) returns the four keys of this thread's vec in round r (docs 4 * (r * NT + tid) .. + 3); it is called again
// in every pass (keys that live in LDS are simply re-read: nothing but the 16-bit `active` mask is held across passes).
template <int TILE_DOCS, int NT, class Get4, class Pos0>
__device__ __forceinline__ HistResult hist_threshold(Get4 get4, const uint32_t k, uint32_t* hist /* [kHistBins] */,
                                                     uint64_t* cand /* [kHistCand] */, HistScratch& hs,
                                                     const uint32_t tid, Pos0 pos0) {
    constexpr int R = TILE_DOCS / (4 * NT);
    constexpr int NW = NT / 64;
    constexpr int BPT = kHistBins / NT;  // bins per thread in the scan
    static_assert(TILE_DOCS <= 8192, "the composite keeps 13 bits for the doc");
    static_assert(kHistBins % NT == 0 && NW <= 16 && R * 4 <= 32, "scan layout / active mask");
    const uint32_t lane = tid & 63, wave = rfl(tid >> 6);
    auto comp = [&](uint32_t key, int r, int e) -> uint64_t {
        const uint32_t local = 4u * ((uint32_t)r * NT + tid) + (uint32_t)e;
        return ((uint64_t)key << 13) | (uint64_t)(TILE_DOCS - 1 - local);
    };
    // visit(f): f(key, r, e, j) for the 4 * R elements of this thread, j = 4 * r + e
    auto visit = [&](auto f) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint4 x = get4(r);
            f(x.x, r, 0, 4 * r + 0);
            f(x.y, r, 1, 4 * r + 1);
            f(x.z, r, 2, 4 * r + 2);
            f(x.w, r, 3, 4 * r + 3);
        }
    };
    HistResult res;
    uint32_t active = 0;  // bit j: element j is present / still a contender for the k-th place
    // ---- present elements: count, largest and smallest key
    {
        uint32_t mx = 0, mn = 0xFFFFFFFFu;
        visit([&](uint32_t key, int, int, int j) {
            mx = max(mx, key);
            mn = min(mn, key ? key : 0xFFFFFFFFu);
            active |= (uint32_t)(key != 0) << j;
        });
        uint32_t c = (uint32_t)__popc(active);
        mx = wave_max_u32(mx);
        mn = ~wave_max_u32(~mn);
        c = wave_sum_u32(c);
        __syncthreads();  // previous users of the scratch are done
        if (lane == 0) {
            hs.wa[wave] = mx;
            hs.wb[wave] = mn;
            hs.wc[wave] = c;
        }
        if (tid == 0) hs.n_cand = 0;
        __syncthreads();
        mx = 0, mn = 0xFFFFFFFFu, c = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            mx = max(mx, hs.wa[w]);
            mn = min(mn, hs.wb[w]);
            c += hs.wc[w];
        }
        res.n = c;
        res.kmax = mx;
        res.kmin = c ? mn : 0u;
    }
    res.T = 1;
    if (res.n <= k) return res;  // (uniform) fewer present elements than k: all belong

    uint32_t need = k;  // the wanted element is the need-th best among the active ones
    bool level0 = res.kmax != res.kmin;
    float scale = 0.f;
    uint64_t lo64 = 0;
    uint32_t shift = 0;
    if (level0) {
        const float top = pos0(res.kmax, res.kmin);  // pos0(kmin, kmin) == 0 by contract
        level0 = top > 0.f;
        scale = ((float)kHistBins - 0.5f) / top;
    }
#pragma unroll 1
    for (int level = 0; level < 8; ++level) {
        if (!level0) {
            // linear in the composite: bin = ((c - lo) >> shift) * scale, with (c - lo) >> shift < 2^31
            uint64_t mn = ~0ull, mx = 0;
            visit([&](uint32_t key, int r, int e, int j) {
                if (active >> j & 1u) {
                    const uint64_t c = comp(key, r, e);
                    mn = c < mn ? c : mn;
                    mx = c > mx ? c : mx;
                }
            });
            mn = ~wave_max_u64(~mn);
            mx = wave_max_u64(mx);
            __syncthreads();
            if (lane == 0) {
                hs.w64a[wave] = mn;
                hs.w64b[wave] = mx;
            }
            __syncthreads();
            mn = ~0ull, mx = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                mn = hs.w64a[w] < mn ? hs.w64a[w] : mn;
                mx = hs.w64b[w] > mx ? hs.w64b[w] : mx;
            }
            lo64 = mn;
            const uint64_t span = mx - mn;  // > 0: composites are unique and at least two are active
            shift = span >> 31 ? (uint32_t)(64 - __clzll((long long)span) - 31) : 0u;
            scale = ((float)kHistBins - 0.5f) / (float)(uint32_t)(span >> shift);
        }
        auto bin_of = [&](uint32_t key, int r, int e) -> uint32_t {
            const float p = level0 ? pos0(key, res.kmin) : (float)(uint32_t)((comp(key, r, e) - lo64) >> shift);
            return min((uint32_t)(kHistBins - 1), (uint32_t)(p * scale));
        };
        // ---- histogram of the active elements
        for (int i = tid; i < kHistBins / 4; i += NT) reinterpret_cast<uint4*>(hist)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        visit([&](uint32_t key, int r, int e, int j) {
            if (active >> j & 1u) atomicAdd(&hist[bin_of(key, r, e)], 1u);
        });
        __syncthreads();
        // ---- the bin of the need-th best: thread t owns bins kHistBins-1 - BPT*t .. (descending), so a prefix sum in
        // thread order counts the elements in HIGHER bins
        uint32_t hb[BPT], s = 0;
#pragma unroll
        for (int b = 0; b < BPT; ++b) {
            hb[b] = hist[kHistBins - 1 - (BPT * (int)tid + b)];
            s += hb[b];
        }
        const uint32_t inc = wave_inclusive_scan_u32(s);
        if (lane == 63) hs.wa[wave] = inc;
        __syncthreads();
        uint32_t above = inc - s;
#pragma unroll
        for (int w = 0; w < NW; ++w) above += (uint32_t)w < wave ? hs.wa[w] : 0u;
        if (above < need && need <= above + s) {  // exactly one thread
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
            hs.bin = (uint32_t)(kHistBins - 1) - (BPT * tid + b);
            hs.above = ab;
            hs.cnt = cnt;
        }
        __syncthreads();
        const uint32_t bsel = hs.bin, cnt = hs.cnt;
        need -= hs.above;  // rank inside the chosen bin, 1-based from the top
        if (cnt <= (uint32_t)kHistCand) {
            // ---- rank the bin's elements exactly
            visit([&](uint32_t key, int r, int e, int j) {
                if ((active >> j & 1u) && bin_of(key, r, e) == bsel) cand[atomicAdd(&hs.n_cand, 1u)] = comp(key, r, e);
            });
            __syncthreads();
            if (cnt <= 64) {
                if (tid < 64) {
                    const uint64_t me = lane < cnt ? cand[lane] : 0ull;
                    const uint32_t lo = (uint32_t)me, hi = (uint32_t)(me >> 32);
                    uint32_t rank = 0;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const uint64_t o = ((uint64_t)rdl(hi, i) << 32) | rdl(lo, i);
                        rank += o > me;
                    }
                    if (lane < cnt && rank == need - 1) hs.T = me;
                }
            } else {
                for (uint32_t i = tid; i < cnt; i += NT) {
                    const uint64_t me = cand[i];
                    uint32_t rank = 0;
                    for (uint32_t o = 0; o < cnt; ++o) rank += cand[o] > me;
                    if (rank == need - 1) hs.T = me;
                }
            }
            __syncthreads();
            res.T = hs.T;
            return res;
        }
        // ---- still too many in one bin (mass ties): keep only that bin's elements and split it again
        uint32_t keep = 0;
        visit([&](uint32_t key, int r, int e, int j) {
            if ((active >> j & 1u) && bin_of(key, r, e) == bsel) keep |= 1u << j;
        });
        active = keep;
        level0 = false;
    }
    return res;  // not reached: every level shrinks the span by ~2^10 and composites are unique
}

}  // namespace msr
