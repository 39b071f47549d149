// Synthetic "encode" step: learned-sparse vectors shaped like the reference's encoder output, for machines
// without the 7-8 B MLLM checkpoints. Mimics  rint(100 * log(1 + relu(logit)))  over the top-`nnz` vocabulary
// entries (src/model.py:104, src/encode.py:69-75) with the generator spec of SURVEY.md §8d:
//   term ids   : `nnz` distinct ids per vector, successive draws from p(r) ~ (r+1)^-s over n_terms, repeats rejected
//   weights    : w = clip(max(1, rint(100*ln(1+x))), 1, 400),  x ~ LogNormal(mu=0.5, sigma=0.6)
// Every row depends only on (seed, row), so the output is identical for any thread count.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "msr_internal.h"

namespace {

struct Rng {  // splitmix64 stream
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

inline uint64_t mix(uint64_t a, uint64_t b) {
    Rng r(a ^ (b * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull));
    r.next();
    return r.next();
}

}  // namespace

extern "C" int msr_synth_vectors(uint64_t n, uint32_t nnz, uint32_t n_terms, double zipf_s, uint64_t seed,
                                 int threads, uint64_t* ptr, uint32_t* term, uint32_t* weight) {
    using namespace msr;
    if (!ptr || (n && nnz && (!term || !weight))) {
        set_error("msr_synth_vectors: null output");
        return MSR_E_INVAL;
    }
    if (nnz > n_terms) {
        set_error("msr_synth_vectors: nnz %u exceeds the vocabulary %u", nnz, n_terms);
        return MSR_E_RANGE;
    }
    threads = clamp_threads(threads);
    std::vector<double> cdf(n_terms);
    double acc = 0;
    for (uint32_t r = 0; r < n_terms; ++r) {
        acc += std::pow((double)r + 1.0, -zipf_s);
        cdf[r] = acc;
    }
    for (uint32_t r = 0; r < n_terms; ++r) cdf[r] /= acc;
    for (uint64_t i = 0; i <= n; ++i) ptr[i] = i * nnz;

    parallel_run(threads, [&](int t) {
        std::vector<uint8_t> mark(n_terms, 0);
        std::vector<std::pair<uint32_t, uint32_t>> row(nnz);
        uint64_t a = n * t / threads, b = n * (t + 1) / threads;
        for (uint64_t i = a; i < b; ++i) {
            Rng rng(mix(seed, i));
            uint32_t got = 0;
            while (got < nnz) {
                double u = rng.uniform();
                uint32_t r = (uint32_t)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
                if (r >= n_terms) r = n_terms - 1;
                if (mark[r]) continue;
                mark[r] = 1;
                double x = std::exp(0.5 + 0.6 * rng.normal());
                double w = std::nearbyint(100.0 * std::log1p(x));
                if (w < 1) w = 1;
                if (w > 400) w = 400;
                row[got++] = {r, (uint32_t)w};
            }
            std::sort(row.begin(), row.end());
            for (uint32_t j = 0; j < nnz; ++j) {
                term[i * nnz + j] = row[j].first;
                weight[i * nnz + j] = row[j].second;
                mark[row[j].first] = 0;
            }
        }
    });
    return MSR_OK;
}
