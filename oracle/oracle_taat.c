/*
 * oracle_taat.c — plain-C restatement of the reference's CPU sparse search.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product library
 * (libmsr.so) never does.  PARITY UNPINNED: the reference's scorer is Lucene behind pyserini
 * (src/search.py:86-87,273-275), which is neither vendored nor runnable here; this file restates the declared
 * contract of SURVEY.md §8a A3 / §8c T1-T5 and is cross-checked against oracle/oracle.py (scipy int64) and the
 * committed fixtures under tests/golden/.
 *
 * What Lucene's impact search does for src/search.py:86-87, restated term-at-a-time:
 *   score(d) = sum over query terms t of q_w(t) * tf(t, d)     (ImpactSimilarity, no norms)
 *   hits     = docs with score > 0, best k by (score desc, doc ordinal asc); ordinals are ranks of the external
 *              doc-id strings, so "ordinal asc" is "doc id string asc" (T1)
 *   query    = OOV terms ignored (T2), terms with df == N dropped when asked (T3), repeated terms add (A4)
 * Threading mirrors Anserini's batch_search: a pool of `threads` workers, one query per task.
 *
 * This is also the CPU baseline timed by bench.py ("kind": "port"): a multithreaded exhaustive TAAT scorer with
 * u32 accumulators and a per-query heap — labelled as such, never as "Lucene". Two more rows of that baseline come from
 * the same file: mode 1 clears only the docs a query touched (the row that uses every core), mode 2 is document-at-a-time
 * MaxScore ("port+pruning"): what a pruning engine like Lucene saves on this workload, verified equal to mode 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct otaat_index {
    uint64_t n_docs;
    uint32_t n_terms;
    uint64_t* term_ptr; /* [n_terms+1] */
    uint32_t* post_doc; /* doc ordinals, ascending inside a term */
    uint32_t* post_w;   /* tf */
    uint32_t* df;
    uint32_t* maxw;
} otaat_index;

void otaat_free(otaat_index* ix) {
    if (!ix) return;
    free(ix->term_ptr);
    free(ix->post_doc);
    free(ix->post_w);
    free(ix->df);
    free(ix->maxw);
    free(ix);
}

/* Rows of the doc-major CSR must already be in doc-ordinal order. A term repeated inside a row adds (tf counts). */
otaat_index* otaat_build(uint64_t n_docs, uint32_t n_terms, const uint64_t* doc_ptr, const uint32_t* term,
                         const uint32_t* weight) {
    otaat_index* ix = (otaat_index*)calloc(1, sizeof(*ix));
    if (!ix) return NULL;
    ix->n_docs = n_docs;
    ix->n_terms = n_terms;
    ix->term_ptr = (uint64_t*)calloc((size_t)n_terms + 2, sizeof(uint64_t));
    ix->df = (uint32_t*)calloc((size_t)n_terms + 1, sizeof(uint32_t));
    ix->maxw = (uint32_t*)calloc((size_t)n_terms + 1, sizeof(uint32_t));
    uint32_t* last_doc = (uint32_t*)malloc(((size_t)n_terms + 1) * sizeof(uint32_t));
    if (!ix->term_ptr || !ix->df || !ix->maxw || !last_doc) goto fail;
    /* pass 1: df (distinct docs per term) */
    memset(last_doc, 0xFF, ((size_t)n_terms + 1) * sizeof(uint32_t));
    for (uint64_t d = 0; d < n_docs; ++d)
        for (uint64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
            if (weight[i] == 0 || term[i] >= n_terms) continue;
            if (last_doc[term[i]] != (uint32_t)d) {
                last_doc[term[i]] = (uint32_t)d;
                ix->df[term[i]]++;
            }
        }
    for (uint32_t t = 0; t < n_terms; ++t) ix->term_ptr[t + 1] = ix->term_ptr[t] + ix->df[t];
    {
        uint64_t np = ix->term_ptr[n_terms];
        ix->post_doc = (uint32_t*)malloc((np ? np : 1) * sizeof(uint32_t));
        ix->post_w = (uint32_t*)calloc(np ? np : 1, sizeof(uint32_t));
        if (!ix->post_doc || !ix->post_w) goto fail;
    }
    /* pass 2: fill; a repeated term inside a doc lands on the same posting */
    {
        uint64_t* cur = (uint64_t*)malloc(((size_t)n_terms + 1) * sizeof(uint64_t));
        if (!cur) goto fail;
        memcpy(cur, ix->term_ptr, ((size_t)n_terms + 1) * sizeof(uint64_t));
        memset(last_doc, 0xFF, ((size_t)n_terms + 1) * sizeof(uint32_t));
        for (uint64_t d = 0; d < n_docs; ++d)
            for (uint64_t i = doc_ptr[d]; i < doc_ptr[d + 1]; ++i) {
                uint32_t t = term[i];
                if (weight[i] == 0 || t >= n_terms) continue;
                if (last_doc[t] != (uint32_t)d) {
                    last_doc[t] = (uint32_t)d;
                    ix->post_doc[cur[t]++] = (uint32_t)d;
                }
                uint64_t p = cur[t] - 1;
                ix->post_w[p] += weight[i];
                if (ix->post_w[p] > ix->maxw[t]) ix->maxw[t] = ix->post_w[p];
            }
        free(cur);
    }
    free(last_doc);
    return ix;
fail:
    free(last_doc);
    otaat_free(ix);
    return NULL;
}

typedef struct {
    uint32_t score;
    uint32_t ord;
} hit_t;

/* a "worse" hit sits at the heap root: lower score, or equal score and higher ordinal */
static int worse(hit_t a, hit_t b) { return a.score < b.score || (a.score == b.score && a.ord > b.ord); }

static void sift_down(hit_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && worse(h[l], h[m])) m = l;
        if (r < n && worse(h[r], h[m])) m = r;
        if (m == i) return;
        hit_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}

typedef struct {
    const otaat_index* ix;
    const int64_t* q_ptr;
    const int32_t* q_term;
    const int32_t* q_w;
    int nq, k, drop;
    int mode; /* 0: exhaustive TAAT, full accumulator scan (the checker); 1: the same, but only the docs a query touched
                 are scanned and cleared (a worker's N-sized accumulator array then costs nothing per query: this is the
                 row that scales to every core); 2: document-at-a-time MaxScore (below) */
    int64_t* out_ord;
    int64_t* out_score;
    int32_t* out_n;
    int* next;  /* shared query counter */
    int* error; /* set to 1 on overflow / allocation failure */
} job_t;

static void emit_heap(job_t* j, int q, hit_t* heap, int hn) {
    /* heap -> best-first order */
    if (hn < j->k)
        for (int i = hn / 2 - 1; i >= 0; --i) sift_down(heap, hn, i);
    int64_t* oo = j->out_ord + (int64_t)q * j->k;
    int64_t* os = j->out_score + (int64_t)q * j->k;
    for (int i = 0; i < j->k; ++i) {
        oo[i] = -1;
        os[i] = 0;
    }
    j->out_n[q] = hn;
    for (int n = hn; n > 0; --n) { /* pop the worst into slot n-1 */
        oo[n - 1] = heap[0].ord;
        os[n - 1] = heap[0].score;
        heap[0] = heap[n - 1];
        sift_down(heap, n - 1, 0);
    }
}

static void offer(hit_t* heap, int* hn, int k, hit_t h) {
    if (*hn < k) {
        heap[(*hn)++] = h;
        if (*hn == k)
            for (int i = *hn / 2 - 1; i >= 0; --i) sift_down(heap, *hn, i);
    } else if (worse(heap[0], h)) {
        heap[0] = h;
        sift_down(heap, *hn, 0);
    }
}

/* ---- mode 2: document-at-a-time MaxScore (Turtle & Flood 1995; the strategy of Lucene's MaxScoreBulkScorer for pure
 * disjunctions, which is what a BooleanQuery of SHOULD BoostQuery(TermQuery) clauses is) — the
 * PRUNING baseline row: same results as the exhaustive scorer (tests/test_oracle.py), fewer postings scored.
 * Terms sorted by upper bound ub(t) = q_w(t) * max tf(t); lists whose bounds sum to <= theta (the k-th best score so
 * far) are non-essential: a doc that occurs only in them cannot enter the top-k (docs are visited in ascending
 * ordinal order, so an EQUAL score never displaces a hit: "<=" is exact under the tie rule T1). Candidate docs come
 * from the essential lists; a candidate's non-essential lists are probed from the largest bound down, stopping as soon
 * as score + remaining bounds <= theta. The essential lists are scored term at a time inside windows of 4096 docs (as
 * Lucene's bulk scorer does), so the accumulators of a window never leave the L1 cache. */
typedef struct {
    const uint32_t* pd;
    const uint32_t* pw;
    uint64_t n, at;
    uint32_t qw;
    uint64_t ub;
} cursor_t;

static int by_ub(const void* a, const void* b) {
    const cursor_t *x = (const cursor_t*)a, *y = (const cursor_t*)b;
    return x->ub < y->ub ? -1 : (x->ub > y->ub ? 1 : 0);
}

/* first position >= at whose doc is >= d (galloping + binary search) */
static uint64_t advance_to(const cursor_t* c, uint32_t d) {
    uint64_t lo = c->at, step = 1;
    if (lo >= c->n || c->pd[lo] >= d) return lo;
    while (lo + step < c->n && c->pd[lo + step] < d) {
        lo += step;
        step <<= 1;
    }
    uint64_t hi = lo + step < c->n ? lo + step : c->n; /* pd[lo] < d, (hi == n or pd[hi] >= d) */
    while (hi - lo > 1) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (c->pd[mid] < d)
            lo = mid;
        else
            hi = mid;
    }
    return hi;
}

#define OT_WINDOW 4096 /* docs per scoring window (Lucene's MaxScoreBulkScorer uses the same inner window) */

static void maxscore_query(job_t* j, int q, cursor_t* cur, uint64_t* prefix, hit_t* heap, uint32_t* wacc) {
    const otaat_index* ix = j->ix;
    const uint64_t N = ix->n_docs;
    int m = 0;
    for (int64_t e = j->q_ptr[q]; e < j->q_ptr[q + 1]; ++e) {
        int32_t t = j->q_term[e], w = j->q_w[e];
        if (t < 0 || (uint32_t)t >= ix->n_terms || w <= 0) continue;
        if (ix->df[t] == 0 || (j->drop && ix->df[t] == N)) continue;
        int dup = -1; /* a repeated query term adds its weights (A4): one cursor per distinct term */
        for (int i = 0; i < m; ++i)
            if (cur[i].pd == ix->post_doc + ix->term_ptr[t]) dup = i;
        if (dup >= 0) {
            cur[dup].qw += (uint32_t)w;
            cur[dup].ub += (uint64_t)w * ix->maxw[t];
            continue;
        }
        cur[m].pd = ix->post_doc + ix->term_ptr[t];
        cur[m].pw = ix->post_w + ix->term_ptr[t];
        cur[m].n = ix->term_ptr[t + 1] - ix->term_ptr[t];
        cur[m].at = 0;
        cur[m].qw = (uint32_t)w;
        cur[m].ub = (uint64_t)w * ix->maxw[t];
        m++;
    }
    qsort(cur, (size_t)m, sizeof(cursor_t), by_ub); /* ascending bound */
    for (int i = 0; i < m; ++i) prefix[i] = (i ? prefix[i - 1] : 0) + cur[i].ub;
    int hn = 0, first_ess = 0; /* lists [first_ess, m) are essential */
    uint64_t theta = 0;        /* k-th best score once the heap is full */
    for (;;) {
        /* next window: the one that holds the smallest current doc of the essential lists */
        uint32_t dmin = 0xFFFFFFFFu;
        for (int i = first_ess; i < m; ++i)
            if (cur[i].at < cur[i].n && cur[i].pd[cur[i].at] < dmin) dmin = cur[i].pd[cur[i].at];
        if (dmin == 0xFFFFFFFFu) break;
        const uint32_t w0 = dmin / OT_WINDOW * OT_WINDOW;
        const uint64_t w1 = (uint64_t)w0 + OT_WINDOW;
        /* essential lists, term at a time inside the window (the accumulators stay in the L1 cache) */
        for (int i = first_ess; i < m; ++i) {
            uint64_t at = cur[i].at;
            const uint32_t qw = cur[i].qw;
            while (at < cur[i].n && cur[i].pd[at] < w1) {
                wacc[cur[i].pd[at] - w0] += qw * cur[i].pw[at];
                at++;
            }
            cur[i].at = at;
        }
        /* candidates of the window in ascending ordinal order; non-essential lists probed from the largest bound down */
        const uint64_t rest = first_ess ? prefix[first_ess - 1] : 0;
        for (uint32_t x = 0; x < OT_WINDOW; ++x) {
            uint64_t score = wacc[x];
            if (!score) continue;
            wacc[x] = 0;
            if (hn == j->k && score + rest <= theta) continue; /* cannot beat the k-th hit */
            const uint32_t d = w0 + x;
            for (int i = first_ess - 1; i >= 0; --i) {
                if (hn == j->k && score + prefix[i] <= theta) break;
                cur[i].at = advance_to(&cur[i], d);
                if (cur[i].at < cur[i].n && cur[i].pd[cur[i].at] == d) score += (uint64_t)cur[i].qw * cur[i].pw[cur[i].at];
            }
            if (hn < j->k || score > theta) {
                hit_t h = {(uint32_t)score, d};
                offer(heap, &hn, j->k, h);
                if (hn == j->k) theta = heap[0].score;
            }
        }
        while (first_ess < m && hn == j->k && prefix[first_ess] <= theta) first_ess++;
    }
    emit_heap(j, q, heap, hn);
}

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    const otaat_index* ix = j->ix;
    const uint64_t N = ix->n_docs;
    if (j->mode == 2) {
        cursor_t* cur = NULL;
        uint64_t* prefix = NULL;
        hit_t* heap2 = (hit_t*)malloc((size_t)(j->k > 0 ? j->k : 1) * sizeof(hit_t));
        uint32_t* wacc = (uint32_t*)calloc(OT_WINDOW, sizeof(uint32_t));
        int cap = 0;
        for (;;) {
            int q = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
            if (q >= j->nq || !heap2 || !wacc) break;
            int len = (int)(j->q_ptr[q + 1] - j->q_ptr[q]);
            if (len > cap) {
                cap = len + 64;
                free(cur);
                free(prefix);
                cur = (cursor_t*)malloc((size_t)cap * sizeof(cursor_t));
                prefix = (uint64_t*)malloc((size_t)cap * sizeof(uint64_t));
            }
            uint64_t bound = 0;
            for (int64_t e = j->q_ptr[q]; e < j->q_ptr[q + 1]; ++e) {
                int32_t t = j->q_term[e], w = j->q_w[e];
                if (t < 0 || (uint32_t)t >= ix->n_terms || w <= 0) continue;
                if (ix->df[t] == 0 || (j->drop && ix->df[t] == N)) continue;
                bound += (uint64_t)w * ix->maxw[t];
            }
            if (bound > 0xFFFFFFFFull || (len && (!cur || !prefix))) {
                __atomic_store_n(j->error, 1, __ATOMIC_RELAXED);
                break;
            }
            maxscore_query(j, q, cur, prefix, heap2, wacc);
        }
        if (!heap2 || !wacc) __atomic_store_n(j->error, 1, __ATOMIC_RELAXED);
        free(wacc);
        free(cur);
        free(prefix);
        free(heap2);
        return NULL;
    }
    uint32_t* touched = j->mode == 1 ? (uint32_t*)malloc((N ? N : 1) * sizeof(uint32_t)) : NULL;
    uint32_t* acc = (uint32_t*)calloc(N ? N : 1, sizeof(uint32_t));
    hit_t* heap = (hit_t*)malloc((size_t)(j->k > 0 ? j->k : 1) * sizeof(hit_t));
    if (!acc || !heap || (j->mode == 1 && !touched)) {
        __atomic_store_n(j->error, 1, __ATOMIC_RELAXED);
        free(acc);
        free(heap);
        free(touched);
        return NULL;
    }
    for (;;) {
        int q = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (q >= j->nq) break;
        uint64_t bound = 0;
        for (int64_t e = j->q_ptr[q]; e < j->q_ptr[q + 1]; ++e) {
            int32_t t = j->q_term[e], w = j->q_w[e];
            if (t < 0 || (uint32_t)t >= ix->n_terms || w <= 0) continue;
            if (ix->df[t] == 0 || (j->drop && ix->df[t] == N)) continue;
            bound += (uint64_t)w * ix->maxw[t];
        }
        if (bound > 0xFFFFFFFFull) {
            __atomic_store_n(j->error, 1, __ATOMIC_RELAXED);
            break;
        }
        /* term at a time */
        uint64_t n_touched = 0;
        for (int64_t e = j->q_ptr[q]; e < j->q_ptr[q + 1]; ++e) {
            int32_t t = j->q_term[e], w = j->q_w[e];
            if (t < 0 || (uint32_t)t >= ix->n_terms || w <= 0) continue;
            if (ix->df[t] == 0 || (j->drop && ix->df[t] == N)) continue;
            const uint32_t* pd = ix->post_doc + ix->term_ptr[t];
            const uint32_t* pw = ix->post_w + ix->term_ptr[t];
            const uint64_t n = ix->term_ptr[t + 1] - ix->term_ptr[t];
            const uint32_t qw = (uint32_t)w;
            if (touched) { /* remember the docs this query reaches: only they are scanned and cleared */
                for (uint64_t i = 0; i < n; ++i) {
                    const uint32_t d = pd[i];
                    if (!acc[d]) touched[n_touched++] = d;
                    acc[d] += qw * pw[i];
                }
            } else {
                for (uint64_t i = 0; i < n; ++i) acc[pd[i]] += qw * pw[i];
            }
        }
        int hn = 0;
        if (touched) {
            /* (any order: `worse` compares (score, ordinal), so the tie rule does not depend on the scan order) */
            for (uint64_t i = 0; i < n_touched; ++i) {
                hit_t h = {acc[touched[i]], touched[i]};
                offer(heap, &hn, j->k, h);
                acc[touched[i]] = 0;
            }
        } else {
            /* top-k of the positive accumulators; ascending scan, so an equal score never displaces a lower ordinal */
            for (uint64_t d = 0; d < N; ++d) {
                uint32_t s2 = acc[d];
                if (!s2) continue;
                hit_t h = {s2, (uint32_t)d};
                offer(heap, &hn, j->k, h);
            }
            memset(acc, 0, N * sizeof(uint32_t));
        }
        emit_heap(j, q, heap, hn);
    }
    free(acc);
    free(heap);
    free(touched);
    return NULL;
}

/* returns 0, or -1 on overflow of the u32 score range / allocation failure. mode: see job_t. */
int otaat_search_mode(const otaat_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int k,
                      int drop_df_eq_n, int threads, int mode, int64_t* out_ord, int64_t* out_score, int32_t* out_n) {
    if (!ix || nq < 0 || k < 1 || mode < 0 || mode > 2) return -1;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    int next = 0, error = 0;
    job_t job = {ix, q_ptr, q_term, q_w, nq, k, drop_df_eq_n, mode, out_ord, out_score, out_n, &next, &error};
    if (threads == 1) {
        worker(&job);
    } else {
        pthread_t th[256];
        int started = 0;
        for (int t = 0; t < threads; ++t)
            if (pthread_create(&th[t], NULL, worker, &job) == 0)
                started++;
            else
                break;
        if (started == 0) worker(&job);
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    }
    return error ? -1 : 0;
}

int otaat_search(const otaat_index* ix, const int64_t* q_ptr, const int32_t* q_term, const int32_t* q_w, int nq, int k,
                 int drop_df_eq_n, int threads, int64_t* out_ord, int64_t* out_score, int32_t* out_n) {
    return otaat_search_mode(ix, q_ptr, q_term, q_w, nq, k, drop_df_eq_n, threads, 0, out_ord, out_score, out_n);
}

uint32_t otaat_df(const otaat_index* ix, uint32_t t) { return t < ix->n_terms ? ix->df[t] : 0; }
uint64_t otaat_n_postings(const otaat_index* ix) { return ix->term_ptr[ix->n_terms]; }
