/*
 * sert_cpu.c -- multithreaded CPU restatement of ONE training step of SERT's LSE model
 * (VectorSpaceLanguageModel, sert/models.py:1024-1118: window mean-pool -> tanh projection ->
 * sigmoid NCE against z sampled negatives, dense L2, dense Adam).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (oracle/): the `cpu_baseline` leg of bench.py times
 * it on the GPU node's host cores and tests/test_cpu_baseline.py checks it against the numpy
 * oracle (oracle/sert_oracle.py).  Nothing under sert_amd/ loads it; the product has no CPU path.
 *
 * "restatement, not Theano": the reference executes this arithmetic inside Theano 0.8.2 /
 * Lasagne 0.1 (requirements.txt:3,11), which cannot run here.  This file is what a
 * good hand-written CPU implementation of the same graph looks like -- OpenMP over rows,
 * vectorisable inner loops, order-fixed segmented sums instead of scatter-adds with duplicates
 * (Theano's AdvancedIncSubtensor1 is a serial row loop), one fused pass for L2 + Adam -- so that
 * the GPU number is compared with a baseline that uses the cores it is given.
 *
 * Arithmetic follows oracle/sert_oracle.py (which cites sert/models.py line by line):
 *   forward   :180, :226, :1057, :1065-1068, :990, :896-900, :1091-1098
 *   loss      :278-282, :754, :773-793
 *   update    lasagne.updates.adam (selected :922, applied :548-549)
 * fp32 throughout (floatX=float32, product-search.sh:95); sums over the batch in fp64 as
 * Theano's Sum does [upstream].
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_SLOTS 16   /* distinct batches whose word index is kept (static slices of the data set) */
#define HEAVY 512
#define CLIP_LO 1e-7f
#define CLIP_HI 0.99999988079071044921875f /* float32(1 - 1e-7) */

typedef struct {
    int B, n, z, Vw, Ve, dw, de;
    float lambda, lr, beta1, beta2, eps;
    long step;
    /* parameters and Adam state, order [R_e, R_w, W, b] (models.py:542-543, :1105) */
    float *Rw, *Re, *W, *b;
    float *m[4], *v[4], *g[4];
    /* activations of one batch */
    float *h, *t, *da, *dh, *coef;
    /* inverted index word -> token positions of the current batch (rebuilt per batch; the GPU
     * engine builds its index once at upload because batches are static slices) */
    int32_t *w_start[MAX_SLOTS], *w_pos[MAX_SLOTS];
    /* heavy words (Zipf: one word can own 15 % of a batch's tokens) are summed in PIECES of
     * HEAVY occurrences by different threads, the partial rows combined in piece order */
    int32_t *pc_lo[MAX_SLOTS], *pc_word[MAX_SLOTS], *w_piece0[MAX_SLOTS];
    int n_pieces[MAX_SLOTS];
    float* pc_part;
    int32_t *e_start, *e_pos, *e_hist;
    float* Wt;          /* W transposed, for dh = da . W^T */
    float* dW_part;     /* per-thread partial dW / db */
    int nthreads;
    double phase_ms[6]; /* last step: forward+NCE+dh, dW, entity grouping, segmented sums, L2+Adam, total */
} cpu_model;

/* zero-filled, pages first touched by the threads that will stream them (static blocks, as the
 * optimiser loop partitions them): on a multi-socket / multi-CCD host a table initialised by one
 * thread lives on one memory node and every other core reads it remotely */
static float* falloc(size_t n) {
    float* p = (float*)aligned_alloc(64, ((n * sizeof(float) + 63) / 64) * 64);
    if (!p) return p;
    const size_t nblk = (n + 4095) / 4096;
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < nblk; ++blk) {
        const size_t lo = blk * 4096, hi = lo + 4096 < n ? lo + 4096 : n;
        memset(p + lo, 0, (hi - lo) * sizeof(float));
    }
    return p;
}
static void pcopy(float* dst, const float* src, size_t n) {
    const size_t nblk = (n + 4095) / 4096;
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < nblk; ++blk) {
        const size_t lo = blk * 4096, hi = lo + 4096 < n ? lo + 4096 : n;
        memcpy(dst + lo, src + lo, (hi - lo) * sizeof(float));
    }
}

cpu_model* sert_cpu_create(int B, int n, int z, int Vw, int Ve, int dw, int de, float lambda,
                           const float* Rw, const float* Re, const float* W, const float* b) {
    cpu_model* m = (cpu_model*)calloc(1, sizeof(cpu_model));
    m->B = B; m->n = n; m->z = z; m->Vw = Vw; m->Ve = Ve; m->dw = dw; m->de = de;
    m->lambda = lambda; m->lr = 1e-3f; m->beta1 = 0.9f; m->beta2 = 0.999f; m->eps = 1e-8f;
    const size_t cnt[4] = {(size_t)Ve * de, (size_t)Vw * dw, (size_t)dw * de, (size_t)de};
    m->Re = falloc(cnt[0]); m->Rw = falloc(cnt[1]); m->W = falloc(cnt[2]); m->b = falloc(cnt[3]);
    pcopy(m->Re, Re, cnt[0]); pcopy(m->Rw, Rw, cnt[1]);
    pcopy(m->W, W, cnt[2]); pcopy(m->b, b, cnt[3]);
    for (int i = 0; i < 4; ++i) { m->m[i] = falloc(cnt[i]); m->v[i] = falloc(cnt[i]); m->g[i] = falloc(cnt[i]); }
    m->h = falloc((size_t)B * dw); m->t = falloc((size_t)B * de);
    m->da = falloc((size_t)B * de); m->dh = falloc((size_t)B * dw);
    m->coef = falloc((size_t)B * (z + 1));
    m->e_start = (int32_t*)calloc((size_t)Ve + 1, 4); m->e_pos = (int32_t*)calloc((size_t)B * (z + 1), 4);
    m->Wt = falloc((size_t)dw * de);
    m->nthreads = omp_get_max_threads();
    m->e_hist = (int32_t*)calloc((size_t)m->nthreads * ((size_t)Ve + 1), 4);
    m->dW_part = falloc((size_t)m->nthreads * ((size_t)dw * de + de));
    return m;
}

void sert_cpu_destroy(cpu_model* m) {
    if (!m) return;
    free(m->Re); free(m->Rw); free(m->W); free(m->b);
    for (int i = 0; i < 4; ++i) { free(m->m[i]); free(m->v[i]); free(m->g[i]); }
    free(m->h); free(m->t); free(m->da); free(m->dh); free(m->coef);
    for (int i = 0; i < MAX_SLOTS; ++i) { free(m->w_start[i]); free(m->w_pos[i]); free(m->pc_lo[i]); free(m->pc_word[i]); free(m->w_piece0[i]); }
    free(m->pc_part);
    free(m->e_start); free(m->e_pos); free(m->e_hist); free(m->Wt); free(m->dW_part);
    free(m);
}

int sert_cpu_threads(const cpu_model* m) { return m->nthreads; }
void sert_cpu_phases(const cpu_model* m, double* out6) { memcpy(out6, m->phase_ms, sizeof m->phase_ms); }

void sert_cpu_get(const cpu_model* m, float* Rw, float* Re, float* W, float* b) {
    memcpy(Re, m->Re, (size_t)m->Ve * m->de * 4); memcpy(Rw, m->Rw, (size_t)m->Vw * m->dw * 4);
    memcpy(W, m->W, (size_t)m->dw * m->de * 4); memcpy(b, m->b, (size_t)m->de * 4);
}

/* the next random row of an activation / embedding table, requested a few iterations ahead (a
 * 512-byte row = 8 cache lines; the hardware prefetcher cannot guess a gather) */
static inline void prefetch_row(const float* row, int floats) {
    for (int k = 0; k < floats; k += 16) __builtin_prefetch(row + k, 0, 1);
}

/* T.nnet.sigmoid, float32 C implementation of Theano 0.8.2 [upstream] */
static inline float theano_sigmoid(float x) {
    if (x < -88.0f) return 0.0f;
    if (x > 15.0f) return 1.0f;
    return 1.0f / (1.0f + expf(-x));
}

/* stable grouping of `count` keys in [0, K): start[k] .. start[k+1] = positions with key k, in
 * increasing position order (=> order-fixed sums).  Every thread owns a contiguous range of
 * keys and scans the whole (cache-resident) key array for them: T-fold redundant reads, no
 * atomics, no serial pass but the prefix sum over K counters. */
static void group_by_key(const int32_t* keys, int count, int K, int32_t* start, int32_t* pos) {
    memset(start, 0, ((size_t)K + 1) * 4);
#pragma omp parallel
    {
        const int T = omp_get_num_threads(), tid = omp_get_thread_num();
        const int klo = (int)((long)K * tid / T), khi = (int)((long)K * (tid + 1) / T);
        for (int i = 0; i < count; ++i) {
            const int k = keys[i];
            if (k >= klo && k < khi) start[k + 1]++;
        }
#pragma omp barrier
#pragma omp single
        for (int k = 0; k < K; ++k) start[k + 1] += start[k];
        /* (implicit barrier) */
        if (khi > klo) {
            int32_t* cur = (int32_t*)malloc((size_t)(khi - klo) * 4);
            memcpy(cur, start + klo, (size_t)(khi - klo) * 4);
            for (int i = 0; i < count; ++i) {
                const int k = keys[i];
                if (k >= klo && k < khi) pos[cur[k - klo]++] = i;
            }
            free(cur);
        }
    }
}

/* The same grouping for a SMALL key space (entity ids at V_e ~ 1000; the negatives change every
 * step, so this runs inside the timed step): classic parallel counting sort -- every thread
 * histograms a contiguous range of positions, offsets come from a (key, thread)-ordered prefix. */
static void group_by_small_key(const int32_t* keys, int count, int K, int32_t* start, int32_t* pos,
                               int32_t* hist, int T) {
    memset(hist, 0, (size_t)T * ((size_t)K + 1) * 4);
#pragma omp parallel num_threads(T)
    {
        const int tid = omp_get_thread_num();
        const int lo = (int)((long)count * tid / T), hi = (int)((long)count * (tid + 1) / T);
        int32_t* h = hist + (size_t)tid * ((size_t)K + 1);
        for (int i = lo; i < hi; ++i) h[keys[i]]++;
#pragma omp barrier
        /* per key: counts of the threads -> exclusive prefix over the threads, total into start[k+1] */
#pragma omp for schedule(static)
        for (int k = 0; k < K; ++k) {
            int run = 0;
            for (int t = 0; t < T; ++t) { int32_t* ht = hist + (size_t)t * ((size_t)K + 1) + k; const int c = *ht; *ht = run; run += c; }
            start[k + 1] = run;
        }
        /* (implicit barrier) exclusive scan of the K totals: K adds, one thread */
#pragma omp single
        {
            start[0] = 0;
            for (int k = 0; k < K; ++k) start[k + 1] += start[k];
        }
        for (int i = lo; i < hi; ++i) { const int k = keys[i]; pos[start[k] + h[k]++] = i; }
    }
}

/* Inverted index word -> token positions of one batch, kept in `slot`.  Batches are static slices
 * of the data set (sert/models.py:322-326), so -- exactly like the GPU engine, which builds its
 * index once at upload -- this runs OUTSIDE the timed step. */
int sert_cpu_index_batch(cpu_model* m, int slot, const int32_t* X) {
    if (slot < 0 || slot >= MAX_SLOTS) return 1;
    if (!m->w_start[slot]) {
        m->w_start[slot] = (int32_t*)calloc((size_t)m->Vw + 1, 4);
        m->w_pos[slot] = (int32_t*)calloc((size_t)m->B * m->n, 4);
    }
    group_by_key(X, m->B * m->n, m->Vw, m->w_start[slot], m->w_pos[slot]);
    const int max_pieces = m->B * m->n / HEAVY + 2;
    if (!m->pc_lo[slot]) {
        m->pc_lo[slot] = (int32_t*)calloc((size_t)max_pieces + 1, 4);
        m->pc_word[slot] = (int32_t*)calloc((size_t)max_pieces + 1, 4);
        m->w_piece0[slot] = (int32_t*)calloc((size_t)m->Vw, 4);
    }
    if (!m->pc_part) m->pc_part = falloc((size_t)max_pieces * m->dw);
    int np = 0;
    const int32_t* st = m->w_start[slot];
    for (int u = 0; u < m->Vw; ++u) {
        m->w_piece0[slot][u] = -1;
        const int cnt = st[u + 1] - st[u];
        if (cnt <= HEAVY) continue;
        m->w_piece0[slot][u] = np;
        for (int lo = st[u]; lo < st[u + 1]; lo += HEAVY) { m->pc_lo[slot][np] = lo; m->pc_word[slot][np] = u; ++np; }
    }
    m->n_pieces[slot] = np;
    return 0;
}

/* One training step on rows X (B, n) int32 ids, y (B) labels, w (B) weights, neg (B, z)
 * negatives; `slot` holds the batch's word index (sert_cpu_index_batch).  Returns the training loss evaluated BEFORE the update (models.py:581-588). */
float sert_cpu_train_step(cpu_model* m, int slot, const int32_t* X, const int32_t* y, const float* w, const int32_t* neg) {
    const int B = m->B, n = m->n, z = m->z, dw = m->dw, de = m->de, Vw = m->Vw, Ve = m->Ve;
    const int c1 = z + 1;
    const float invB = 1.0f / (float)B, invn = 1.0f / (float)n;
    double loss_sum = 0.0;

    const double t0 = omp_get_wtime();
    /* W^T for the backward projection */
    for (int k = 0; k < dw; ++k)
        for (int j = 0; j < de; ++j) m->Wt[(size_t)j * dw + k] = m->W[(size_t)k * de + j];

    /* forward + NCE + d(loss)/da, one pass per row */
#pragma omp parallel for schedule(static) reduction(+ : loss_sum)
    for (int i = 0; i < B; ++i) {
        float* h = m->h + (size_t)i * dw;
        float acc[512];
        for (int k = 0; k < dw; ++k) acc[k] = 0.f;
        if (i + 1 < B)
            for (int q = 0; q < n; ++q) prefetch_row(m->Rw + (size_t)X[(size_t)(i + 1) * n + q] * dw, dw);
        for (int q = 0; q < n; ++q) {                       /* models.py:180, :226 */
            const float* row = m->Rw + (size_t)X[(size_t)i * n + q] * dw;
#pragma omp simd
            for (int k = 0; k < dw; ++k) acc[k] += row[k];
        }
        for (int k = 0; k < dw; ++k) h[k] = acc[k] * invn;
        float a[512];
        for (int j = 0; j < de; ++j) a[j] = m->b[j];
        for (int k = 0; k < dw; ++k) {                      /* models.py:1057 */
            const float hk = h[k];
            const float* wr = m->W + (size_t)k * de;
#pragma omp simd
            for (int j = 0; j < de; ++j) a[j] += hk * wr[j];
        }
        float* t = m->t + (size_t)i * de;
        float p[512];
        for (int j = 0; j < de; ++j) {
            t[j] = tanhf(a[j]);
            p[j] = fminf(fmaxf(t[j], -CLIP_HI), CLIP_HI);  /* models.py:1065-1068 */
        }
        const float gi = w[i] * invB;
        float dp[512];
        for (int j = 0; j < de; ++j) dp[j] = 0.f;
        float rowloss = 0.f;
        for (int c = 0; c < c1; ++c) {                      /* models.py:990, :896-900, :1091-1098 */
            const int e = c == 0 ? y[i] : neg[(size_t)i * z + c - 1];
            const float* er = m->Re + (size_t)e * de;
            float u = 0.f;
#pragma omp simd reduction(+ : u)
            for (int j = 0; j < de; ++j) u += er[j] * p[j];
            const float sig = theano_sigmoid(u);
            const float s = fminf(fmaxf(sig, CLIP_LO), CLIP_HI);
            const float inside = (sig >= CLIP_LO && sig <= CLIP_HI) ? 1.0f : 0.0f;   /* Clip.grad inclusive */
            float du;
            if (c == 0) { rowloss -= logf(s); du = -(gi / s) * inside * sig * (1.0f - sig); }
            else        { rowloss -= logf(1.0f - s); du = (gi / (1.0f - s)) * inside * sig * (1.0f - sig); }
            m->coef[(size_t)i * c1 + c] = du;
#pragma omp simd
            for (int j = 0; j < de; ++j) dp[j] += du * er[j];
        }
        loss_sum += (double)(rowloss * w[i]);
        float* da = m->da + (size_t)i * de;
        for (int j = 0; j < de; ++j) {
            const float in = (t[j] >= -CLIP_HI && t[j] <= CLIP_HI) ? 1.0f : 0.0f;
            da[j] = dp[j] * in * (1.0f - t[j] * t[j]);
        }
        float* dh = m->dh + (size_t)i * dw;                 /* dh = da . W^T */
        for (int k = 0; k < dw; ++k) acc[k] = 0.f;
        for (int j = 0; j < de; ++j) {
            const float dj = da[j];
            const float* wt = m->Wt + (size_t)j * dw;
#pragma omp simd
            for (int k = 0; k < dw; ++k) acc[k] += dj * wt[k];
        }
        for (int k = 0; k < dw; ++k) dh[k] = acc[k];
    }

    const double t1 = omp_get_wtime();
    /* dW = h^T . da and db = sum_i da_i: per-thread partials over a static row range, combined in
     * thread order (deterministic for a given thread count) */
    const size_t mn = (size_t)dw * de, stride = mn + de;
    memset(m->dW_part, 0, (size_t)m->nthreads * stride * 4);
#pragma omp parallel
    {
        const int T = omp_get_num_threads(), tid = omp_get_thread_num();
        const int lo = (int)((long)B * tid / T), hi = (int)((long)B * (tid + 1) / T);
        float* part = m->dW_part + (size_t)tid * stride;
        for (int i = lo; i < hi; ++i) {
            const float* h = m->h + (size_t)i * dw;
            const float* da = m->da + (size_t)i * de;
            for (int k = 0; k < dw; ++k) {
                const float hk = h[k];
                float* pr = part + (size_t)k * de;
#pragma omp simd
                for (int j = 0; j < de; ++j) pr[j] += hk * da[j];
            }
            float* pb = part + mn;
#pragma omp simd
            for (int j = 0; j < de; ++j) pb[j] += da[j];
        }
    }
    {
        const int T = m->nthreads;
#pragma omp parallel for schedule(static)
        for (size_t o = 0; o < stride; ++o) {
            float s = 0.f;
            for (int t2 = 0; t2 < T; ++t2) s += m->dW_part[(size_t)t2 * stride + o];
            if (o < mn) m->g[2][o] = s; else m->g[3][o - mn] = s;
        }
    }

    const double t2 = omp_get_wtime();
    /* dR_w[word] = sum over its occurrences of dh_i / n, dR_e[e] = sum of coef * p: segmented
     * sums in occurrence order over a per-batch inverted index (no scatter-add with duplicates) */
    {
        int32_t* keys = (int32_t*)malloc((size_t)B * c1 * 4);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < B; ++i) {
            keys[(size_t)i * c1] = y[i];
            for (int c = 1; c < c1; ++c) keys[(size_t)i * c1 + c] = neg[(size_t)i * z + c - 1];
        }
        if ((size_t)Ve * m->nthreads <= ((size_t)1 << 22))
            group_by_small_key(keys, B * c1, Ve, m->e_start, m->e_pos, m->e_hist, m->nthreads);
        else
            group_by_key(keys, B * c1, Ve, m->e_start, m->e_pos);
        free(keys);
    }
    const double t3 = omp_get_wtime();
    const int32_t* w_start = m->w_start[slot];
    const int32_t* w_pos = m->w_pos[slot];
    const int32_t *pc_lo = m->pc_lo[slot], *pc_word = m->pc_word[slot], *w_piece0 = m->w_piece0[slot];
    const int n_pieces = m->n_pieces[slot];
#pragma omp parallel for schedule(dynamic, 1)
    for (int pc = 0; pc < n_pieces; ++pc) {
        const int u = pc_word[pc];
        const int lo = pc_lo[pc], hi = lo + HEAVY < w_start[u + 1] ? lo + HEAVY : w_start[u + 1];
        float* g = m->pc_part + (size_t)pc * dw;
        for (int k = 0; k < dw; ++k) g[k] = 0.f;
        for (int q = lo; q < hi; ++q) {
            if (q + 4 < hi) prefetch_row(m->dh + (size_t)(w_pos[q + 4] / n) * dw, dw);
            const float* dh = m->dh + (size_t)(w_pos[q] / n) * dw;
#pragma omp simd
            for (int k = 0; k < dw; ++k) g[k] += dh[k] * invn;
        }
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int u = 0; u < Vw; ++u) {
        float* g = m->g[1] + (size_t)u * dw;
        for (int k = 0; k < dw; ++k) g[k] = 0.f;
        if (w_piece0[u] >= 0) {
            const int cnt = w_start[u + 1] - w_start[u];
            for (int pc = w_piece0[u]; pc < w_piece0[u] + (cnt + HEAVY - 1) / HEAVY; ++pc) {
                const float* pr = m->pc_part + (size_t)pc * dw;
#pragma omp simd
                for (int k = 0; k < dw; ++k) g[k] += pr[k];
            }
            continue;
        }
        for (int q = w_start[u]; q < w_start[u + 1]; ++q) {
            if (q + 4 < w_start[u + 1]) prefetch_row(m->dh + (size_t)(w_pos[q + 4] / n) * dw, dw);
            const float* dh = m->dh + (size_t)(w_pos[q] / n) * dw;
#pragma omp simd
            for (int k = 0; k < dw; ++k) g[k] += dh[k] * invn;
        }
    }
#pragma omp parallel for schedule(dynamic, 8)
    for (int e = 0; e < Ve; ++e) {
        float* g = m->g[0] + (size_t)e * de;
        for (int j = 0; j < de; ++j) g[j] = 0.f;
        for (int q = m->e_start[e]; q < m->e_start[e + 1]; ++q) {
            if (q + 4 < m->e_start[e + 1]) prefetch_row(m->t + (size_t)(m->e_pos[q + 4] / c1) * de, de);
            const int pos = m->e_pos[q];
            const float cf = m->coef[pos];
            const float* t = m->t + (size_t)(pos / c1) * de;
#pragma omp simd
            for (int j = 0; j < de; ++j) g[j] += cf * fminf(fmaxf(t[j], -CLIP_HI), CLIP_HI);
        }
    }

    const double t4 = omp_get_wtime();
    /* dense L2 + dense Adam over every element of every tensor (models.py:764-795, :548-549), one
     * fused pass; sum of squares of the pre-update values for the returned loss */
    m->step += 1;
    const float tt = (float)m->step;
    const float a_t = m->lr * sqrtf(1.0f - powf(m->beta2, tt)) / (1.0f - powf(m->beta1, tt));
    const float l2k = m->lambda > 0.f ? m->lambda / (float)B : 0.f;
    const float b1 = m->beta1, b2 = m->beta2, eps = m->eps;
    float* P[4] = {m->Re, m->Rw, m->W, m->b};
    const size_t cnt[4] = {(size_t)Ve * de, (size_t)Vw * dw, (size_t)dw * de, (size_t)de};
    double sq = 0.0;
    for (int ti = 0; ti < 4; ++ti) {
        float *p = P[ti], *g = m->g[ti], *mm = m->m[ti], *vv = m->v[ti];
        const float k = ti == 3 ? 0.f : l2k;     /* bias: not regularised [upstream] */
        const size_t N = cnt[ti];
        double part = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : part)
        for (size_t blk = 0; blk < (N + 4095) / 4096; ++blk) {
            const size_t lo = blk * 4096, hi = lo + 4096 < N ? lo + 4096 : N;
            float s = 0.f;
#pragma omp simd reduction(+ : s)
            for (size_t i = lo; i < hi; ++i) {
                const float pv = p[i];
                const float gv = g[i] + k * pv;
                s += pv * pv;
                const float mv = b1 * mm[i] + (1.0f - b1) * gv;
                const float v2 = b2 * vv[i] + (1.0f - b2) * gv * gv;
                mm[i] = mv; vv[i] = v2;
                p[i] = pv - a_t * mv / (sqrtf(v2) + eps);
            }
            part += (double)s;
        }
        if (ti != 3) sq += part;
    }
    const double t5 = omp_get_wtime();
    m->phase_ms[0] = 1e3 * (t1 - t0); m->phase_ms[1] = 1e3 * (t2 - t1); m->phase_ms[2] = 1e3 * (t3 - t2);
    m->phase_ms[3] = 1e3 * (t4 - t3); m->phase_ms[4] = 1e3 * (t5 - t4); m->phase_ms[5] = 1e3 * (t5 - t0);
    const float reg = m->lambda > 0.f ? (m->lambda / (2.0f * (float)B)) * (float)sq : 0.f;
    return (float)(loss_sum) * invB + reg;
}

/* Scoring baseline (bin/query.py:239-370, batched): cosines of Q projections against V unit
 * entity rows and the k best per query by partial selection; returns indices only. */
void sert_cpu_score_topk(const float* E_unit, int V, int d, const float* P_unit, int Q, int k, int32_t* idx_out) {
#pragma omp parallel
    {
        float* sc = (float*)malloc((size_t)V * 4);
        int32_t* best_i = (int32_t*)malloc((size_t)k * 4);
        float* best_s = (float*)malloc((size_t)k * 4);
#pragma omp for schedule(dynamic, 4)
        for (int q = 0; q < Q; ++q) {
            const float* p = P_unit + (size_t)q * d;
            for (int e = 0; e < V; ++e) {
                const float* er = E_unit + (size_t)e * d;
                float u = 0.f;
#pragma omp simd reduction(+ : u)
                for (int j = 0; j < d; ++j) u += er[j] * p[j];
                sc[e] = u;
            }
            /* k best by insertion into a sorted list (k = 100 << V): ties keep the lower index */
            int filled = 0;
            for (int e = 0; e < V; ++e) {
                const float s = sc[e];
                if (filled == k && !(s > best_s[k - 1])) continue;
                int pos = filled < k ? filled : k - 1;
                while (pos > 0 && best_s[pos - 1] < s) { best_s[pos] = best_s[pos - 1]; best_i[pos] = best_i[pos - 1]; --pos; }
                best_s[pos] = s; best_i[pos] = e;
                if (filled < k) ++filled;
            }
            memcpy(idx_out + (size_t)q * k, best_i, (size_t)k * 4);
        }
        free(sc); free(best_i); free(best_s);
    }
}
