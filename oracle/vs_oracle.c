/*
 * vs_oracle.c -- CPU restatement of the vsearch vocabulary-space retrieval path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vsearch_amd/ (the product) may import, link or call
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only
 * as the checker.  Parity of this restatement is PINNED against golden vectors produced by
 * running the reference itself in the build container (tools/gen_golden.py -> tests/golden/,
 * checked by tests/test_oracle_golden.py).
 *
 * The reference (jzhoubu/vsearch @ 2024-12-18) is pure Python; the arithmetic of this path lives in
 * torch 2.3.0 (poetry.lock:6091): sparse-CSR addmm, dense GEMM, topk, scatter_.  Each function
 * below cites the reference call site it restates.  Conventions fixed here (the reference leaves
 * them implementation-defined): fp32 products accumulated left-to-right in CSR / column order;
 * top-k order = (score descending, id ascending).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; no dependencies)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VSO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* canonical top-k: (score desc, id asc). Binary min-heap of the k best seen so far.            */
typedef struct { float s; int64_t id; } vso_cand;

static inline int cand_better(vso_cand a, vso_cand b) {          /* a ranks before b */
    return (a.s > b.s) || (a.s == b.s && a.id < b.id);
}

static void heap_sift_down(vso_cand* h, int64_t n, int64_t i) {  /* root = worst kept */
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && cand_better(h[w], h[l])) w = l;
        if (r < n && cand_better(h[w], h[r])) w = r;
        if (w == i) return;
        vso_cand t = h[i]; h[i] = h[w]; h[w] = t; i = w;
    }
}

static int cand_cmp(const void* a, const void* b) {
    const vso_cand* x = (const vso_cand*)a; const vso_cand* y = (const vso_cand*)b;
    if (cand_better(*x, *y)) return -1;
    if (cand_better(*y, *x)) return 1;
    return 0;
}

/* scores[0..n) -> out_ids/out_scores[0..k): restates Tensor.topk(k) (index.py:92), sorted, largest. */
static void topk_row(const float* scores, int64_t n, int64_t k, int64_t id0, int64_t* out_ids, float* out_scores) {
    vso_cand* h = (vso_cand*)malloc(sizeof(vso_cand) * (size_t)k);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        vso_cand c = { scores[i], id0 + i };
        if (m < k) {
            h[m++] = c;
            if (m == k) for (int64_t j = k / 2 - 1; j >= 0; --j) heap_sift_down(h, k, j);
        } else if (cand_better(c, h[0])) {
            h[0] = c; heap_sift_down(h, k, 0);
        }
    }
    qsort(h, (size_t)m, sizeof(vso_cand), cand_cmp);
    for (int64_t j = 0; j < m; ++j) { out_ids[j] = h[j].id; out_scores[j] = h[j].s; }
    free(h);
}

/* ------------------------------------------------------------------------------------------ */
/* SparseIndex.search / BoTIndex.search  (src/ir/retriever/index.py:88-94):
 *   scores = matmul(q [B,V], vector_csr[N,V].t());  topk(k).
 * data == NULL  => binary index (BoT, values == 1, retriever.py:242).
 * acc64 != 0    => accumulate in double, round once (the "exact" variant used to state tolerances).
 * all_scores (optional, [B,N]) receives the dense score matrix the reference materialises.      */
VSO_API int vso_csr_search(const int64_t* indptr, const int32_t* indices, const float* data,
                           int64_t n_rows, int32_t n_cols, const float* q, int32_t B, int64_t k,
                           int acc64, int64_t* out_ids, float* out_scores, float* all_scores) {
    if (k > n_rows || k <= 0) return -1;           /* torch.topk raises RuntimeError (index.py:92) */
    float* s = (float*)malloc(sizeof(float) * (size_t)n_rows);
    for (int32_t b = 0; b < B; ++b) {
        const float* qb = q + (size_t)b * n_cols;
        for (int64_t r = 0; r < n_rows; ++r) {
            if (acc64) {
                double a = 0.0;
                for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p)
                    a += (double)qb[indices[p]] * (double)(data ? data[p] : 1.0f);
                s[r] = (float)a;
            } else {
                float a = 0.0f;
                for (int64_t p = indptr[r]; p < indptr[r + 1]; ++p) {
                    float prod = qb[indices[p]] * (data ? data[p] : 1.0f);   /* rounded product */
                    a += prod;                                                 /* rounded sum     */
                }
                s[r] = a;
            }
        }
        if (all_scores) memcpy(all_scores + (size_t)b * n_rows, s, sizeof(float) * (size_t)n_rows);
        topk_row(s, n_rows, k, 0, out_ids + (size_t)b * k, out_scores + (size_t)b * k);
    }
    free(s);
    return 0;
}

/* Index.search on a dense index (index.py:88-94): matmul(q, vector.t()).topk(k). */
VSO_API int vso_dense_search(const float* mat, int64_t n_rows, int32_t n_cols, const float* q, int32_t B,
                             int64_t k, int acc64, int64_t* out_ids, float* out_scores) {
    if (k > n_rows || k <= 0) return -1;
    float* s = (float*)malloc(sizeof(float) * (size_t)n_rows);
    for (int32_t b = 0; b < B; ++b) {
        const float* qb = q + (size_t)b * n_cols;
        for (int64_t r = 0; r < n_rows; ++r) {
            const float* pr = mat + (size_t)r * n_cols;
            if (acc64) {
                double a = 0.0;
                for (int32_t c = 0; c < n_cols; ++c) a += (double)qb[c] * (double)pr[c];
                s[r] = (float)a;
            } else {
                float a = 0.0f;
                for (int32_t c = 0; c < n_cols; ++c) { float prod = qb[c] * pr[c]; a += prod; }
                s[r] = a;
            }
        }
        topk_row(s, n_rows, k, 0, out_ids + (size_t)b * k, out_scores + (size_t)b * k);
    }
    free(s);
    return 0;
}

/* Row-sharded search merge (new in the build): concatenate per-shard top-k
 * candidate lists (global ids) and re-select the canonical top-k. cand_* are [B, n_cand].       */
VSO_API int vso_merge_topk(const int64_t* cand_ids, const float* cand_scores, int32_t B, int64_t n_cand,
                           int64_t k, int64_t* out_ids, float* out_scores) {
    if (k > n_cand) return -1;
    vso_cand* c = (vso_cand*)malloc(sizeof(vso_cand) * (size_t)n_cand);
    for (int32_t b = 0; b < B; ++b) {
        for (int64_t i = 0; i < n_cand; ++i) { c[i].s = cand_scores[(size_t)b * n_cand + i]; c[i].id = cand_ids[(size_t)b * n_cand + i]; }
        qsort(c, (size_t)n_cand, sizeof(vso_cand), cand_cmp);
        for (int64_t j = 0; j < k; ++j) { out_ids[(size_t)b * k + j] = c[j].id; out_scores[(size_t)b * k + j] = c[j].s; }
    }
    free(c);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* elu1p (src/ir/utils/sparse.py:6): F.elu(x) + 1  ==  x > 0 ? x + 1 : expm1(x) + 1.            */
VSO_API void vso_elu1p(const float* x, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = x[i] > 0.0f ? x[i] + 1.0f : expm1f(x[i]) + 1.0f;
}

/* build_topk_mask (sparse.py:8-14): bool mask of the k largest per row. Ties at the k-th value are
 * implementation-defined in torch; canonical here: lower column index wins.                      */
VSO_API int vso_topk_mask(const float* x, int32_t B, int32_t V, int32_t k, uint8_t* mask) {
    if (k > V || k < 0) return -1;
    int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(k > 0 ? k : 1));
    float* sc = (float*)malloc(sizeof(float) * (size_t)(k > 0 ? k : 1));
    memset(mask, 0, (size_t)B * V);
    for (int32_t b = 0; b < B && k > 0; ++b) {
        topk_row(x + (size_t)b * V, V, k, 0, ids, sc);
        for (int32_t j = 0; j < k; ++j) mask[(size_t)b * V + ids[j]] = 1;
    }
    free(ids); free(sc);
    return 0;
}

/* build_bow_mask (sparse.py:21-29): multi-hot over `vocab`, drop the first `shift` columns,
 * optional L2 row normalisation (F.normalize: x / max(||x||_2, 1e-12)).  ids: [B, L] int64.     */
VSO_API int vso_bow_mask(const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift, int norm, float* out) {
    int32_t V = vocab - shift;
    memset(out, 0, sizeof(float) * (size_t)B * V);
    for (int32_t b = 0; b < B; ++b) {
        for (int32_t l = 0; l < L; ++l) {
            int64_t t = ids[(size_t)b * L + l];
            if (t < 0 || t >= vocab) return -1;                   /* scatter_ would raise */
            if (t >= shift) out[(size_t)b * V + (t - shift)] = 1.0f;
        }
        if (norm) {
            float ss = 0.0f;
            for (int32_t c = 0; c < V; ++c) ss += out[(size_t)b * V + c] * out[(size_t)b * V + c];
            float d = sqrtf(ss); if (d < 1e-12f) d = 1e-12f;
            for (int32_t c = 0; c < V; ++c) out[(size_t)b * V + c] /= d;
        }
    }
    return 0;
}

/* VDREncoder.embed mask logic (src/ir/encoder/vdr.py:152-169):
 *   bow -> emb = bow_mask;  else topk==0 -> zeros; topk==-1 -> ones; else build_topk_mask;
 *   mask = bow | topk (if activate_lexical) else topk;  emb *= mask.   emb: [B,V] in/out.        */
VSO_API int vso_embed_mask(float* emb, const int64_t* ids, int32_t B, int32_t L, int32_t vocab, int32_t shift,
                           int32_t topk, int activate_lexical, int bow) {
    int32_t V = vocab - shift;
    float* bm = (float*)malloc(sizeof(float) * (size_t)B * V);
    uint8_t* tm = (uint8_t*)malloc((size_t)B * V);
    int rc = vso_bow_mask(ids, B, L, vocab, shift, 0, bm);
    if (rc == 0) {
        if (bow) memcpy(emb, bm, sizeof(float) * (size_t)B * V);
        else {
            if (topk == 0) memset(tm, 0, (size_t)B * V);
            else if (topk < 0) memset(tm, 1, (size_t)B * V);
            else rc = vso_topk_mask(emb, B, V, topk, tm);
            if (rc == 0)
                for (int64_t i = 0; i < (int64_t)B * V; ++i) {
                    int m = tm[i] || (activate_lexical && bm[i] != 0.0f);
                    emb[i] = m ? emb[i] : emb[i] * 0.0f;
                }
        }
    }
    free(bm); free(tm);
    return rc;
}

/* Encoder head tail (vdr.py:73-75): elu1p over [B,L,V] logits then max over L (pad positions
 * included, no attention mask).                                                                 */
VSO_API void vso_head_pool(const float* logits, int32_t B, int32_t L, int32_t V, float* out) {
    for (int32_t b = 0; b < B; ++b)
        for (int32_t c = 0; c < V; ++c) {
            float m = -INFINITY;
            for (int32_t l = 0; l < L; ++l) {
                float x = logits[((size_t)b * L + l) * V + c];
                float e = x > 0.0f ? x + 1.0f : expm1f(x) + 1.0f;
                if (e > m) m = e;
            }
            out[(size_t)b * V + c] = m;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Retriever._build_bot_vectors (src/ir/retriever/retriever.py:208-253) + get_first_unique_n
 * (index_utils.py:11-21), intended single-batch semantics:
 *   per doc: token ids (already truncated by the tokenizer), optional first-`max_token`-unique cap
 *   (counting every id, [CLS] included), multi-hot over `vocab`, drop ids < shift, CSR with sorted
 *   columns, all values 1.  tokens: flat int32, offsets: [n_docs+1].
 * Two-call protocol: indices == NULL -> only indptr is filled (so the caller can size indices).   */
VSO_API int vso_bot_build(const int32_t* tokens, const int64_t* offsets, int64_t n_docs, int32_t vocab,
                          int32_t shift, int32_t max_token, int64_t* indptr, int32_t* indices) {
    uint8_t* seen = (uint8_t*)calloc((size_t)vocab, 1);
    int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)vocab);
    indptr[0] = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        int32_t nt = 0;
        for (int64_t p = offsets[d]; p < offsets[d + 1]; ++p) {
            int32_t t = tokens[p];
            if (t < 0 || t >= vocab) { free(seen); free(touched); return -1; }
            if (seen[t]) continue;
            seen[t] = 1; touched[nt++] = t;
            if (max_token > 0 && nt == max_token) break;
        }
        int64_t cnt = 0;
        for (int32_t i = 0; i < nt; ++i) if (touched[i] >= shift) ++cnt;
        indptr[d + 1] = indptr[d] + cnt;
        if (indices) {
            int32_t* dst = indices + indptr[d];
            int64_t w = 0;
            for (int32_t i = 0; i < nt; ++i) if (touched[i] >= shift) dst[w++] = touched[i] - shift;
            for (int64_t i = 1; i < w; ++i) {                 /* insertion sort: rows are short */
                int32_t v = dst[i]; int64_t j = i - 1;
                while (j >= 0 && dst[j] > v) { dst[j + 1] = dst[j]; --j; }
                dst[j + 1] = v;
            }
        }
        for (int32_t i = 0; i < nt; ++i) seen[touched[i]] = 0;
    }
    free(seen); free(touched);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic corpus generator: bit-exact twin of vsearch_amd/synth.py (numpy) and
 * vsearch_amd/csrc/synth.hip (device).                                                          */
static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t hash2(uint64_t seed, uint64_t a) { return splitmix64(splitmix64(seed) ^ (a * 0xD1342543DE82EF95ull)); }
static inline uint64_t hash3(uint64_t seed, uint64_t a, uint64_t b) { return splitmix64(hash2(seed, a) + b * 0x2545F4914F6CDD1Dull); }
static inline uint32_t feistel16(uint32_t x, uint64_t key) {
    uint32_t L = x >> 8, R = x & 0xFF;
    for (int i = 0; i < 4; ++i) {
        uint32_t k = (uint32_t)((key >> (16 * i)) & 0xFFFF);
        uint32_t t = (R ^ k) * 0x9E3779B1u + k;
        uint32_t F = (t >> 24) & 0xFF;
        uint32_t nl = R; R = L ^ F; L = nl;
    }
    return (L << 8) | R;
}
static inline uint32_t perm_col(uint64_t key, uint32_t j, uint32_t n_cols) {
    uint32_t x = feistel16(j, key);
    while (x >= n_cols) x = feistel16(x, key);
    return x;
}
/* kind 2: skewed column popularity, twin of synth_skew_col (vsearch_amd/csrc/synth_device.h) */
static inline uint32_t skew_col(uint64_t key, uint32_t j, uint32_t len, uint32_t n_cols) {
    const uint32_t kHead = 127;
    uint32_t rank, kmax = 0;
    while ((2u << kmax) <= n_cols) ++kmax;
    if (len <= kHead + 8 || kmax < 8) {
        rank = j + 1;
    } else if (j < kHead) {
        rank = j + 1;
    } else {
        const uint32_t n_oct = kmax - 7 + 1, per = (len - kHead) / n_oct, t = j - kHead;
        uint32_t o = t / per;
        if (o > n_oct - 1) o = n_oct - 1;
        const uint32_t i = t - o * per, k = 7 + o;
        const uint32_t lo = 1u << k, size = (2u << k) <= n_cols + 1 ? lo : n_cols + 1 - lo, mask = lo - 1;
        const uint64_t h = splitmix64(key + 0x9E3779B97F4A7C15ull * (k + 1));
        const uint32_t odd = (uint32_t)h | 1u, add = (uint32_t)(h >> 32);
        uint32_t x = i;
        do { x = (x * odd + add) & mask; } while (x >= size);
        rank = lo + x;
    }
    return perm_col(0x5A495046534B4557ull, rank - 1, n_cols);
}
static inline int64_t row_len(uint64_t seed, int64_t row, int kind, int32_t nnz, int32_t n_cols) {
    int64_t len = nnz;
    if (kind == 1) {
        uint64_t h = hash3(seed, (uint64_t)row, 0x4C454Eull);
        int64_t s = 0;
        for (int i = 0; i < 4; ++i) s += (int64_t)((h >> (16 * i)) & 0xFFFF);
        len = 1 + (s * (int64_t)(nnz - 1)) / (2 * 65536);
    }
    return len < n_cols ? len : n_cols;
}
static inline float synth_val(uint64_t seed, int64_t row, uint32_t col, int val_law) {
    uint64_t h = hash3(seed ^ 0x56414Cull, (uint64_t)row, (uint64_t)col);
    if (val_law == 0) return (164.0f + (float)(h % 49152ull)) / 16384.0f;
    if (val_law == 1) return (1.0f + (float)(h % 255ull)) / 64.0f;
    return 1.0f;
}

/* indices == NULL -> fill indptr only. kind: 0 = fixed nnz, 1 = BoT lengths (binary), 2 = fixed nnz, skewed column popularity. */
VSO_API int vso_synth_csr(uint64_t seed, int64_t row0, int64_t n_rows, int32_t n_cols, int32_t nnz, int kind,
                          int val_law, int64_t* indptr, int32_t* indices, float* data) {
    if (n_cols <= 0 || n_cols > 65536) return -1;
    indptr[0] = 0;
    for (int64_t r = 0; r < n_rows; ++r) indptr[r + 1] = indptr[r] + row_len(seed, row0 + r, kind, nnz, n_cols);
    if (!indices) return 0;
    /* columns of a row are distinct: set them in a bitmap, then read the bitmap in ascending order
       (same result as sorting; far fewer cycles than qsort for 768 entries out of 29 523) */
    const int words = (n_cols + 63) / 64;
    uint64_t* tmp = (uint64_t*)calloc((size_t)words, sizeof(uint64_t));
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t row = row0 + r, len = indptr[r + 1] - indptr[r];
        uint64_t key = hash3(seed, (uint64_t)row, 0x4B4559ull);
        for (int64_t j = 0; j < len; ++j) {
            uint32_t c = kind == 2 ? skew_col(key, (uint32_t)j, (uint32_t)len, (uint32_t)n_cols) : perm_col(key, (uint32_t)j, (uint32_t)n_cols);
            tmp[c >> 6] |= 1ull << (c & 63);
        }
        int64_t w = indptr[r];
        for (int i = 0; i < words; ++i) {
            uint64_t bits = tmp[i];
            tmp[i] = 0;
            while (bits) {
                uint32_t c = (uint32_t)i * 64 + (uint32_t)__builtin_ctzll(bits);
                bits &= bits - 1;
                indices[w] = (int32_t)c;
                if (data) data[w] = (kind == 1) ? 1.0f : synth_val(seed, row, c, val_law);
                ++w;
            }
        }
    }
    free(tmp);
    return 0;
}
