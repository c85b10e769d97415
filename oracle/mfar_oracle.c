/*
 * mfar_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's dense multi-field scoring path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (multifield-adaptive-retrieval_amd/) never does.
 *
 * What it restates (citations relative to /root/reference):
 *   - DenseFlatIndex.retrieve_batch   mfar/data/index.py:181-222  (exhaustive per-field top-k, running
 *     top-k seeded with k x (row 0, score 0.0): the "zero sentinel", index.py:192-193)
 *   - DenseFlatIndex.score_batch      mfar/data/index.py:227-232  (gather rows, dot with the query)
 *   - LinearWeights.forward           mfar/modeling/weighting.py:17-29 (softmax(q @ W), weighted field sum)
 *   - RetrievalTrainingModule.trec_eval_step  mfar/modeling/contrastive.py:669-704 (stage 1 per field,
 *     union of ids :678-679, stage 2 re-score :681-683, mask :685-686, mix :694, topk(100) :696)
 *
 * Parity pinning: tests/test_oracle_golden.py checks every function here against the golden vectors in
 * tests/golden/ (npz files), which tools/gen_golden.py captured by running the reference's own unmodified
 * functions (ids equal under the canonical tie-break, scores <= 1e-4).
 *
 * Arithmetic contract (this is what makes GPU-vs-oracle comparisons BIT-EXACT):
 *   - every query.doc dot product is ONE fp32 fused-multiply-add chain, acc = fmaf(q[k], v[k], acc),
 *     starting from 0, visiting k in the order the gfx950 kernel's MFMA stream visits it: inside each
 *     aligned group of 8 dims the order is 0,4,1,5,2,6,3,7 (lanes 0-31 of v_mfma_f32_32x32x2_f32 hold
 *     dims 8g..8g+3, lanes 32-63 hold 8g+4..8g+7; step s multiplies dim 8g+s then 8g+4+s).
 *     v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain, so the HIP kernels reproduce these bits.
 *   - the field-weight head uses a natural-order fmaf chain for q@W, a polynomial exp built from
 *     fmaf/mul only (mfar_oracle_exp), a left-to-right sum and one IEEE division per weight;
 *   - the mixed score is acc = fmaf(w_f, x_f * mask_f, acc) for f = 0..F-1.
 *   - ties: canonical order (score desc, doc id asc) everywhere (the reference leaves ties unspecified:
 *     unstable torch.topk + hash-ordered python set).
 * Compile with -ffp-contract=off (see oracle/Makefile) so the compiler adds no fusions of its own.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MFAR_ORACLE_VERSION 1

static const int CHAIN_PERM[8] = {0, 4, 1, 5, 2, 6, 3, 7};
/* 0: the fp32-slab contract above (MFMA order).  1: natural dim order -- the contract of the bf16 slab, whose stage-2
 * kernel walks the dims in order over bf16 values widened exactly to fp32 (feed this oracle the bf16-ROUNDED corpus:
 * mfar_oracle_bf16_round). */
static int g_chain_natural = 0;
void mfar_oracle_set_chain(int natural) { g_chain_natural = natural; }

int mfar_oracle_version(void) { return MFAR_ORACLE_VERSION; }

/* position p in the chain -> dim index */
static inline int chain_dim(int p, int E) {
    if (g_chain_natural) return p;
    int g = p >> 3;
    if ((g << 3) + 8 <= E) return (g << 3) + CHAIN_PERM[p & 7];
    return p; /* ragged tail (E % 8 != 0): natural order; the GPU path rejects such E */
}

float mfar_oracle_dot(const float* q, const float* v, int E) {
    float acc = 0.0f;
    for (int p = 0; p < E; ++p) {
        int k = chain_dim(p, E);
        acc = fmaf(q[k], v[k], acc);
    }
    return acc;
}

/* scores[Q, D] = q[Q,E] . V[D,E]^T with the chain order.  Vectorises over queries. */
void mfar_oracle_scores(const float* V, int64_t D, int E, const float* q, int Q, float* out) {
    /* qT[p][Q]: query values in chain order, transposed so the inner loop runs over queries */
    float* qT = (float*)malloc((size_t)E * Q * sizeof(float));
    for (int p = 0; p < E; ++p) {
        int k = chain_dim(p, E);
        for (int i = 0; i < Q; ++i) qT[(size_t)p * Q + i] = q[(size_t)i * E + k];
    }
#pragma omp parallel
    {
        float* acc = (float*)malloc((size_t)Q * sizeof(float));
#pragma omp for schedule(static)
        for (int64_t d = 0; d < D; ++d) {
            const float* v = V + (size_t)d * E;
            for (int i = 0; i < Q; ++i) acc[i] = 0.0f;
            for (int p = 0; p < E; ++p) {
                const float vk = v[chain_dim(p, E)];
                const float* qp = qT + (size_t)p * Q;
                for (int i = 0; i < Q; ++i) acc[i] = __builtin_fmaf(qp[i], vk, acc[i]);
            }
            for (int i = 0; i < Q; ++i) out[(size_t)i * D + d] = acc[i];
        }
        free(acc);
    }
    free(qT);
}

/* fp32 -> bf16 -> fp32, round to nearest even: what the bf16 slab stores (same integer formula as the HIP kernel) */
void mfar_oracle_bf16_round(const float* src, float* dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        union { float f; uint32_t u; } v;
        v.f = src[i];
        if ((v.u & 0x7FFFFFFFu) > 0x7F800000u) v.u = 0x7FC00000u;
        else v.u = ((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16) << 16;
        dst[i] = v.f;
    }
}

/* canonical order: a ranks before b */
static inline int before(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

typedef struct {
    float s;
    int64_t id;
} ent_t;

static int ent_cmp(const void* pa, const void* pb) {
    const ent_t* a = (const ent_t*)pa;
    const ent_t* b = (const ent_t*)pb;
    if (before(a->s, a->id, b->s, b->id)) return -1;
    if (before(b->s, b->id, a->s, a->id)) return 1;
    return 0;
}

/* top-k of n entries under the canonical order (bounded insertion into a sorted array) */
static int select_topk(const float* s, const int64_t* ids, int64_t n, int64_t id_base, int k, int strict_positive,
                       ent_t* out) {
    int cnt = 0;
    for (int64_t j = 0; j < n; ++j) {
        float sj = s[j];
        int64_t ij = ids ? ids[j] : id_base + j;
        if (sj != sj) continue;                       /* NaN never selected */
        if (strict_positive && !(sj > 0.0f)) continue; /* zero sentinel: index.py:192-193 */
        if (cnt == k && !before(sj, ij, out[k - 1].s, out[k - 1].id)) continue;
        int pos = cnt < k ? cnt : k - 1;
        while (pos > 0 && before(sj, ij, out[pos - 1].s, out[pos - 1].id)) {
            out[pos] = out[pos - 1];
            --pos;
        }
        out[pos].s = sj;
        out[pos].id = ij;
        if (cnt < k) ++cnt;
    }
    return cnt;
}

/*
 * Per-field exhaustive top-k == DenseFlatIndex.retrieve_batch (index.py:181-222) for one field matrix V[D,E].
 * sentinel != 0: lists are seeded with k x (row 0, 0.0) exactly like index.py:192-193, so only strictly positive
 *   scores enter and short lists are padded with (id 0, 0.0).
 * sentinel == 0: "clean" mode, padded with (id -1, -inf) when D < k.
 * ids are row_offset + local row.  Output [Q,k] sorted canonically.
 */
int mfar_oracle_retrieve(const float* V, int64_t D, int E, const float* q, int Q, int k, int sentinel,
                         int64_t row_offset, int64_t* ids, float* scores) {
    if (k <= 0 || Q < 0 || D < 0 || E <= 0) return -1;
    float* sc = (float*)malloc((size_t)(Q > 0 ? Q : 1) * (size_t)(D > 0 ? D : 1) * sizeof(float));
    if (!sc) return -3;
    mfar_oracle_scores(V, D, E, q, Q, sc);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < Q; ++i) {
        ent_t* best = (ent_t*)malloc((size_t)k * sizeof(ent_t));
        int cnt = select_topk(sc + (size_t)i * D, NULL, D, row_offset, k, sentinel, best);
        for (int r = 0; r < k; ++r) {
            if (r < cnt) {
                ids[(size_t)i * k + r] = best[r].id;
                scores[(size_t)i * k + r] = best[r].s;
            } else if (sentinel) {
                ids[(size_t)i * k + r] = 0;
                scores[(size_t)i * k + r] = 0.0f;
            } else {
                ids[(size_t)i * k + r] = -1;
                scores[(size_t)i * k + r] = -INFINITY;
            }
        }
        free(best);
    }
    free(sc);
    return 0;
}

/* Deterministic expf for x <= 0 built only from IEEE mul / fma / rint and exponent-field arithmetic,
 * so the HIP device function of the same name returns the same bits. */
float mfar_oracle_exp(float x) {
    if (!(x > -80.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;
    const float t = x * 1.44269504088896341f;
    const float n = rintf(t);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723e-6f, r);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, r, 1.0f / 720.0f);
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    union {
        uint32_t u;
        float f;
    } sc;
    sc.u = (uint32_t)((int)n + 127) << 23;
    return p * sc.f;
}

/* field weights = softmax(q @ W) (query_cond) or softmax(W^T) (not query_cond; W then has F entries).
 * weighting.py:24-28 */
void mfar_oracle_gate(const float* q, const float* W, int E, int F, int query_cond, float* w /*[F]*/) {
    float m = -INFINITY;
    for (int f = 0; f < F; ++f) {
        float z;
        if (query_cond) {
            z = 0.0f;
            for (int e = 0; e < E; ++e) z = fmaf(q[e], W[(size_t)e * F + f], z);
        } else {
            z = W[f];
        }
        w[f] = z;
        if (z > m) m = z;
    }
    float sum = 0.0f;
    for (int f = 0; f < F; ++f) {
        w[f] = mfar_oracle_exp(w[f] - m);
        sum = sum + w[f];
    }
    for (int f = 0; f < F; ++f) w[f] = w[f] / sum;
}

/* LinearWeights.forward for one query: out[c] = sum_f w_f * (x[c,f] * mask_f)   (weighting.py:29, contrastive.py:686) */
void mfar_oracle_mix(const float* x /*[C,F]*/, int C, int F, const float* w, const float* mask, float* out) {
    for (int c = 0; c < C; ++c) {
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) {
            float xm = x[(size_t)c * F + f] * (mask ? mask[f] : 1.0f);
            acc = fmaf(w[f], xm, acc);
        }
        out[c] = acc;
    }
}

/* score_batch (index.py:227-232) over the whole slab: out[Q,C,F] for candidate ids cand[Q,C] (global ids;
 * entries < 0 or outside [row_offset, row_offset+D) give NaN). */
int mfar_oracle_score_candidates(const float* slab /*[F,D,E]*/, int F, int64_t D, int E, int64_t row_offset,
                                 const float* q, int Q, const int64_t* cand, int C, float* out) {
    for (int i = 0; i < Q; ++i)
        for (int c = 0; c < C; ++c) {
            int64_t id = cand[(size_t)i * C + c] - row_offset;
            for (int f = 0; f < F; ++f) {
                float v = NAN;
                if (id >= 0 && id < D) v = mfar_oracle_dot(q + (size_t)i * E, slab + ((size_t)f * D + id) * E, E);
                out[((size_t)i * C + c) * F + f] = v;
            }
        }
    return 0;
}

static int i64_cmp(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/*
 * The whole trec_eval_step scorer (contrastive.py:669-704) for a batch of Q queries.
 *   slab [F,D,E] fp32, q [Q,E], W [E,F] (or [F] when !query_cond), mask [F] or NULL.
 *   out: ids/scores [Q,k2] canonical; n_valid[Q] = min(C_q, k2) (the reference raises when C_q < k2,
 *        contrastive.py:696; rows beyond n_valid are (-1, -inf));
 *   optional field_ids/field_scores [Q,F,k1] (stage-1 lists), optional n_cand[Q].
 */
int mfar_oracle_two_stage(const float* slab, int F, int64_t D, int E, const float* q, int Q, const float* W,
                          int query_cond, const float* mask, int k1, int k2, int sentinel, int64_t* ids, float* scores,
                          int32_t* n_valid, int64_t* field_ids, float* field_scores, int32_t* n_cand) {
    if (F <= 0 || k1 <= 0 || k2 <= 0) return -1;
    size_t lsz = (size_t)Q * F * k1;
    int64_t* fid = field_ids ? field_ids : (int64_t*)malloc(lsz * sizeof(int64_t));
    float* fsc = field_scores ? field_scores : (float*)malloc(lsz * sizeof(float));
    int64_t* tid = (int64_t*)malloc((size_t)Q * k1 * sizeof(int64_t));
    float* tsc = (float*)malloc((size_t)Q * k1 * sizeof(float));
    /* stage 1: contrastive.py:672-674 */
    for (int f = 0; f < F; ++f) {
        int rc = mfar_oracle_retrieve(slab + (size_t)f * D * E, D, E, q, Q, k1, sentinel, 0, tid, tsc);
        if (rc) return rc;
        for (int i = 0; i < Q; ++i)
            for (int r = 0; r < k1; ++r) {
                fid[((size_t)i * F + f) * k1 + r] = tid[(size_t)i * k1 + r];
                fsc[((size_t)i * F + f) * k1 + r] = tsc[(size_t)i * k1 + r];
            }
    }
    free(tid);
    free(tsc);
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < Q; ++i) {
        /* union of ids: contrastive.py:678-679 (sorted ascending here; the reference's set order is arbitrary) */
        int n = F * k1;
        int64_t* u = (int64_t*)malloc((size_t)n * sizeof(int64_t));
        memcpy(u, fid + (size_t)i * F * k1, (size_t)n * sizeof(int64_t));
        qsort(u, (size_t)n, sizeof(int64_t), i64_cmp);
        int C = 0;
        for (int j = 0; j < n; ++j) {
            if (u[j] < 0) continue; /* clean-mode padding */
            if (C == 0 || u[j] != u[C - 1]) u[C++] = u[j];
        }
        /* stage 2: contrastive.py:681-686 */
        float* x = (float*)malloc((size_t)(C > 0 ? C : 1) * F * sizeof(float));
        for (int c = 0; c < C; ++c)
            for (int f = 0; f < F; ++f)
                x[(size_t)c * F + f] = mfar_oracle_dot(q + (size_t)i * E, slab + ((size_t)f * D + u[c]) * E, E);
        float* w = (float*)malloc((size_t)F * sizeof(float));
        float* mixed = (float*)malloc((size_t)(C > 0 ? C : 1) * sizeof(float));
        mfar_oracle_gate(q + (size_t)i * E, W, E, F, query_cond, w);
        mfar_oracle_mix(x, C, F, w, mask, mixed);
        /* final topk: contrastive.py:696 */
        ent_t* best = (ent_t*)malloc((size_t)k2 * sizeof(ent_t));
        int cnt = select_topk(mixed, u, C, 0, k2, 0, best);
        for (int r = 0; r < k2; ++r) {
            ids[(size_t)i * k2 + r] = r < cnt ? best[r].id : -1;
            scores[(size_t)i * k2 + r] = r < cnt ? best[r].s : -INFINITY;
        }
        if (n_valid) n_valid[i] = cnt;
        if (n_cand) n_cand[i] = C;
        free(best);
        free(mixed);
        free(w);
        free(x);
        free(u);
    }
    if (!field_ids) free(fid);
    if (!field_scores) free(fsc);
    return 0;
}

/* Merge S per-shard stage-1 lists into the global list (canonical top-k of the union; sentinel entries
 * (id 0, 0.0) and clean-mode padding (id -1) need no special casing other than dropping id < 0). */
int mfar_oracle_merge_lists(const int64_t* ids /*[S,k]*/, const float* scores, int S, int k, int sentinel,
                            int64_t* out_ids, float* out_scores) {
    ent_t* all = (ent_t*)malloc((size_t)S * k * sizeof(ent_t));
    int n = 0;
    for (int j = 0; j < S * k; ++j) {
        if (ids[j] < 0) continue;
        all[n].s = scores[j];
        all[n].id = ids[j];
        ++n;
    }
    qsort(all, (size_t)n, sizeof(ent_t), ent_cmp);
    for (int r = 0; r < k; ++r) {
        if (r < n) {
            out_ids[r] = all[r].id;
            out_scores[r] = all[r].s;
        } else {
            out_ids[r] = sentinel ? 0 : -1;
            out_scores[r] = sentinel ? 0.0f : -INFINITY;
        }
    }
    free(all);
    return 0;
}
