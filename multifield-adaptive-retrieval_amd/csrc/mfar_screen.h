// mfar_screen.h -- certified fp16 screening for fp32 indexes.
//
// Stage 1 on an fp32 slab is bound by the fp32 MFMA rate.  The screen turns it into an HBM-bound pass at half the bytes
// WITHOUT changing a single output bit:
//
//   1. the index keeps a second, fp16 copy of its rows (the "screen slab", same tiled layout as the bf16 slab): per field
//      the rows are CENTRED on the field's mean vector m (real embeddings share a large common component; q.m shifts every
//      score of the field by the same amount, so it cannot change the ranking, and the error bound then scales with
//      |d - m| instead of |d|) and scaled by a power of two so that the largest |d_i - m_i| lands in [2^13, 2^14);
//   2. stage 1 runs on the screen slab (v_mfma_f32_32x32x16_f16, queries scaled per query by a power of two and split into
//      two fp16 terms) and keeps the k' = min(k + 92, 192) best APPROXIMATE scores per (query, field);
//   3. those k' rows are re-scored from the fp32 slab with the exact fma chain of the arithmetic contract
//      (mfar_score_candidates_kernel, per-field mode) and the exact top-k is taken from them;
//   4. the result is CERTIFIED: with eps(q, f) a rigorous bound of |approx - exact| for every row of the field, every row
//      outside the k' has exact score <= approx_k' + eps; when that is < the exact k-th best, no outside row can enter
//      or tie into the top-k, so the list equals the exhaustive fp32 result bit for bit.  Otherwise the field is flagged
//      and the exact fp32 MFMA pass (mfar_stage1_kernel) re-runs for that field only -- always launched, its workgroups
//      exit at once when the flag is clear, so there is no host round trip.
//
// Error bound (K = dim, u16 = 2^-11, u32 = 2^-24; c = fl(d - m) the centred row, any vector m is valid; scaled operands
// Qi = qi * sq, Di = ci * sf, powers of two = exact; approx = q.m (fp32) + MFMA sum / (sq sf)):
//   centring          |fl(di - mi) - (di - mi)| <= u32 |ci|;  q.m computed in fp32: K u32 sum|qi||mi|
//   doc rounding      |fp16(Di) - Di| <= u16 |Di| + 2^-25            (normal / subnormal fp16)
//   query split       Qi = A + B + r,  |r| <= u16^2 |Qi| + 2^-25
//   accumulation      the MFMA sums 2K exact products in fp32; we allow 2 u32 per addition in ANY order: (4K + 64) u32
//   exact chain       the contract's fma chain itself: K u32
//   => |approx - exact| <= [1.02 u16 + (4K + 66) u32] * sum|qi||ci| + K u32 (sum|qi||di| + sum|qi||mi|)
//                          + 2^-24 (|q|_1 / sf + |c|_1 / sq)
//   with sum|qi||xi| <= |q|_2 |x|_2,  |x|_1 <= sqrt(K) |x|_2,  |c|_2 <= the field's largest centred row norm,  |d| <= |c| + |m|.
// The constant is multiplied by SCREEN_SLACK for margin; tests measure the real error (about 30x below the bound).
#pragma once
#include "mfar_device.h"
#include "mfar_stage1.h"

#define SCREEN_EXTRA_MIN 64       // the screen is used only when at least this margin fits: k + 64 <= SCREEN_MAX_KP
#define SCREEN_EXTRA 92          // k' = min(k + SCREEN_EXTRA, SCREEN_MAX_KP): 192 for the reference's k = 100 (a wider margin
                                 // costs ~1 % and makes a failed certificate -- a 0.9 ms exact pass per field -- rarer)
#define SCREEN_MAX_KP S1_MAX_DEPTH   // the stage-1 lists compact to k', which must leave room for one tile of appends
#define SCREEN_SLACK 1.25f
#define SCREEN_FLAGS (4 * MFAR_MAX_FIELDS + 12)  // ints of a batch's certificate flags (CertifyParams::fail)
#define SCREEN_FLAG_T2_WANT (4 * MFAR_MAX_FIELDS + 11)      // per batch: a list asked for the rescan while it was not armed (it went to the exact
                                                            // pass; the policy arms the rescan: mfar_policy.h; cleared by the query kernel)
#define SCREEN_STAT_T2_RESCAN (3 * MFAR_MAX_FIELDS + 9)     // statistics: lists of tier 2 whose candidates needed the RESCAN ...
#define SCREEN_STAT_T2_FIRST (3 * MFAR_MAX_FIELDS + 10)     // ... / were taken from the chunk lists the batch's own scan had written
#define SCREEN_T2_RESCAN_FIELDS (3 * MFAR_MAX_FIELDS + 11)  // [MFAR_MAX_FIELDS] per batch: field f holds such a list (selects the rescan's
                                                            // fields; cleared by the query kernel)
#define SCREEN_STAT_T2_OVF (3 * MFAR_MAX_FIELDS + 5)   // statistics [4]: why tier 2 passed lists on -- a chunk list reached its depth / more than
                                                       // T2_CAP_IN rows above the threshold / more than T2_CAP candidates in the band / ties at the cut
#define SCREEN_T2_FIELDS (2 * MFAR_MAX_FIELDS + 5)     // [MFAR_MAX_FIELDS] per batch: field f has lists for tier 2 (the policy's input; cleared by the query kernel)
#define SCREEN_FLAG_T1 (2 * MFAR_MAX_FIELDS + 2)       // per batch: some list failed the FIRST certificate (tier 2 had work; cleared by the query kernel)
#define SCREEN_STAT_T2_LISTS (2 * MFAR_MAX_FIELDS + 3) // statistics: lists handed to tier 2 ...
#define SCREEN_STAT_T2_FAILED (2 * MFAR_MAX_FIELDS + 4)   // ... and lists tier 2 could not finish either (overflow: the exact pass decides)
// TIER 2 of the certified screen ("threshold rescan", below): candidates one list can hold; lists that need more go to the exact pass
#define T2_CAP 2048

// struct ScreenField: mfar_device.h (shared with the gather-slab kernels of mfar_select.h)
// struct ScreenQuery: mfar_device.h

// ---------------------------------------------------------------------------------------------------------
// Build, step 1: the per-field mean vector (row-major [F][E]).  Its exact value is irrelevant for correctness -- ANY
// vector m works, it only has to be the same one everywhere -- so plain float atomics are fine.
//   grid = (ceil(n_blk / 8), F), block 256; thread eq owns dims 4 eq .. 4 eq + 3 and walks the rows of 8 blocks.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfar_screen_mean_kernel(const float* __restrict__ slab, long long field_stride, int n_steps,
                                                               long long n_blk, long long n_rows, float* __restrict__ acc) {
    const int f = blockIdx.y, E = n_steps * 16;
    const long long b0 = (long long)blockIdx.x * 8, b1 = b0 + 8 < n_blk ? b0 + 8 : n_blk;
    for (int eq = threadIdx.x; eq < E / 4; eq += blockDim.x) {
        const int e = eq * 4;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (long long b = b0; b < b1; ++b) {
            const long long nr = n_rows - b * 64 < 64 ? n_rows - b * 64 : 64;
            for (int rr = 0; rr < nr; ++rr) sum += *(const f32x4*)(slab + (size_t)f * field_stride + tiled_offset(n_steps, b * 64 + rr, e));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(&acc[(size_t)f * E + e + i], sum[i]);
    }
}
__global__ void mfar_screen_mean_finish_kernel(float* __restrict__ acc, int n, long long n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = n_rows > 0 ? acc[i] / (float)n_rows : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// Build, step 2: per-field statistics of the CENTRED rows, then the conversion.  grid = (n_blk, F), block 256.
// stats[2f] = max |value| bits, stats[2f+1] = max row norm^2 bits (non-negative floats order like their bit patterns;
// NaN bits are above inf bits, so a non-finite value poisons the field's maximum as intended).
// ---------------------------------------------------------------------------------------------------------
//   rnorm [n_rows] or nullptr: the centred 2-norm of every row (leaning up like dnorm_max);  nsum: sum of those norms (any order: a statistic)
__global__ void __launch_bounds__(256) mfar_screen_stats_kernel(const float* __restrict__ slab, long long field_stride, int n_steps,
                                                                long long n_rows, const float* __restrict__ mean,
                                                                u32* __restrict__ stats, float* __restrict__ rnorm, float* __restrict__ nsum) {
    const int f = blockIdx.y;
    const int rr = threadIdx.x >> 2, pp = threadIdx.x & 3;   // thread = quarter pp (8 floats) of row rr's 128-byte line per k-step pair
    const bool live = (long long)blockIdx.x * 64 + rr < n_rows;   // padding rows are not part of the field
    const float* tile = slab + (size_t)f * field_stride + (size_t)blockIdx.x * n_steps * 1024 + rr * 32 + pp * 8;
    const float* mrow = mean + (size_t)f * n_steps * 16 + pp * 8;
    float amax = 0.0f, ss = 0.0f;
    if (live)
        for (int pr = 0; pr < (n_steps >> 1); ++pr) {
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const f32x4 v = *(const f32x4*)(tile + (size_t)pr * 2048 + hq * 4) - *(const f32x4*)(mrow + pr * 32 + hq * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    amax = fmaxf(amax, fabsf(v[i]));
                    ss = __builtin_fmaf(v[i], v[i], ss);
                }
            }
        }
    // threads 4r .. 4r+3 hold the four quarters of row r
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    const float rn = sqrtf(ss) * 1.0001f;
    if (rnorm && live && pp == 0) rnorm[(long long)blockIdx.x * 64 + rr] = rn;
    float part = (live && pp == 0) ? rn : 0.0f;
    u32 a = __float_as_uint(amax), n = __float_as_uint(ss) & 0x7FFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) {
        a = max(a, (u32)__shfl_xor((int)a, off));
        n = max(n, (u32)__shfl_xor((int)n, off));
        part += __shfl_xor(part, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[2 * f], a);
        atomicMax(&stats[2 * f + 1], n);
        if (nsum) atomicAdd(nsum, part);
    }
}
// rows of the screen slab are UNIQUE rows: out[u] = rnorm[urep[u]] (padding rows: 0).  grid = ceil(n_pad / 256), block 256.
__global__ void __launch_bounds__(256) mfar_rownorm_gather_kernel(const float* __restrict__ rnorm, const int* __restrict__ urep, int n_unique,
                                                                  long long n_pad, float* __restrict__ out) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n_pad) out[u] = u < n_unique ? rnorm[urep[u]] : 0.0f;
}

// fp32 index: the centred 2-norm of every row, as a 10-bit code in the top bits of its u_of entry (unique number + 1 < 2^22):
// |c_row| <= dnorm_max * (code + 1) / 1024.  The score dump's look-up reads the entry anyway and gets a PER-ROW error bound for free
// (mfar_select.h mfar_s2_lookup_kernel).  Rows without a norm (no table) keep code 1023 = the field-wide bound.
__global__ void __launch_bounds__(256) mfar_uof_norm_code_kernel(long long n, const float* __restrict__ rnorm, const ScreenField* __restrict__ sf,
                                                                 u32* __restrict__ uof) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 code = 1023u;
    const float dmax = sf->dnorm_max;
    if (rnorm && dmax > 0.0f && dmax < __builtin_inff()) {
        const float x = rnorm[i] * (1024.0f / dmax) * 1.0001f;       // code + 1 >= x  =>  the coded norm is an upper bound
        if (x >= 0.0f && x < 1023.0f) code = (u32)x;
    }
    uof[i] = (uof[i] & UOF_INDEX_MASK) | (code << UOF_NORM_SHIFT);
}

// power-of-two scale that puts `amax` into [2^13, 2^14); 1 for zero / non-finite input
__device__ __forceinline__ float screen_pow2_scale(float amax) {
    const u32 b = __float_as_uint(amax) & 0x7FFFFFFFu;
    if (b == 0u || b >= 0x7F800000u) return 1.0f;
    int e = 13 - ((int)(b >> 23) - 127);
    e = max(-100, min(100, e));
    return __uint_as_float((u32)(e + 127) << 23);
}

__global__ void mfar_screen_scale_kernel(const u32* __restrict__ stats, const float* __restrict__ mean, int F, int E,
                                         ScreenField* __restrict__ sf, const float* __restrict__ nsum, long long n_rows, int row_mode_allowed) {
    const int f = threadIdx.x;
    if (f >= F) return;
    const float amax = __uint_as_float(stats[2 * f]);
    const float n2 = __uint_as_float(stats[2 * f + 1]);
    float m2 = 0.0f;
    for (int e = 0; e < E; ++e) m2 = __builtin_fmaf(mean[(size_t)f * E + e], mean[(size_t)f * E + e], m2);
    ScreenField o;
    o.scale = screen_pow2_scale(amax);
    o.inv_scale = 1.0f / o.scale;
    // the fp32 sums of squares can be low by K u32 relative: lean up, the bound must not shrink
    o.dnorm_max = sqrtf(n2) * 1.0001f;
    o.mnorm = sqrtf(m2) * 1.0001f;
    o.dnorm_mean = (nsum && n_rows > 0) ? nsum[f] / (float)n_rows : o.dnorm_max;
    // ROW MODE (below): worth its epilogue cost only when the largest norm is far above the typical one
    o.row_mode = (row_mode_allowed && o.dnorm_max < __builtin_inff() && o.dnorm_max > 1.5f * o.dnorm_mean) ? 1.0f : 0.0f;
    sf[f] = o;
}

// fp32 tiled slab -> fp16 tiled screen slab of ONE field (centred, scaled), unique rows only: output row u is the
// document urep[u].  One thread per 16-byte output granule (8 dims).  grid = ceil(n_blk_u * n_steps * 128 / 256).
__global__ void __launch_bounds__(256) mfar_screen_build_kernel(const float* __restrict__ field, _Float16* __restrict__ out,
                                                                long long n_granules, int n_steps, int n_unique,
                                                                const int* __restrict__ urep, const float* __restrict__ mean,
                                                                const ScreenField* __restrict__ sf) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_granules) return;
    const float sc = sf->scale;
    const long long tile = g >> 7;           // 128 granules per [64][16] fp16 tile = (block, k-step)
    const int rr = (int)(g >> 1) & 63, c8 = (int)g & 1;
    const int step = (int)(tile % n_steps);
    const long long blk = tile / n_steps;
    const long long u = blk * 64 + rr;
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)0.0f;
    if (u < n_unique) {
        const float* src = field + tiled_offset(n_steps, urep[u], step * 16 + c8 * 8);
        const float* m = mean + step * 16 + c8 * 8;
        const f32x4 a = *(const f32x4*)src - *(const f32x4*)m;
        const f32x4 b = *(const f32x4*)(src + 4) - *(const f32x4*)(m + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = (_Float16)(a[i] * sc);
            o[4 + i] = (_Float16)(b[i] * sc);
        }
    }
    *(f16x8*)(out + (size_t)tile * 1024 + rr * 16 + ((c8 ^ ((rr >> 3) & 1)) << 3)) = o;
}

// ---------------------------------------------------------------------------------------------------------
// Unique rows.  Real fields are full of bit-identical rows: a field a document lacks is encoded from the empty string
// (format.py:58-59), low-cardinality fields (STaRK-prime `type` / `source`, amazon `brand`; schema.py:11-53) repeat a handful
// of texts over the whole corpus, and the host encodes every distinct text once, so repeated texts ARE bit-identical rows.
// Equal rows get equal approximate scores: a list full of them cannot separate its k-th from its k'-th entry, and every
// certificate of such a field would fail.  The screen therefore scans each DISTINCT vector of a field once:
//
//   build   hash every row (mfar_row_hash_kernel) -> stable radix sort of (hash, row) -> a row starts a new group when its
//           hash or -- compared bit for bit -- its vector differs from its predecessor's (mfar_group_heads_kernel; a hash
//           collision merely splits a group, which costs a little bandwidth and never correctness) -> groups are numbered in
//           the order of their lowest row ("unique row" u of the field; u ascending == representative ascending).
//           Kept per field: urep[u] (lowest member = representative document), ustart[u] / ucount[u] into `members` (the
//           rows sorted by group, ascending inside a group), and the fp16 screen slab built from the unique rows only.
//   scan    the screened pass ranks unique rows; fields with few distinct vectors collapse to a few tiles.
//   certify the exact top-k of the re-scored unique rows is EXPANDED to documents: a unique row with exact score s stands for
//           ucount documents with score s; the k best (score desc, doc id asc) of the expanded entries are the list.  The
//           certificate compares the bound on every unscanned unique row with the k-th best DOCUMENT.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 row_mix(u64 x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// grid = n_blk, block 256 (thread = quarter pp of row rr's 128-byte line per k-step pair); field = base of one field
__global__ void __launch_bounds__(256) mfar_row_hash_kernel(const float* __restrict__ field, int n_steps, long long n_rows,
                                                            u64* __restrict__ keys, u32* __restrict__ vals) {
    const int rr = threadIdx.x >> 2, pp = threadIdx.x & 3;
    const long long row = (long long)blockIdx.x * 64 + rr;
    const float* tile = field + (size_t)blockIdx.x * n_steps * 1024 + rr * 32 + pp * 8;
    u64 h = 0;
    if (row < n_rows)
        for (int pr = 0; pr < (n_steps >> 1); ++pr) {
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const f32x4 v = *(const f32x4*)(tile + (size_t)pr * 2048 + hq * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    h += row_mix((u64)__float_as_uint(v[i]) + (u64)(pr * 32 + pp * 8 + hq * 4 + i + 1) * 0x9E3779B97F4A7C15ull);
            }
        }
    // threads 4r .. 4r+3 hold the four quarters of row r (the sum is order independent)
    h += ((u64)(u32)__shfl_xor((int)(u32)h, 1)) | ((u64)(u32)__shfl_xor((int)(u32)(h >> 32), 1) << 32);
    h += ((u64)(u32)__shfl_xor((int)(u32)h, 2)) | ((u64)(u32)__shfl_xor((int)(u32)(h >> 32), 2) << 32);
    if (pp == 0 && row < n_rows) {
        keys[row] = h;
        vals[row] = (u32)row;
    }
}

// head[i] = 1 when sorted position i starts a new group.  8 threads per position.  grid = ceil(n * 8 / 256), block 256.
__global__ void __launch_bounds__(256) mfar_group_heads_kernel(const float* __restrict__ field, int n_steps, long long n,
                                                               const u64* __restrict__ keys, const u32* __restrict__ vals,
                                                               u32* __restrict__ head) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = gid >> 3;
    const int part = (int)gid & 7;
    bool differ = false, live = i < n;
    if (live) {
        if (i == 0 || keys[i] != keys[i - 1]) differ = true;
        else {
            const long long a = vals[i], b = vals[i - 1];
            const int E = n_steps * 16;
            for (int e = part * 4; e < E && !differ; e += 32) {
                const f32x4 x = *(const f32x4*)(field + tiled_offset(n_steps, a, e));
                const f32x4 y = *(const f32x4*)(field + tiled_offset(n_steps, b, e));
#pragma unroll
                for (int c = 0; c < 4; ++c) differ = differ || __float_as_uint(x[c]) != __float_as_uint(y[c]);
            }
        }
    }
    int d = differ ? 1 : 0;
    d |= __shfl_xor(d, 1);
    d |= __shfl_xor(d, 2);
    d |= __shfl_xor(d, 4);
    if (live && part == 0) head[i] = (u32)d;
}

// gid = inclusive scan of head (1-based group number per sorted position).  For every group: its start in the sorted order
// and its lowest row (the sort is stable, rows ascend inside a group); the lowest row is flagged in is_rep[row].
__global__ void __launch_bounds__(256) mfar_group_starts_kernel(long long n, const u32* __restrict__ head, const u32* __restrict__ gid,
                                                                const u32* __restrict__ vals, int* __restrict__ gstart,
                                                                u32* __restrict__ is_rep) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    gstart[gid[i] - 1] = (int)i;
    is_rep[vals[i]] = 1u;
}
// urank = inclusive scan of is_rep over the rows: unique number (1-based) of a representative row.
__global__ void __launch_bounds__(256) mfar_unique_table_kernel(long long n, int n_groups, const int* __restrict__ gstart,
                                                                const u32* __restrict__ vals, const u32* __restrict__ urank,
                                                                int* __restrict__ urep, int* __restrict__ ustart, int* __restrict__ ucount) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int s = gstart[g], e = g + 1 < n_groups ? gstart[g + 1] : (int)n;
    const int rep = (int)vals[s];
    const int u = (int)urank[rep] - 1;
    urep[u] = rep;
    ustart[u] = s;
    ucount[u] = e - s;
}

// repof[row] = the representative (lowest row) of row's group of bit-identical rows.  Stage 2 gathers the representative
// instead of the row itself (same bits): the rows a low-cardinality field repeats over the corpus -- the empty-string vector
// of a field most documents lack -- then come from the caches instead of HBM.  grid = ceil(n / 256), block 256.
__global__ void __launch_bounds__(256) mfar_rep_of_kernel(long long n, const u32* __restrict__ gid, const int* __restrict__ gstart,
                                                          const u32* __restrict__ vals, int* __restrict__ repof) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    repof[vals[i]] = (int)vals[gstart[gid[i] - 1]];
}

// ---------------------------------------------------------------------------------------------------------
// Queries of one block of qw = 64 or 128 queries: per-query scale + norm, the fp16 tiles, and per (field, query): eps (real
// units) and the starting threshold of the screened pass (scaled units).
//   qw == 64:  two-term split (hi + lo = 22 significant bits), tiles [n_steps][2 terms][64][16];
//   qw == 128: ONE fp16 term per query (the wide pass, mfar_stage1.h), tiles [n_steps][2 query blocks][64][16]; the query
//              rounding error |fp16(Qi) - Qi| <= u16 |Qi| + 2^-25 enters the bound at first order:
//              |A D' - Qi Di| <= (2 u16 + u16^2) |Qi||Di| + 2^-25 (1 + u16) (|Qi| + |Di|).
//   grid = qw (one workgroup per query row), block 256.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfar_screen_queries_kernel(const float* __restrict__ q, _Float16* __restrict__ qt,
                                                                  ScreenQuery* __restrict__ qinfo, const ScreenField* __restrict__ sf,
                                                                  float* __restrict__ eps, float* __restrict__ tau_base,
                                                                  int* __restrict__ fail_flags, int q0, int Q, int E, int F,
                                                                  float eps_mult, int qw, int direct, float* __restrict__ arow,
                                                                  float* __restrict__ eps_cert, u32 row_mask, float* __restrict__ dump_inv,
                                                                  float* __restrict__ dump_step, float* __restrict__ eps_dump,
                                                                  float* __restrict__ dump_arel) {
    __shared__ float red_a[4], red_s[4];
    const int r = blockIdx.x;
    // a new batch: clear the certificate flags of the fields and the "any" flag ([MFAR_MAX_FIELDS + 1] keeps accumulating statistics)
    if (r == 0 && (int)threadIdx.x <= MFAR_MAX_FIELDS) fail_flags[threadIdx.x] = 0;
    if (r == 0 && (int)threadIdx.x < MFAR_MAX_FIELDS) fail_flags[MFAR_MAX_FIELDS + 2 + threadIdx.x] = 0;      // ... and the probe flags
    if (r == 0 && threadIdx.x == 0) fail_flags[SCREEN_FLAG_T1] = fail_flags[SCREEN_FLAG_T2_WANT] = 0;
    if (r == 0 && (int)threadIdx.x < MFAR_MAX_FIELDS) fail_flags[SCREEN_T2_FIELDS + threadIdx.x] = fail_flags[SCREEN_T2_RESCAN_FIELDS + threadIdx.x] = 0;
    const bool live = q0 + r < Q;
    const float* row = q + (size_t)(q0 + (live ? r : 0)) * E;
    // ONE round of global loads (this kernel opens a batch on the critical path, usually while the previous batch's gathers
    // saturate the memory system: every dependent round trip costs tens of microseconds there): each thread keeps its up to
    // 8 granules of 8 query values in registers, the field constants are fetched alongside
    constexpr int NG = 8;   // dim <= 8 * 256 * NG
    const int gpr = E >> 3;
    f32x4 va[NG][2];
    ScreenField fld = {};
    if ((int)threadIdx.x < F) fld = sf[threadIdx.x];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int g = (int)threadIdx.x + 256 * i;
        va[i][0] = va[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live && g < gpr) {
            va[i][0] = *(const f32x4*)(row + g * 8);
            va[i][1] = *(const f32x4*)(row + g * 8 + 4);
        }
    }
    float amax = 0.0f, ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = va[i][j >> 2][j & 3];
            amax = fmaxf(amax, fabsf(v));
            ss = __builtin_fmaf(v, v, ss);
        }
    for (int off = 32; off > 0; off >>= 1) {
        amax = fmaxf(amax, __shfl_xor(amax, off));
        ss += __shfl_xor(ss, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red_a[threadIdx.x >> 6] = amax;
        red_s[threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    amax = fmaxf(fmaxf(red_a[0], red_a[1]), fmaxf(red_a[2], red_a[3]));
    ss = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const float sq = screen_pow2_scale(amax);
    const float qn = sqrtf(ss) * 1.0001f;   // the fp32 sum of squares can be low by K u32 relative: lean up
    // split tiles
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int g = (int)threadIdx.x + 256 * i;
        if (g >= gpr) continue;
        const int e = g << 3;
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = va[i][j >> 2][j & 3] * sq;
            const _Float16 a = (_Float16)x;
            hi[j] = a;
            lo[j] = (_Float16)(x - (float)a);
        }
        const int step = e >> 4;
        const size_t in_tile = tiled_offset_bf16(E >> 4, r & 63, e) - (size_t)step * 1024;
        _Float16* base = qt + (size_t)step * 2048;   // 2 tiles of 1024 halves per k-step
        if (qw == 128) *(f16x8*)(base + (r >> 6) * 1024 + in_tile) = hi;      // tile = query block
        else {
            *(f16x8*)(base + in_tile) = hi;                                   // tile = term
            *(f16x8*)(base + 1024 + in_tile) = lo;
        }
    }
    if (threadIdx.x == 0) {
        ScreenQuery o;
        o.scale = sq;
        o.inv_scale = 1.0f / sq;
        o.norm = qn;
        o.pad = 0.0f;
        qinfo[r] = o;
    }
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        const ScreenField s = fld;
        const float K = (float)E, u32f = 5.9604645e-8f;
        // direct (the converted-docs pass over a bf16 slab, qw == 128): the docs are exact -- no rounding, no centring (mnorm = 0) -- up
        // to the clamp of magnitudes below 2^-14 scaled units; what remains is the one-term query rounding
        const float c16 = direct ? 1.02f : (qw == 128 ? 2.04f : 1.02f);
        const float c_rel = c16 * 4.8828125e-4f + (4.0f * K + 66.0f) * u32f;
        const float c_abs = u32f * sqrtf(K) * 1.0001f;
        const float c_doc = direct ? 6.1035156e-5f * 1.001f * sqrtf(K) * 1.0001f : c_abs;      // 2^-14 (1 + u16) |q|_1 / sf
        const float e_rest = K * u32f * qn * (s.dnorm_max + 2.0f * s.mnorm) + c_doc * qn * s.inv_scale + c_abs * s.dnorm_max / sq;
        float e_ = SCREEN_SLACK * (c_rel * qn * s.dnorm_max + e_rest);
        e_ *= eps_mult;
        if (!live) e_ = 0.0f;
        eps[f * qw + r] = e_;
        // ROW MODE.  The dominant term of the bound is c_rel |q|_2 |c_row|_2 -- per ROW once the row's own norm replaces the field's
        // largest.  The scan then adds arow * |c_row| to every score (scaled units; mfar_stage1.h s1_row_bound), so its lists are ranked
        // by an UPPER bound of the exact score up to the row-independent rest, and the certificate compares the k'-th upper bound +
        // eps_cert with the exact k-th best.  eps (above) stays the field-wide bound: |upper bound - exact| <= 2 eps for every row, which
        // is all the re-scoring prefix rule and stage 2 need.  (1.001: the fp32 rounding of the added term itself.)
        if (arow) {
            const bool rm = ((row_mask >> f) & 1u) != 0u && live;         // (the fields whose scans add the row term in this batch)
            arow[f * qw + r] = rm ? 1.001f * SCREEN_SLACK * eps_mult * c_rel * qn * (sq * s.scale) : 0.0f;
            eps_cert[f * qw + r] = rm ? SCREEN_SLACK * eps_mult * e_rest : e_;
        }
        // SCORE DUMP (S1Params::dump): the scan stores a / B as a 16-bit signed-normalised code, B >= |a| for every row: the MFMA sums
        // products of the fp16 query tile (2-norm <= sq |q| (1 + u16)) and fp16 rows (2-norm <= sf max|c| (1 + u16) + tiny) in fp32.  The
        // code's error, B / 65534 in scaled units, joins the bound of stage 2's approximate level (eps_dump, real units).
        if (dump_inv) {
            const float B = 1.01f * (sq * qn) * (s.scale * s.dnorm_max);
            dump_inv[f * qw + r] = live && B > 0.0f && B < __builtin_inff() ? 1.0f / B : 0.0f;
            dump_step[f * qw + r] = live && B > 0.0f ? B * (1.0f / 32767.0f) : 0.0f;        // (B = inf: codes 0 x inf = NaN = "unknown", survives)
            eps_dump[f * qw + r] = e_ + eps_mult * SCREEN_SLACK * 1.03f * qn * s.dnorm_max * (1.0f / 65534.0f);
            // the part of the bound that follows the ROW's norm, per 1/1024 of the field's largest: a row whose coded norm is
            // (code + 1) / 1024 of it gets eps_dump - dump_arel * (1023 - code)   (0.9999: the subtraction's own rounding)
            dump_arel[f * qw + r] = live ? 0.9999f * eps_mult * SCREEN_SLACK * c_rel * qn * s.dnorm_max * (1.0f / 1024.0f) : 0.0f;
        }
        // starting threshold of the screened pass: none for live queries (the zero sentinel of index.py:192-193 is applied
        // to the EXACT scores by the certify kernel; deciding it here would need q.m on the critical path), +inf for the
        // padding queries of a short batch so that they append nothing
        tau_base[f * qw + r] = live ? -__builtin_inff() : __builtin_inff();
    }
}

// ---------------------------------------------------------------------------------------------------------
// bf16 indexes: the certified pass needs NO second copy of the rows.  The docs of the scan are the index's own bf16 values
// (exact), the query is split into two bf16 terms Q = hi + mid (|q_i - Q_i| <= 2^-16 |q_i|), every product is exact in fp32:
//   |approx - exact chain| <= [2^-16 + (4 K + 66) u32 + K u32] sum|q_i||d_i| + tiny        (query residual; fp32 accumulation of
//   2 K products in any order at 2 u32 per addition; the exact natural-order chain's own K u32; "tiny" covers bf16 / fp32
//   subnormals that an MFMA may flush: 2^-118 (sqrt(K) (|q|_2 + |d|_2) + K)), with sum|q||d| <= |q|_2 max_row |d|_2, times
//   SCREEN_SLACK.  No centring, no scaling (bf16 has fp32's exponent range): ScreenField = {1, 1, largest row norm, 0}, the
//   field mean is the zero vector and ScreenQuery = {1, 1, |q|_2}, so the certify kernel runs unchanged.
// The pass scans every document but ranks UNIQUE rows: mfar_rep_bits_kernel packs "real row and representative of its group"
// into one bit per row (S1Params::rep_bits) and uof maps a representative's row to its unique number (CertifyParams::uof).
// ---------------------------------------------------------------------------------------------------------
// largest row norm^2 of one bf16 field -> stats[1] (bits; NaN / inf poison the maximum as in mfar_screen_stats_kernel).
// grid = n_blk, block 128: thread = granule (tid & 1) of row (tid >> 1); the sum of squares does not care about the tile swizzle.
__global__ void __launch_bounds__(128) mfar_direct_stats_kernel(const unsigned short* __restrict__ field, int n_steps, long long n_rows,
                                                                u32* __restrict__ stats) {
    const int rr = threadIdx.x >> 1;
    const bool live = (long long)blockIdx.x * 64 + rr < n_rows;
    const unsigned short* tile = field + (size_t)blockIdx.x * n_steps * 1024 + threadIdx.x * 8;
    float ss = 0.0f;
    u32 a = 0;
    if (live)
        for (int s = 0; s < n_steps; ++s) {
            const bf16x8 v = *(const bf16x8*)(tile + (size_t)s * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float x = bf2f((unsigned short)v[i]);
                ss = __builtin_fmaf(x, x, ss);
                a = max(a, __float_as_uint(x) & 0x7FFFFFFFu);      // |x| as bits (NaN bits sit above inf: a non-finite value poisons the maximum)
            }
        }
    ss += __shfl_xor(ss, 1);
    u32 n = __float_as_uint(ss) & 0x7FFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) {
        n = max(n, (u32)__shfl_xor((int)n, off));
        a = max(a, (u32)__shfl_xor((int)a, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[0], a);
        atomicMax(&stats[1], n);
    }
}
// sf1: the constants of the two-term passes (scale 1);  sfc / cvt: those of the converted-docs pass -- the power-of-two scale that puts
// the field's largest |value| into [2^13, 2^14) as an fp16 number, and the packed-integer constants of the in-register conversion
// (mfar_stage1.h s1_bf16x2_to_f16x2): fp16 exponent field = bf16 exponent field - rebias, rebias = 112 - log2(scale); a magnitude
// below (rebias + 1) << 7 would not be a normal fp16 number and is clamped up to it.
__global__ void mfar_direct_fields_kernel(const u32* __restrict__ stats, int F, ScreenField* __restrict__ sf1, ScreenField* __restrict__ sfc,
                                          uint2* __restrict__ cvt) {
    const int f = threadIdx.x;
    if (f >= F) return;
    ScreenField o;
    o.scale = o.inv_scale = 1.0f;
    o.dnorm_max = sqrtf(__uint_as_float(stats[2 * f + 1])) * 1.0001f;   // (the fp32 sum of squares can be low by K u32 relative)
    o.mnorm = 0.0f;
    o.dnorm_mean = o.dnorm_max;
    o.row_mode = 0.0f;
    sf1[f] = o;
    o.scale = screen_pow2_scale(__uint_as_float(stats[2 * f]));
    o.inv_scale = 1.0f / o.scale;
    sfc[f] = o;
    const int rebias = 112 - ((int)(__float_as_uint(o.scale) >> 23) - 127);       // in [12, 212]: screen_pow2_scale keeps |log2| <= 100
    const u32 b = (u32)rebias << 7, t = (u32)(rebias + 1) << 7;
    cvt[f] = make_uint2(t | (t << 16), b | (b << 16));
}
// bits[b] bit r = row 64 b + r exists and (is_rep == nullptr or is_rep[row]).  grid = n_words, block 64.
__global__ void __launch_bounds__(64) mfar_rep_bits_kernel(const u32* __restrict__ is_rep, long long n_rows, u64* __restrict__ bits) {
    const long long row = (long long)blockIdx.x * 64 + threadIdx.x;
    const u64 m = __ballot(row < n_rows && (!is_rep || is_rep[row] != 0u));
    if (threadIdx.x == 0) bits[blockIdx.x] = m;
}
__global__ void mfar_iota1_kernel(u32* __restrict__ a, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (u32)i + 1u;
}

// Queries of one block of qw = 64 / 128 queries for the certified bf16 pass: two bf16 terms per query,
//   qw == 64:  tiles [n_steps][2 terms][64][16]                 (mfar_stage1_bf16s_kernel);
//   qw == 128: tiles [n_steps][2 terms][2 query blocks][64][16] (mfar_stage1_bf16w_kernel);
// eps per (field, query) in real units, tau_base, the batch's fail flags -- as mfar_screen_queries_kernel.  grid = qw, block 256.
__global__ void __launch_bounds__(256) mfar_direct_queries_kernel(const float* __restrict__ q, unsigned short* __restrict__ qt,
                                                                  ScreenQuery* __restrict__ qinfo, const ScreenField* __restrict__ sf,
                                                                  float* __restrict__ eps, float* __restrict__ tau_base,
                                                                  int* __restrict__ fail_flags, int q0, int Q, int E, int F,
                                                                  float eps_mult, int qw) {
    __shared__ float red_s[4];
    const int r = blockIdx.x;
    if (r == 0 && (int)threadIdx.x <= MFAR_MAX_FIELDS) fail_flags[threadIdx.x] = 0;
    if (r == 0 && (int)threadIdx.x < MFAR_MAX_FIELDS) fail_flags[MFAR_MAX_FIELDS + 2 + threadIdx.x] = 0;      // ... and the probe flags
    if (r == 0 && threadIdx.x == 0) fail_flags[SCREEN_FLAG_T1] = fail_flags[SCREEN_FLAG_T2_WANT] = 0;
    if (r == 0 && (int)threadIdx.x < MFAR_MAX_FIELDS) fail_flags[SCREEN_T2_FIELDS + threadIdx.x] = fail_flags[SCREEN_T2_RESCAN_FIELDS + threadIdx.x] = 0;
    const bool live = q0 + r < Q;
    const float* row = q + (size_t)(q0 + (live ? r : 0)) * E;
    ScreenField fld = {};
    if ((int)threadIdx.x < F) fld = sf[threadIdx.x];
    float ss = 0.0f;
    const int n_tiles_step = qw == 128 ? 4 : 2;                     // 2 KB tiles per k-step
    for (int g = threadIdx.x; g < (E >> 3); g += blockDim.x) {
        const int e = g << 3;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
        if (live) {
            a = *(const f32x4*)(row + e);
            b = *(const f32x4*)(row + e + 4);
        }
        bf16x8 hi, mid;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = i < 4 ? a[i] : b[i - 4];
            ss = __builtin_fmaf(x, x, ss);
            const unsigned short hh = f2bf(x);
            hi[i] = (short)hh;
            mid[i] = (short)f2bf(x - bf2f(hh));                     // the residual of a round-to-nearest split is exact in fp32
        }
        const int step = e >> 4;
        const size_t in_tile = tiled_offset_bf16(E >> 4, r & 63, e) - (size_t)step * 1024;
        unsigned short* base = qt + (size_t)step * 1024 * n_tiles_step;
        if (qw == 128) {                                            // tile = term * 2 + query block
            *(bf16x8*)(base + (size_t)(r >> 6) * 1024 + in_tile) = hi;
            *(bf16x8*)(base + (size_t)(2 + (r >> 6)) * 1024 + in_tile) = mid;
        } else {                                                    // tile = term
            *(bf16x8*)(base + in_tile) = hi;
            *(bf16x8*)(base + 1024 + in_tile) = mid;
        }
    }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red_s[threadIdx.x >> 6] = ss;
    __syncthreads();
    ss = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const float qn = sqrtf(ss) * 1.0001f;
    if (threadIdx.x == 0) {
        ScreenQuery o;
        o.scale = o.inv_scale = 1.0f;
        o.norm = qn;
        o.pad = 0.0f;
        qinfo[r] = o;
    }
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        const float K = (float)E, u32f = 5.9604645e-8f;
        const float c_rel = 1.01f * 1.52587890625e-5f + (5.0f * K + 66.0f) * u32f;
        const float tiny = 3.0e-36f * (sqrtf(K) * (qn + fld.dnorm_max) + K);    // 2^-118 = 3.01e-36
        float e_ = SCREEN_SLACK * (c_rel * qn * fld.dnorm_max + tiny);
        e_ *= eps_mult;
        if (!live) e_ = 0.0f;
        eps[f * qw + r] = e_;
        tau_base[f * qw + r] = live ? -__builtin_inff() : __builtin_inff();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Certify: exact top-k DOCUMENTS of the k' re-scored unique rows of one (query, field) + the certificate.
//   grid = Qt * nf, block 256.
// ---------------------------------------------------------------------------------------------------------
#define CERT_EXPAND_CAP 1024    // expanded (score, doc) entries a block can rank; more -> the exact pass decides
struct CertifyParams {
    const long long* sid;     // [64, nf, kp] unique-row numbers of the screened lists (-1 = empty)
    const float* ssc;         // [64, nf, kp] approximate scores (scaled units), descending
    const int* scnt;          // [64 * nf] entries per screened list
    const float* sx;          // [64, nf, kp] exact scores of those rows' representatives (NaN = not scored)
    const ScreenField* sf;    // [F]
    const ScreenQuery* qinfo;
    const float* eps;         // [F, qw] field-wide bound
    const float* eps_cert;    // [F, qw] or nullptr (= eps): what the certificate adds to the k'-th list score (ROW MODE: the lists hold upper
                              // bounds that already carry the row-dependent part)
    const float* q;           // [Qt, E] the block's queries (row-major)
    const float* mean;        // [F, E] field means: q . mean is added back to the centred approximate scores
    int E;
    long long* out_ids;       // [Q, nf, k]
    float* out_scores;
    int* fail;                // [F] field flags, [MFAR_MAX_FIELDS] = any, [MFAR_MAX_FIELDS + 1] = failed (query, field) pairs (statistics),
                              // [MFAR_MAX_FIELDS + 2 + f] = probe flags (quiet_mask), then the tier-2 flags / statistics (SCREEN_FLAG_T1 ...):
                              // SCREEN_FLAGS ints in all
    u32 skip_mask;            // bit f: field f was not screened in this batch (switched off: the exact pass wrote its lists) -- nothing to do
    u32 quiet_mask;           // bit f: field f is switched off but was screened as a PROBE: evaluate the certificate, record a failure in
                              // the probe flags only, write no lists (the exact pass's lists stand)
    // unique-row tables (per field f at offset f * ustride): ustart / ucount index `members` (local rows, grouped, ascending)
    const int* ustart;
    const int* ucount;
    const int* members;
    const u32* uof;           // [F][ustride] or nullptr.  The certified pass over a bf16 slab scans documents: its lists hold LOCAL
                              // ROWS of group representatives, and uof[f][row] - 1 is the row's unique number (mfar_direct_* below);
                              // nullptr: the lists hold unique-row numbers (the fp16 screen slab of an fp32 index)
    long long ustride;
    long long row_offset;
    int f0, nf, k, kp, q0, sentinel;
    int qw;                   // query columns of the screened pass (64 / 128): stride of eps
    float* dbg;               // diagnostics (MFAR_CERT_DEBUG) or nullptr: per list {ok, bound, T_k, a_real, eps, cnt, m_out, overflow}
    // TIER 2 (below).  First certificate (pass2 == 0) with tau2 != nullptr: a list whose proof fails is not flagged for the exact pass yet --
    // lfail[ql * nf + fo] = 1 and tau2[f * qw + ql] = the scan-unit threshold every row that can still reach its exact top-k must pass
    // (+inf for lists that are done).  Second certificate (pass2 != 0): only lists with lfail != 0 are looked at; their sid / sx / scnt now
    // hold the best-by-exact-score rows of the COMPLETE candidate set (1) or nothing usable (2 = tier 2 overflowed: flagged for the exact pass).
    float* tau2;              // [F, qw] or nullptr
    int* lfail;               // [qw * nf] or nullptr
    int pass2;
    u32 deep_mask;            // DEEP SCAN fields (below): no first certificate -- every list goes to tier 2's back half (no rescan: the scan
                              // itself collected the complete set)
};
__global__ void __launch_bounds__(256) mfar_screen_certify_kernel(const CertifyParams p) {
    __shared__ u64 keys[CERT_EXPAND_CAP], sel[256], sorted[256];
    __shared__ int red[36], ucnt_s[256], upre_s[256];
    __shared__ float qm_s[4];
    __shared__ int total_s, overflow_s;
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    if ((p.skip_mask >> f) & 1u) {                       // workgroup-uniform
        if (p.tau2 && !p.pass2 && threadIdx.x == 0) {
            p.tau2[f * p.qw + ql] = __builtin_inff();
            p.lfail[ql * p.nf + fo] = 0;
        }
        return;
    }
    const bool quiet = ((p.quiet_mask >> f) & 1u) != 0u;
    if (!p.pass2 && ((p.deep_mask >> f) & 1u)) {         // workgroup-uniform
        if (threadIdx.x == 0) {
            p.tau2[f * p.qw + ql] = __builtin_inff();
            p.lfail[ql * p.nf + fo] = 1;
            atomicOr(&p.fail[SCREEN_FLAG_T1], 1);        // (keeps tier 2 armed)
            atomicAdd(&p.fail[SCREEN_STAT_T2_LISTS], 1);
        }
        return;
    }
    if (p.pass2) {                                       // workgroup-uniform: only the lists tier 2 worked on
        const int lf = p.lfail[ql * p.nf + fo];
        if (lf == 0) return;
        if (lf != 1) {                                   // tier 2 overflowed: the exact pass decides
            if (threadIdx.x == 0) {
                atomicOr(&p.fail[f], 1);
                atomicOr(&p.fail[MFAR_MAX_FIELDS], 1);
                atomicAdd(&p.fail[MFAR_MAX_FIELDS + 1], 1);
                atomicAdd(&p.fail[SCREEN_STAT_T2_FAILED], 1);
            }
            return;
        }
    }
    const size_t lb = ((size_t)ql * p.nf + fo) * p.kp;
    const int cnt = min(p.scnt[ql * p.nf + fo], p.kp);
    const float tau0 = p.sentinel ? 0.0f : -__builtin_inff();
    const int* ustart = p.ustart + (size_t)f * p.ustride;
    const int* ucount = p.ucount + (size_t)f * p.ustride;
    const int* members = p.members + (size_t)f * p.ustride;
    if (threadIdx.x == 0) {
        red[32] = 0;
        overflow_s = 0;
    }
    __syncthreads();
    // 1. the unique rows that pass the sentinel (every thread keeps its candidate), ranked by (exact score desc, unique number asc)
    u32 my_u = 0, my_bits = 0;
    bool mine = false;
    if ((int)threadIdx.x < cnt) {
        long long u = p.sid[lb + threadIdx.x];
        const float s = p.sx[lb + threadIdx.x];
        if (u >= 0 && p.uof) u = (long long)p.uof[(size_t)f * p.ustride + u] - 1;
        if (u >= 0 && s > tau0) {
            mine = true;
            my_u = (u32)u;
            my_bits = f2ord(s);
            keys[lds_add_rtn(&red[32], 1)] = make_key(s, my_u);
        }
    }
    __syncthreads();
    const int n = red[32];
    const int m = block_topk_sorted<1>(keys, n, p.k, sel, sorted, red);      // k unique rows always cover k documents
    // q . mean(field) (any summation order: part of the approximation, budgeted in eps)
    float pm = 0.0f;
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) pm = __builtin_fmaf(p.q[(size_t)ql * p.E + e], p.mean[(size_t)f * p.E + e], pm);
    for (int off = 32; off > 0; off >>= 1) pm += __shfl_xor(pm, off);
    if ((threadIdx.x & 63) == 0) qm_s[threadIdx.x >> 6] = pm;
    // 2. expansion.  A unique row with exact score s stands for min(ucount, k) documents with score s.  Walking the ranked
    //    rows, the first prefix that covers k documents ends at row j*: the k best documents are among the documents of all
    //    candidates that score AT LEAST what row j* scores (rows that TIE with j* included, wherever they rank: their
    //    documents may have lower ids -- exact fp32 ties between distinct vectors at the cut are rare but real).
    ucnt_s[threadIdx.x] = (int)threadIdx.x < m ? min(ucount[key_id(sorted[threadIdx.x])], p.k) : 0;
    __syncthreads();
    if (threadIdx.x == 0) {      // m <= 128: a serial prefix is a few hundred cycles
        int run = 0, jstar = m - 1;
        for (int i = 0; i < m; ++i) {
            run += ucnt_s[i];
            if (run >= p.k) {
                jstar = i;
                break;
            }
        }
        total_s = m > 0 ? (int)(u32)(sorted[jstar] >> 32) : -1;     // orderable score bits of row j* (0 = no candidate)
    }
    __syncthreads();
    const u32 tie_bits = (u32)total_s;
    const bool in = mine && m > 0 && my_bits >= tie_bits;
    const int c = in ? min(ucount[my_u], p.k) : 0;
    // block-wide exclusive prefix of c (256 threads = 4 waves)
    int incl = c;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if ((int)(threadIdx.x & 63) >= off) incl += v;
    }
    __syncthreads();                                  // total_s was read by everybody
    if ((threadIdx.x & 63) == 63) upre_s[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wbase += upre_s[w];
    const int total_all = upre_s[0] + upre_s[1] + upre_s[2] + upre_s[3];
    const bool fits = total_all <= CERT_EXPAND_CAP;   // more tied groups than the block can rank: the exact pass decides
    const int total = fits ? total_all : 0;
    if (fits && c > 0) {
        const int off0 = wbase + incl - c;
        const float s = ord2f(my_bits);
        const int* mem = members + ustart[my_u];
        for (int e = 0; e < c; ++e) keys[off0 + e] = make_key(s, (u32)(p.row_offset + mem[e]));
    }
    if (threadIdx.x == 0 && !fits) overflow_s = 1;
    __syncthreads();
    const int m_out = block_topk_sorted<4>(keys, total, p.k, sel, sorted, red);
    // 3. certificate
    if (threadIdx.x == 0) {
        // (a non-finite bound -- non-finite query or rows -- proves nothing, not even that a short list is complete)
        bool ok = overflow_s == 0 && p.eps[f * p.qw + ql] < __builtin_inff();
        float a_real = 0.0f, bound = 0.0f;
        if (ok && cnt == p.kp) {  // the list is full: unique rows outside it exist
            const float qm = (qm_s[0] + qm_s[1]) + (qm_s[2] + qm_s[3]);
            a_real = (p.ssc[lb + p.kp - 1] * p.qinfo[ql].inv_scale) * p.sf[f].inv_scale + qm;
            bound = a_real + (p.eps_cert ? p.eps_cert : p.eps)[f * p.qw + ql];   // every outside row scores <= bound (exactly)
            if (m_out == p.k) ok = bound < key_score(sorted[p.k - 1]); // ... strictly below the exact k-th best DOCUMENT
            else ok = bound <= tau0;                               // ... or cannot pass the sentinel at all
        }
        if (p.dbg) {
            float* d = p.dbg + (size_t)blockIdx.x * 8;
            d[0] = ok ? 1.0f : 0.0f;
            d[1] = bound;
            d[2] = m_out > 0 ? key_score(sorted[m_out - 1]) : 0.0f;
            d[3] = a_real;
            d[4] = p.eps[f * p.qw + ql];
            d[5] = (float)cnt;
            d[6] = (float)m_out;
            d[7] = (float)overflow_s + 10.0f * (float)n + 10000.0f * (float)total;
        }
        const bool to_t2 = !ok && !quiet && p.tau2 && !p.pass2 && overflow_s == 0 && p.eps[f * p.qw + ql] < __builtin_inff();
        if (p.tau2 && !p.pass2) {
            // TIER 2 threshold: every document with exact score >= E_k (the true k-th best) has exact >= e_k (the k-th best found so far,
            // a lower bound of E_k; without k documents past the sentinel: the sentinel itself), hence approx >= e_k - eps: in scan units
            // (e_k - eps - q.m) sq sf, leaning down by the roundings of this very expression (eps carries 25 % slack for the approximation's own)
            float t2 = __builtin_inff();
            if (to_t2) {
                const float qm = (qm_s[0] + qm_s[1]) + (qm_s[2] + qm_s[3]);
                const float ek = m_out == p.k ? key_score(sorted[p.k - 1]) : tau0;
                const float e_ = p.eps[f * p.qw + ql];
                float t = (ek - e_) - qm;
                t -= (fabsf(ek) + e_ + fabsf(qm)) * 4.0e-7f;
                t *= p.qinfo[ql].scale * p.sf[f].scale;             // powers of two
                t2 = t - fabsf(t) * 2.0e-7f;
                if (!(t2 == t2)) t2 = -__builtin_inff();            // (-inf - ...: no sentinel and a short list: everything qualifies -> overflow)
            }
            p.tau2[f * p.qw + ql] = t2;
            p.lfail[ql * p.nf + fo] = to_t2 ? 1 : 0;
        }
        if (!ok && quiet) atomicOr(&p.fail[MFAR_MAX_FIELDS + 2 + f], 1);
        else if (to_t2) {                                // not a failure yet: tier 2 takes the list (its field is rescanned)
            atomicOr(&p.fail[SCREEN_T2_FIELDS + f], 1);
            atomicOr(&p.fail[SCREEN_FLAG_T1], 1);
            atomicAdd(&p.fail[SCREEN_STAT_T2_LISTS], 1);
        } else if (!ok) {
            atomicOr(&p.fail[f], 1);
            atomicOr(&p.fail[MFAR_MAX_FIELDS], 1);
            atomicAdd(&p.fail[MFAR_MAX_FIELDS + 1], 1);
            if (p.pass2) atomicAdd(&p.fail[SCREEN_STAT_T2_FAILED], 1);
        } else if (p.pass2) p.lfail[ql * p.nf + fo] = 0;
    }
    if (quiet) return;
    const size_t ob = ((size_t)(p.q0 + ql) * p.nf + fo) * p.k;
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        if (r < m_out) {
            p.out_ids[ob + r] = (long long)key_id(sorted[r]);
            p.out_scores[ob + r] = key_score(sorted[r]);
        } else {
            p.out_ids[ob + r] = p.sentinel ? 0 : -1;
            p.out_scores[ob + r] = p.sentinel ? 0.0f : -__builtin_inff();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// TIER 2 of the certified screen: the complete candidate set above a proven threshold -- from the launch's own chunk lists, or by a
// THRESHOLD RESCAN (round 6; counted first: profiles/r06_tier2_population.txt).
//
// The first certificate fails when more than k' - k unique rows of a field score within ~2 eps of the list's k-th best: near-duplicate
// rows, or simply a narrow cone of vectors (mean-pooled transformer outputs: cosine 0.86 between the rows of a field) whose score
// density near the cut is high against eps.  Until round 5 such a field went to the exact fp32 pass: bound by the fp32 MFMA rate, ~7x the
// time of its screened scan.  But the first attempt leaves a USABLE fact behind: e_k, the exact k-th best document among the re-scored
// rows, is a lower bound of the true k-th best E_k -- so every document that belongs to the exact top-k (or ties with its last entry)
// has exact >= e_k, hence approximate score >= T(q, f) = e_k - eps.  Tier 2 therefore needs EVERY row of the field scoring >= T:
//   1. usually the launch's own scan already holds them (mfar_t2_collect_kernel pass A, no second scan).  That scan appended every row
//      scoring >= tg(q, f), its sample threshold, to the chunk lists -- and tg is loose: the k'-th best of a few percent of the rows, so
//      thousands of rows per list pass it, while T sits a few hundred rows deep.  Where tg <= T and no chunk list of the list was ever
//      compacted (both checked per list on the device), the candidates are the entries >= T of those lists.  (Found late in round 6:
//      until then every failed list paid step 2.  Hostile corpus 26 k -> 40 k q/s, no list rescanned.)
//   2. otherwise (pass A marks the list, flags its field): the flagged fields are RESCANNED with the SAME screened kernel over the SAME
//      fp16 slab (HBM-bound, half the bytes of any pass over the fp32 rows), list depth k', no sample pass, and the fixed non-strict
//      threshold T per failed list (+inf for the lists that are done: they append nothing) -- the chunk lists then hold every row
//      above T; pass B concatenates them (<= T2_CAP rows; a chunk list that reached depth k' may have dropped rows: overflow);
//      The rescan is ARMED by the policy (mfar_policy.h): enqueued only while a list of the last 256 launches asked for it -- a list
//      that asks while it is not armed goes to the exact pass and arms it -- because even an idle rescan is a full-width scan kernel in
//      the launch's tail, and its workgroups cannot start before the next launch's scan lets go of the register file;
//   3. re-scores those rows from the fp32 slab with the contract's chain (mfar_score_rows_kernel, per-list counts);
//   4. mfar_t2_select_kernel keeps the k' - 1 best by EXACT score in the screened-list format, and the certify kernel runs again on
//      those lists only: a list shorter than k' is complete by construction, so it expands unique rows to documents, applies the
//      sentinel and writes the final list -- no proof needed, the set is exhaustive above E_k;
//   5. lists that overflowed (or tie across the k' - 1 cut) keep their flag: the exact pass decides, as before.
// Counted on encoder-produced and clustered corpora: the candidate sets hold 200 - 500 rows (max ~1 150), i.e. ~1 MB of gathers per
// failed list against the 11 MB per list an exact pass of the field reads at the fp32 MFMA rate.  (The alternative tier -- hi + lo fp16
// terms over the fp32 slab -- certifies the encoder corpora too but almost none of the clustered lists: its eps keeps the rigorous
// fp32-accumulation terms, 0.27 of tier 1's; and it reads twice the bytes.)
// ---------------------------------------------------------------------------------------------------------
struct T2CollectParams {
    const uint2* lists;       // [n_chunks * qw][S1_CAP] chunk lists -- of the batch's own scan (pass A) / of the rescan (pass B): (score bits, unique row)
    const int* list_cnt;      // [n_chunks * qw]
    const int* fchunk;        // [F + 1] chunk ranges of the scan's table
    const int* lfail;         // [qw * nf]
    long long* cand;          // [qw, nf, T2_CAP] out: unique-row numbers
    int* cnt;                 // [qw * nf] out: candidates of the list (0 for lists tier 2 does not handle)
    int* lfail_out;           // = lfail (1 collected, 2 overflow: the exact pass decides, 3 = pass A could not vouch for the set: rescan, pass B)
    int f0, nf, qw, kp;
    int* stats;               // the batch's flag array (SCREEN_STAT_T2_*; SCREEN_T2_RESCAN_FIELDS)
    // PASS A (pass_b == 0), lists with lfail == 1: NO SECOND SCAN when the batch's own scan already holds the set.  That scan appended every
    // row of a chunk scoring >= tg = scan_tau(q, f) (non-strict: S1_PASS1) to the chunk's list, and a chunk list that never reached its
    // depth kp was never compacted (a compaction leaves kp entries): it is complete above tg.  The sample's threshold is loose -- the
    // k'-th best of a few percent of the rows, i.e. thousands of rows per list pass it -- so tg <= T (tau2) is the normal case: the
    // candidates are then simply the entries >= T of the lists the scan wrote.  Checked here, per list, at run time; otherwise
    // (tg > T, or a chunk at depth) the list is marked 3, its field is flagged, the rescan runs for the flagged fields and pass B collects.
    // PASS B (pass_b != 0), lists with lfail == 3: the rescan's chunk lists (everything there is >= T).
    int pass_b;
    int rescan_on;            // pass A: the rescan follows (armed: mfar_policy.h).  0: a list pass A cannot vouch for goes to the exact pass
    int no_first;             // diagnostic (MFAR_T2_FIRST_SCAN=0): pass A vouches for nothing -- every list takes the rescan, as in the first design
    const float* tau2;        // [F, qw] T in scan units
    const float* scan_tau;    // [F, qw] thresholds the batch's own scan ran with, or nullptr (none: every row was appended)
    // DEEP SCAN fields: the chunk lists hold everything above a sample-derived threshold (a few thousand rows); the candidates are the rows
    // within the band of the k-th best APPROXIMATE score of that complete set (same argument as for the threshold: mfar_sample_tau_kernel)
    u32 deep_mask;
    int k, sentinel;
    const float4* info;       // [F, qw] {band, eps, position of exact 0} in scan units
};
#define T2_CAP_IN 8192        // entries a deep list's chunk lists may hold in all
#define T2_COLLECT_LDS_BYTES SEL_LDS_BYTES(T2_CAP_IN)
#define T2_COLLECT_THREADS 1024
// Passes A and B for the lists of ordinary (not DEEP SCAN) fields.  grid = Qt * nf, block T2_COLLECT_THREADS, no dynamic LDS: a list's
// chunk lists are many (hundreds) and short, one wave per chunk list and sixteen waves per list keep enough loads in flight (with four
// waves and the deep kernel's 70 KB of LDS per workgroup this pass took 1.2 ms per launch of the hostile corpus: a chain of 128 dependent
// count -> entries loads per wave, two workgroups per CU).
__global__ void __launch_bounds__(T2_COLLECT_THREADS) mfar_t2_collect_kernel(const T2CollectParams p) {
    __shared__ int m_s, ovf_s;
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    const int li = ql * p.nf + fo;
    const int lf = p.lfail[li];
    const bool deep = ((p.deep_mask >> f) & 1u) != 0u;
    if (lf != (p.pass_b ? 3 : 1) || (deep && !p.pass_b)) {      // workgroup-uniform (a deep list: mfar_t2_collect_deep_kernel)
        if (threadIdx.x == 0 && !p.pass_b && !(deep && lf == 1)) p.cnt[li] = 0;
        return;
    }
    if (threadIdx.x == 0) m_s = ovf_s = 0;
    __syncthreads();
    const int c_lo = p.fchunk[f], n_chunks = p.fchunk[f + 1] - c_lo;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    long long* out = p.cand + (size_t)li * T2_CAP;
    const float T = p.tau2[f * p.qw + ql];
    const bool covers = p.pass_b || (!p.no_first && (!p.scan_tau || p.scan_tau[f * p.qw + ql] <= T));     // (NaN: false)
    for (int c = w; c < n_chunks && covers; c += nw) {        // one wave per chunk list
        const size_t lq = (size_t)(c_lo + c) * p.qw + ql;
        const int n = min(p.list_cnt[lq], S1_CAP);
        if (n >= p.kp) {                                 // compacted to its depth (or exactly full): rows above the threshold may be gone
            if (lane == 0) ovf_s = 1;
            continue;
        }
        for (int e0 = 0; e0 < n; e0 += 64) {
            const int e = e0 + lane;
            uint2 v = make_uint2(0u, 0u);
            if (e < n) v = p.lists[lq * S1_CAP + e];
            const bool keep = e < n && __uint_as_float(v.x) >= T;
            const int pos = wave_reserve(&m_s, keep);
            if (keep && pos < T2_CAP) out[pos] = (long long)v.y;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool full = ovf_s != 0 || !covers;
        if (!p.pass_b && full && p.rescan_on) {          // the rescan decides
            p.cnt[li] = 0;
            p.lfail_out[li] = 3;
            atomicOr(&p.stats[SCREEN_T2_RESCAN_FIELDS + f], 1);
            atomicAdd(&p.stats[SCREEN_STAT_T2_RESCAN], 1);
        } else if (full || m_s > T2_CAP) {
            if (!p.pass_b && full) atomicOr(&p.stats[SCREEN_FLAG_T2_WANT], 1);      // (no rescan enqueued behind this launch: arm it)
            p.cnt[li] = 0;
            p.lfail_out[li] = 2;
            atomicAdd(&p.stats[SCREEN_STAT_T2_OVF + (full ? 0 : 2)], 1);
        } else {
            p.cnt[li] = m_s;
            p.lfail_out[li] = 1;
            if (!p.pass_b) atomicAdd(&p.stats[SCREEN_STAT_T2_FIRST], 1);
        }
    }
}
// The lists of DEEP SCAN fields (launched only when the batch has such fields): everything their chunk lists hold, narrowed to the band
// around the k-th best approximate score.  grid = Qt * nf, block 256, dynamic LDS = T2_COLLECT_LDS_BYTES
__global__ void __launch_bounds__(256) mfar_t2_collect_deep_kernel(const T2CollectParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, T2_CAP_IN);
    int& n_s = L.misc[0];
    int& ovf_s = L.misc[1];
    int& m_s = L.misc[2];
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    const int li = ql * p.nf + fo;
    if (p.lfail[li] != 1 || ((p.deep_mask >> f) & 1u) == 0u) return;      // workgroup-uniform
    if (threadIdx.x == 0) n_s = ovf_s = m_s = 0;
    __syncthreads();
    const int c_lo = p.fchunk[f], n_chunks = p.fchunk[f + 1] - c_lo;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long* out = p.cand + (size_t)li * T2_CAP;
    for (int c = w; c < n_chunks; c += 4) {              // one wave per chunk list -> (approx score, unique row) keys in LDS
        const size_t lq = (size_t)(c_lo + c) * p.qw + ql;
        const int n = min(p.list_cnt[lq], S1_CAP);
        if (n >= p.kp) {
            if (lane == 0) ovf_s = 1;
            continue;
        }
        int base = 0;
        if (lane == 0 && n > 0) base = atomicAdd(&n_s, n);
        base = __shfl(base, 0);
        if (base + n > T2_CAP_IN) {
            if (lane == 0) ovf_s = 1;
            continue;
        }
        for (int e = lane; e < n; e += 64) {
            const uint2 v = p.lists[lq * S1_CAP + e];
            L.keys[base + e] = make_key(__uint_as_float(v.x), v.y);
        }
    }
    __syncthreads();
    const int n = min(n_s, T2_CAP_IN);
    bool ovf = ovf_s != 0 || n_s > T2_CAP_IN;
    if (threadIdx.x == 0 && ovf) atomicAdd(&p.stats[SCREEN_STAT_T2_OVF + (n_s > T2_CAP_IN ? 1 : 0)], 1);
    if (!ovf) {
        // the k-th best approximate score a_k of the complete set: k rows score at least a_k, so E_k >= a_k - eps and every row that can
        // reach the exact top-k has approx >= a_k - band (or, without a proof of k positive documents, can be positive at all)
        float lo = -__builtin_inff();
        if (n > p.k) {
            const float4 in = p.info[f * p.qw + ql];
            const int m = block_topk_sorted<T2_CAP_IN / 256>(L.keys, n, p.k, L.sel, L.sorted, L.red);
            const float ak = key_score(L.sorted[m - 1]);
            lo = ak - in.x;
            lo -= fabsf(lo) * 2.0e-7f;
            if (p.sentinel && !(ak - in.x > in.z)) lo = fminf(lo, in.z - 1.01f * in.y);
            if (!(in.y < __builtin_inff())) lo = -__builtin_inff();       // non-finite bound: keep everything (overflows unless the set is small)
        }
        __syncthreads();
        for (int i0 = 0; i0 < n; i0 += blockDim.x) {
            const int i = i0 + threadIdx.x;
            const bool keep = i < n && !(key_score(L.keys[i]) < lo);
            const int pos = wave_reserve(&m_s, keep);
            if (keep && pos < T2_CAP) out[pos] = (long long)key_id(L.keys[i]);
        }
        __syncthreads();
        if (m_s > T2_CAP) {
            ovf = true;
            if (threadIdx.x == 0) atomicAdd(&p.stats[SCREEN_STAT_T2_OVF + 2], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.cnt[li] = ovf ? 0 : m_s;
        if (ovf) p.lfail_out[li] = 2;
    }
}

struct T2SelectParams {
    const long long* cand;    // [qw, nf, T2_CAP] unique rows
    const float* sx2;         // [qw, nf, T2_CAP] their exact scores
    const int* cnt;           // [qw * nf]
    int* lfail;               // [qw * nf]
    long long* sid;           // [qw, nf, kp] out: the screened-list format the certify kernel reads
    float* sx;                // [qw, nf, kp]
    int* scnt;                // [qw * nf]
    int* stats;               // the batch's flag array
    int nf, kp, k;
};
// grid = Qt * nf, block 256, dynamic LDS = SEL_LDS_BYTES(T2_CAP)
__global__ void __launch_bounds__(256) mfar_t2_select_kernel(const T2SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int li = blockIdx.x;
    if (p.lfail[li] != 1) return;                        // workgroup-uniform
    u64* keys = (u64*)smem;
    u64* sel = keys + T2_CAP;
    u64* sorted = sel + 256;
    int* red = (int*)(sorted + 256);
    const int n = min(p.cnt[li], T2_CAP);
    const long long* cd = p.cand + (size_t)li * T2_CAP;
    const float* sc = p.sx2 + (size_t)li * T2_CAP;
    // (every candidate was scored: a NaN score -- non-finite data -- orders above everything and ends in the certify kernel's own checks)
    for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = make_key(sc[i], (u32)cd[i]);
    const int keep = p.kp - 1;                           // a list SHORTER than k' needs no proof (mfar_screen_certify_kernel)
    const int m = block_topk_sorted<T2_CAP / 256>(keys, n, keep, sel, sorted, red);
    // the cut must not fall inside a run of equal scores that reaches up into the top-k: rows tied with the k-th would be missing
    bool bad = false;
    if (n > keep && m == keep && p.k <= keep) bad = (u32)(sorted[keep - 1] >> 32) == (u32)(sorted[p.k - 1] >> 32);
    const size_t lb = (size_t)li * p.kp;
    for (int r = threadIdx.x; r < p.kp; r += blockDim.x) {
        p.sid[lb + r] = r < m ? (long long)key_id(sorted[r]) : -1;
        p.sx[lb + r] = r < m ? key_score(sorted[r]) : __builtin_nanf("");
    }
    if (threadIdx.x == 0) {
        p.scnt[li] = m;
        if (bad) {
            p.lfail[li] = 2;
            atomicAdd(&p.stats[SCREEN_STAT_T2_OVF + 3], 1);
        }
    }
}
#define T2_SELECT_LDS_BYTES ((size_t)T2_CAP * 8 + 2 * 256 * 8 + 36 * 4)
