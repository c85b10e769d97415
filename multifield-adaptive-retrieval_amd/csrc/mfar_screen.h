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

struct ScreenField {     // per field, written by mfar_screen_scale_kernel
    float scale;         // sf = 2^e: fp16 value = (fp32 value - mean) * sf
    float inv_scale;
    float dnorm_max;     // largest 2-norm of a centred row of the field (inf / NaN when the field holds non-finite values)
    float mnorm;         // 2-norm of the field's mean vector
};
struct ScreenQuery {     // per query of the current 64-query block
    float scale, inv_scale, norm, pad;
};

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
__global__ void __launch_bounds__(256) mfar_screen_stats_kernel(const float* __restrict__ slab, long long field_stride, int n_steps,
                                                                long long n_rows, const float* __restrict__ mean,
                                                                u32* __restrict__ stats) {
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
    u32 a = __float_as_uint(amax), n = __float_as_uint(ss) & 0x7FFFFFFFu;
    for (int off = 32; off > 0; off >>= 1) {
        a = max(a, (u32)__shfl_xor((int)a, off));
        n = max(n, (u32)__shfl_xor((int)n, off));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&stats[2 * f], a);
        atomicMax(&stats[2 * f + 1], n);
    }
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
                                         ScreenField* __restrict__ sf) {
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
    sf[f] = o;
}

// fp32 tiled slab -> fp16 tiled screen slab (centred, scaled).  One thread per 16-byte output granule (8 dims).
// grid = (ceil(n_blk * n_steps * 128 / 256), F)
__global__ void __launch_bounds__(256) mfar_screen_build_kernel(const float* __restrict__ slab, _Float16* __restrict__ screen,
                                                                long long field_stride, long long n_granules, int n_steps,
                                                                const float* __restrict__ mean, const ScreenField* __restrict__ sf) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_granules) return;
    const int f = blockIdx.y;
    const float sc = sf[f].scale;
    const long long tile = g >> 7;           // 128 granules per [64][16] fp16 tile = (block, k-step)
    const int rr = (int)(g >> 1) & 63, c8 = (int)g & 1;
    const int step = (int)(tile % n_steps);
    const long long blk = tile / n_steps;
    // source: row rr's 128-byte line of k-step pair step / 2, half step & 1, dims 8 c8 .. 8 c8 + 7 of the step
    const float* src = slab + (size_t)f * field_stride + (size_t)(blk * (n_steps >> 1) + (step >> 1)) * 2048 + rr * 32 + (step & 1) * 16 + c8 * 8;
    const float* m = mean + (size_t)f * n_steps * 16 + step * 16 + c8 * 8;
    const f32x4 a = *(const f32x4*)src - *(const f32x4*)m;
    const f32x4 b = *(const f32x4*)(src + 4) - *(const f32x4*)(m + 4);
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = (_Float16)(a[i] * sc);
        o[4 + i] = (_Float16)(b[i] * sc);
    }
    *(f16x8*)(screen + (size_t)f * field_stride + (size_t)tile * 1024 + rr * 16 + ((c8 ^ ((rr >> 3) & 1)) << 3)) = o;
}

// ---------------------------------------------------------------------------------------------------------
// Duplicate rows.  A field that a document lacks is encoded from the empty string (format.py:58-59), so real corpora hold
// one huge group of IDENTICAL rows per field.  When that group reaches the top of a list, equal approximate scores cannot
// separate the k-th from the k'-th entry and every certificate of the field would fail.  The screen therefore finds the
// field's largest group of bit-identical rows at build time, scans only its lowest-id member (the others are masked out of
// the screened pass by a row bitmap) and re-inserts the masked members -- same exact score, ids ascending -- when the
// representative makes it into the exact top-k (mfar_screen_certify_kernel).  Masked rows can never be missed: they score
// exactly what their representative scores and lose every tie against it.
//   1. mfar_dup_sample_kernel   hashes DUP_SAMPLES pseudo-random rows of the field; the most frequent hash names a candidate row
//   2. mfar_dup_compare_kernel  compares EVERY row with the candidate bit for bit -> bitmap of equal rows, their count, the lowest id
//   3. mfar_dup_finish_kernel   drops groups below DUP_MIN_GROUP, clears the representative's bit, lists the DUP_MEMBERS lowest masked ids
// ---------------------------------------------------------------------------------------------------------
#define DUP_SAMPLES 4096
#define DUP_MIN_GROUP 64
#define DUP_MEMBERS 128        // >= MFAR_MAX_K: a list can never need more members than its depth
struct DupGroup {              // per field
    int rep;                   // local row of the lowest-id member, -1 = no group
    int n_masked;              // masked members (all of them, not only the listed ones)
    int cand;                  // scratch: the sampled candidate row
    int count;                 // scratch: rows equal to the candidate
    int members[DUP_MEMBERS];  // the lowest masked local rows, ascending; -1 padded
};

__device__ __forceinline__ u64 dup_mix(u64 x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// grid = F, block 256, dynamic LDS = DUP_SAMPLES * 8
__global__ void __launch_bounds__(256) mfar_dup_sample_kernel(const float* __restrict__ slab, long long field_stride, int n_steps,
                                                              long long n_rows, DupGroup* __restrict__ grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* keys = (u64*)smem;   // (hash with the low 12 bits replaced by the sample index)
    const int f = blockIdx.x, E = n_steps * 16;
    for (int i = threadIdx.x; i < DUP_SAMPLES; i += blockDim.x) {
        const long long row = n_rows > 0 ? (long long)(((u64)i * 0x9E3779B97F4A7C15ull) % (u64)n_rows) : 0;
        u64 h = 0;
        for (int e = 0; e < E; e += 4) {
            const f32x4 v = *(const f32x4*)(slab + (size_t)f * field_stride + tiled_offset(n_steps, row, e));
#pragma unroll
            for (int j = 0; j < 4; ++j) h += dup_mix((u64)__float_as_uint(v[j]) + (u64)(e + j + 1) * 0x9E3779B97F4A7C15ull);
        }
        keys[i] = (h & ~0xFFFull) | (u64)i;
    }
    __syncthreads();
    for (int size = 2; size <= DUP_SAMPLES; size <<= 1)      // bitonic sort ascending
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < DUP_SAMPLES / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const u64 x = keys[lo], y = keys[hi];
                if ((x > y) == up) {
                    keys[lo] = y;
                    keys[hi] = x;
                }
            }
            __syncthreads();
        }
    // longest run of equal hashes (thread t looks at runs starting at positions t, t + 256, ...)
    __shared__ unsigned long long best;   // (run length << 32) | start
    if (threadIdx.x == 0) best = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < DUP_SAMPLES; i += blockDim.x) {
        const u64 hk = keys[i] >> 12;
        if (i > 0 && (keys[i - 1] >> 12) == hk) continue;   // not a run start
        int j = i + 1;
        while (j < DUP_SAMPLES && (keys[j] >> 12) == hk) ++j;
        atomicMax(&best, ((unsigned long long)(j - i) << 32) | (unsigned)i);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int len = (int)(best >> 32), start = (int)(best & 0xFFFFFFFFu);
        DupGroup* g = grp + f;
        g->rep = 0x7FFFFFFF;
        g->n_masked = 0;
        g->count = 0;
        g->cand = -1;
        if (len >= 2 && n_rows >= DUP_MIN_GROUP) {
            const int si = (int)(keys[start] & 0xFFFull);
            g->cand = (int)(((u64)si * 0x9E3779B97F4A7C15ull) % (u64)n_rows);
        }
    }
}

// grid = (n_blk, F), block 256 (thread layout of mfar_screen_stats_kernel).  eq: [F][n_blk * 2] words, zeroed by the host.
__global__ void __launch_bounds__(256) mfar_dup_compare_kernel(const float* __restrict__ slab, long long field_stride, int n_steps,
                                                               long long n_rows, DupGroup* __restrict__ grp, u32* __restrict__ eq,
                                                               long long words_per_field) {
    const int f = blockIdx.y;
    const int cand = grp[f].cand;
    if (cand < 0) return;   // workgroup-uniform
    const int rr = threadIdx.x >> 2, pp = threadIdx.x & 3;   // thread = quarter pp (8 floats) of row rr's line per k-step pair
    const long long row = (long long)blockIdx.x * 64 + rr;
    const float* tile = slab + (size_t)f * field_stride + (size_t)blockIdx.x * n_steps * 1024 + rr * 32 + pp * 8;
    const float* cnd = slab + (size_t)f * field_stride + tiled_offset(n_steps, cand, pp * 8);
    bool same = row < n_rows;
    for (int pr = 0; pr < (n_steps >> 1); ++pr) {
        if (!__any(same)) break;   // wave-uniform
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const f32x4 a = *(const f32x4*)(tile + (size_t)pr * 2048 + hq * 4);
            const f32x4 b = *(const f32x4*)(cnd + (size_t)pr * 2048 + hq * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) same = same && __float_as_uint(a[i]) == __float_as_uint(b[i]);
        }
    }
    // threads 4r .. 4r+3 hold the four quarters of row r
    same = (__shfl_xor((int)same, 1) & (int)same) != 0;
    same = (__shfl_xor((int)same, 2) & (int)same) != 0;
    const u64 m = __ballot(same && pp == 0);   // bit 4r set: row r of this wave's 16 rows equals the candidate
    if ((threadIdx.x & 63) == 0 && m) {
        u32 bits = 0;
        for (int r = 0; r < 16; ++r) bits |= (u32)((m >> (4 * r)) & 1ull) << r;
        const int wv = threadIdx.x >> 6;                                     // wave: rows 16 wv .. 16 wv + 15 of the block
        atomicOr(&eq[(size_t)f * words_per_field + blockIdx.x * 2 + (wv >> 1)], bits << (16 * (wv & 1)));
        atomicAdd(&grp[f].count, __popc(bits));
        atomicMin(&grp[f].rep, (int)(blockIdx.x * 64 + wv * 16 + __builtin_ctz(bits)));
    }
}

// grid = F, block 64
__global__ void __launch_bounds__(64) mfar_dup_finish_kernel(DupGroup* __restrict__ grp, u32* __restrict__ eq, long long words_per_field) {
    const int f = blockIdx.x, lane = threadIdx.x;
    DupGroup* g = grp + f;
    u32* w = eq + (size_t)f * words_per_field;
    const bool have = g->cand >= 0 && g->count >= DUP_MIN_GROUP;
    if (!have) {
        for (long long i = lane; i < words_per_field; i += 64) w[i] = 0u;
        if (lane == 0) {
            g->rep = -1;
            g->n_masked = 0;
        }
        for (int i = lane; i < DUP_MEMBERS; i += 64) g->members[i] = -1;
        return;
    }
    const int rep = g->rep;
    if (lane == 0) {
        w[rep >> 5] &= ~(1u << (rep & 31));   // the representative stays in the scan
        g->n_masked = g->count - 1;
    }
    __syncthreads();
    // the DUP_MEMBERS lowest masked rows, ascending: walk the bitmap 64 words at a time
    int found = 0;
    for (long long base = 0; base < words_per_field && found < DUP_MEMBERS; base += 64) {
        const u32 word = base + lane < words_per_field ? w[base + lane] : 0u;
        const int cnt = __popc(word);
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        int pos = found + incl - cnt;
        u32 bitsw = word;
        while (bitsw && pos < DUP_MEMBERS) {
            const int b = __builtin_ctz(bitsw);
            bitsw &= bitsw - 1;
            g->members[pos++] = (int)((base + lane) * 32 + b);
        }
        found += __shfl(incl, 63);
    }
    for (int i = (found < DUP_MEMBERS ? found : DUP_MEMBERS) + lane; i < DUP_MEMBERS; i += 64) g->members[i] = -1;
}

// ---------------------------------------------------------------------------------------------------------
// Queries of one 64-query block: per-query scale + norm, two-term fp16 split tiles [n_steps][2][64][16], and per
// (field, query): eps (real units) and the starting threshold of the screened pass (scaled units).
//   grid = 64 (one workgroup per query row), block 256.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfar_screen_queries_kernel(const float* __restrict__ q, _Float16* __restrict__ qt,
                                                                  ScreenQuery* __restrict__ qinfo, const ScreenField* __restrict__ sf,
                                                                  float* __restrict__ eps, float* __restrict__ tau_base,
                                                                  int* __restrict__ fail_flags, int q0, int Q, int E, int F,
                                                                  float eps_mult) {
    __shared__ float red_a[4], red_s[4];
    const int r = blockIdx.x;
    // a new batch: clear the certificate flags of the fields and the "any" flag ([F + 1] keeps accumulating statistics)
    if (r == 0 && (int)threadIdx.x <= F) fail_flags[threadIdx.x] = 0;
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
        const size_t in_tile = tiled_offset_bf16(E >> 4, r, e) - (size_t)step * 1024;
        _Float16* base = qt + (size_t)step * 2048;   // 2 tiles of 1024 halves per k-step
        *(f16x8*)(base + in_tile) = hi;
        *(f16x8*)(base + 1024 + in_tile) = lo;
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
        const float c_rel = 1.02f * 4.8828125e-4f + (4.0f * K + 66.0f) * u32f;
        const float c_abs = u32f * sqrtf(K) * 1.0001f;
        float e_ = SCREEN_SLACK * (c_rel * qn * s.dnorm_max + K * u32f * qn * (s.dnorm_max + 2.0f * s.mnorm) +
                                   c_abs * (qn * s.inv_scale + s.dnorm_max / sq));
        e_ *= eps_mult;
        if (!live) e_ = 0.0f;
        eps[f * 64 + r] = e_;
        // starting threshold of the screened pass: none for live queries (the zero sentinel of index.py:192-193 is applied
        // to the EXACT scores by the certify kernel; deciding it here would need q.m on the critical path), +inf for the
        // padding queries of a short batch so that they append nothing
        tau_base[f * 64 + r] = live ? -__builtin_inff() : __builtin_inff();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Certify: exact top-k of the k' re-scored rows of one (query, field) + the certificate.  grid = Qt * F, block 256.
// ---------------------------------------------------------------------------------------------------------
struct CertifyParams {
    const long long* sid;     // [64, F, kp] global ids of the screened lists (-1 = empty)
    const float* ssc;         // [64, F, kp] approximate scores (scaled units), descending
    const int* scnt;          // [64 * F] entries per screened list
    const float* sx;          // [64, F, kp] exact scores of those rows (NaN = not scored)
    const ScreenField* sf;
    const ScreenQuery* qinfo;
    const float* eps;         // [F, 64]
    const float* q;           // [Qt, E] the block's queries (row-major)
    const float* mean;        // [F, E] field means: q . mean is added back to the centred approximate scores
    int E;
    long long* out_ids;       // [Q, F, k]
    float* out_scores;
    int* fail;                // [F] field flags, [F] = any, [F+1] = failed (query, field) pairs (statistics)
    const DupGroup* grp;      // [F] duplicate groups (local rows) or nullptr
    long long row_offset;
    int F, k, kp, q0, sentinel;
};
__global__ void __launch_bounds__(256) mfar_screen_certify_kernel(const CertifyParams p) {
    __shared__ u64 keys[256], sel[256], sorted[256];
    __shared__ int red[36];
    const int ql = blockIdx.x / p.F, f = blockIdx.x - ql * p.F;
    const size_t lb = ((size_t)ql * p.F + f) * p.kp;
    const int cnt = min(p.scnt[ql * p.F + f], p.kp);
    const float tau0 = p.sentinel ? 0.0f : -__builtin_inff();
    if (threadIdx.x == 0) red[32] = 0;
    __syncthreads();
    if ((int)threadIdx.x < cnt) {
        const long long id = p.sid[lb + threadIdx.x];
        const float s = p.sx[lb + threadIdx.x];
        if (id >= 0 && s > tau0) keys[lds_add_rtn(&red[32], 1)] = make_key(s, (u32)id);
    }
    __syncthreads();
    const int n = red[32];
    const int m = block_topk_sorted<1>(keys, n, p.k, sel, sorted, red);
    // q . mean(field) (any summation order: part of the approximation, budgeted in eps)
    float pm = 0.0f;
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) pm = __builtin_fmaf(p.q[(size_t)ql * p.E + e], p.mean[(size_t)f * p.E + e], pm);
    for (int off = 32; off > 0; off >>= 1) pm += __shfl_xor(pm, off);
    __shared__ float qm_s[4];
    if ((threadIdx.x & 63) == 0) qm_s[threadIdx.x >> 6] = pm;
    __syncthreads();
    // certificate
    if (threadIdx.x == 0) {
        bool ok = true;
        if (cnt == p.kp) {  // the list is full: rows outside it exist
            const float qm = (qm_s[0] + qm_s[1]) + (qm_s[2] + qm_s[3]);
            const float a_real = (p.ssc[lb + p.kp - 1] * p.qinfo[ql].inv_scale) * p.sf[f].inv_scale + qm;
            const float bound = a_real + p.eps[f * 64 + ql];       // every outside row scores <= bound (exactly)
            if (m == p.k) ok = bound < key_score(sorted[p.k - 1]); // ... strictly below the exact k-th best
            else ok = bound <= tau0;                               // ... or cannot pass the sentinel at all
        }
        if (!ok) {
            atomicOr(&p.fail[f], 1);
            atomicOr(&p.fail[p.F], 1);
            atomicAdd(&p.fail[p.F + 1], 1);
        }
    }
    // duplicate group: when its representative is among the k best, the masked members (same exact score, higher ids) take
    // their canonical places right behind it -- select the k best of (the k selected) + (the group's lowest masked members)
    int m_out = m;
    if (p.grp && p.grp[f].rep >= 0 && m > 0) {
        const DupGroup* g = p.grp + f;
        const u32 rep_id = (u32)(p.row_offset + g->rep);
        __shared__ int hit;
        if (threadIdx.x == 0) hit = -1;
        __syncthreads();
        if ((int)threadIdx.x < m && key_id(sorted[threadIdx.x]) == rep_id) hit = threadIdx.x;
        __syncthreads();
        if (hit >= 0) {   // workgroup-uniform
            const float s_rep = key_score(sorted[hit]);
            const int n_add = min(min(g->n_masked, DUP_MEMBERS), p.k);
            if ((int)threadIdx.x < m) keys[threadIdx.x] = sorted[threadIdx.x];
            if ((int)threadIdx.x < n_add) keys[m + threadIdx.x] = make_key(s_rep, (u32)(p.row_offset + g->members[threadIdx.x]));
            __syncthreads();
            m_out = block_topk_sorted<1>(keys, m + n_add, p.k, sel, sorted, red);
        }
    }
    const size_t ob = ((size_t)(p.q0 + ql) * p.F + f) * p.k;
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
