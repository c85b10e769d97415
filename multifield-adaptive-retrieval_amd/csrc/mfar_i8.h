// mfar_i8.h -- an int8 FIRST LEVEL under the certified fp16 screen (wide pass only).
//
// The wide screened pass (mfar_stage1.h, s1_body_f16w) is bound by the bytes of the fp16 screen slab: 2 bytes per scanned
// element and 128 queries.  This level scans ONE byte per element instead and hands the fp16 level a short, CERTIFIED superset of
// every list's rows; the fp16 values of only those rows are then gathered (the fp16 gather slab, mfar_select.h) and everything
// behind that -- k' best approximate scores, exact re-scoring, certificate, repair -- is the screen's own machinery, unchanged.
//
//   build   per field, the unique rows h (the screen's centred + scaled fp16 values) are quantised with ONE step per SEGMENT:
//           d8 = clamp(rint(h / lam), +-127), stored biased (d8 + 128).  Segment 0 holds the rows whose largest |h| fits 127 lam_0
//           (lam_0 from a high percentile of the per-row maxima), segment 1 the few rows beyond it, with lam_1 from the field's
//           maximum: a single step for all rows would be set by the one most extreme element of millions, and the bound below
//           is proportional to the step.  Every (field, segment) is a PSEUDO-FIELD pf = 2 f + segment of the int8 slab, with its
//           own chunks in the pass's table; map8[pf][local row] = the unique-row number.  err[pf] = the largest residual norm
//           |h - lam d8|_2 over the segment's rows, computed from the stored values.
//   scan    the wide pass's structure (4 waves x 64 rows x 128 query columns, docs in a register ring, queries in an LDS ring),
//           the docs arrive as bytes and become EXACT fp16 integers in registers (v_perm_b32 builds 0x6400 | byte = 1024 + byte,
//           v_pk_add_f16 subtracts 1152), two k-steps per 16-byte load.  acc8 = sum_i Q_i d8_i with Q the wide pass's fp16 query
//           tile; products and (up to the fp32 accumulation) sums are exact, so
//               | lam acc8 - sum_i Q_i h_i |  <=  |Q|_2 err[pf]  +  accumulation              (Cauchy-Schwarz on the residual)
//           and with the screen's own bound eps16 on |exact - approx16|:  |exact - a8| <= eps8(q, pf) for EVERY row.
//   filter  l(r) = a8 - eps8, u(r) = a8 + eps8.  L_k = the k-th largest l over the field's rows: k rows have exact >= L_k, so a
//           row with u(r) < L_k is STRICTLY below k unique rows (which cover at least k documents) and can neither enter nor
//           tie into the top-k.  The pass appends every row with u(r) >= tg, tg = (k-th largest sampled l) <= L_k, to the chunk
//           lists WITHOUT compaction (a list that would overflow marks the (query, field) as failed: the exact repair pass
//           decides, as for a failed certificate); mfar_i8_filter_kernel computes L_k from the appended rows and keeps
//           S = {u >= L_k}.  |S| is a few hundred rows per (query, field).
//   level 2 mfar_score_rows_kernel<SRC_F16G> gathers the fp16 rows of S (1.5 KB per row at 768 dims) -> approximate scores
//           with the screen's bound; mfar_i8_select_kernel sorts the k' best into the screen's list format.  Rows of S outside
//           the k' are covered by the certificate exactly as before (their approximate score is <= the k'-th); rows outside S
//           are excluded by the argument above.
#pragma once
#include "mfar_device.h"
#include "mfar_stage1.h"
#include "mfar_select.h"
#include "mfar_screen.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// two bytes of d (selected by sel) -> two exact fp16 integers (byte - 128)
__device__ __forceinline__ u32 i8_cvt2(u32 d, u32 sel) {
    const u32 p = __builtin_amdgcn_perm(0x64646464u, d, sel);
    const f16x2 h = __builtin_bit_cast(f16x2, p) - f16x2{(_Float16)1152.0f, (_Float16)1152.0f};
    return __builtin_bit_cast(u32, h);
}
__device__ __forceinline__ f16x8 i8_cvt8(u32 lo, u32 hi) {
    const u32x4 o = {i8_cvt2(lo, 0x04010400u), i8_cvt2(lo, 0x04030402u), i8_cvt2(hi, 0x04010400u), i8_cvt2(hi, 0x04030402u)};
    return __builtin_bit_cast(f16x8, o);
}

// int8 slab layout of one pseudo-field: [n_blk 64-row blocks][n_pairs k-step PAIRS][2 doc blocks of 32 rows][32 rows][32 bytes]; the 32
// bytes of (row, pair) are [h = 0: dims 0-7 | 16-23][h = 1: dims 8-15 | 24-31] of the pair, so that lane (j, h) of the MFMA's A
// operand finds its 8 bytes of k-step 0 and its 8 bytes of k-step 1 of the pair in ONE 16-byte load.
__host__ __device__ __forceinline__ size_t i8_offset(int64_t n_pairs, int64_t row, int e) {
    const int64_t blk = row >> 6;
    const int rr = (int)(row & 63), d = e & 31, h = (d >> 3) & 1, ks = d >> 4;
    return (size_t)((blk * n_pairs + (e >> 5)) * 2048 + (rr >> 5) * 1024 + (rr & 31) * 32 + h * 16 + ks * 8 + (d & 7));
}

// ---------------------------------------------------------------------------------------------------------------------
// The scan.  R = doc register ring (pair stages), RQ = query LDS ring (pair stages of 8 KB); the loop is unrolled lcm(R, RQ)
// times so that both slot indices are compile-time constants: n_pairs % lcm == 0 (host: i8_ring()).
// ---------------------------------------------------------------------------------------------------------------------
#define S1I_SCAP 32
template <int R, int RQ>
struct S1I {
    static constexpr int U = (R % RQ == 0) ? R : ((RQ % R == 0) ? RQ : R * RQ / 2);   // lcm for the pairs used: (6,4) -> 12, (4,4) -> 4
    static constexpr int Q_STAGE = 8192;
    static constexpr int LOADS = 4;
    static constexpr int LDS_BYTES = RQ * Q_STAGE + 2 * S1_STATE_BYTES_(S1I_SCAP);
};

template <int R, int RQ>
__device__ __forceinline__ void s1_body_i8w(const S1Params& p, const int chunk_id) {
    typedef S1I<R, RQ> X;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem;
    const S1State stA = s1_state(smem + RQ * X::Q_STAGE);
    const S1State stB = s1_state(smem + RQ * X::Q_STAGE + S1_STATE_BYTES_(S1I_SCAP));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform; ck.f = pseudo-field
    const int f = ck.f;
    const int t0 = ck.t0;
    int t1 = ck.t1;
    if (p.sample) t1 = min(t1, t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    const size_t wgq0 = (size_t)chunk_id * p.qw;      // qw == 128
    s1_state_init(stA, p, f, 0);
    s1_state_init(stB, p, f, 64);

    const int n_pairs = p.n_steps >> 1;
    const int qoff = j * 32 + ((h ^ ((j >> 3) & 1)) << 4);   // query fragment inside a 1 KB (32 queries x 16 dims) LDS block
    const size_t stage_bytes = 2048;
    const size_t tile_jump = (size_t)3 * n_pairs * stage_bytes;
    const char* dnext = (const char*)p.slab + (size_t)ck.base + ((size_t)(4 * t0 + w) * n_pairs) * stage_bytes + j * 32 + h * 16;
    const char* const dlast = (const char*)p.slab + (size_t)ck.base + ((size_t)(4 * (t1 - 1) + w) * n_pairs + (n_pairs - 1)) * stage_bytes + j * 32 + h * 16;
    const char* const qbase = (const char*)p.qt + lane * 16 + w * 1024;   // wave w loads pieces w and w + 4 of the 8 KB stage
    int sd_next = 0, sq_next = 0;
    u32x4 dr0[R], dr1[R];
#define S1I_QDMA(S, D)                                                                                           \
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(S), "s"(__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(D))) : "memory")
#define S1I_ISSUE_D(SLOT)                                                                                 \
    do {                                                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dr0[SLOT]) : "v"(dnext) : "memory");    \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(dr1[SLOT]) : "v"(dnext) : "memory"); \
        const char* nx_ = dnext + stage_bytes;                                                            \
        if (++sd_next == n_pairs) {                                                                       \
            sd_next = 0;                                                                                  \
            nx_ += tile_jump;                                                                             \
        }                                                                                                 \
        dnext = (unsigned long long)nx_ <= (unsigned long long)dlast ? nx_ : dlast;                       \
    } while (0)
#define S1I_ISSUE_Q(SLOT)                                                                                 \
    do {                                                                                                  \
        const char* qs_ = qbase + (size_t)sq_next * X::Q_STAGE;                                           \
        char* qd_ = qring + (SLOT) * X::Q_STAGE + w * 1024;                                               \
        S1I_QDMA(qs_, qd_);                                                                               \
        S1I_QDMA(qs_ + 4096, qd_ + 4096);                                                                 \
        if (++sq_next == n_pairs) sq_next = 0;                                                            \
    } while (0)
    // prologue: the issue order of the steady state (iteration i issues docs of stage i + R - 1, then queries of stage i + RQ - 1)
#pragma unroll
    for (int i = -(R - 1); i < 0; ++i) {
        S1I_ISSUE_D((i + R - 1) % R);
        if (i + RQ - 1 >= 0) S1I_ISSUE_Q((i + RQ - 1) % RQ);
    }

    for (int t = t0; t < t1; ++t) {
        f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, b00 = {0}, b01 = {0}, b10 = {0}, b11 = {0};
        for (int s0 = 0; s0 < n_pairs; s0 += X::U) {
#pragma unroll
            for (int u = 0; u < X::U; ++u) {
                // the stage's query pieces landed (own pieces: counted wait -- the RQ - 2 younger stages may stay in flight; the other
                // waves' pieces: the barrier); its doc registers were loaded before those pieces
                asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(dr0[u % R]), "+v"(dr1[u % R]) : "n"((RQ - 2) * X::LOADS) : "memory");
                const char* curq = qring + (u % RQ) * X::Q_STAGE + qoff;
                const u32x4 x0 = dr0[u % R], x1 = dr1[u % R];
                S1I_ISSUE_D((u + R - 1) % R);
                S1I_ISSUE_Q((u + RQ - 1) % RQ);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const f16x8 qa0 = *(const f16x8*)(curq + ks * 4096), qa1 = *(const f16x8*)(curq + ks * 4096 + 1024);
                    const f16x8 qb0 = *(const f16x8*)(curq + ks * 4096 + 2048), qb1 = *(const f16x8*)(curq + ks * 4096 + 3072);
                    const f16x8 e0 = i8_cvt8(x0[2 * ks], x0[2 * ks + 1]), e1 = i8_cvt8(x1[2 * ks], x1[2 * ks + 1]);
                    a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa0, a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa1, a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa0, a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa1, a11, 0, 0, 0);
                    b00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb0, b00, 0, 0, 0);
                    b01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb1, b01, 0, 0, 0);
                    b10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb0, b10, 0, 0, 0);
                    b11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb1, b11, 0, 0, 0);
                }
            }
        }
        if (p.dbg & 1) {
            asm volatile("" ::"v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(b00), "v"(b01), "v"(b10), "v"(b11));
            continue;
        }
        if (p.sample == 2) {
            s1_sample_top2(p, ck, t - t0, t, w, a00, a01, a10, a11, 0);
            s1_sample_top2(p, ck, t - t0, t, w, b00, b01, b10, b11, 64);
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier A
        s1_epilogue_append<S1I_SCAP>(p, stA, ck.n_rows, t, w, wgq0, a00, a01, a10, a11);
        s1_epilogue_append<S1I_SCAP>(p, stB, ck.n_rows, t, w, wgq0 + 64, b00, b01, b10, b11);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier B
        s1_epilogue_finish<S1I_SCAP, true>(p, stA, w, wgq0);
        s1_epilogue_finish<S1I_SCAP, true>(p, stB, w, wgq0 + 64);
    }
#undef S1I_ISSUE_D
#undef S1I_ISSUE_Q
#undef S1I_QDMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.sample != 2) {
        s1_flush<S1I_SCAP, true>(p, stA, w, wgq0, 0);
        s1_flush<S1I_SCAP, true>(p, stB, w, wgq0 + 64, 64);
    }
}
// (a query ring as deep as the doc ring -- <6, 6> with 16 staged survivors per query, so that it still fits twice per CU -- was
//  measured: the k-loop alone takes 1.51 ms either way, the whole kernel 2.87 ms against 2.14 (149 spilled VGPRs): the loop is
//  not short of loads in flight, it is bound by MFMA issue + fragment reads at 2 waves per SIMD)
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_i8w_kernel(const S1Params p) { s1_body_i8w<6, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_i8w_sample_kernel(const S1Params p) { s1_body_i8w<6, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_i8w4_kernel(const S1Params p) { s1_body_i8w<4, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_i8w4_sample_kernel(const S1Params p) { s1_body_i8w<4, 4>(p, p.chunk0 + (int)blockIdx.x); }
#define S1IW_LDS_BYTES (S1I<6, 4>::LDS_BYTES)

// ---------------------------------------------------------------------------------------------------------------------
// Build.  Source = the fp16 screen slab of one field (unique rows, tiled_offset_bf16 layout).
// ---------------------------------------------------------------------------------------------------------------------
struct I8Field {          // per pseudo-field pf = 2 f + segment
    float lam, inv_lam;   // quantisation step (units of the screen's fp16 values)
    float err;            // largest residual norm |h - lam d8|_2 of the segment's rows (leaned up)
    float ratio;          // lam / lam of segment 0 of the same field: acc8 of this segment * ratio = segment-0 units
    int seg_base;         // first entry of the segment in the field's map8 table
    int n_rows;           // rows of the segment
    int pad0, pad1;
};

// rmax[u] = largest |h| of unique row u.  grid = n_blk_u (64-row blocks), block 256: thread (rr = tid >> 2, pp = tid & 3) walks
// k-steps pp, pp + 4, ... of row rr (32 contiguous bytes per (row, k-step)).
__global__ void __launch_bounds__(256) mfar_i8_rowmax_kernel(const _Float16* __restrict__ scr, int n_steps, int n_unique, float* __restrict__ rmax) {
    const int rr = threadIdx.x >> 2, pp = threadIdx.x & 3;
    const long long u = (long long)blockIdx.x * 64 + rr;
    float m = 0.0f;
    if (u < n_unique)
        for (int s = pp; s < n_steps; s += 4) {
            const _Float16* g = scr + ((size_t)blockIdx.x * n_steps + s) * 1024 + rr * 16;
            const f16x8 a = *(const f16x8*)g, b = *(const f16x8*)(g + 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) m = fmaxf(m, fmaxf(fabsf((float)a[i]), fabsf((float)b[i])));
        }
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    if (pp == 0 && u < n_unique) rmax[u] = m;
}
// flag[u] = 1 when row u goes to segment 1 (largest |h| above the cap of segment 0)
__global__ void __launch_bounds__(256) mfar_i8_flag_kernel(const float* __restrict__ rmax, int n, float cap0, u32* __restrict__ flag) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < n) flag[u] = rmax[u] > cap0 ? 1u : 0u;
}
// map8: segment 0 = the unflagged rows in order (entries 0 .. n0), segment 1 = the flagged rows in order (entries n0 ..); ex = exclusive
// scan of flag
__global__ void __launch_bounds__(256) mfar_i8_map_kernel(const u32* __restrict__ flag, const u32* __restrict__ ex, int n, int n0, int* __restrict__ map8) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    if (flag[u]) map8[n0 + (int)ex[u]] = u;
    else map8[u - (int)ex[u]] = u;
}
// One segment: quantise + residual norms.  grid = 2 * n_blk8 (32-row doc blocks), block 256: thread (j = tid >> 3, part = tid & 7)
// -> h = part & 1, pairs (part >> 1), (part >> 1) + 4, ...; err2max = max over the rows of the squared residual norm (float bits).
__global__ void __launch_bounds__(256) mfar_i8_quant_kernel(const _Float16* __restrict__ scr, int n_steps, const int* __restrict__ map8seg, int n_rows,
                                                            float lam, unsigned char* __restrict__ out, u32* __restrict__ err2max) {
    const int j = threadIdx.x >> 3, part = threadIdx.x & 7, h = part & 1;
    const int n_pairs = n_steps >> 1;
    const long long row = (long long)blockIdx.x * 32 + j;      // local row of the segment
    const bool live = row < n_rows;
    const long long u = live ? map8seg[row] : 0;
    const float inv = 1.0f / lam;
    float e2 = 0.0f;
    for (int pr = part >> 1; pr < n_pairs; pr += 4) {
        u32 wds[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
        if (live) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 v = *(const f16x8*)(scr + tiled_offset_bf16(n_steps, u, pr * 32 + ks * 16 + h * 8));
                u32 lo = 0, hi = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float x = (float)v[i];
                    float d = __builtin_rintf(x * inv);
                    d = fminf(127.0f, fmaxf(-127.0f, d));
                    if (!(d == d)) d = 0.0f;                    // NaN input: the residual below poisons err (the level is then not used)
                    const float r = x - d * lam;
                    e2 = __builtin_fmaf(r, r, e2);
                    const u32 b = (u32)((int)d + 128) & 0xFFu;
                    if (i < 4) lo |= b << (8 * i);
                    else hi |= b << (8 * (i - 4));
                }
                wds[2 * ks] = lo;
                wds[2 * ks + 1] = hi;
            }
        }
        *(u32x4*)(out + i8_offset(n_pairs, row, pr * 32) + h * 16) = u32x4{wds[0], wds[1], wds[2], wds[3]};
    }
    e2 += __shfl_xor(e2, 1);
    e2 += __shfl_xor(e2, 2);
    e2 += __shfl_xor(e2, 4);
    u32 b = __float_as_uint(e2) & 0x7FFFFFFFu;               // NaN bits order above inf: a non-finite row poisons the maximum
    for (int off = 8; off < 64; off <<= 1) b = max(b, (u32)__shfl_xor((int)b, off));
    if ((threadIdx.x & 63) == 0 && b) atomicMax(err2max, b);
}
__global__ void mfar_i8_fields_kernel(I8Field* __restrict__ fl, const u32* __restrict__ err2max, int n_pf, int E) {
    const int pf = threadIdx.x;
    if (pf >= n_pf) return;
    // the fp32 sum of squares can be low by (K + 2) u32 relative: lean up; every residual was formed from the ROUNDED product
    // d8 * lam (|d8| <= 127): up to 127 lam 2^-23 per element off the real one
    fl[pf].err = sqrtf(__uint_as_float(err2max[pf])) * 1.001f + sqrtf((float)E) * 127.0f * fl[pf].lam * 1.2e-7f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Per launch: eps8 and the scan thresholds.  One thread per (pseudo-field, query).  grid = ceil(2 F qw / 256).
//   vk     [2 F, qw] k-th largest sampled acc8 of the pseudo-field (own units; mfar_sample_tau_kernel), -inf = no bound
//   epsA   [2 F, qw] out: eps8 in SEGMENT-0 acc units (the units the filter compares in)
//   tg     [2 F, qw] out: non-strict thresholds of the scan in the pseudo-field's OWN acc units
// Bound (scaled-16 units = units of sum_i Q_i h_i; Qn >= |Q|_2, Hn >= |h|_2 of any row, K = dim, u32 = 2^-24):
//   |exact - a8| <= eps16  +  Qn err  +  2 K u32 Qn (Hn + err)   [fp32 accumulation of K exact products, any order]
//                   + 8 u32 Qn Hn                                [the scalings / additions of a8 below this line, 1 ulp each]
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfar_i8_prep_kernel(const float* __restrict__ eps16, const ScreenQuery* __restrict__ qinfo,
                                                           const ScreenField* __restrict__ sf, const I8Field* __restrict__ fl,
                                                           const float* __restrict__ vk, float* __restrict__ epsA, float* __restrict__ tg, int F,
                                                           int qw, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * F * qw) return;
    const int pf = i / qw, q = i - pf * qw, f = pf >> 1;
    const float K = (float)E, u32f = 5.9604645e-8f;
    const ScreenQuery qi = qinfo[q];
    const ScreenField s = sf[f];
    const I8Field me = fl[pf], s0 = fl[2 * f];
    const float Qn = qi.norm * qi.scale * 1.001f + sqrtf(K) * 5.9604645e-8f;
    const float Hn = s.dnorm_max * s.scale * 1.001f + sqrtf(K) * 5.9604645e-8f;
    auto eps_of = [&](const I8Field& g) {
        const float e16 = eps16[f * qw + q] * (qi.scale * s.scale) * 1.000001f;
        const float e = e16 + 1.01f * Qn * g.err + 2.02f * K * u32f * Qn * (Hn + g.err) + 8.0f * u32f * Qn * Hn;
        return e * s0.inv_lam * 1.00001f;
    };
    const float e_me = eps_of(me), e_0 = eps_of(s0);
    epsA[i] = e_me;
    const float v = vk[(2 * f) * qw + q];       // segment 0's sample only
    float t = -__builtin_inff();
    if (v > -__builtin_inff()) {
        // rows of this segment qualify when  a ratio + eps_me >= v - eps_0  (segment-0 units)
        t = (v - e_0 - e_me) / me.ratio;
        t -= fabsf(t) * 1e-5f + 1e-30f;
    }
    if (!(e_me < __builtin_inff())) t = -__builtin_inff();     // non-finite bound: everything qualifies (the lists overflow, the repair decides)
    tg[i] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// Filter: the certified superset S of one (query, field).  grid = Qt * nf, block 256.
// ---------------------------------------------------------------------------------------------------------------------
#define I8_CAP 2048                 // rows of S the second level takes per (query, field); more -> the exact pass decides
#define I8_MAX_CHUNKS 1024
struct I8FilterParams {
    const uint2* lists;             // [n_chunks * qw][S1_CAP]  (acc8 bits, local row of the pseudo-field)
    const int* list_cnt;            // [n_chunks * qw]; -1 = the list was closed (overflow)
    const int* fchunk;              // [2 F + 1]
    const I8Field* fl;              // [2 F]
    const float* epsA;              // [2 F, qw]
    const int* map8;                // [F][ustride]
    long long ustride;
    long long* out_ids;             // [Qt, nf, I8_CAP] unique-row numbers
    int* out_cnt;                   // [Qt * nf]
    int* fail;                      // certificate flags of the batch (mfar_screen_certify_kernel)
    int* stats;                     // [4]: lists, appended entries, survivors, failed lists (statistics)
    int f0, nf, k, qw;
};
template <int NPT>
__global__ void __launch_bounds__(256) mfar_i8_filter_kernel(const I8FilterParams p) {
    __shared__ int pre[I8_MAX_CHUNKS + 1];
    __shared__ int red[36], wsum[4], bad_s, nsurv_s;
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    const int c_lo = p.fchunk[2 * f], c_mid = p.fchunk[2 * f + 1], c_hi = p.fchunk[2 * f + 2];
    const int nch = c_hi - c_lo;
    if (threadIdx.x == 0) {
        bad_s = nch > I8_MAX_CHUNKS ? 1 : 0;
        nsurv_s = 0;
    }
    __syncthreads();
    // exclusive prefix of the chunk counts: thread t owns chunks 4 t .. 4 t + 3
    int cn[4], mine = 0;
    bool closed = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * (int)threadIdx.x + i;
        int v = 0;
        if (c < nch && c < I8_MAX_CHUNKS) {
            v = p.list_cnt[(size_t)(c_lo + c) * p.qw + ql];
            if (v < 0) {
                closed = true;
                v = 0;
            }
        }
        cn[i] = v;
        mine += v;
    }
    if (closed) bad_s = 1;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * (int)threadIdx.x + i;
        if (c < I8_MAX_CHUNKS) pre[c] = base;
        base += cn[i];
    }
    if (threadIdx.x == 255) pre[I8_MAX_CHUNKS] = base;
    __syncthreads();
    const int N = pre[I8_MAX_CHUNKS];
    const bool bad = bad_s != 0 || N > NPT * 256;
    const size_t ob = (size_t)ql * p.nf + fo;
    if (bad) {      // workgroup-uniform
        if (threadIdx.x == 0) {
            p.out_cnt[ob] = 0;
            atomicOr(&p.fail[f], 1);
            atomicOr(&p.fail[MFAR_MAX_FIELDS], 1);
            atomicAdd(&p.fail[MFAR_MAX_FIELDS + 1], 1);
            if (p.stats) {
                atomicAdd(&p.stats[0], 1);
                atomicAdd(&p.stats[3], 1);
            }
        }
        return;
    }
    const I8Field f0s = p.fl[2 * f], f1s = p.fl[2 * f + 1];
    const float e0 = p.epsA[(2 * f) * p.qw + ql], e1 = p.epsA[(2 * f + 1) * p.qw + ql];
    const int nchc = min(nch, I8_MAX_CHUNKS);
    u32 hi[NPT], idl[NPT];
    float uu[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int e = (int)threadIdx.x + 256 * i;
        hi[i] = 0u;
        idl[i] = 0u;
        uu[i] = -__builtin_inff();
        if (e < N) {
            int lo_ = 0, hi_ = nchc;                          // largest c with pre[c] <= e
            while (hi_ - lo_ > 1) {
                const int mid = (lo_ + hi_) >> 1;
                if (pre[mid] <= e) lo_ = mid;
                else hi_ = mid;
            }
            const uint2 en = p.lists[((size_t)(c_lo + lo_) * p.qw + ql) * S1_CAP + (e - pre[lo_])];
            const bool seg1 = c_lo + lo_ >= c_mid;
            const float a = __uint_as_float(en.x) * (seg1 ? f1s.ratio : 1.0f);
            const float ee = seg1 ? e1 : e0;
            const float l = a - ee;
            hi[i] = l == l ? max(f2ord(l), 1u) : 1u;          // 0 marks an empty slot; NaN ranks lowest
            uu[i] = a == a ? a + ee : __builtin_inff();       // a NaN score survives (the next level sees its row)
            idl[i] = en.y | (seg1 ? 0x80000000u : 0u);
        }
        asm volatile("" : "+v"(hi[i]));
    }
    // L_k = k-th largest l (radix descent on the orderable bits; ballot counts, see block_topk_regs)
    float Lk = -__builtin_inff();
    if (N >= p.k) {
        u32 T = 0;
        int parity = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NPT; ++i) c += __popcll(__ballot(hi[i] >= cand));
            const int tot = block_sum_uniform(c, red, parity);
            parity ^= 1;
            if (tot >= p.k) T = cand;
        }
        Lk = T > 1u ? ord2f(T) : -__builtin_inff();
    }
    long long* out = p.out_ids + ob * I8_CAP;
    const int* map0 = p.map8 + (size_t)f * p.ustride + f0s.seg_base;
    const int* map1 = p.map8 + (size_t)f * p.ustride + f1s.seg_base;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const bool keep = hi[i] != 0u && uu[i] >= Lk;
        const int pos = wave_reserve(&nsurv_s, keep);
        if (keep && pos < I8_CAP) out[pos] = (long long)((idl[i] & 0x80000000u) ? map1[idl[i] & 0x7FFFFFFFu] : map0[idl[i]]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int S = nsurv_s;
        p.out_cnt[ob] = S <= I8_CAP ? S : 0;
        if (S > I8_CAP) {
            atomicOr(&p.fail[f], 1);
            atomicOr(&p.fail[MFAR_MAX_FIELDS], 1);
            atomicAdd(&p.fail[MFAR_MAX_FIELDS + 1], 1);
        }
        if (p.stats) {
            atomicAdd(&p.stats[0], 1);
            atomicAdd(&p.stats[1], N);
            atomicAdd(&p.stats[2], min(S, I8_CAP));
            if (S > I8_CAP) atomicAdd(&p.stats[3], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Select: the k' best of S by the fp16 level's approximate score, in the screen's list format.  grid = Qt * nf, block 256.
// ---------------------------------------------------------------------------------------------------------------------
struct I8SelectParams {
    const long long* ids;           // [Qt, nf, I8_CAP]
    const float* a16;               // [Qt, nf, I8_CAP] sum_i q_i h_i (query unscaled; mfar_score_rows_kernel<SRC_F16G>, qm == nullptr)
    const int* cnt;                 // [Qt * nf]
    const ScreenQuery* qinfo;
    long long* sid;                 // [Qt, nf, kp] out
    float* ssc;                     // [Qt, nf, kp] out: a16 * query scale = the units of the screened pass
    int* scnt;                      // [Qt * nf] out
    int nf, kp;
};
__global__ void __launch_bounds__(256) mfar_i8_select_kernel(const I8SelectParams p) {
    __shared__ u64 keys[I8_CAP], sel[SEL_MAX_K], sorted[SEL_MAX_K];
    __shared__ int red[36];
    const int ql = blockIdx.x / p.nf;
    const size_t lb = (size_t)blockIdx.x;
    const int n = min(p.cnt[lb], I8_CAP);
    const float sq = p.qinfo[ql].scale;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float a = p.a16[lb * I8_CAP + i];
        const long long id = p.ids[lb * I8_CAP + i];
        keys[i] = make_key(a == a ? a * sq : -__builtin_inff(), (u32)id);
    }
    const int m = block_topk_sorted<I8_CAP / 256>(keys, n, p.kp, sel, sorted, red);
    for (int r = threadIdx.x; r < p.kp; r += blockDim.x) {
        p.sid[lb * p.kp + r] = r < m ? (long long)key_id(sorted[r]) : -1;
        p.ssc[lb * p.kp + r] = r < m ? key_score(sorted[r]) : -__builtin_inff();
    }
    if (threadIdx.x == 0) p.scnt[lb] = m;
}
