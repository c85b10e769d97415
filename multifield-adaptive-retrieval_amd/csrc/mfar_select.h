// mfar_select.h -- the small kernels around stage 1: list merge, candidate union, candidate scoring (stage 2),
// field-weight softmax head + weighted sum + final top-k, and the row-major <-> tiled layout converters.
#pragma once
#include "mfar_device.h"
#include "mfar_stage1.h"

// ---------------------------------------------------------------------------------------------------------
// Layout converters
// ---------------------------------------------------------------------------------------------------------
// src[n, E] row-major fp32 -> tiled slab rows [row0, row0+n) of one field.  One thread per 16-byte granule.
__global__ void mfar_tile_rows_kernel(const float* __restrict__ src, float* __restrict__ field_base, long long row0,
                                      long long n, int E) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 2;  // granules per row
    if (gid >= n * gpr) return;
    const long long r = gid / gpr;
    const int e = (int)(gid - r * gpr) << 2;
    const f32x4 v = *(const f32x4*)(src + r * E + e);
    *(f32x4*)(field_base + tiled_offset(E >> 4, row0 + r, e)) = v;
}
// inverse: tiled rows -> dst[n, E] row-major
__global__ void mfar_untile_rows_kernel(const float* __restrict__ field_base, float* __restrict__ dst, long long row0,
                                        long long n, int E) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 2;
    if (gid >= n * gpr) return;
    const long long r = gid / gpr;
    const int e = (int)(gid - r * gpr) << 2;
    *(f32x4*)(dst + r * E + e) = *(const f32x4*)(field_base + tiled_offset(E >> 4, row0 + r, e));
}
// queries q[Q, E] (rows q0 .. q0+63, zero beyond Q) -> tiled [n_steps][64][16]
__global__ void mfar_tile_queries_kernel(const float* __restrict__ q, float* __restrict__ qt, int q0, int Q, int E) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 2;
    if (gid >= 64 * gpr) return;
    const int r = gid / gpr;
    const int e = (gid - r * gpr) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q0 + r < Q) v = *(const f32x4*)(q + (size_t)(q0 + r) * E + e);
    *(f32x4*)(qt + lds_image_offset(E >> 4, r, e)) = v;
}

// bf16 slab variants: rows are rounded to bf16 (RNE) on the way in and widened exactly on the way out
__global__ void mfar_tile_rows_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ field_base,
                                           long long row0, long long n, int E) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 3;  // 16-byte granules (8 bf16) per row
    if (gid >= n * gpr) return;
    const long long r = gid / gpr;
    const int e = (int)(gid - r * gpr) << 3;
    const f32x4 a = *(const f32x4*)(src + r * E + e), b = *(const f32x4*)(src + r * E + e + 4);
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = (short)f2bf(a[i]);
        o[4 + i] = (short)f2bf(b[i]);
    }
    *(bf16x8*)(field_base + tiled_offset_bf16(E >> 4, row0 + r, e)) = o;
}
__global__ void mfar_untile_rows_bf16_kernel(const unsigned short* __restrict__ field_base, float* __restrict__ dst,
                                             long long row0, long long n, int E) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 3;
    if (gid >= n * gpr) return;
    const long long r = gid / gpr;
    const int e = (int)(gid - r * gpr) << 3;
    const bf16x8 v = *(const bf16x8*)(field_base + tiled_offset_bf16(E >> 4, row0 + r, e));
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[r * E + e + i] = bf2f((unsigned short)v[i]);
}
// queries q[Q, E] -> per k-step [4][64][16] bf16: three EXACT terms hi + mid + lo = q (bf16 keeps 8 significant bits, the
// residual of a round-to-nearest split is exactly representable in fp32, and three terms cover fp32's 24 bits), 4th = 0
__global__ void mfar_tile_queries_bf16_kernel(const float* __restrict__ q, unsigned short* __restrict__ qt, int q0, int Q, int E) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 3;
    if (gid >= 64 * gpr) return;
    const int r = gid / gpr;
    const int e = (gid - r * gpr) << 3;
    bf16x8 hi, mid, lo, zero;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = (q0 + r < Q) ? q[(size_t)(q0 + r) * E + e + i] : 0.0f;
        const unsigned short h = f2bf(x);
        const float r1 = x - bf2f(h);
        const unsigned short m = f2bf(r1);
        const float r2 = r1 - bf2f(m);
        hi[i] = (short)h;
        mid[i] = (short)m;
        lo[i] = (short)f2bf(r2);
        zero[i] = 0;
    }
    const int step = e >> 4;
    const size_t in_tile = tiled_offset_bf16(E >> 4, r, e) - (size_t)step * 1024;   // offset inside one [64][16] tile
    unsigned short* base = qt + (size_t)step * 4096;                                 // 4 tiles of 1024 bf16 per k-step
    *(bf16x8*)(base + in_tile) = hi;
    *(bf16x8*)(base + 1024 + in_tile) = mid;
    *(bf16x8*)(base + 2048 + in_tile) = lo;
    *(bf16x8*)(base + 3072 + in_tile) = zero;
}

// ---------------------------------------------------------------------------------------------------------
// K_B: merge the per-workgroup lists of one (query, field) into the final sorted top-k list.
//   grid = Qt * F workgroups (Qt <= 64 queries of this pass), dynamic LDS = n_chunks * k * 8 bytes.
// ---------------------------------------------------------------------------------------------------------
struct MergeParams {
    const uint2* lists;   // [n_chunks_total * qw][S1_CAP]
    const int* list_cnt;  // [n_chunks_total * qw]
    const int* fchunk;    // [F + 1] first chunk of every field in the pass's chunk table (chunks of a field are consecutive)
    long long* out_ids;   // [Q, nf, k] global ids (nullptr: threshold-only pass)
    float* out_scores;    // [Q, nf, k]
    float* tau_out;       // [F, qw] or nullptr: k-th best score of the merged list (-inf when fewer than k entries)
    int* cnt_out;         // [Qt * nf] or nullptr: entries of the merged list
    const int* only_failed;  // [F] or nullptr: only fields whose flag is set are merged (screen fall-back pass)
    long long row_offset;
    int f0, nf;           // this launch merges fields [f0, f0 + nf): grid = Qt * nf, output rows are nf wide
    int max_chunks;       // largest chunk count of a field (sizes the LDS staging of the non-register variant)
    int k, q0, sentinel;
    int qw;               // query columns of the stage-1 pass (64 / 128): stride of the per-chunk tables and of tau_out
    // two-level merge (fields cut into more chunks than one merge can hold): level 1 merges GROUPS of chunks -- "field" f is
    // then a group, fchunk the group boundaries, gfield[f] the real field (for only_failed) -- and writes its result in the
    // chunk-list format (out_lists / out_cnt, list (f, query)), which level 2 merges like chunk lists
    const int* gfield;    // [n_groups] or nullptr
    uint2* out_lists;     // [n_groups * qw][S1_CAP] or nullptr
    int* out_cnt;         // [n_groups * qw]
    u32 skip_mask;        // bit f: field f has no chunks in this pass's table (switched off, mfar_hip.hip AUTO-OFF): its lists are
                          // written EMPTY without touching the chunk lists (a field without chunks has no slots there to read)
};
// the outputs of an empty merged list (query ql, output field slot fo, field f)
__device__ __forceinline__ void merge_write_empty(const MergeParams& p, int ql, int fo, int f) {
    if (p.tau_out && threadIdx.x == 0) p.tau_out[f * p.qw + ql] = -__builtin_inff();
    if (p.cnt_out && threadIdx.x == 0) p.cnt_out[ql * p.nf + fo] = 0;
    if (p.out_lists) {
        if (threadIdx.x == 0) p.out_cnt[(size_t)f * p.qw + ql] = 0;
        return;
    }
    if (!p.out_ids) return;
    const size_t ob = ((size_t)(p.q0 + ql) * p.nf + fo) * p.k;
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        p.out_ids[ob + r] = p.sentinel ? 0 : -1;
        p.out_scores[ob + r] = p.sentinel ? 0.0f : -__builtin_inff();
    }
}
// LDS carve-up shared by the selection kernels (everything in the dynamic region: 16-byte aligned base)
//   keys[n_keys] u64 | sel[SEL_MAX_K] u64 | sorted[SEL_MAX_K] u64 | red[32] int | misc[4] int
#define SEL_MAX_K 256   // internal depth limit (the screened stage-1 pass keeps k + 64 rows; the ABI limit is MFAR_MAX_K)
#define SEL_LDS_BYTES(n_keys) ((size_t)(n_keys) * 8 + 2 * SEL_MAX_K * 8 + 36 * 4)
#define MIX_LDS_BYTES(C, E, F) (SEL_LDS_BYTES(C) + 3 * MFAR_MAX_FIELDS * 4 + (size_t)(E) * 4 + (size_t)(E) * (F) * 4)
struct SelLds {
    u64* keys;
    u64* sel;
    u64* sorted;
    int* red;
    int* misc;
};
__device__ __forceinline__ SelLds sel_lds(char* smem, int n_keys) {
    SelLds s;
    s.keys = (u64*)smem;
    s.sel = s.keys + n_keys;
    s.sorted = s.sel + SEL_MAX_K;
    s.red = (int*)(s.sorted + SEL_MAX_K);
    s.misc = s.red + 32;
    return s;
}

template <int NPT>
__global__ void __launch_bounds__(256) mfar_merge_lists_kernel(const MergeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, p.max_chunks * p.k);
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    if (p.only_failed && !p.only_failed[p.gfield ? p.gfield[f] : f]) return;   // workgroup-uniform
    if ((p.skip_mask >> (p.gfield ? p.gfield[f] : f)) & 1u) {                  // workgroup-uniform
        merge_write_empty(p, ql, fo, f);
        return;
    }
    const int c_lo = p.fchunk[f], n_chunks = p.fchunk[f + 1] - c_lo;
    // chunk counts first (LDS), then the entries eight chunks at a time: the global loads of a round are independent,
    // so their latencies overlap instead of adding up
    int* cnts = (int*)L.sel;   // the selection scratch is free until block_topk_sorted runs (<= 128 chunks per field)
    for (int c = threadIdx.x; c < n_chunks; c += blockDim.x) cnts[c] = min(p.list_cnt[(size_t)(c_lo + c) * p.qw + ql], p.k);
    if (threadIdx.x == 0) L.misc[0] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        for (int c0 = 0; c0 < n_chunks; c0 += 8) {
            uint2 e[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u;
                ok[u] = c < n_chunks && r < cnts[c < n_chunks ? c : 0];
                e[u] = make_uint2(0u, 0u);
                if (ok[u]) e[u] = p.lists[((size_t)(c_lo + c) * p.qw + ql) * S1_CAP + r];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int pos = wave_reserve(&L.misc[0], ok[u]);
                if (ok[u]) L.keys[pos] = make_key(__uint_as_float(e[u].x), e[u].y);
            }
        }
    }
    __syncthreads();
    const int n = L.misc[0];
    // the lists are usually far from full (the sample pass's threshold keeps most chunks short): pick the register
    // budget of the selection by the ACTUAL key count (workgroup-uniform)
    int m;
    if (NPT > 8 && n <= 8 * 256) m = block_topk_sorted<8>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    else if (NPT > 16 && n <= 16 * 256) m = block_topk_sorted<16>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    else if (NPT > 32 && n <= 32 * 256) m = block_topk_sorted<32>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    else m = block_topk_sorted<NPT>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    if (p.tau_out && threadIdx.x == 0) p.tau_out[f * p.qw + ql] = m == p.k ? key_score(L.sorted[p.k - 1]) : -__builtin_inff();
    if (p.cnt_out && threadIdx.x == 0) p.cnt_out[ql * p.nf + fo] = m;
    if (p.out_lists) {   // level 1 of a two-level merge: the group's list, in chunk-list format
        uint2* ol = p.out_lists + ((size_t)f * p.qw + ql) * S1_CAP;
        for (int r = threadIdx.x; r < m; r += blockDim.x) ol[r] = make_uint2(__float_as_uint(key_score(L.sorted[r])), key_id(L.sorted[r]));
        if (threadIdx.x == 0) p.out_cnt[(size_t)f * p.qw + ql] = m;
        return;
    }
    if (!p.out_ids) return;
    const size_t ob = ((size_t)(p.q0 + ql) * p.nf + fo) * p.k;
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        if (r < m) {
            p.out_ids[ob + r] = p.row_offset + (long long)key_id(L.sorted[r]);
            p.out_scores[ob + r] = key_score(L.sorted[r]);
        } else {
            p.out_ids[ob + r] = p.sentinel ? 0 : -1;
            p.out_scores[ob + r] = p.sentinel ? 0.0f : -__builtin_inff();
        }
    }
}

// Register-resident variant for n_chunks * k <= 256 * NPT and n_chunks <= 128: slot s = c * k + r of the concatenated lists
// goes to thread s % 256, register s / 256, so the keys never pass through LDS (5 KB of static LDS instead of
// n_chunks * k * 8 bytes) and all loads of a thread are independent.  Same results as mfar_merge_lists_kernel.
// One selection over the N real entries of the field's chunk lists, NPT keys per thread (N <= NPT * TPB): entry e of the
// concatenated lists goes to thread e % TPB, register e / TPB; its chunk is found by bisection in the exclusive prefix `pre` of the
// chunk counts (LDS).  All loads of a thread are independent.
#ifdef MFAR_TRACE
__shared__ unsigned long long g_tr_merge0;
#endif
template <int NPT, int TPB>
__device__ __forceinline__ int merge_regs_select(const MergeParams& p, const int* pre, int n_chunks, int c_lo, int ql, int N, u64* sel, u64* sorted,
                                                 int* red) {
    uint2 e[NPT];
    // chunk of entry en = largest c with pre[c] <= en (empty chunks repeat their neighbour's prefix).  n_chunks <= 128: seven branch-free
    // bisection steps, written STEP-major over groups of eight entries -- the eight LDS reads of a step are independent and wait once;
    // entry-major, the compiler waited on every one of the 7 * NPT reads in turn (14 of the merge's 23 us load phase).
#pragma unroll
    for (int g = 0; g < NPT; g += 8) {
        int lo_[8], en[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s_ = (int)threadIdx.x + TPB * (g + j);
            en[j] = s_ < N ? s_ : 0;
            lo_[j] = 0;
        }
#pragma unroll
        for (int st = 64; st > 0; st >>= 1) {
            int pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = pre[lo_[j] + st < n_chunks ? lo_[j] + st : 0];
#pragma unroll
            for (int j = 0; j < 8; ++j) lo_[j] = (lo_[j] + st < n_chunks && pv[j] <= en[j]) ? lo_[j] + st : lo_[j];
        }
        int pb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[j] = pre[lo_[j]];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (g + j < NPT) e[g + j] = p.lists[((size_t)(c_lo + lo_[j]) * p.qw + ql) * S1_CAP + (N > 0 ? en[j] - pb[j] : 0)];
    }
    u32 hi[NPT], lo[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const bool ok = (int)threadIdx.x + TPB * i < N;
        hi[i] = ok ? f2ord(__uint_as_float(e[i].x)) : 0u;
        lo[i] = ok ? 0xFFFFFFFFu - e[i].y : 0u;
        asm volatile("" : "+v"(hi[i]), "+v"(lo[i]));   // keep the keys materialised: the selection re-reads them 34 times
    }
#ifdef MFAR_TRACE
    const unsigned long long tr_l = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) trace_put(11, (int)blockIdx.x, N, g_tr_merge0);
    const int mm_ = block_topk_regs<NPT>(hi, lo, N, p.k, sel, sorted, red);
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) trace_put(12, (int)blockIdx.x, NPT, tr_l);
    return mm_;
#else
    return block_topk_regs<NPT>(hi, lo, N, p.k, sel, sorted, red);
#endif
}

// Register-resident variant for n_chunks * k <= 256 * NPT and n_chunks <= 128: the keys never pass through LDS (6 KB of static LDS
// instead of n_chunks * k * 8 bytes).  The lists are usually far from full (the sample pass's threshold keeps most chunks short, a
// small shard's chunks hold a few entries each): only the N entries that exist are loaded, and the selection runs with the
// smallest register budget that holds them (workgroup-uniform) -- at the 125 k-row shard 1 400 of 12 288 slots are real and the
// merge took 156 us of every launch's ~700.  Same results as mfar_merge_lists_kernel.
template <int NPT, int TPB>
__device__ __forceinline__ void merge_lists_regs_body(const MergeParams& p) {
    __shared__ u64 sel[SEL_MAX_K], sorted[SEL_MAX_K];
    __shared__ int red[36], pre[129], wtot[2];
    const int ql = blockIdx.x / p.nf, fo = blockIdx.x - ql * p.nf, f = p.f0 + fo;
    if (p.only_failed && !p.only_failed[p.gfield ? p.gfield[f] : f]) return;   // workgroup-uniform
    if ((p.skip_mask >> (p.gfield ? p.gfield[f] : f)) & 1u) {                  // workgroup-uniform
        merge_write_empty(p, ql, fo, f);
        return;
    }
    const int c_lo = p.fchunk[f], n_chunks = p.fchunk[f + 1] - c_lo;
#ifdef MFAR_TRACE
    const unsigned long long tr_a = wall_clock64();
#endif
    // exclusive prefix of the chunk counts (n_chunks <= 128: the first two waves)
    if (threadIdx.x < 128) {
        const int c = (int)threadIdx.x < n_chunks ? max(0, min(p.list_cnt[(size_t)(c_lo + threadIdx.x) * p.qw + ql], p.k)) : 0;
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane_id() >= off) incl += v;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        pre[threadIdx.x] = incl - c;                  // wave-local; the second wave adds the first wave's total below
    }
    __syncthreads();
    if (threadIdx.x >= 64 && threadIdx.x < 128) pre[threadIdx.x] += wtot[0];
    if (threadIdx.x == 0) pre[128] = wtot[0] + wtot[1];
    __syncthreads();
    const int N = pre[128];
#ifdef MFAR_TRACE
    if (threadIdx.x == 0) g_tr_merge0 = wall_clock64();
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) trace_put(10, (int)blockIdx.x, N, tr_a);
    __syncthreads();
#endif
    int m;
    if (NPT > 8 && N <= 8 * TPB) m = merge_regs_select<8, TPB>(p, pre, n_chunks, c_lo, ql, N, sel, sorted, red);
    else if (NPT > 16 && N <= 16 * TPB) m = merge_regs_select<(NPT > 16 ? 16 : NPT), TPB>(p, pre, n_chunks, c_lo, ql, N, sel, sorted, red);
    else if (NPT > 32 && N <= 32 * TPB) m = merge_regs_select<(NPT > 32 ? 32 : NPT), TPB>(p, pre, n_chunks, c_lo, ql, N, sel, sorted, red);
    else m = merge_regs_select<NPT, TPB>(p, pre, n_chunks, c_lo, ql, N, sel, sorted, red);
    if (p.tau_out && threadIdx.x == 0) p.tau_out[f * p.qw + ql] = m == p.k ? key_score(sorted[p.k - 1]) : -__builtin_inff();
    if (p.cnt_out && threadIdx.x == 0) p.cnt_out[ql * p.nf + fo] = m;
    if (p.out_lists) {   // level 1 of a two-level merge
        uint2* ol = p.out_lists + ((size_t)f * p.qw + ql) * S1_CAP;
        for (int r = threadIdx.x; r < m; r += blockDim.x) ol[r] = make_uint2(__float_as_uint(key_score(sorted[r])), key_id(sorted[r]));
        if (threadIdx.x == 0) p.out_cnt[(size_t)f * p.qw + ql] = m;
        return;
    }
    if (!p.out_ids) return;
#ifdef MFAR_TRACE
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) trace_put(13, (int)blockIdx.x, m, tr_a);
#endif
    const size_t ob = ((size_t)(p.q0 + ql) * p.nf + fo) * p.k;
    for (int i = threadIdx.x; i < p.k; i += blockDim.x) {
        if (i < m) {
            p.out_ids[ob + i] = p.row_offset + (long long)key_id(sorted[i]);
            p.out_scores[ob + i] = key_score(sorted[i]);
        } else {
            p.out_ids[ob + i] = p.sentinel ? 0 : -1;
            p.out_scores[ob + i] = p.sentinel ? 0.0f : -__builtin_inff();
        }
    }
}
template <int NPT>
__global__ void __launch_bounds__(256, 2) mfar_merge_lists_regs_kernel(const MergeParams p) { merge_lists_regs_body<NPT, 256>(p); }
// tau[f][q] = max(base[f][q], k-th largest of the n_vals scores published by the light sample pass for (q, f)); the k-th
// largest counts only scores above tau0 and is -inf when there are fewer than k of them (then the sample gives no bound).
// One WAVE per (query, field), NV values per lane (n_vals <= 64 * NV): the 32-step radix descent on the score bits is
// NV ballots + scalar popcounts per step, no LDS and no barriers.  grid = ceil(qw * F / 4), block 256 (qw = query columns
// of the pass, 64 / 128).
// DEEP SCAN fields (S1DeepDev::mask; mfar_screen.h): the threshold is not a starting point for a running top-k' but the COMPLETE-SET
// threshold T = tau_k - band: tau_k = the k-th (not k'-th) largest published sample value -- k distinct rows score at least that, so the
// true k-th best exact score E_k >= tau_k - eps and every row of the exact top-k has approximate score >= tau_k - 2 eps (ROW MODE fields
// publish upper bounds, each within 2 eps of its exact score: tau_k - 4 eps).  With the zero sentinel and no proof of k positive
// documents (tau_k - eps <= 0 in real units) the set must hold every row whose exact score can be positive: approx >= -eps.
// info[f * qw + q] = {band, eps, position of the exact score 0, -} in scan units, for mfar_t2_collect_kernel.
struct S1DeepDev {
    u32 mask, row_mask;
    int k, sentinel, E;
    int Q;                         // live query columns of the block (the padding columns of a short block have no query row to read)
    const float* eps;              // [F, qw] real units
    const ScreenQuery* qinfo;      // [qw]
    const ScreenField* sf;         // [F]
    const float* q;                // [Qt, E] the block's queries
    const float* mean;             // [F, E]
    float4* info;                  // [F, qw]
};
template <int NV>
__global__ void __launch_bounds__(256) mfar_sample_tau_kernel(const float* __restrict__ samp, const int* __restrict__ samp_n,
                                                              int samp_stride, int f0, int nf, int k, float tau0,
                                                              const float* __restrict__ base, float* __restrict__ tau_out, int qw,
                                                              const S1DeepDev dp) {
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pair >= qw * nf) return;   // wave-uniform
    const int q = pair / nf, f = f0 + pair - q * nf;
    const bool deep = ((dp.mask >> f) & 1u) != 0u && q < dp.Q;       // wave-uniform (a padding column keeps its +inf base threshold)
    if (deep) k = dp.k;
    const int n_wave_blocks = samp_n[f];      // wave blocks the sample pass published for this field
    const int n_vals = n_wave_blocks * 2;
    u32 hi[NV];
    int n = 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int i = lane + j * 64;
        const int ii = i < n_vals ? i : 0;
        const float v = samp[((size_t)f * samp_stride + (ii >> 1)) * (size_t)(2 * qw) + q * 2 + (ii & 1)];
        const bool ok = i < n_vals && v > tau0;
        hi[j] = ok ? f2ord(v) : 0u;     // 0 = empty: below every real score's key
        n += __popcll(__ballot(ok));
    }
    float t = -__builtin_inff();
    if (n >= k) {
        u32 T = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int j = 0; j < NV; ++j) c += __popcll(__ballot(hi[j] >= cand));
            if (c >= k) T = cand;
        }
        t = ord2f(T);
    }
    if (deep) {
        const float sc = dp.qinfo[q].scale * dp.sf[f].scale;       // scan units per real unit (powers of two)
        const float e_s = dp.eps[f * qw + q] * sc;
        const float band = (((dp.row_mask >> f) & 1u) ? 4.04f : 2.02f) * e_s;
        float qm = 0.0f;                                            // q . mean(f): where the exact score 0 sits on the scan's axis
        for (int e = lane; e < dp.E; e += 64) qm = __builtin_fmaf(dp.q[(size_t)q * dp.E + e], dp.mean[(size_t)f * dp.E + e], qm);
        for (int off = 32; off > 0; off >>= 1) qm += __shfl_xor(qm, off);
        const float zero_s = -qm * sc - (fabsf(qm) * sc) * 1.0e-5f;   // (the fp32 sum of q . m: K u32 relative at most, leaning down)
        float T = t - band;
        T -= fabsf(T) * 2.0e-7f;
        if (!(t > -__builtin_inff()) || !(e_s < __builtin_inff())) T = -__builtin_inff();       // fewer than k sampled rows / a non-finite bound: no threshold
        else if (dp.sentinel && !(t - band > zero_s)) T = fminf(T, zero_s - 1.01f * e_s);        // (k rows above the band prove k positive documents)
        t = T;
        if (lane == 0) dp.info[f * qw + q] = make_float4(band, e_s, zero_s, 0.0f);
    }
    if (lane == 0) {
        if (base) t = fmaxf(t, base[f * qw + q]);
        tau_out[f * qw + q] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// M1: merge S shard lists (multi-GPU).  src ids/scores are addressed through a stride so the payload buffers of
// all shards can be read in place.   grid = Q * F, dynamic LDS = S * k * 8.
// ---------------------------------------------------------------------------------------------------------
struct ShardMergeParams {
    const char* payloads;      // S payloads, `payload_stride` bytes apart
    long long payload_stride;
    long long ids_off, scores_off;  // byte offsets of field_ids / field_scores inside a payload
    long long* out_ids;
    float* out_scores;
    int S, F, k, sentinel;
};
template <int NPT>
__global__ void __launch_bounds__(256) mfar_merge_shards_kernel(const ShardMergeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, p.S * p.k);
    u64* keys = L.keys;
    u64 *sel = L.sel, *sorted = L.sorted;
    int* red = L.red;
    int& n_s = L.misc[0];
    const int q = blockIdx.x / p.F, f = blockIdx.x - q * p.F;
    if (threadIdx.x == 0) n_s = 0;
    __syncthreads();
    const size_t lb = ((size_t)q * p.F + f) * p.k;
    for (int i = threadIdx.x; i < p.S * p.k; i += blockDim.x) {
        const int s = i / p.k, r = i - s * p.k;
        const char* pl = p.payloads + (size_t)s * p.payload_stride;
        const long long id = ((const long long*)(pl + p.ids_off))[lb + r];
        const float sc = ((const float*)(pl + p.scores_off))[lb + r];
        // padding entries ((0, 0.0) sentinels or (-1, -inf)) are re-created after the selection
        const bool pad = (id < 0) || (p.sentinel && id == 0 && sc == 0.0f);
        if (!pad) keys[atomicAdd(&n_s, 1)] = make_key(sc, (u32)id);
    }
    __syncthreads();
    const int n = n_s;
    const int m = block_topk_sorted<NPT>(keys, n, p.k, sel, sorted, red);
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        if (r < m) {
            p.out_ids[lb + r] = (long long)key_id(sorted[r]);
            p.out_scores[lb + r] = key_score(sorted[r]);
        } else {
            p.out_ids[lb + r] = p.sentinel ? 0 : -1;
            p.out_scores[lb + r] = p.sentinel ? 0.0f : -__builtin_inff();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// K_C: union of the F per-field id lists of one query (contrastive.py:678-679) -> sorted unique candidate ids.
//   grid = Q, block 256, LDS u32[4096].  cand [Q, Cmax] (Cmax = F*k), padded with -1.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mfar_union_kernel(const long long* __restrict__ field_ids, int F, int k,
                                                         long long* __restrict__ cand, int* __restrict__ n_cand) {
    __shared__ u32 a[4096];
    __shared__ int wsum[4], total_s;
    const int q = blockIdx.x;
    const int n = F * k;
    int N = 256;
    while (N < n) N <<= 1;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        u32 v = MFAR_INVALID_ID;
        if (i < n) {
            const long long id = field_ids[(size_t)q * n + i];
            if (id >= 0) v = (u32)id;
        }
        a[i] = v;
    }
    __syncthreads();
    // bitonic sort ascending
    for (int size = 2; size <= N; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (N >> 1); t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const u32 x = a[lo], y = a[hi];
                if ((x > y) == up) {
                    a[lo] = y;
                    a[hi] = x;
                }
            }
            __syncthreads();
        }
    }
    // unique + compaction (each thread owns a contiguous run of N/256 elements)
    const int per = N / 256;
    const int b = threadIdx.x * per;
    int mine = 0;
    for (int i = b; i < b + per; ++i) mine += (a[i] != MFAR_INVALID_ID && (i == 0 || a[i] != a[i - 1])) ? 1 : 0;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int ww = 0; ww < (int)(threadIdx.x >> 6); ++ww) wbase += wsum[ww];
    if (threadIdx.x == 0) total_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int pos = wbase + incl - mine;
    long long* out = cand + (size_t)q * n;
    for (int i = b; i < b + per; ++i)
        if (a[i] != MFAR_INVALID_ID && (i == 0 || a[i] != a[i - 1])) out[pos++] = (long long)a[i];
    __syncthreads();
    const int C = total_s;
    for (int i = C + threadIdx.x; i < n; i += blockDim.x) out[i] = -1;
    if (threadIdx.x == 0) n_cand[q] = C;
}

// ---------------------------------------------------------------------------------------------------------
// K_D: stage 2 == DenseFlatIndex.score_batch for all fields (index.py:227-232; contrastive.py:681-683).
//   out[q, c, f] = chain-ordered dot(q, slab[f, cand[q,c]]).  One thread per (c, f); query row staged in LDS.
//   grid = (ceil(C*F/256), Q).
// ---------------------------------------------------------------------------------------------------------
struct ScoreParams {
    const void* slab;        // fp32 or bf16 tiled slab
    long long field_stride;  // elements
    const float* q;          // [Q, E] row-major
    const long long* cand;   // [Q, C]
    const int* n_cand;       // [Q] or nullptr
    const int* n_cand_pf;    // [Q, F] or nullptr.  per_field mode of mfar_score_rows_kernel, C a multiple of its block size: entries of
                             // list (q, j); rows past the count are neither gathered nor written (tier 2 of the certified screen)
    float* out;              // [Q, C, F]
    long long row_offset;
    int n_rows, n_steps, E, F, C;
    int per_field;           // != 0: cand is [Q, F, C] (one list per field), out is [Q, F, C]: row (f, c) is scored for field f only
    // per_field mode of the certified screen: cand holds UNIQUE-row numbers of the field's screen slab; the row that is
    // gathered from the fp32 slab is the unique row's representative document urep[f * ustride + u] (local row, u < nuniq[f]).
    // nullptr: cand holds global doc ids.
    const int* urep;
    const int* nuniq;        // [F]
    long long ustride;
    int f0;                  // per_field mode: list (q, j) belongs to field f0 + j (F = number of fields in this launch)
    // all-fields mode: repof[f * ustride + local row] = representative of the row's group of bit-identical rows (mfar_screen.h),
    // gathered in its place; nullptr: every row is gathered itself
    const int* repof;
    // 16-bit gather slab (mfar_score_rows_kernel<SRC_F16G / SRC_BF16G>): row-major [F][n_rows] rows of g_row_bytes
    const void* gslab;
    long long g_row_bytes;
    const ScreenField* sfld;         // [F] SRC_F16G: the field's power-of-two scale
    const float* qm;         // [Q, qm_stride] SRC_F16G: q . mean(field)
    int qm_stride;
    // per_field mode of the certified screen (exact re-scoring of the k' screened rows of every list): a list is sorted by
    // APPROXIMATE score a (scaled units), |a - exact| <= eps for every row.  The k rows with the largest a all have exact >= a_(k-1) - eps,
    // so a row with a + eps < a_(k-1) - eps is strictly below k rows of its own list: it can neither enter nor tie into the exact
    // top-k and is NOT gathered (output NaN = "not scored", which mfar_screen_certify_kernel skips).  ~125 of the 192 rows are
    // gathered on the bench corpora.  nullptr: every row is gathered.
    // all-fields mode of the two-level stage 2: KNOWN pairs.  A candidate's score in the field whose stage-1 list it came from is
    // already exact (stage 1 and stage 2 walk the same fma chain: identical bits), mfar_s2_known_kernel wrote it into the
    // approximate table and set bit f of kmask[q, c].  Such a pair is not gathered: the approximate launch (kval == nullptr) leaves
    // the table entry alone, the exact launch over the SURVIVORS copies it (row c of this launch is row ksrc[q, c] of the table).
    const u32* kmask;        // [Q, C] or nullptr
    const int* ksrc;         // [Q, C] or nullptr (identity)
    const float* kval;       // [Q, C, F] or nullptr
    const float* pre_sc;     // [Q, F, C] approximate scores of the lists
    const float* pre_eps;    // [n_fields, pre_qw] eps in real units (mfar_screen_queries_kernel)
    const ScreenQuery* pre_qinfo;   // [pre_qw]
    int pre_k, pre_qw;
};
// Each wave owns 64 (candidate, field) rows, one per lane, gathers their segments cooperatively by LDS-DMA into a private
// two-slot LDS ring, and every lane then walks ITS row's segment from LDS in chain order.  No barriers: the ring is private
// to the wave, ordered by counted vmcnt waits.
// bf16 slab (this kernel, DT = 1): 32-byte segments (16 dims) 2 KB apart, 2 lanes per segment, ring slot = 2 k-steps (4 KB
//            per wave); the chain walks the dims in natural order and widens bf16 -> fp32 exactly, so scores equal the
//            oracle's bits.  (Each segment costs a whole 128-byte line: the bf16 slab is laid out for the scan.)
// fp32 slab: mfar_score_rows_f32_kernel below (whole-line gathers).
#ifndef SC_AUX
#define SC_AUX 0                             // cache policy of the row gathers (2 = non-temporal)
#endif
#define SC_SLOT_BYTES 4096
#define SC_WAVE_BYTES (2 * SC_SLOT_BYTES)   // two-slot ring; 35 KB per workgroup with the query row: fits beside two
                                            // resident stage-1 workgroups when batches are pipelined
#define SCORE_LDS_BYTES(E) ((size_t)4 * SC_WAVE_BYTES + (size_t)(E) * 4)
template <int DT>
__global__ void __launch_bounds__(256) mfar_score_candidates_kernel(const ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)(smem + 4 * SC_WAVE_BYTES);
    constexpr int SEG = DT ? 32 : 64;                 // bytes of one row per k-step
    constexpr int STEPS = DT ? 2 : 1;                 // k-steps per ring slot
    constexpr int LPS = SEG / 16;                     // lanes per segment
    constexpr int RPI = 64 / LPS;                     // rows per load instruction
    constexpr int IPS = 64 / RPI;                     // load instructions per k-step
    constexpr size_t STEP_STRIDE = DT ? 2048 : 4096;  // bytes between consecutive k-steps of a block
    const int qi = blockIdx.y;
    const int nc = p.n_cand ? p.n_cand[qi] : p.C;
    const int first = blockIdx.x * blockDim.x;
    const int idx = first + threadIdx.x;
    if (first >= nc * p.F) {
        // slots past the candidate count: define the output (NaN) so downstream never reads garbage
        if (idx < p.C * p.F) p.out[(size_t)qi * p.C * p.F + idx] = __builtin_nanf("");
        return;
    }
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) qs[e] = p.q[(size_t)qi * p.E + e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = smem + w * SC_WAVE_BYTES;
    // this lane's row
    bool valid = false;
    int rr = 0;
    const char* rowbase = (const char*)p.slab;  // harmless in-bounds address for invalid rows
    if (idx < p.C * p.F) {
        const int c = p.per_field ? idx % p.C : idx / p.F;
        const int fl = p.per_field ? idx / p.C : idx - c * p.F;     // list / field slot of this launch
        const int f = p.per_field ? p.f0 + fl : fl;                 // field of the slab
        if (c < nc) {
            long long id = p.cand[p.per_field ? ((size_t)qi * p.F + fl) * p.C + c : (size_t)qi * p.C + c];
            if (p.urep) id = (id >= 0 && id < p.nuniq[f]) ? (long long)p.urep[(size_t)f * p.ustride + id] : -1;   // unique row -> its document
            else id -= p.row_offset;
            if (id >= 0 && id < p.n_rows) {
                valid = true;
                if (p.repof && !p.per_field) id = p.repof[(size_t)f * p.ustride + id];
                rr = (int)(id & 63);
                rowbase = (const char*)p.slab + ((size_t)f * p.field_stride + (size_t)(id >> 6) * p.n_steps * 1024) * (DT ? 2 : 4) + rr * SEG;
            }
        }
    }
    // lane l fetches piece (l % LPS) of the segments of rows (l / LPS) + RPI * i
    const char* src[IPS];
#pragma unroll
    for (int i = 0; i < IPS; ++i) {
        const int r = lane / LPS + RPI * i;
        const unsigned long long b = (unsigned long long)rowbase;
        const u32 lo = __shfl((int)(u32)b, r), hi = __shfl((int)(u32)(b >> 32), r);
        src[i] = (const char*)(((unsigned long long)hi << 32) | lo) + (lane % LPS) * 16;
    }
    const int n_groups = p.n_steps / STEPS;
#define SC_ISSUE(G, SLOT)                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < STEPS; ++s_) _Pragma("unroll") for (int i = 0; i < IPS; ++i)           \
        __builtin_amdgcn_global_load_lds(                                                                         \
            (const __attribute__((address_space(1))) void*)(src[i] + (size_t)((G) * STEPS + s_) * STEP_STRIDE),    \
            (__attribute__((address_space(3))) void*)(ring + (SLOT) * SC_SLOT_BYTES + s_ * (64 * SEG) + i * 1024), 16, 0, SC_AUX)
    SC_ISSUE(0, 0);
    float acc = 0.0f;
    for (int g = 0; g < n_groups; ++g) {
        const int slot = g & 1;
        if (g + 1 < n_groups) {
            SC_ISSUE(g + 1, slot ^ 1);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the 4 loads of the next slot may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* seg = ring + slot * SC_SLOT_BYTES + lane * SEG;
        if (DT == 0) {
            const int sw = (rr >> 2) & 3;
            const float* t = (const float*)seg;
            const f32x4 c0 = *(const f32x4*)(t + ((0 ^ sw) << 2));
            const f32x4 c1 = *(const f32x4*)(t + ((1 ^ sw) << 2));
            const f32x4 c2 = *(const f32x4*)(t + ((2 ^ sw) << 2));
            const f32x4 c3 = *(const f32x4*)(t + ((3 ^ sw) << 2));
            const float* qq = qs + g * 16;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc = __builtin_fmaf(qq[x], c0[x], acc);
                acc = __builtin_fmaf(qq[4 + x], c1[x], acc);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc = __builtin_fmaf(qq[8 + x], c2[x], acc);
                acc = __builtin_fmaf(qq[12 + x], c3[x], acc);
            }
        } else {
            const int swb = (rr >> 3) & 1;
#pragma unroll
            for (int s_ = 0; s_ < STEPS; ++s_) {
                const char* t = seg + s_ * (64 * SEG);
                const bf16x8 c0 = *(const bf16x8*)(t + ((0 ^ swb) << 4));
                const bf16x8 c1 = *(const bf16x8*)(t + ((1 ^ swb) << 4));
                const float* qq = qs + (g * STEPS + s_) * 16;
#pragma unroll
                for (int x = 0; x < 8; ++x) acc = __builtin_fmaf(qq[x], bf2f((unsigned short)c0[x]), acc);
#pragma unroll
                for (int x = 0; x < 8; ++x) acc = __builtin_fmaf(qq[8 + x], bf2f((unsigned short)c1[x]), acc);
            }
        }
        // the LDS reads of this slot have been consumed before the slot is re-filled two iterations later
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef SC_ISSUE
    if (idx < p.C * p.F) p.out[(size_t)qi * p.C * p.F + idx] = valid ? acc : __builtin_nanf("");
}

// Row gathers in whole 128-byte lines.  A wave gathers 64 rows x 128 B = 8 KB per ring slot: 8 lanes per row, 8 rows per 1 KB
// LDS-DMA instruction, 8 instructions per slot; two slots per wave.  A workgroup is 2 waves (32 KB + the query row: fits
// beside two resident stage-1 workgroups).  The 16-byte pieces of a row are stored at position piece ^ ((owner lane >> 1) & 7)
// inside its 128 bytes: conflict-free ds_read_b128 for any rows.  Three sources (template SRC):
//   SRC_F32   the fp32 tiled slab: every (row, k-step pair) is one full line (mfar_device.h), lines of a row 8 KB apart.  Chain
//             order = the arithmetic contract (inside every aligned group of 8 dims: 0,4,1,5,2,6,3,7): stage-1 bits.
//   SRC_F16G  the fp16 GATHER slab of an fp32 index (row-major [F][rows][row_bytes], the screen's centred + scaled values, see
//             mfar_gslab_build_kernel): a row is E * 2 contiguous bytes, HALF the bytes of the fp32 row.  out = the APPROXIMATE
//             score acc / sf + q.m of the certified two-level stage 2 (mfar_s2_prune_kernel owns the error bound).
//   SRC_BF16G the row-major bf16 companion of a bf16 index (exact copies of the slab's values): the scan-ordered bf16 slab keeps
//             32-byte row segments 2 KB apart, every one a whole 128-byte line for a gather (4x the useful bytes); here a row is
//             contiguous.  Natural-order chain, bf16 widened exactly: the bf16 contract's bits (same as DT = 1 above).
enum { SRC_F32 = 0, SRC_F16G = 1, SRC_BF16G = 2 };
#define SCF_THREADS 128
#define SCF_SLOT_BYTES 8192
#define SCF_WAVE_BYTES (2 * SCF_SLOT_BYTES)
#define SCORE_F32_LDS_BYTES(E) ((size_t)(SCF_THREADS / 64) * SCF_WAVE_BYTES + (size_t)(E) * 4)
#define GSLAB_ROW_BYTES(E) ((size_t)(((E) + 63) / 64) * 128)   // 16-bit row-major rows padded to whole lines
template <int SRC>
__global__ void __launch_bounds__(SCF_THREADS) mfar_score_rows_kernel(const ScoreParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MFAR_TRACE
    const unsigned long long tr_t0 = wall_clock64();
    struct TrEnd { unsigned long long t0; int k; __device__ ~TrEnd() { if (threadIdx.x == 0 && (blockIdx.x & 7) == 0 && (blockIdx.y & 3) == 0) trace_put(k, (int)(blockIdx.y * gridDim.x + blockIdx.x), 0, t0); } } tr_end_ = {tr_t0, 3 + SRC};
#endif
    float* qs = (float*)(smem + (SCF_THREADS / 64) * SCF_WAVE_BYTES);
    const int qi = blockIdx.y;
    const int nc = p.n_cand ? p.n_cand[qi] : p.C;
    const int first = blockIdx.x * blockDim.x;
    const int idx = first + threadIdx.x;
    if (first >= nc * p.F) {
        // slots past the candidate count: define the output (NaN) so downstream never reads garbage
        if (idx < p.C * p.F) p.out[(size_t)qi * p.C * p.F + idx] = __builtin_nanf("");
        return;
    }
    int nc_pf = p.C;
    if (p.per_field && p.n_cand_pf) {                    // the workgroup lies inside ONE list (C % SCF_THREADS == 0): workgroup-uniform
        const int fl0 = first / p.C;
        nc_pf = p.n_cand_pf[(size_t)qi * p.F + fl0];
        if (first - fl0 * p.C >= nc_pf) return;
    }
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) qs[e] = p.q[(size_t)qi * p.E + e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = smem + w * SCF_WAVE_BYTES;
    // this lane's row
    bool valid = false, known = false;
    int rr = 0, fld = 0, kc = 0;
    const char* rowbase = (const char*)(SRC == SRC_F32 ? p.slab : p.gslab);  // harmless in-bounds address for invalid rows
    if (idx < p.C * p.F) {
        const int c = p.per_field ? idx % p.C : idx / p.F;
        const int fl = p.per_field ? idx / p.C : idx - c * p.F;     // list / field slot of this launch
        const int f = p.per_field ? p.f0 + fl : fl;                 // field of the slab
        fld = f;
        if (c < nc && c < nc_pf) {
            long long id = p.cand[p.per_field ? ((size_t)qi * p.F + fl) * p.C + c : (size_t)qi * p.C + c];
            if (p.urep) id = (id >= 0 && id < p.nuniq[f]) ? (long long)p.urep[(size_t)f * p.ustride + id] : -1;   // unique row -> its document
            else id -= p.row_offset;
            if (p.kmask && !p.per_field) {                      // a pair stage 1 already scored exactly: no gather
                kc = p.ksrc ? p.ksrc[(size_t)qi * p.C + c] : c;
                known = kc >= 0 && ((p.kmask[(size_t)qi * p.C + kc] >> f) & 1u);
                if (known) id = -1;
            }
            if (p.per_field && p.pre_sc && c >= p.pre_k) {      // provably outside the exact top-k of its list: skip the gather
                const size_t lb = ((size_t)qi * p.F + fl) * p.C;
                const float e_sc = p.pre_eps[f * p.pre_qw + qi] * (p.pre_qinfo[qi].scale * p.sfld[f].scale);
                if (p.pre_sc[lb + c] < p.pre_sc[lb + p.pre_k - 1] - 2.01f * e_sc) id = -1;
            }
            if (id >= 0 && id < p.n_rows) {
                valid = true;
                if (p.repof && !p.per_field) id = p.repof[(size_t)f * p.ustride + id];
                rr = (int)(id & 63);
                if (SRC == SRC_F32) rowbase = (const char*)p.slab + ((size_t)f * p.field_stride + (size_t)(id >> 6) * p.n_steps * 1024) * 4 + rr * 128;
                else rowbase = (const char*)p.gslab + ((size_t)f * p.n_rows + (size_t)id) * p.g_row_bytes;
            }
        }
    }
    // lane l fetches, for the rows owned by lanes r = (l / 8) + 8 i, the piece that belongs at LDS position l % 8:
    // piece (l % 8) ^ key(r), key(r) = (r >> 1) & 7.  The key follows the OWNER LANE, not the row number: the 16 lanes a
    // ds_read_b128 serves per pass (two lanes per 64-bank row pair) then always hit 16 distinct bank groups, whatever rows
    // the candidates are (keyed by the row number, random candidates collided: SQ_LDS_BANK_CONFLICT was 9 % of the wave cycles)
    const char* src[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (lane >> 3) + 8 * i;
        const unsigned long long b = (unsigned long long)rowbase;
        const u32 lo = __shfl((int)(u32)b, r), hi = __shfl((int)(u32)(b >> 32), r);
        src[i] = (const char*)(((unsigned long long)hi << 32) | lo) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    // ring slots: fp32 = k-step pairs (32 dims, lines 8 KB apart); 16-bit row-major = 64 dims, consecutive lines
    constexpr size_t SLOT_STRIDE = SRC == SRC_F32 ? 8192 : 128;
    const int n_slots = SRC == SRC_F32 ? (p.n_steps >> 1) : (p.E + 63) >> 6;
#define SCF_ISSUE(G, SLOT)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds(                                \
        (const __attribute__((address_space(1))) void*)(src[i] + (size_t)(G) * SLOT_STRIDE),                       \
        (__attribute__((address_space(3))) void*)(ring + (SLOT) * SCF_SLOT_BYTES + i * 1024), 16, 0, SC_AUX)
    SCF_ISSUE(0, 0);
    float acc = 0.0f;
    const int sw = (lane >> 1) & 7;
    for (int g = 0; g < n_slots; ++g) {
        const int slot = g & 1;
        if (g + 1 < n_slots) {
            SCF_ISSUE(g + 1, slot ^ 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 loads of the next slot may stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (SRC == SRC_F32) {
            const float* t = (const float*)(ring + slot * SCF_SLOT_BYTES + lane * 128);
            const float* qq = qs + g * 32;
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {   // the two k-steps of the pair: pieces 4 hs .. 4 hs + 3 = dims 16 hs .. 16 hs + 15
                const f32x4 c0 = *(const f32x4*)(t + (((4 * hs + 0) ^ sw) << 2));
                const f32x4 c1 = *(const f32x4*)(t + (((4 * hs + 1) ^ sw) << 2));
                const f32x4 c2 = *(const f32x4*)(t + (((4 * hs + 2) ^ sw) << 2));
                const f32x4 c3 = *(const f32x4*)(t + (((4 * hs + 3) ^ sw) << 2));
                const float* qh = qq + 16 * hs;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    acc = __builtin_fmaf(qh[x], c0[x], acc);
                    acc = __builtin_fmaf(qh[4 + x], c1[x], acc);
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    acc = __builtin_fmaf(qh[8 + x], c2[x], acc);
                    acc = __builtin_fmaf(qh[12 + x], c3[x], acc);
                }
            }
        } else {
            // 64 dims of the row (the last slot of a row whose dim is not a multiple of 64 holds 32: rows are padded to whole lines)
            const char* t = ring + slot * SCF_SLOT_BYTES + lane * 128;
            const float* qq = qs + g * 64;
            const int np = (p.E - g * 64) >= 64 ? 8 : 4;      // 16-byte pieces (8 dims each) that hold real dims: wave-uniform
            for (int pc = 0; pc < np; pc += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int piece = pc + u;
                    if (SRC == SRC_F16G) {
                        const f16x8 c = *(const f16x8*)(t + ((piece ^ sw) << 4));
#pragma unroll
                        for (int x = 0; x < 8; ++x) acc = __builtin_fmaf(qq[piece * 8 + x], (float)c[x], acc);
                    } else {
                        const bf16x8 c = *(const bf16x8*)(t + ((piece ^ sw) << 4));
#pragma unroll
                        for (int x = 0; x < 8; ++x) acc = __builtin_fmaf(qq[piece * 8 + x], bf2f((unsigned short)c[x]), acc);
                    }
                }
            }
        }
        // the LDS reads of this slot have been consumed before the slot is re-filled two iterations later
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef SCF_ISSUE
    if (idx < p.C * p.F) {
        float o = acc;
        // un-scale (power of two), add q . mean
        if (SRC == SRC_F16G && valid) o = acc * p.sfld[fld].inv_scale + p.qm[(size_t)qi * p.qm_stride + fld];
        if (known) {
            if (p.kval) p.out[(size_t)qi * p.C * p.F + idx] = p.kval[((size_t)qi * p.C + kc) * p.F + fld];      // exact launch: stage 1's bits
        } else {
            p.out[(size_t)qi * p.C * p.F + idx] = valid ? o : __builtin_nanf("");
        }
    }
}

// Gather slabs: row-major 16-bit copies of a field's rows, rows padded to whole 128-byte lines (GSLAB_ROW_BYTES).
//   fp32 index -> fp16 of (value - mean[f]) * scale[f], the screen's centring and power-of-two scale (mfar_screen.h);
//   bf16 index -> the slab's bf16 values, bit for bit.
// One thread per 16-byte output granule (8 dims).  grid = ceil(n_rows * (row_bytes / 16) / 256).
__global__ void __launch_bounds__(256) mfar_gslab_build_f16_kernel(const float* __restrict__ field, char* __restrict__ out, long long n_rows,
                                                                   int n_steps, int row_bytes, const float* __restrict__ mean,
                                                                   const ScreenField* __restrict__ sf) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = row_bytes >> 4;
    if (g >= n_rows * gpr) return;
    const long long r = g / gpr;
    const int e = (int)(g - r * gpr) << 3;
    const int E = n_steps * 16;
    const float sc = sf->scale;
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)0.0f;
    if (e < E) {
        const float* src = field + tiled_offset(n_steps, r, e);
        const f32x4 a = *(const f32x4*)src - *(const f32x4*)(mean + e);
        const f32x4 b = *(const f32x4*)(src + 4) - *(const f32x4*)(mean + e + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = (_Float16)(a[i] * sc);
            o[4 + i] = (_Float16)(b[i] * sc);
        }
    }
    *(f16x8*)(out + (size_t)r * row_bytes + (size_t)e * 2) = o;
}
__global__ void __launch_bounds__(256) mfar_gslab_build_bf16_kernel(const unsigned short* __restrict__ field, char* __restrict__ out,
                                                                    long long n_rows, int n_steps, int row_bytes) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = row_bytes >> 4;
    if (g >= n_rows * gpr) return;
    const long long r = g / gpr;
    const int e = (int)(g - r * gpr) << 3;
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0;
    if (e < n_steps * 16) o = *(const bf16x8*)(field + tiled_offset_bf16(n_steps, r, e));
    *(bf16x8*)(out + (size_t)r * row_bytes + (size_t)e * 2) = o;
}

// ---------------------------------------------------------------------------------------------------------
// M3 (multi-GPU): fetch the F-score vector of every global candidate from its owner shard's payload.
//   grid = (ceil(C/256), Q).  The owner is found from the shard row ranges, the slot by binary search in the owner's
//   sorted candidate list.
// ---------------------------------------------------------------------------------------------------------
struct LookupParams {
    const char* payloads;
    long long payload_stride, hdr_off, cand_off, ncand_off, x_off;
    const long long* cand;  // [Q, C] global union
    const int* n_cand;      // [Q]
    float* out;             // [Q, C, F]
    int S, F, C;
};
struct PayloadHeader {  // 64 bytes at the start of every payload
    int magic, Q, F, k1;
    long long row_offset, n_rows;
    int sentinel, pad[7];
};
__global__ void __launch_bounds__(256) mfar_lookup_kernel(const LookupParams p) {
    const int qi = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    float* o = p.out + ((size_t)qi * p.C + c) * p.F;
    const int nc = p.n_cand[qi];
    int owner = -1, slot = -1;
    if (c < nc) {
        const long long id = p.cand[(size_t)qi * p.C + c];
        for (int s = 0; s < p.S; ++s) {
            const PayloadHeader* h = (const PayloadHeader*)(p.payloads + (size_t)s * p.payload_stride + p.hdr_off);
            if (id >= h->row_offset && id < h->row_offset + h->n_rows) owner = s;
        }
        if (owner >= 0) {
            const char* pl = p.payloads + (size_t)owner * p.payload_stride;
            const long long* oc = (const long long*)(pl + p.cand_off) + (size_t)qi * p.C;
            int lo = 0, hi = ((const int*)(pl + p.ncand_off))[qi] - 1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const long long v = oc[mid];
                if (v == id) {
                    slot = mid;
                    break;
                }
                if (v < id) lo = mid + 1; else hi = mid - 1;
            }
        }
    }
    if (slot >= 0) {
        const float* x = (const float*)(p.payloads + (size_t)owner * p.payload_stride + p.x_off) + ((size_t)qi * p.C + slot) * p.F;
        for (int f = 0; f < p.F; ++f) o[f] = x[f];
    } else {
        for (int f = 0; f < p.F; ++f) o[f] = __builtin_nanf("");
    }
}

// ---------------------------------------------------------------------------------------------------------
// K_E: mask (contrastive.py:686) + LinearWeights.forward (weighting.py:17-29) + topk (contrastive.py:696).
//   grid = Q, block 256.  C <= 4096 candidates per query.
// ---------------------------------------------------------------------------------------------------------
struct MixParams {
    const float* x;         // [Q, C, F]
    const long long* cand;  // [Q, C]  (< 0 = empty)
    const int* n_cand;      // [Q] or nullptr
    const float* q;         // [Q, E]
    const float* W;         // [E, F] or [F]
    const float* mask;      // [F] or nullptr
    long long* ids;         // [Q, k]
    float* scores;          // [Q, k]
    int* n_valid;           // [Q] or nullptr
    int C, F, E, k, query_cond;
    const float* wgt;       // [Q, MFAR_MAX_FIELDS] or nullptr: the field weights mfar_s2_gate_kernel computed for these queries (the SAME code as
                            // mix_gate_weights below, hence the same bits) -- the mixer then stages neither q nor W
};
// Field weights of one query, shared by the mixer and the prune kernel (the SAME instruction sequence => the same weight bits):
// gate logits = natural-order fma chain per field (q and W staged in LDS by all threads first, so the F serial chains read LDS,
// unrolled by 16, instead of paying a global-memory round trip per element), softmax over <= 32 fields in fixed order.
// LDS: z | wgt | msk (3 * MFAR_MAX_FIELDS floats), qs [E], Ws [E * F].  Ends with a barrier; msk = mask or ones.
__device__ __forceinline__ void mix_gate_weights(const float* __restrict__ q_row, const float* __restrict__ W, const float* __restrict__ mask,
                                                 int query_cond, int E, int F, float* z, float* wgt, float* msk, float* qs, float* Ws) {
    if (query_cond) {
        for (int e = threadIdx.x; e < E; e += blockDim.x) qs[e] = q_row[e];
        for (int i = threadIdx.x; i < E * F; i += blockDim.x) Ws[i] = W[i];
    }
    __syncthreads();
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        float acc;
        if (query_cond) {
            acc = 0.0f;
            int e = 0;
            for (; e + 16 <= E; e += 16) {
                float qv[16], wv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    qv[i] = qs[e + i];
                    wv[i] = Ws[(e + i) * F + f];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = __builtin_fmaf(qv[i], wv[i], acc);
            }
            for (; e < E; ++e) acc = __builtin_fmaf(qs[e], Ws[e * F + f], acc);
        } else {
            acc = W[f];
        }
        z[f] = acc;
        msk[f] = mask ? mask[f] : 1.0f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // softmax over <= 32 fields, fixed order
        float m = -__builtin_inff();
        for (int f = 0; f < F; ++f) m = z[f] > m ? z[f] : m;
        float sum = 0.0f;
        for (int f = 0; f < F; ++f) {
            wgt[f] = mfar_exp(z[f] - m);
            sum = sum + wgt[f];
        }
        for (int f = 0; f < F; ++f) wgt[f] = wgt[f] / sum;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) mfar_mix_topk_kernel(const MixParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef MFAR_TRACE
    const unsigned long long tr_t0 = wall_clock64();
#endif
    const SelLds L = sel_lds(smem, p.C);
    u64* keys = L.keys;
    u64 *sel = L.sel, *sorted = L.sorted;
    int* red = L.red;
    int& n_s = L.misc[0];
    float* z = (float*)(L.misc + 4);  // z | wgt | msk : 3 * MFAR_MAX_FIELDS floats after the select scratch
    float* wgt = z + MFAR_MAX_FIELDS;
    float* msk = wgt + MFAR_MAX_FIELDS;
    const int qi = blockIdx.x;
    const int nc = min(p.n_cand ? p.n_cand[qi] : p.C, p.C);
    float* qs = msk + MFAR_MAX_FIELDS;   // [E]
    float* Ws = qs + p.E;                // [E * F]
    if (threadIdx.x == 0) n_s = 0;
    if (p.wgt) {
        if ((int)threadIdx.x < p.F) {
            wgt[threadIdx.x] = p.wgt[(size_t)qi * MFAR_MAX_FIELDS + threadIdx.x];
            msk[threadIdx.x] = p.mask ? p.mask[threadIdx.x] : 1.0f;
        }
        __syncthreads();
    } else mix_gate_weights(p.q + (size_t)qi * p.E, p.W, p.mask, p.query_cond, p.E, p.F, z, wgt, msk, qs, Ws);
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        const long long id = p.cand[(size_t)qi * p.C + c];
        if (id < 0) continue;
        const float* xr = p.x + ((size_t)qi * p.C + c) * p.F;
        float acc = 0.0f;
        for (int f = 0; f < p.F; ++f) acc = __builtin_fmaf(wgt[f], xr[f] * msk[f], acc);
        if (acc != acc) continue;  // NaN (candidate not scored) never selected
        keys[atomicAdd(&n_s, 1)] = make_key(acc, (u32)id);
    }
    __syncthreads();
    const int n = n_s;
    const int m = block_topk_sorted<16>(keys, n, p.k, sel, sorted, red);
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        p.ids[(size_t)qi * p.k + r] = r < m ? (long long)key_id(sorted[r]) : -1;
        p.scores[(size_t)qi * p.k + r] = r < m ? key_score(sorted[r]) : -__builtin_inff();
    }
    if (threadIdx.x == 0 && p.n_valid) p.n_valid[qi] = m;
#ifdef MFAR_TRACE
    if (threadIdx.x == 0) trace_put(2, (int)blockIdx.x, 0, tr_t0);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Certified two-level stage 2 of an fp32 index (no reference counterpart; same output bits as gathering every row).
//
// Stage 2 needs C * F (candidate, field) scores per query and C grows with F: F^2 * k row gathers of E * 4 bytes (22 fields:
// 148 MB per query against 34 MB scanned).  The bound the threshold algorithm would use -- an unknown s_cf is at most the field's
// k-th list score -- prunes NOTHING on independent fields (profiles/r03_stage2_bound_survival.txt), so the bytes are cut instead:
//   A. every (candidate, field) pair is scored APPROXIMATELY from the fp16 gather slab (mfar_score_rows_kernel<SRC_F16G>: half
//      the bytes of the fp32 row) with a rigorous error bound eps(q, f) (mfar_s2_prep_kernel);
//   B. mfar_s2_prune_kernel evaluates the mixer's own fma chain on the interval ends: the chain acc = fma(w_f, x_f * m_f, acc)
//      is monotone in every x_f (w_f >= 0; a negative mask entry swaps the ends), and fp32 rounding is monotone, so
//      LB_c <= mixed_c <= UB_c holds in the COMPUTED arithmetic with no further slack.  T = the k2-th largest LB; a candidate with
//      UB_c < T is strictly below k2 candidates and can neither enter nor tie into the top-k2.  NaN bounds survive;
//   C. the survivors (~1.3 k2 on the bench corpora) are gathered from the fp32 slab exactly as before and mixed: ids and score
//      bits of the top-k2 are those of the full gather.  A sweep of masks keeps the union of the masks' survivors.
//
// eps (K = dim, u16 = 2^-11, u32 = 2^-24; c = fl(d - m) the centred row, h = fp16(c * sf), approx = fl(A / sf + fl(q.m)),
// A = the fp32 fma chain of q_i * h_i in any order):
//   |approx - exact chain| <= [u16 + 1.01 (K + 2) u32] sum|q_i||c_i| + 1.01 (K + 1) u32 sum|q_i||d_i| + 1.01 K u32 sum|q_i||m_i|
//                             + 2^-24 |q|_1 / sf
//   (fp16 rounding of the row; the centring; the fp32 chain over exactly representable products; the final add; the exact chain's
//   own K u32; q.m in fp32; subnormal fp16 values), with sum|q||x| <= |q|_2 |x|_2, |c|_2 <= the field's largest centred row norm,
//   |d| <= |c| + |m|, |q|_1 <= sqrt(K) |q|_2, times SCREEN_SLACK.  Non-finite data makes eps inf / NaN: everything survives.
// ---------------------------------------------------------------------------------------------------------
#define S2_SLACK 1.25f
struct S2PrepParams {
    const float* q;          // [Q, E]
    const float* mean;       // [F, E] field means
    const ScreenField* sf;   // [F]
    float* qm;               // [Q, MFAR_MAX_FIELDS] q . mean(field), fp32, any order (budgeted in eps)
    float* eps;              // [Q, MFAR_MAX_FIELDS]
    int E, F;
    float eps_mult;          // test knob (mfar_set_screen): scales the bound; 1 = rigorous
    const float* eps_src;    // [F, eps_qw] or nullptr: the approximate level comes from the scan's SCORE DUMP (mfar_s2_lookup_kernel) -- its
    int eps_qw;              // bound is the screened pass's own eps(q, f) (mfar_screen_queries_kernel, real units, eps_mult applied there)
};
__global__ void __launch_bounds__(256) mfar_s2_prep_kernel(const S2PrepParams p) {
    __shared__ float part[MFAR_MAX_FIELDS + 1][4];
    const int qi = blockIdx.x;
    const float* qr = p.q + (size_t)qi * p.E;
    float ss = 0.0f;
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) ss = __builtin_fmaf(qr[e], qr[e], ss);
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) part[MFAR_MAX_FIELDS][threadIdx.x >> 6] = ss;
    for (int f = 0; f < p.F; ++f) {
        float pm = 0.0f;
        for (int e = threadIdx.x; e < p.E; e += blockDim.x) pm = __builtin_fmaf(qr[e], p.mean[(size_t)f * p.E + e], pm);
        for (int off = 32; off > 0; off >>= 1) pm += __shfl_xor(pm, off);
        if ((threadIdx.x & 63) == 0) part[f][threadIdx.x >> 6] = pm;
    }
    __syncthreads();
    if ((int)threadIdx.x < p.F) {
        const int f = threadIdx.x;
        const ScreenField s = p.sf[f];
        const float qn = sqrtf((part[MFAR_MAX_FIELDS][0] + part[MFAR_MAX_FIELDS][1]) + (part[MFAR_MAX_FIELDS][2] + part[MFAR_MAX_FIELDS][3])) * 1.0001f;
        const float K = (float)p.E, u16f = 4.8828125e-4f, u32f = 5.9604645e-8f;
        const float c_rel = u16f + 1.01f * (K + 2.0f) * u32f;
        float e_ = S2_SLACK * (c_rel * qn * s.dnorm_max + 1.01f * (K + 1.0f) * u32f * qn * (s.dnorm_max + s.mnorm) + 1.01f * K * u32f * qn * s.mnorm +
                               u32f * sqrtf(K) * 1.0001f * qn * s.inv_scale);
        e_ *= p.eps_mult;
        if (p.eps_src) e_ = p.eps_src[f * p.eps_qw + qi];
        p.eps[(size_t)qi * MFAR_MAX_FIELDS + f] = e_;
        p.qm[(size_t)qi * MFAR_MAX_FIELDS + f] = (part[f][0] + part[f][1]) + (part[f][2] + part[f][3]);
    }
}

// Approximate level of stage 2 from the scan's SCORE DUMP (S1Params::dump).  Stage 2 needs C * F (candidate, field) scores per query;
// when queries x candidates exceeds the rows of a field -- many fields, small corpora or shards: 129 k x 22 gathers 7.9 GB of fp16 rows
// per 128 queries, 1.9 x what the scan itself reads -- it is cheaper to let the wide screened pass WRITE every score it computes anyway
// (rows x 128 x 2 bytes per field: 16-bit codes) and to pick the pairs out of that table: one 2-byte read per pair instead of a 1.5 KB row.
//   xa[q, c, f] = dump[row u of field f][q] / (sq sf) + q . mean(f),   u = unique row of the candidate's group in field f;
// the bound on |xa - exact| is the screened pass's own eps(q, f) (the certificate's) + the code's quantisation step, so
// mfar_s2_prune_kernel runs unchanged.
// Known pairs (kmask) keep the exact score mfar_s2_known_kernel wrote.  grid = (ceil(C F / 256), Q), block 256.
struct S2LookupParams {
    const unsigned short* dump;    // [row pairs of the screen slab][128][2] signed-normalised 16-bit codes (S1Params::dump)
    const float* dump_step;        // [F, 128] B / 32767: code -> scaled units
    const float* eps_dump;         // [F, 128] the level's field-wide bound (real units)
    const float* dump_arel;        // [F, 128] its row-norm share per 1/1024 of the field's largest norm
    float* xe;                     // [Q, C, F] out: the PER-PAIR bound, from the 10-bit norm code in the pair's u_of entry
    int uof_packed;                // the entries carry norm codes (else: plain unique numbers, field-wide bound)
    const long long* dump_base;    // [F] first row of a field
    const long long* cand;         // [Q, C]
    const int* n_cand;             // [Q]
    const int* repof;              // [F][ustride] representative of a row's group (unused: uof covers every row)
    const u32* uof;                // [F][ustride] unique number + 1 of the row's group
    long long ustride, row_offset;
    const ScreenField* sf;
    const ScreenQuery* qinfo;      // [128]
    const float* qm;               // [Q, MFAR_MAX_FIELDS]
    const u32* kmask;              // [Q, C] or nullptr
    float* xa;                     // [Q, C, F]
    int n_rows, F, C;
};
__global__ void __launch_bounds__(256) mfar_s2_lookup_kernel(const S2LookupParams p) {
    const int qi = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.C * p.F) return;
    const int c = idx / p.F, f = idx - c * p.F;
    float o = __builtin_nanf("");
    if (c < min(p.n_cand[qi], p.C)) {
        if (p.kmask && ((p.kmask[(size_t)qi * p.C + c] >> f) & 1u)) return;       // exact already
        const long long id = p.cand[(size_t)qi * p.C + c] - p.row_offset;
        if (id >= 0 && id < p.n_rows) {
            const u32 ent = p.uof[(size_t)f * p.ustride + id];                            // the row's group (mfar_uof_all_kernel) + norm code
            const long long u = (long long)(p.uof_packed ? (ent & UOF_INDEX_MASK) : ent) - 1;
            const u32 code = p.uof_packed ? (ent >> UOF_NORM_SHIFT) : 1023u;
            p.xe[(size_t)qi * p.C * p.F + idx] = p.eps_dump[f * 128 + qi] - p.dump_arel[f * 128 + qi] * (float)(1023u - code);
            const size_t ru = (size_t)p.dump_base[f] + (size_t)u;
            const float a = (float)(short)p.dump[(ru >> 1) * 256 + (size_t)qi * 2 + (ru & 1)] * p.dump_step[f * 128 + qi];
            o = (a * p.qinfo[qi].inv_scale) * p.sf[f].inv_scale + p.qm[(size_t)qi * MFAR_MAX_FIELDS + f];
        }
    }
    p.xa[(size_t)qi * p.C * p.F + idx] = o;
}

// Known pairs (see ScoreParams::kmask): every REAL entry (field f, doc d, exact score s) of the query's stage-1 lists -- not the
// (0, 0.0) padding of a short zero-sentinel list, not the (-1, -inf) padding of the clean variant -- is looked up in the sorted
// candidate list; xa[q, c, f] = s, bit f of kmask[q, c] set.  Entries whose document is not a candidate of this call (a row another
// shard owns) are skipped.  grid = Q, block 256; kmask must be zero on entry.
struct KnownParams {
    const long long* fid;    // [Q, F, k]
    const float* fsc;        // [Q, F, k]
    const long long* cand;   // [Q, C] sorted unique ids, -1 padded
    const int* n_cand;       // [Q]
    float* xa;               // [Q, C, F]
    u32* kmask;              // [Q, C]
    int F, k, C, sentinel;
};
__global__ void __launch_bounds__(256) mfar_s2_known_kernel(const KnownParams p) {
    const int qi = blockIdx.x;
    const int nc = min(p.n_cand[qi], p.C);
    const long long* cq = p.cand + (size_t)qi * p.C;
    for (int i = threadIdx.x; i < p.F * p.k; i += blockDim.x) {
        const long long id = p.fid[(size_t)qi * p.F * p.k + i];
        const float sc = p.fsc[(size_t)qi * p.F * p.k + i];
        if (id < 0 || (p.sentinel && !(sc > 0.0f))) continue;      // padding
        int lo = 0, hi = nc - 1, c = -1;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const long long v = cq[mid];
            if (v == id) {
                c = mid;
                break;
            }
            if (v < id) lo = mid + 1; else hi = mid - 1;
        }
        if (c < 0) continue;
        const int f = i / p.k;
        p.xa[((size_t)qi * p.C + c) * p.F + f] = sc;
        atomicOr(&p.kmask[(size_t)qi * p.C + c], 1u << f);
    }
}

struct PruneParams {
    const float* xa;         // [Q, C, F] approximate scores (NaN = not scored)
    const long long* cand;   // [Q, C] sorted unique candidate ids (< 0 = empty)
    const int* n_cand;       // [Q]
    const float* eps;        // [Q, MFAR_MAX_FIELDS]
    const float* xe;         // [Q, C, F] or nullptr: per-PAIR bounds (the score dump's level: row norms), used instead of eps
    const float* q;          // [Q, E]
    const float* W;          // [E, F] or [F]
    const float* masks;      // [n_masks, F] or nullptr (ones)
    const u32* kmask;        // [Q, C] or nullptr: bit f = xa[q, c, f] is EXACT (eps = 0 for that pair)
    long long* cand2;        // [Q, C] the survivors in ascending id order, padded with -1
    int* src2;               // [Q, C] or nullptr: index of survivor j in the candidate list (-1 padded)
    int* n_cand2;            // [Q]
    unsigned long long* stats;   // [2] or nullptr: candidates seen / survivors kept (mfar_stage2_stats)
    int C, F, E, k, query_cond, n_masks;
};
#define PRUNE_LDS_BYTES(C, E, F) (MIX_LDS_BYTES(C, E, F) + (((size_t)(C) + 15) & ~(size_t)15))
__device__ __forceinline__ float s2_nextdown(float x) { return -s1_nextup(-x); }
__global__ void __launch_bounds__(256) mfar_s2_prune_kernel(const PruneParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, p.C);
    u64* keys = L.keys;
    int& n_s = L.misc[0];
    float* z = (float*)(L.misc + 4);
    float* wgt = z + MFAR_MAX_FIELDS;
    float* msk = wgt + MFAR_MAX_FIELDS;
    float* qs = msk + MFAR_MAX_FIELDS;
    float* Ws = qs + (p.query_cond ? p.E : 0);
    unsigned char* surv = (unsigned char*)(Ws + (p.query_cond ? (size_t)p.E * p.F : 0));
    // no static LDS: the dynamic region may be the whole 160 KB.  eps lives in z[] (the gate logits are dead once the weights
    // exist), the compaction's wave sums in the selection's reduction scratch
    float* eps_s = z;
    int* wsum = L.red;
    const int qi = blockIdx.x;
    const int nc = min(p.n_cand[qi], p.C);
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) surv[c] = 0;
    mix_gate_weights(p.q + (size_t)qi * p.E, p.W, nullptr, p.query_cond, p.E, p.F, z, wgt, msk, qs, Ws);
    if ((int)threadIdx.x < p.F) eps_s[threadIdx.x] = p.eps[(size_t)qi * MFAR_MAX_FIELDS + threadIdx.x];
    const float* xq = p.xa + (size_t)qi * p.C * p.F;
    const long long* cq = p.cand + (size_t)qi * p.C;
    for (int m = 0; m < p.n_masks; ++m) {
        if ((int)threadIdx.x < p.F) msk[threadIdx.x] = p.masks ? p.masks[(size_t)m * p.F + threadIdx.x] : 1.0f;
        if (threadIdx.x == 0) n_s = 0;
        __syncthreads();
        // lower ends: the mixer's chain on x - eps (x + eps under a negative mask entry), rounded outwards
        for (int c = threadIdx.x; c < nc; c += blockDim.x) {
            const long long id = cq[c];
            if (id < 0) continue;
            const float* xr = xq + (size_t)c * p.F;
            const float* er = p.xe ? p.xe + ((size_t)qi * p.C + c) * p.F : nullptr;
            const u32 km = p.kmask ? p.kmask[(size_t)qi * p.C + c] : 0u;
            float acc = 0.0f;
            for (int f = 0; f < p.F; ++f) {
                const float v = xr[f], e = ((km >> f) & 1u) ? 0.0f : (er ? er[f] : eps_s[f]), mf = msk[f];
                const float end = mf >= 0.0f ? s2_nextdown(v - e) : s1_nextup(v + e);
                acc = __builtin_fmaf(wgt[f], end * mf, acc);
            }
            if (acc != acc) continue;       // NaN: no lower bound (the candidate survives below)
            keys[atomicAdd(&n_s, 1)] = make_key(acc, (u32)id);
        }
        __syncthreads();
        const int n = n_s;
        const int mm = block_topk_sorted<16>(keys, n, p.k, L.sel, L.sorted, L.red);
        const float T = mm == p.k ? key_score(L.sorted[p.k - 1]) : -__builtin_inff();
        for (int c = threadIdx.x; c < nc; c += blockDim.x) {
            if (cq[c] < 0 || surv[c]) continue;
            const float* xr = xq + (size_t)c * p.F;
            const float* er = p.xe ? p.xe + ((size_t)qi * p.C + c) * p.F : nullptr;
            const u32 km = p.kmask ? p.kmask[(size_t)qi * p.C + c] : 0u;
            float acc = 0.0f;
            for (int f = 0; f < p.F; ++f) {
                const float v = xr[f], e = ((km >> f) & 1u) ? 0.0f : (er ? er[f] : eps_s[f]), mf = msk[f];
                const float end = mf >= 0.0f ? s1_nextup(v + e) : s2_nextdown(v - e);
                acc = __builtin_fmaf(wgt[f], end * mf, acc);
            }
            if (!(acc < T)) surv[c] = 1;    // NaN upper ends survive
        }
        __syncthreads();                    // T (L.sorted), keys and n_s are reused by the next mask
    }
    // compaction in candidate order (every thread owns a contiguous run)
    const int per = (p.C + 255) / 256;
    const int b = threadIdx.x * per, e_ = min(b + per, nc);
    int mine = 0;
    for (int i = b; i < e_; ++i) mine += surv[i] ? 1 : 0;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int ww = 0; ww < (int)(threadIdx.x >> 6); ++ww) wbase += wsum[ww];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int pos = wbase + incl - mine;
    long long* out = p.cand2 + (size_t)qi * p.C;
    int* osrc = p.src2 ? p.src2 + (size_t)qi * p.C : nullptr;
    for (int i = b; i < e_; ++i)
        if (surv[i]) {
            if (osrc) osrc[pos] = i;
            out[pos++] = cq[i];
        }
    for (int i = total + threadIdx.x; i < p.C; i += blockDim.x) {
        out[i] = -1;
        if (osrc) osrc[i] = -1;
    }
    if (threadIdx.x == 0) {
        p.n_cand2[qi] = total;
        if (p.stats) {
            atomicAdd(&p.stats[0], (unsigned long long)nc);
            atomicAdd(&p.stats[1], (unsigned long long)total);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: the tail of a launch as FEWER, WIDER kernels (at 129 375 x 22 the one-workgroup-per-query kernels -- union,
// prep, known pairs, prune, mixer -- held 128 of 256 CUs for ~0.5 ms of every 1.66 ms period).
//   mfar_s2_gate_kernel     per query, ONCE: the field weights (mix_gate_weights: the mixer's own code, the mixer's bits), q . mean(f) and the
//                           approximate level's eps(q, f) (what mfar_s2_prep_kernel computed).  Depends on q and W only, so the pipeline
//                           enqueues it BEFORE the tail waits for the scan: off the critical chain; prune and mixer read the weights.
//   mfar_s2_front_kernel    candidate union + known pairs in one kernel, by BITMAP when the ids span <= 2^20 (every BASELINE shape on one
//                           GPU): set bits, prefix of popcounts, emit the ids in order; a known pair's slot is its id's rank in the bitmap
//                           -- no sort, no binary search in global memory, no kmask memset.
//   mfar_s2_bounds_kernel   the interval ends of every candidate's mixed score, S2_BOUND_SPLIT workgroups per query;
//   mfar_s2_select_kernel   per query: T = k-th largest lower end, survivors, compaction (what is left of mfar_s2_prune_kernel).
// Same results as the kernels they replace (tests/test_gpu_stage2.py runs both families against each other and the oracle).
// ---------------------------------------------------------------------------------------------------------
struct GateParams {
    const float* q;          // [Q, E]
    const float* W;          // [E, F] or [F]
    float* wgt;              // [Q, MFAR_MAX_FIELDS] out
    // the approximate level of the two-level stage 2 (mean == nullptr: weights only)
    const float* mean;       // [F, E]
    const ScreenField* sf;   // [F]
    float* qm;               // [Q, MFAR_MAX_FIELDS]
    float* eps;              // [Q, MFAR_MAX_FIELDS]
    float eps_mult;
    int E, F, query_cond;
};
#define GATE_LDS_BYTES(E, F) ((size_t)3 * MFAR_MAX_FIELDS * 4 + (size_t)(E) * 4 + (size_t)(E) * (F) * 4)
// grid = Q, block 256, dynamic LDS = GATE_LDS_BYTES(query_cond ? E : 0, F) (query_cond == 0: q is still staged: E floats)
__global__ void __launch_bounds__(256) mfar_s2_gate_kernel(const GateParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float part[MFAR_MAX_FIELDS + 1][4];
    float* z = (float*)smem;
    float* wgt = z + MFAR_MAX_FIELDS;
    float* msk = wgt + MFAR_MAX_FIELDS;
    float* qs = msk + MFAR_MAX_FIELDS;
    float* Ws = qs + p.E;
    const int qi = blockIdx.x;
    const float* qr = p.q + (size_t)qi * p.E;
    mix_gate_weights(qr, p.W, nullptr, p.query_cond, p.E, p.F, z, wgt, msk, qs, Ws);
    if ((int)threadIdx.x < p.F) p.wgt[(size_t)qi * MFAR_MAX_FIELDS + threadIdx.x] = wgt[threadIdx.x];
    if (!p.mean) return;
    // (as mfar_s2_prep_kernel)
    float ss = 0.0f;
    for (int e = threadIdx.x; e < p.E; e += blockDim.x) ss = __builtin_fmaf(qr[e], qr[e], ss);
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) part[MFAR_MAX_FIELDS][threadIdx.x >> 6] = ss;
    for (int f = 0; f < p.F; ++f) {
        float pm = 0.0f;
        for (int e = threadIdx.x; e < p.E; e += blockDim.x) pm = __builtin_fmaf(qr[e], p.mean[(size_t)f * p.E + e], pm);
        for (int off = 32; off > 0; off >>= 1) pm += __shfl_xor(pm, off);
        if ((threadIdx.x & 63) == 0) part[f][threadIdx.x >> 6] = pm;
    }
    __syncthreads();
    if ((int)threadIdx.x < p.F) {
        const int f = threadIdx.x;
        const ScreenField s = p.sf[f];
        const float qn = sqrtf((part[MFAR_MAX_FIELDS][0] + part[MFAR_MAX_FIELDS][1]) + (part[MFAR_MAX_FIELDS][2] + part[MFAR_MAX_FIELDS][3])) * 1.0001f;
        const float K = (float)p.E, u16f = 4.8828125e-4f, u32f = 5.9604645e-8f;
        const float c_rel = u16f + 1.01f * (K + 2.0f) * u32f;
        float e_ = S2_SLACK * (c_rel * qn * s.dnorm_max + 1.01f * (K + 1.0f) * u32f * qn * (s.dnorm_max + s.mnorm) + 1.01f * K * u32f * qn * s.mnorm +
                               u32f * sqrtf(K) * 1.0001f * qn * s.inv_scale);
        e_ *= p.eps_mult;
        p.eps[(size_t)qi * MFAR_MAX_FIELDS + f] = e_;
        p.qm[(size_t)qi * MFAR_MAX_FIELDS + f] = (part[f][0] + part[f][1]) + (part[f][2] + part[f][3]);
    }
}

// Candidate union + known pairs by bitmap.  ids in [0, span), span <= S2_FRONT_MAX_SPAN; negative ids are padding.
//   grid = Q, block 256, dynamic LDS = S2_FRONT_LDS_BYTES(span, C)
#define S2_FRONT_MAX_SPAN (1 << 20)
#define S2_DYN_LDS_MAX (158 * 1024)     // dynamic LDS a kernel with a few hundred bytes of static LDS may ask for (160 KB per CU)
#define S2_FRONT_LDS_BYTES(span, C) ((size_t)(((span) + 511) / 512) * 64 + (size_t)(((span) + 511) / 512) * 4 + (size_t)(C) * 4 + 64)
struct FrontParams {
    const long long* fid;    // [Q, F, k] the stage-1 lists
    const float* fsc;        // [Q, F, k] their exact scores, or nullptr: no known pairs (xa / kmask untouched)
    long long* cand;         // [Q, C] out: sorted unique ids, -1 padded (C = F * k)
    int* n_cand;             // [Q]
    float* xa;               // [Q, C, F] known pairs' scores (other entries untouched)
    u32* kmask;              // [Q, C] written for every slot when fsc != nullptr
    int F, k, sentinel, span;
};
__global__ void __launch_bounds__(256) mfar_s2_front_kernel(const FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int wsum[4];
    const int nw16 = (p.span + 511) / 512;                   // groups of 16 words (512 ids)
    const int n_words = nw16 * 16;
    u32* bits = (u32*)smem;                                  // [n_words]
    int* pre = (int*)(bits + n_words);                       // [nw16] ids set before the group
    u32* km = (u32*)(pre + nw16);                            // [C]
    const int qi = blockIdx.x, C = p.F * p.k;
    const long long* fq = p.fid + (size_t)qi * C;
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) bits[i] = 0u;
    for (int i = threadIdx.x; i < C; i += blockDim.x) km[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const long long id = fq[i];
        if (id >= 0 && id < p.span) atomicOr(&bits[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    // prefix over the 16-word groups: thread t owns a contiguous run of groups
    const int per = (nw16 + 255) / 256;
    const int g0 = min((int)threadIdx.x * per, nw16), g1 = min(g0 + per, nw16);
    int mine = 0;
    for (int g = g0; g < g1; ++g) {
        int c = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) c += __popc(bits[g * 16 + w]);
        pre[g] = c;                                          // (count for now)
        mine += c;
    }
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = incl - mine;
    for (int ww = 0; ww < (int)(threadIdx.x >> 6); ++ww) base += wsum[ww];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    long long* out = p.cand + (size_t)qi * C;
    for (int g = g0; g < g1; ++g) {                           // exclusive prefix in place + the ids of the group, ascending
        const int c = pre[g];
        pre[g] = base;
        if (c) {
            int pos = base;
            for (int w = 0; w < 16; ++w) {
                u32 b = bits[g * 16 + w];
                while (b) {
                    const int t = __builtin_ctz(b);
                    b &= b - 1;
                    out[pos++] = (long long)(g * 512 + w * 32 + t);
                }
            }
        }
        base += c;
    }
    for (int i = total + threadIdx.x; i < C; i += blockDim.x) out[i] = -1;
    if (threadIdx.x == 0) p.n_cand[qi] = total;
    if (!p.fsc) return;                                      // kernel-uniform
    __syncthreads();
    // known pairs: slot of id = ids set below it
    const float* sq = p.fsc + (size_t)qi * C;
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        const long long id = fq[i];
        const float sc = sq[i];
        if (id < 0 || id >= p.span || (p.sentinel && !(sc > 0.0f))) continue;      // padding
        const int wd = (int)(id >> 5), g = wd >> 4;
        int c = pre[g];
        for (int w = g * 16; w < wd; ++w) c += __popc(bits[w]);
        c += __popc(bits[wd] & ((1u << (id & 31)) - 1u));
        const int f = i / p.k;
        p.xa[((size_t)qi * C + c) * p.F + f] = sc;
        atomicOr(&km[c], 1u << f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) p.kmask[(size_t)qi * C + i] = km[i];
}

// Interval ends of the mixed scores: PruneParams as for mfar_s2_prune_kernel, + wgt (precomputed) and lbub [n_masks, Q, C, 2].
#define S2_BOUND_SPLIT 4
struct BoundsParams {
    PruneParams pr;
    const float* wgt;        // [Q, MFAR_MAX_FIELDS]
    float* lbub;             // [n_masks, Q, C, 2]
    int Q;
};
// grid = (S2_BOUND_SPLIT, Q), block 256
__global__ void __launch_bounds__(256) mfar_s2_bounds_kernel(const BoundsParams b) {
    __shared__ float wgt[MFAR_MAX_FIELDS], eps_s[MFAR_MAX_FIELDS], msk[MFAR_MAX_FIELDS];
    const PruneParams& p = b.pr;
    const int qi = blockIdx.y;
    const int nc = min(p.n_cand[qi], p.C);
    const int chunk = (nc + S2_BOUND_SPLIT - 1) / S2_BOUND_SPLIT;
    const int c0 = blockIdx.x * chunk, c1 = min(c0 + chunk, nc);
    if (c0 >= c1) return;                                    // workgroup-uniform
    if ((int)threadIdx.x < p.F) {
        wgt[threadIdx.x] = b.wgt[(size_t)qi * MFAR_MAX_FIELDS + threadIdx.x];
        eps_s[threadIdx.x] = p.eps[(size_t)qi * MFAR_MAX_FIELDS + threadIdx.x];
    }
    const float* xq = p.xa + (size_t)qi * p.C * p.F;
    const long long* cq = p.cand + (size_t)qi * p.C;
    for (int m = 0; m < p.n_masks; ++m) {
        __syncthreads();
        if ((int)threadIdx.x < p.F) msk[threadIdx.x] = p.masks ? p.masks[(size_t)m * p.F + threadIdx.x] : 1.0f;
        __syncthreads();
        float* o = b.lbub + (((size_t)m * b.Q + qi) * p.C) * 2;
        for (int c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
            float lo = __builtin_nanf(""), hi = __builtin_nanf("");
            if (cq[c] >= 0) {
                const float* xr = xq + (size_t)c * p.F;
                const float* er = p.xe ? p.xe + ((size_t)qi * p.C + c) * p.F : nullptr;
                const u32 km = p.kmask ? p.kmask[(size_t)qi * p.C + c] : 0u;
                lo = hi = 0.0f;
                for (int f = 0; f < p.F; ++f) {              // the mixer's chain on both interval ends (see mfar_s2_prune_kernel)
                    const float v = xr[f], e = ((km >> f) & 1u) ? 0.0f : (er ? er[f] : eps_s[f]), mf = msk[f];
                    const float dn = s2_nextdown(v - e), up = s1_nextup(v + e);
                    lo = __builtin_fmaf(wgt[f], (mf >= 0.0f ? dn : up) * mf, lo);
                    hi = __builtin_fmaf(wgt[f], (mf >= 0.0f ? up : dn) * mf, hi);
                }
            }
            *(float2*)(o + (size_t)c * 2) = make_float2(lo, hi);
        }
    }
}
// The score dump's level and the interval ends in ONE kernel (one mask): a thread per candidate walks the F fields -- table entry + 16-bit
// code per pair, as mfar_s2_lookup_kernel reads them -- and feeds the mixer's chain at once; neither the approximate table xa nor the
// per-pair bounds xe are written (2 x Q C F x 4 bytes each way at 129 375 x 22: 0.2 GB per launch).  The table is the TRANSPOSED u_of
// (uof_t [n_rows][F]: the F entries of a document are 4 F contiguous bytes -- two 64-byte sectors at F = 22 where the [F][n_rows] table
// costs 22; built with the screen when the shape wants a dump).  Known pairs (kmask) read their exact score from xa, eps = 0.
struct LookupBoundsParams {
    S2LookupParams lp;       // dump, steps, eps tables, cand, kmask, xa (known pairs' scores) ...
    const u32* uof_t;        // [n_rows][F]
    const float* wgt;        // [Q, MFAR_MAX_FIELDS]
    const float* mask;       // [F] or nullptr
    float* lbub;             // [Q, C, 2]
};
// grid = (ceil(C / 256), Q), block 256
__global__ void __launch_bounds__(256) mfar_s2_lookup_bounds_kernel(const LookupBoundsParams b) {
    __shared__ float wgt[MFAR_MAX_FIELDS], msk[MFAR_MAX_FIELDS], step_s[MFAR_MAX_FIELDS], eps_s[MFAR_MAX_FIELDS], arel_s[MFAR_MAX_FIELDS], qm_s[MFAR_MAX_FIELDS],
        isf_s[MFAR_MAX_FIELDS];
    __shared__ long long base_s[MFAR_MAX_FIELDS];
    const S2LookupParams& p = b.lp;
    const int qi = blockIdx.y;
    const int nc = min(p.n_cand[qi], p.C);
    if ((int)(blockIdx.x * blockDim.x) >= nc) return;        // workgroup-uniform
    if ((int)threadIdx.x < p.F) {
        const int f = threadIdx.x;
        wgt[f] = b.wgt[(size_t)qi * MFAR_MAX_FIELDS + f];
        msk[f] = b.mask ? b.mask[f] : 1.0f;
        step_s[f] = p.dump_step[f * 128 + qi];
        eps_s[f] = p.eps_dump[f * 128 + qi];
        arel_s[f] = p.dump_arel[f * 128 + qi];
        qm_s[f] = p.qm[(size_t)qi * MFAR_MAX_FIELDS + f];
        isf_s[f] = p.sf[f].inv_scale;
        base_s[f] = p.dump_base[f];
    }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    const long long gid = p.cand[(size_t)qi * p.C + c];
    float lo = __builtin_nanf(""), hi = __builtin_nanf("");
    if (gid >= 0) {
        const long long id = gid - p.row_offset;
        const bool valid = id >= 0 && id < p.n_rows;
        const u32 km = p.kmask ? p.kmask[(size_t)qi * p.C + c] : 0u;
        const float isq = p.qinfo[qi].inv_scale;
        const float* xr = p.xa + ((size_t)qi * p.C + c) * p.F;
        const u32* ut = b.uof_t + (size_t)(valid ? id : 0) * p.F;
        lo = hi = 0.0f;
        for (int f = 0; f < p.F; ++f) {
            float v = __builtin_nanf(""), e = 0.0f;
            if ((km >> f) & 1u) v = xr[f];                               // exact already (stage 1 scored the pair)
            else if (valid) {
                const u32 ent = ut[f];
                const long long u = (long long)(p.uof_packed ? (ent & UOF_INDEX_MASK) : ent) - 1;
                const u32 code = p.uof_packed ? (ent >> UOF_NORM_SHIFT) : 1023u;
                e = eps_s[f] - arel_s[f] * (float)(1023u - code);
                const size_t ru = (size_t)base_s[f] + (size_t)u;
                const float a = (float)(short)p.dump[(ru >> 1) * 256 + (size_t)qi * 2 + (ru & 1)] * step_s[f];
                v = (a * isq) * isf_s[f] + qm_s[f];
            }
            const float mf = msk[f];
            const float dn = s2_nextdown(v - e), up = s1_nextup(v + e);
            lo = __builtin_fmaf(wgt[f], (mf >= 0.0f ? dn : up) * mf, lo);
            hi = __builtin_fmaf(wgt[f], (mf >= 0.0f ? up : dn) * mf, hi);
        }
    }
    *(float2*)(b.lbub + ((size_t)qi * p.C + c) * 2) = make_float2(lo, hi);
}
// uof_t[row][f] = uof[f][row]: grid = ceil(n_rows * F / 256), block 256
__global__ void __launch_bounds__(256) mfar_uof_transpose_kernel(const u32* __restrict__ uof, long long n_rows, int F, u32* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * F) return;
    const long long row = i / F;
    const int f = (int)(i - row * F);
    out[i] = uof[(size_t)f * n_rows + row];
}

// grid = Q, block 256, dynamic LDS = SEL_LDS_BYTES(C) + C bytes
#define S2_SELECT_LDS_BYTES(C) (SEL_LDS_BYTES(C) + (((size_t)(C) + 15) & ~(size_t)15))
__global__ void __launch_bounds__(256) mfar_s2_select_kernel(const BoundsParams b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PruneParams& p = b.pr;
    const SelLds L = sel_lds(smem, p.C);
    u64* keys = L.keys;
    int& n_s = L.misc[0];
    unsigned char* surv = (unsigned char*)(L.misc + 4);
    int* wsum = L.red;
    const int qi = blockIdx.x;
    const int nc = min(p.n_cand[qi], p.C);
    const long long* cq = p.cand + (size_t)qi * p.C;
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) surv[c] = 0;
    for (int m = 0; m < p.n_masks; ++m) {
        const float2* lu = (const float2*)(b.lbub + (((size_t)m * b.Q + qi) * p.C) * 2);
        if (threadIdx.x == 0) n_s = 0;
        __syncthreads();
        for (int c = threadIdx.x; c < nc; c += blockDim.x) {
            const long long id = cq[c];
            if (id < 0) continue;
            const float lo = lu[c].x;
            if (lo != lo) continue;                          // NaN: no lower bound (the candidate survives below)
            keys[atomicAdd(&n_s, 1)] = make_key(lo, (u32)id);
        }
        __syncthreads();
        const int n = n_s;
        const int mm = block_topk_sorted<16>(keys, n, p.k, L.sel, L.sorted, L.red);
        const float T = mm == p.k ? key_score(L.sorted[p.k - 1]) : -__builtin_inff();
        for (int c = threadIdx.x; c < nc; c += blockDim.x) {
            if (cq[c] < 0 || surv[c]) continue;
            if (!(lu[c].y < T)) surv[c] = 1;                 // NaN upper ends survive
        }
        __syncthreads();
    }
    // compaction in candidate order (every thread owns a contiguous run)
    const int per = (p.C + 255) / 256;
    const int bb = threadIdx.x * per, e_ = min(bb + per, nc);
    int mine = 0;
    for (int i = bb; i < e_; ++i) mine += surv[i] ? 1 : 0;
    int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int ww = 0; ww < (int)(threadIdx.x >> 6); ++ww) wbase += wsum[ww];
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int pos = wbase + incl - mine;
    long long* out = p.cand2 + (size_t)qi * p.C;
    int* osrc = p.src2 ? p.src2 + (size_t)qi * p.C : nullptr;
    for (int i = bb; i < e_; ++i)
        if (surv[i]) {
            if (osrc) osrc[pos] = i;
            out[pos++] = cq[i];
        }
    for (int i = total + threadIdx.x; i < p.C; i += blockDim.x) {
        out[i] = -1;
        if (osrc) osrc[i] = -1;
    }
    if (threadIdx.x == 0) {
        p.n_cand2[qi] = total;
        if (p.stats) {
            atomicAdd(&p.stats[0], (unsigned long long)nc);
            atomicAdd(&p.stats[1], (unsigned long long)total);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Lists-first multi-GPU exchange (two small collectives instead of one large one):
//   every rank all-gathers only its stage-1 LISTS; then every rank merges them, forms the same global candidate union,
//   re-scores and mixes ONLY the candidates whose rows it owns, and the ranks exchange their local top-k.
// mfar_filter_owned_kernel: the union is sorted, so the candidates of rows [row_lo, row_hi) are one contiguous run.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) mfar_filter_owned_kernel(const long long* __restrict__ cand, const int* __restrict__ n_cand,
                                                               int C, long long row_lo, long long row_hi,
                                                               long long* __restrict__ owned, int* __restrict__ n_owned) {
    const int q = blockIdx.x;
    const long long* c = cand + (size_t)q * C;
    const int n = n_cand[q];
    int lo = 0, hi = n;                      // first index with c[i] >= row_lo
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < row_lo) lo = mid + 1; else hi = mid;
    }
    const int lb = lo;
    hi = n;                                  // first index with c[i] >= row_hi
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < row_hi) lo = mid + 1; else hi = mid;
    }
    const int m = lo - lb;
    for (int i = threadIdx.x; i < C; i += blockDim.x) owned[(size_t)q * C + i] = i < m ? c[lb + i] : -1;
    if (threadIdx.x == 0) n_owned[q] = m;
}

// final merge of the per-rank local top-k lists: [S][Q][k] (ids, scores) -> top-k; n_valid = min(global candidates, k)
struct TopkMergeParams {
    const char* payloads;        // S payloads, `stride` bytes apart: ids[Q,k] int64 | scores[Q,k] f32 | n_cand_global[Q] int32
    long long stride, ids_off, scores_off, ncand_off, flag_off;
    int* any_fail;               // or nullptr: OR of the S ranks' certificate flags (every rank derives the same value)
    long long* ids;
    float* scores;
    int* n_valid;
    int S, k;
};
template <int NPT>
__global__ void __launch_bounds__(256) mfar_merge_topk_kernel(const TopkMergeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, p.S * p.k);
    const int q = blockIdx.x;
    if (threadIdx.x == 0) L.misc[0] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < p.S * p.k; i += blockDim.x) {
        const int s = i / p.k, r = i - s * p.k;
        const char* pl = p.payloads + (size_t)s * p.stride;
        const long long id = ((const long long*)(pl + p.ids_off))[(size_t)q * p.k + r];
        if (id >= 0) L.keys[lds_add_rtn(&L.misc[0], 1)] = make_key(((const float*)(pl + p.scores_off))[(size_t)q * p.k + r], (u32)id);
    }
    __syncthreads();
    const int n = L.misc[0];
    const int m = block_topk_sorted<NPT>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        p.ids[(size_t)q * p.k + r] = r < m ? (long long)key_id(L.sorted[r]) : -1;
        p.scores[(size_t)q * p.k + r] = r < m ? key_score(L.sorted[r]) : -__builtin_inff();
    }
    if (threadIdx.x == 0 && p.n_valid) {
        const int ng = ((const int*)(p.payloads + p.ncand_off))[q];      // every rank derived the same global union
        p.n_valid[q] = min(ng, p.k);
    }
    if (q == 0 && threadIdx.x == 0 && p.any_fail) {
        int any = 0;
        for (int s = 0; s < p.S; ++s) any |= *(const int*)(p.payloads + (size_t)s * p.stride + p.flag_off);
        *p.any_fail = any;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fused mode (mfar_search_fused): sum_f g_f(q) mask_f <q, d_f> = <[g_1 m_1 q; ...; g_F m_F q], [d_1; ...; d_F]>, i.e. plain
// inner-product search in F * E dims with the gate folded into the query (weighting.py:25-29).
//   mfar_concat_rows_kernel  field f of the multi-field slab -> columns [f E, (f + 1) E) of the one-field companion slab
//                            (both tiled; one thread per 16-byte granule);
//   mfar_fold_queries_kernel gate logits (natural-order fma chain), deterministic softmax (same code as the mixer),
//                            folded query qf[q, f E + e] = (g_f * mask_f) * q[e].   grid = Q, block 256.
// ---------------------------------------------------------------------------------------------------------
// companion layout: G interleaved row groups stored as G "fields" of ceil(n_rows / G) rows -- row r lives in group r % G at
// local row r / G -- so that the stage-1 grid is cut into G x (chunks per field) workgroups (one field alone is limited to
// the chunks one list merge can hold)
__global__ void mfar_concat_rows_kernel(const float* __restrict__ src_field, float* __restrict__ dst, long long dst_field_stride,
                                        long long n_rows, int E, int FE, int col0, int G) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gpr = E >> 2;
    if (gid >= n_rows * gpr) return;
    const long long r = gid / gpr;
    const int e = (int)(gid - r * gpr) << 2;
    *(f32x4*)(dst + (size_t)(r % G) * dst_field_stride + tiled_offset(FE >> 4, r / G, col0 + e)) =
        *(const f32x4*)(src_field + tiled_offset(E >> 4, r, e));
}

// final top-k of the G group lists of one query: ids arrive as row_offset + local row, the document is
// row_offset + local * G + g; the one padding row a group may hold (documents beyond n_rows) is dropped here, which is why
// the group lists are one entry deeper than k.   grid = Q, block 256, dynamic LDS = SEL_LDS_BYTES(G * kg).
struct GroupMergeParams {
    const long long* gids;   // [Q, G, kg]
    const float* gsc;        // [Q, G, kg]
    long long* ids;          // [Q, k]
    float* scores;           // [Q, k]
    long long row_offset, n_rows;
    int G, kg, k;
};
__global__ void __launch_bounds__(256) mfar_merge_groups_kernel(const GroupMergeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SelLds L = sel_lds(smem, p.G * p.kg);
    const int q = blockIdx.x;
    if (threadIdx.x == 0) L.misc[0] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < p.G * p.kg; i += blockDim.x) {
        const int g = i / p.kg;
        const long long id = p.gids[(size_t)q * p.G * p.kg + i];
        if (id < 0) continue;
        const long long doc = (id - p.row_offset) * p.G + g;
        if (doc >= p.n_rows) continue;
        L.keys[lds_add_rtn(&L.misc[0], 1)] = make_key(p.gsc[(size_t)q * p.G * p.kg + i], (u32)(p.row_offset + doc));
    }
    __syncthreads();
    const int n = L.misc[0];
    const int m = block_topk_sorted<8>(L.keys, n, p.k, L.sel, L.sorted, L.red);
    for (int r = threadIdx.x; r < p.k; r += blockDim.x) {
        p.ids[(size_t)q * p.k + r] = r < m ? (long long)key_id(L.sorted[r]) : -1;
        p.scores[(size_t)q * p.k + r] = r < m ? key_score(L.sorted[r]) : -__builtin_inff();
    }
}

__global__ void __launch_bounds__(256) mfar_fold_queries_kernel(const float* __restrict__ q, const float* __restrict__ W,
                                                                const float* __restrict__ mask, int query_cond, int F, int E,
                                                                float* __restrict__ qf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* z = (float*)smem;                 // z | g : 2 * MFAR_MAX_FIELDS floats, then q [E], W [E * F]
    float* g = z + MFAR_MAX_FIELDS;
    float* qs = g + MFAR_MAX_FIELDS;
    float* Ws = qs + E;
    const int qi = blockIdx.x;
    const float* qr = q + (size_t)qi * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) qs[e] = qr[e];
    if (query_cond)
        for (int i = threadIdx.x; i < E * F; i += blockDim.x) Ws[i] = W[i];
    __syncthreads();
    if ((int)threadIdx.x < F) {
        const int f = threadIdx.x;
        float acc;
        if (query_cond) {
            acc = 0.0f;
            for (int e = 0; e < E; ++e) acc = __builtin_fmaf(qs[e], Ws[e * F + f], acc);
        } else acc = W[f];
        z[f] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = -__builtin_inff();
        for (int f = 0; f < F; ++f) m = z[f] > m ? z[f] : m;
        float sum = 0.0f;
        for (int f = 0; f < F; ++f) {
            g[f] = mfar_exp(z[f] - m);
            sum = sum + g[f];
        }
        for (int f = 0; f < F; ++f) g[f] = (g[f] / sum) * (mask ? mask[f] : 1.0f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < F * E; i += blockDim.x) {
        const int f = i / E;
        qf[(size_t)qi * F * E + i] = g[f] * qs[i - f * E];
    }
}
#define FOLD_LDS_BYTES(E, F) ((size_t)2 * MFAR_MAX_FIELDS * 4 + (size_t)(E) * 4 + (size_t)(E) * (F) * 4)
