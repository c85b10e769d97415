// mfar_device.h -- device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
//
// Arithmetic contract (mirrored bit for bit by oracle/mfar_oracle.c, which is test infrastructure):
//   * a query.doc score is ONE fp32 fma chain over the embedding dims; inside every aligned group of 8 dims the
//     visiting order is 0,4,1,5,2,6,3,7 -- the order in which v_mfma_f32_32x32x2_f32 consumes the fragments
//     staged by the stage-1 kernel (lanes 0-31 hold dims 8g..8g+3, lanes 32-63 hold 8g+4..8g+7).
//   * ordering is (score desc, doc id asc): keys are (orderable(score) << 32) | (0xFFFFFFFF - id).
//   * mfar_exp() is a fixed fma/mul polynomial so the softmax head is reproducible on the host.
// Build with -ffp-contract=off: the only fusions are the explicit __builtin_fmaf calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mfar_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef unsigned int u32;

#define MFAR_INVALID_ID 0xFFFFFFFFu

// Per-field constants of the 16-bit copies of an fp32 index (the fp16 screen slab, mfar_screen.h; the fp16 gather slab of
// the two-level stage 2, mfar_select.h); written by mfar_screen_scale_kernel.
struct ScreenField {
    float scale;         // sf = 2^e: fp16 value = (fp32 value - mean) * sf
    float inv_scale;
    float dnorm_max;     // largest 2-norm of a centred row of the field (inf / NaN when the field holds non-finite values)
    float mnorm;         // 2-norm of the field's mean vector
    float dnorm_mean;    // mean 2-norm of the field's centred rows
    float row_mode;      // != 0: heavy-tailed row norms (dnorm_max > 1.5 dnorm_mean): the screened pass ranks rows by their UPPER bound
                         // approx + eps(row norm) instead of approx, and the certificate needs no global norm (mfar_screen.h "ROW MODE")
};
struct ScreenQuery {     // per query of the current block of 64 / 128 queries (mfar_screen_queries_kernel)
    float scale, inv_scale, norm, pad;
};

// monotone float -> uint map (-0.0 folded onto +0.0 so that float '==' and key '==' agree)
__device__ __forceinline__ u32 f2ord(float s) {
    u32 u = __float_as_uint(s);
    if (s == 0.0f) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(u32 o) {
    u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(float s, u32 id) { return ((u64)f2ord(s) << 32) | (u64)(0xFFFFFFFFu - id); }
__device__ __forceinline__ float key_score(u64 k) { return ord2f((u32)(k >> 32)); }
__device__ __forceinline__ u32 key_id(u64 k) { return 0xFFFFFFFFu - (u32)(k & 0xFFFFFFFFull); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// number of set bits of `m` below this lane
__device__ __forceinline__ int mbcnt(u64 m) {
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// LDS atomic add with return, issued and waited for inside one asm statement (hipcc otherwise
// drains in-flight LDS-DMA before an LDS atomic it can see, and expands atomics on generic pointers)
__device__ __forceinline__ int lds_add_rtn(int* lds_ptr, int v) {
    int old;
    // the low 32 bits of a generic pointer into LDS are the LDS byte offset (flat aperture base lives in the high half)
    const u32 addr = (u32)(uintptr_t)lds_ptr;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}

// Global stores the compiler does not track.  gfx9-family vmcnt counts loads AND stores, which may retire out of order
// with respect to each other, so hipcc answers any pending store with a full vmcnt(0) drain in front of the next use of a
// loaded register.  The stage-1 loops keep several k-steps of loads in flight across the selection epilogue; their few
// stores go through these helpers.  Safe: an untracked store only adds to what a counted wait has to drain (a wait for
// "at most N younger loads outstanding" still implies that every older load has landed, loads retire in order).
__device__ __forceinline__ void store_untracked_b64(void* ptr, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void store_untracked_b32(void* ptr, u32 v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(ptr), "v"(v) : "memory");
}

// Wave-aggregated slot reservation in an LDS counter: every lane of the wave must call it (convergent); lanes with
// `pred` get distinct consecutive slots, one LDS atomic per wave.
__device__ __forceinline__ int wave_reserve(int* counter, bool pred) {
    const u64 m = __ballot(pred);
    if (!m) return 0;
    const int leader = __builtin_ctzll(m);
    int base = 0;
    if (lane_id() == leader) base = lds_add_rtn(counter, __popcll(m));
    base = __shfl(base, leader);
    return base + mbcnt(m);
}

// Deterministic exp(x) for x <= 0 (same operation sequence as mfar_oracle_exp in the oracle).
__device__ __forceinline__ float mfar_exp(float x) {
    if (!(x > -80.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;
    const float t = x * 1.44269504088896341f;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(n, -0.693145751953125f, x);
    r = __builtin_fmaf(n, -1.42860682030941723e-6f, r);
    float p = 1.0f / 5040.0f;
    p = __builtin_fmaf(p, r, 1.0f / 720.0f);
    p = __builtin_fmaf(p, r, 1.0f / 120.0f);
    p = __builtin_fmaf(p, r, 1.0f / 24.0f);
    p = __builtin_fmaf(p, r, 1.0f / 6.0f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const float sc = __uint_as_float((u32)((int)n + 127) << 23);
    return p * sc;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 slab layout.  One field of the index is stored as
//     [n_blk 64-row blocks][n_steps / 2 k-step PAIRS][64 rows][32 floats]
// i.e. 8 KB tiles in which every row owns one full 128-byte line (dims 32*pair .. 32*pair + 31, natural order).
//   * the exact fp32 pass streams a block tile by tile: per k-step each lane fetches one 16-byte granule (LDS-DMA), the
//     bank swizzle of the LDS image is applied by the lane -> granule mapping, not by the memory layout; the two k-steps
//     of a pair touch the same lines, the second time in L2;
//   * the row gathers (exact re-scoring of the screened lists, stage 2) fetch whole lines: E/32 lines per (row, field),
//     no half of a fetched line belongs to another row (the former [64][16] tiles wasted every second 64 bytes fetched:
//     the gathers moved 2x the bytes they used).
// The query tile of the exact pass keeps the 4 KB LDS-image layout ([n_steps][64 queries][16 floats], granule position
// p = c ^ ((rr >> 2) & 3)): lds_image_offset().
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ size_t tiled_offset(int64_t n_steps, int64_t row, int e) {
    const int64_t blk = row >> 6;
    const int rr = (int)(row & 63);
    return (size_t)((blk * (n_steps >> 1) + (e >> 5)) * 2048 + rr * 32 + (e & 31));
}
__host__ __device__ __forceinline__ size_t lds_image_offset(int64_t n_steps, int64_t row, int e) {
    const int64_t blk = row >> 6;
    const int rr = (int)(row & 63);
    const int step = e >> 4;
    const int c = (e >> 2) & 3;
    const int p = c ^ ((rr >> 2) & 3);
    return (size_t)((blk * n_steps + step) * 1024 + rr * 16 + p * 4 + (e & 3));
}

// bf16 slab: tile = [64 rows][16 dims] bf16 (2 KB); granule (8 dims) position p = c ^ ((rr >> 3) & 1)
__host__ __device__ __forceinline__ size_t tiled_offset_bf16(int64_t n_steps, int64_t row, int e) {
    const int64_t blk = row >> 6;
    const int rr = (int)(row & 63);
    const int step = e >> 4;
    const int c = (e >> 3) & 1;
    const int p = c ^ ((rr >> 3) & 1);
    return (size_t)((blk * n_steps + step) * 1024 + rr * 16 + p * 8 + (e & 7));
}
// fp32 -> bf16, round to nearest even (same integer formula as the oracle; NaN stays NaN)
__host__ __device__ __forceinline__ unsigned short f2bf(float x) {
    union { float f; u32 u; } v;
    v.f = x;
    if ((v.u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    return (unsigned short)((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16);
}
__host__ __device__ __forceinline__ float bf2f(unsigned short b) {
    union { float f; u32 u; } v;
    v.u = (u32)b << 16;
    return v.f;
}

// ---------------------------------------------------------------------------------------------------------
// Block-level exact top-k of n unique 64-bit keys held in LDS (256 threads, n <= 256 * NPT).
// Result: sorted[0..m) descending, m = min(n, k) returned.  Every thread keeps its NPT keys in registers, so one
// radix step is NPT compares + a wave reduction + one barrier.  The descent runs on the score half first (32 steps)
// and only refines on the id half when the k-th score is tied.  LDS scratch: sel u64[MFAR_MAX_K], red int[32].
// ---------------------------------------------------------------------------------------------------------
template <int NPT>
__device__ __forceinline__ int block_sum(int c, int* red, int parity) {
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane_id() == 0) red[parity * 12 + (threadIdx.x >> 6)] = c;   // up to 12 waves per workgroup
    __syncthreads();
    int tot = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) tot += red[parity * 12 + w];
    return tot;
}

// Sum over the workgroup of a WAVE-UNIFORM count (every lane of a wave passes the same c, e.g. a sum of ballot popcounts):
// no lane reduction at all -- one LDS word per wave, one barrier.
__device__ __forceinline__ int block_sum_uniform(int c, int* red, int parity) {
    if (lane_id() == 0) red[parity * 12 + (threadIdx.x >> 6)] = c;
    __syncthreads();
    int tot = 0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) tot += red[parity * 12 + w];
    return tot;
}

// Core: every thread holds NPT keys in registers (hi = orderable score, lo = inverted id; hi == lo == 0 marks an empty slot,
// which no real key can be), n = number of non-empty slots in the workgroup.
// Every radix step counts with wave BALLOTS (one v_cmp + scalar popcount per key, the wave's count lives in an SGPR) instead of
// per-lane counters and a 6-shuffle lane reduction: the steps are latency chains -- 34 of them per selection, in every small
// kernel of a launch's tail -- and the shuffles were most of each link.
template <int NPT>
__device__ __forceinline__ int block_topk_regs(const u32 (&hi)[NPT], const u32 (&lo)[NPT], int n, int k, u64* sel, u64* sel_sorted,
                                               int* red) {
    int m = n;
    u32 T = 0, TL = 0;
    if (n > k) {
        m = k;
        int parity = 0;
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = T | (1u << bit);
            int c = 0;
#pragma unroll
            for (int i = 0; i < NPT; ++i) c += __popcll(__ballot(hi[i] >= cand));
            const int tot = block_sum_uniform(c, red, parity);
            parity ^= 1;
            if (tot >= k) T = cand;
        }
        int cg = 0, ce = 0;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            cg += __popcll(__ballot(hi[i] > T));
            ce += __popcll(__ballot(hi[i] == T && (hi[i] | lo[i]) != 0u));
        }
        const int tg = block_sum_uniform(cg, red, parity);
        parity ^= 1;
        const int te = block_sum_uniform(ce, red, parity);
        parity ^= 1;
        if (tg + te > k) {  // tie on the k-th score: keep the smallest ids (largest inverted ids)
            const int need = k - tg;
            for (int bit = 31; bit >= 0; --bit) {
                const u32 cand = TL | (1u << bit);
                int c = 0;
#pragma unroll
                for (int i = 0; i < NPT; ++i) c += __popcll(__ballot(hi[i] == T && lo[i] >= cand));
                const int tot = block_sum_uniform(c, red, parity);
                parity ^= 1;
                if (tot >= need) TL = cand;
            }
        }
    }
    if (threadIdx.x == 0) red[24] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const bool keep = (hi[i] | lo[i]) != 0u && (hi[i] > T || (hi[i] == T && lo[i] >= TL));
        const int pos = wave_reserve(&red[24], keep);
        if (keep && pos < k) sel[pos] = ((u64)hi[i] << 32) | lo[i];
    }
    __syncthreads();
    // rank by counting (keys are unique)
    if ((int)threadIdx.x < m) {
        const u64 mine = sel[threadIdx.x];
        int rank = 0;
        for (int j = 0; j < m; ++j) rank += (sel[j] > mine) ? 1 : 0;
        sel_sorted[rank] = mine;
    }
    __syncthreads();
    return m;
}

template <int NPT>
__device__ __forceinline__ int block_topk_sorted(const u64* keys, int n, int k, u64* sel, u64* sel_sorted, int* red) {
    __syncthreads();
    u32 hi[NPT], lo[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int idx = threadIdx.x + i * 256;
        const u64 key = idx < n ? keys[idx] : 0ull;
        hi[i] = (u32)(key >> 32);
        lo[i] = (u32)key;
    }
    return block_topk_regs<NPT>(hi, lo, n, k, sel, sel_sorted, red);
}
