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
// u_of entries of an fp32 index (mfar_screen.h mfar_uof_norm_code_kernel): low 22 bits = unique number + 1, top 10 bits = norm code
#define UOF_NORM_SHIFT 22
#define UOF_INDEX_MASK ((1u << UOF_NORM_SHIFT) - 1u)
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

// The same for `cnt` slots per lane (every lane calls it; cnt may be 0): a lane's slots are consecutive, one LDS atomic per wave.
__device__ __forceinline__ int wave_reserve_n(int* counter, int cnt) {
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane_id() >= off) incl += v;
    }
    int base = 0;
    if (lane_id() == 63) base = lds_add_rtn(counter, incl);
    base = __shfl(base, 63);
    return base + incl - cnt;
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
//
// The k-th largest score is found by HISTOGRAM levels instead of one counting step per bit (round 4; a workgroup-level trace of the
// list merge showed 35 us of its 60 in the 25-34 counting steps, each a ballot sweep over every key plus a barrier):
//   1. bits on which ALL keys agree need no work: scores of one list share sign, exponent and usually the first mantissa bits;
//   2. one sweep builds a 256-bin histogram (LDS atomics) of the next 8 bits below the highest differing one; wave 0 scans it
//      from the top and finds the bin that holds the k-th key and how many keys lie above it;  repeated on the next 8 bits while that
//      bin holds more than 512 keys (clustered scores, mixed signs);
//   3. the bin's keys (<= 512) are compacted to LDS and wave 0 alone finishes the descent on their remaining bits with ballots --
//      no barriers, 8 values per lane;
//   4. ties on the k-th score (rare) refine on the id half with the counting steps below.
// Four sweeps over the keys and a handful of barriers replace 34 of each; results are identical (the k-th key is unique).
template <int NPT>
__device__ __forceinline__ int block_topk_regs(const u32 (&hi)[NPT], const u32 (&lo)[NPT], int n, int k, u64* sel, u64* sel_sorted,
                                               int* red) {
    int m = n;
    u32 T = 0, TL = 0;
    if (n > k) {
        m = k;
        int parity = 0;
        int* const hist = (int*)sel_sorted;     // [256]; the output arrays are free until the selection is known
        u32* const binv = (u32*)sel;            // [512]
        u32 any1 = 0u, all1 = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const bool live = (hi[i] | lo[i]) != 0u;
            any1 |= live ? hi[i] : 0u;
            all1 &= live ? hi[i] : 0xFFFFFFFFu;
        }
        for (int off = 32; off > 0; off >>= 1) {
            any1 |= (u32)__shfl_xor((int)any1, off);
            all1 &= (u32)__shfl_xor((int)all1, off);
        }
        if (lane_id() == 0) {               // the two parity rows of the counting steps double as scratch here
            red[threadIdx.x >> 6] = (int)any1;
            red[12 + (threadIdx.x >> 6)] = (int)all1;
        }
        __syncthreads();
        {
            const int nw = (blockDim.x + 63) >> 6;
            any1 = 0u;
            all1 = 0xFFFFFFFFu;
            for (int w = 0; w < nw; ++w) {
                any1 |= (u32)red[w];
                all1 &= (u32)red[12 + w];
            }
        }
        const u32 diff = any1 ^ all1;
        int tg = 0, te = n;                 // keys above T / equal to T once T is final (diff == 0: every score is the same)
        T = all1;
        if (diff != 0u) {                   // workgroup-uniform
            int sh_top = 32 - __builtin_clz(diff);      // bits [31 .. sh_top) are common
            T = sh_top < 32 ? (all1 >> sh_top) << sh_top : 0u;
            int above = 0, nb = n, sh = sh_top;
            while (true) {
                const int W = sh_top < 8 ? sh_top : 8;
                sh = sh_top - W;
                for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
                if (threadIdx.x == 0) red[28] = 0;
                __syncthreads();            // also orders the reads of red[0 .. 24) above before the next writers
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const bool in = (hi[i] | lo[i]) != 0u && (sh_top >= 32 || ((hi[i] ^ T) >> sh_top) == 0u);
                    if (in) atomicAdd(&hist[(hi[i] >> sh) & ((1u << W) - 1u)], 1);
                }
                __syncthreads();
                if (threadIdx.x < 64) {     // wave 0: the bin of the (k - above)-th largest key among the nb keys under the prefix
                    const int need = k - above;
                    const int l = (int)threadIdx.x;
                    const int h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
                    const int mine = h0 + h1 + h2 + h3;
                    int suf = mine;         // inclusive suffix sum over lanes >= l
                    for (int off = 1; off < 64; off <<= 1) {
                        const int v = __shfl_down(suf, off);
                        if (l + off < 64) suf += v;
                    }
                    const u64 ok = __ballot(suf >= need);      // lanes 0 .. L (suf never grows with the lane); need <= nb: lane 0 is set
                    const int L = 63 - __builtin_clzll(ok);
                    if (l == L) {
                        int ab = suf - mine, B;
                        if (ab + h3 >= need) B = 3;
                        else if ((ab += h3) + h2 >= need) B = 2;
                        else if ((ab += h2) + h1 >= need) B = 1;
                        else { ab += h1; B = 0; }
                        red[25] = 4 * l + B;
                        red[26] = ab;
                        red[27] = B == 3 ? h3 : B == 2 ? h2 : B == 1 ? h1 : h0;
                    }
                }
                __syncthreads();
                T |= (u32)red[25] << sh;
                above += red[26];
                nb = red[27];
                sh_top = sh;
                if (sh == 0 || nb <= 512) break;
            }
            tg = above;
            te = nb;
            if (sh > 0) {                   // finish on the bin's keys alone
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < NPT; ++i) cnt += ((hi[i] | lo[i]) != 0u && ((hi[i] ^ T) >> sh) == 0u) ? 1 : 0;
                int pos = wave_reserve_n(&red[28], cnt);      // one LDS atomic per wave, not one per key
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const bool in = (hi[i] | lo[i]) != 0u && ((hi[i] ^ T) >> sh) == 0u;
                    if (in) binv[pos++] = hi[i];
                }
                __syncthreads();
                if (threadIdx.x < 64) {
                    u32 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (int)threadIdx.x + 64 * j < nb ? binv[threadIdx.x + 64 * j] : 0u;   // 0 < every key of the bin
                    const int need = k - above;
                    u32 t = T;
                    for (int bit = sh - 1; bit >= 0; --bit) {
                        const u32 cand = t | (1u << bit);
                        int c = 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) c += __popcll(__ballot(v[j] >= cand));
                        if (c >= need) t = cand;
                    }
                    int cg = 0, ce = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        cg += __popcll(__ballot(v[j] > t));
                        ce += __popcll(__ballot(v[j] == t));
                    }
                    if (threadIdx.x == 0) {
                        red[29] = (int)t;
                        red[30] = above + cg;
                        red[31] = ce;
                    }
                }
                __syncthreads();
                T = (u32)red[29];
                tg = red[30];
                te = red[31];
            }
        }
        __syncthreads();
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
    {
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < NPT; ++i) cnt += ((hi[i] | lo[i]) != 0u && (hi[i] > T || (hi[i] == T && lo[i] >= TL))) ? 1 : 0;
        int pos = wave_reserve_n(&red[24], cnt);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const bool keep = (hi[i] | lo[i]) != 0u && (hi[i] > T || (hi[i] == T && lo[i] >= TL));
            if (keep) {
                if (pos < k) sel[pos] = ((u64)hi[i] << 32) | lo[i];
                ++pos;
            }
        }
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
