// mfar_stage1.h -- stage 1 of the scorer: per-(query, field) exhaustive top-k over the shard's rows.
//
// Replaces DenseFlatIndex.retrieve_batch (reference mfar/data/index.py:181-222): matmul(query, vectors^T) over
// 1 048 576-row chunks + cat + topk.  Here the [64 queries] x [rows] scores never leave registers:
//
//   * one workgroup = 4 waves = one contiguous chunk of 256-row tiles of ONE field, all 64 queries of the batch;
//     each wave owns one 64-row block per tile and streams it from the tiled slab (mfar_device.h) with linear
//     global_load_lds_dwordx4 (LDS-DMA) -- no shared operand, hence no barrier inside the K loop;
//   * v_mfma_f32_32x32x2_f32: A = doc fragment (32 rows), B = query fragment (32 queries), 2x2 blocks per wave,
//     K = 16 per step; the accumulation is one fma chain per score in the order the oracle documents;
//   * epilogue: every lane owns ONE query column, its threshold tau (current k-th best of this workgroup's chunk)
//     sits in a register; survivors (rare after warm-up) are appended to the workgroup's candidate list in HBM
//     through an LDS slot counter; a list is compacted by a wave-level radix select when it could overflow;
//   * the per-workgroup lists are merged by mfar_merge_lists_kernel (mfar_select.h).
//
// The kernel is bound by the fp32 MFMA rate (2*D*F*E flops per query, 157 TFLOP/s peak); at Q = 64 the slab read
// (D*F*E*4 bytes per batch) needs ~4.9 TB/s at full MFMA rate, so both roofs are close.
#pragma once
#include "mfar_device.h"

#define S1_THREADS 256
#define S1_TILE_ROWS 256                     // rows per workgroup tile (4 waves x 64)
#define S1_CAP 512                           // list capacity per (workgroup, query)
#define S1_TRIG (S1_CAP - S1_TILE_ROWS)      // compact when more than this many entries are held
#define S1_WAVE_LDS 16384                    // 2 buffers x (4 KB doc tile + 4 KB query tile)
#define S1_LDS_BYTES (4 * S1_WAVE_LDS + 512) // + tau[64] + cnt[64]

struct S1Params {
    const float* slab;      // tiled, [F][n_blk][n_steps][64][16]
    const float* qt;        // tiled queries [n_steps][64][16] (rows >= Q are zero)
    uint2* lists;           // [F * n_chunks * 64][S1_CAP]  (score bits, local row)
    int* list_cnt;          // [F * n_chunks * 64]
    long long field_stride; // floats between fields
    int n_rows;             // valid rows of this shard
    int n_steps;            // E / 16
    int n_tiles;            // n_blk / 4
    int n_chunks;           // workgroups per field
    int Q;                  // valid queries (<= 64)
    int k;                  // list depth (<= MFAR_MAX_K)
    float tau0;             // 0 (zero sentinel, index.py:192-193) or -inf
};

// Wave-level compaction of one list: keep the k best of n (k < n <= S1_CAP) entries, return the k-th best score.
__device__ __forceinline__ float s1_compact(uint2* list, int n, int k) {
    const int lane = lane_id();
    u32 hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = lane + 64 * j;
        uint2 e = make_uint2(0u, 0u);
        if (idx < n) e = list[idx];
        const bool ok = idx < n;
        hi[j] = ok ? f2ord(__uint_as_float(e.x)) : 0u;
        lo[j] = ok ? (0xFFFFFFFFu - e.y) : 0u;
    }
    u32 T = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const u32 cand = T | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) c += __popcll(__ballot(hi[j] >= cand));
        if (c >= k) T = cand;
    }
    int cgt = 0, ceq = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cgt += __popcll(__ballot(hi[j] > T));
        ceq += __popcll(__ballot(hi[j] == T));
    }
    u32 TL = 0;
    if (cgt + ceq > k) {  // ties on the k-th score: keep the smallest rows (largest inverted ids)
        const int need = k - cgt;
        for (int bit = 31; bit >= 0; --bit) {
            const u32 cand = TL | (1u << bit);
            int c = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) c += __popcll(__ballot(hi[j] == T && lo[j] >= cand));
            if (c >= need) TL = cand;
        }
    }
    int base = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool keep = (hi[j] > T) || (hi[j] == T && lo[j] >= TL);
        const u64 m = __ballot(keep);
        if (keep) {
            const int pos = base + mbcnt(m);
            if (pos < S1_CAP) list[pos] = make_uint2(__float_as_uint(ord2f(hi[j])), 0xFFFFFFFFu - lo[j]);
        }
        base += __popcll(m);
    }
    return ord2f(T);
}

__device__ __forceinline__ void s1_issue(const char* dsrc, const char* qsrc, char* buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + p * 1024),
                                         (__attribute__((address_space(3))) void*)(buf + p * 1024), 16, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(qsrc + p * 1024),
                                         (__attribute__((address_space(3))) void*)(buf + 4096 + p * 1024), 16, 0, 0);
}

__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_kernel(const S1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tau_s = (float*)(smem + 4 * S1_WAVE_LDS);
    int* cnt_s = (int*)(smem + 4 * S1_WAVE_LDS + 256);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int f = blockIdx.x / p.n_chunks;
    const int chunk = blockIdx.x - f * p.n_chunks;
    const int t0 = (int)(((long long)chunk * p.n_tiles) / p.n_chunks);
    const int t1 = (int)(((long long)(chunk + 1) * p.n_tiles) / p.n_chunks);
    const size_t wgq0 = (size_t)blockIdx.x * 64;

    if (tid < 64) {
        tau_s[tid] = tid < p.Q ? p.tau0 : __builtin_inff();
        cnt_s[tid] = 0;
    }
    __syncthreads();

    // fragment read offsets inside a 4 KB tile: row (32*blk + j), dims 8g + 4h .. +3  (chunk c = 2g + h)
    const int sw = (j >> 2) & 3;
    const int off_g0 = j * 64 + (((0 + h) ^ sw) << 4);
    const int off_g1 = j * 64 + (((2 + h) ^ sw) << 4);

    char* const mybuf = smem + w * S1_WAVE_LDS;
    const size_t step_bytes = 4096;
    const size_t tile_jump = (size_t)3 * p.n_steps * step_bytes;  // from the end of block b to the start of block b+4
    const char* dptr = (const char*)p.slab + ((size_t)f * (size_t)p.field_stride) * 4 +
                       ((size_t)(4 * t0 + w) * p.n_steps) * step_bytes + lane * 16;
    const char* const qbase = (const char*)p.qt + lane * 16;

    if (t0 < t1) {
        s1_issue(dptr, qbase, mybuf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int it = 0;
    for (int t = t0; t < t1; ++t) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};  // [doc block][query block]
        for (int s = 0; s < p.n_steps; ++s, ++it) {
            char* cur = mybuf + (it & 1) * 8192;
            char* nxt = mybuf + ((it + 1) & 1) * 8192;
            const f32x4 d00 = *(const f32x4*)(cur + off_g0);
            const f32x4 d01 = *(const f32x4*)(cur + off_g1);
            const f32x4 d10 = *(const f32x4*)(cur + 2048 + off_g0);
            const f32x4 d11 = *(const f32x4*)(cur + 2048 + off_g1);
            const f32x4 q00 = *(const f32x4*)(cur + 4096 + off_g0);
            const f32x4 q01 = *(const f32x4*)(cur + 4096 + off_g1);
            const f32x4 q10 = *(const f32x4*)(cur + 4096 + 2048 + off_g0);
            const f32x4 q11 = *(const f32x4*)(cur + 4096 + 2048 + off_g1);
            const bool last_step = (s == p.n_steps - 1);
            dptr += step_bytes + (last_step ? tile_jump : 0);
            if (!(last_step && t == t1 - 1)) {
                const char* qn = qbase + (last_step ? 0 : (size_t)(s + 1) * step_bytes);
                s1_issue(dptr, qn, nxt);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(d00[x], q00[x], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(d00[x], q10[x], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(d10[x], q00[x], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(d10[x], q10[x], acc11, 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(d01[x], q01[x], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(d01[x], q11[x], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(d11[x], q01[x], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(d11[x], q11[x], acc11, 0, 0, 0);
            }
            // the prefetch issued above has had this whole step of MFMAs to land: wait for it only now
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---------------- epilogue: threshold filter + append ----------------
        __syncthreads();  // compactions of the previous tile are complete: tau / cnt are stable
        const float tq0 = tau_s[j], tq1 = tau_s[32 + j];
        bool any = false;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            any = any || (acc00[r] > tq0) || (acc10[r] > tq0) || (acc01[r] > tq1) || (acc11[r] > tq1);
        if (__any(any)) {
            const int row_w = t * S1_TILE_ROWS + w * 64 + 4 * h;
#define S1_APPEND(ACC, DB, QB, TQ)                                                          \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                        \
        const float v = ACC[r];                                                             \
        if (v > TQ) {                                                                       \
            const int row = row_w + 32 * DB + (r & 3) + 8 * (r >> 2);                       \
            if (row < p.n_rows) {                                                           \
                const int qq = 32 * QB + j;                                                 \
                const int slot = atomicAdd(&cnt_s[qq], 1);                                  \
                if (slot < S1_CAP)                                                          \
                    p.lists[(wgq0 + qq) * S1_CAP + slot] = make_uint2(__float_as_uint(v), (u32)row); \
            }                                                                               \
        }                                                                                   \
    }
            S1_APPEND(acc00, 0, 0, tq0)
            S1_APPEND(acc10, 1, 0, tq0)
            S1_APPEND(acc01, 0, 1, tq1)
            S1_APPEND(acc11, 1, 1, tq1)
#undef S1_APPEND
        }
        __syncthreads();  // every append of this tile is visible
        // ---------------- compaction: wave w serves queries 16w .. 16w+15 ----------------
        for (int qq = 16 * w; qq < 16 * w + 16; ++qq) {
            const int n = __builtin_amdgcn_readfirstlane(min(cnt_s[qq], S1_CAP));
            if (n > S1_TRIG) {  // wave-uniform
                const float nt = s1_compact(p.lists + (wgq0 + qq) * S1_CAP, n, p.k);
                if (lane == 0) {
                    tau_s[qq] = nt;
                    cnt_s[qq] = p.k;
                }
            }
        }
    }
    // ---------------- flush: leave at most k entries per query, publish the counts ----------------
    __syncthreads();
    for (int qq = 16 * w; qq < 16 * w + 16; ++qq) {
        int n = __builtin_amdgcn_readfirstlane(min(cnt_s[qq], S1_CAP));
        if (n > p.k) {
            s1_compact(p.lists + (wgq0 + qq) * S1_CAP, n, p.k);
            n = p.k;
        }
        if (lane == 0) p.list_cnt[wgq0 + qq] = n;
    }
}
