// mfar_exact16.h -- the EXHAUSTIVE chain pass of a bf16 index: what stands behind the certified stage 1 of a bf16 slab when a certificate
// fails (and what scans a field the adaptive policy has switched off).
//
// The contract of a bf16 index is the natural-order fp32 fma chain over the stored bf16 values (oracle: c_retrieve under chain("natural");
// stage 2 and the screen's re-scoring walk it with VALU fmas).  No MFMA pass can reproduce those bits -- v_mfma_f32_32x32x16_bf16 sums its 16
// exact products in an order of its own -- so the plain three-term MFMA pass (mfar_stage1_bf16r_kernel) agrees with the chain to ~1e-4 only:
// repaired lists would carry other bits than certified ones, and a row shard where one list fails would no longer equal the unsharded run.
// This pass computes the chain itself for EVERY document of a flagged field and takes the top-k by (score desc, doc id asc):
//   mfar_chain_scan_bf16_kernel     one lane per row (coalesced 32-byte segments of the scan-ordered slab: a wave reads whole 2 KB tiles),
//                                   QT queries per LDS-resident query tile, one accumulator per (row, query): acc = fma(q_i, d_i, acc) in
//                                   dim order -> scores [QB][n_pad] (scratch);
//   mfar_chain_select_kernel        one workgroup per query: the k-th largest key by four 8-bit histogram rounds over the scores, then the
//                                   entries above it + the ties with the smallest ids, sorted -> the final list.
// VALU-bound (2 K flops per (row, query) at the fp32 FMA rate): ~30x the time of the certified scan per field -- a repair path, not a scan
// path.  Both kernels exit at once for fields whose flag is clear, so they can be enqueued without knowing the flags on the host.
#pragma once

#define CHAIN_QB 64                 // queries per scratch block
struct ChainScanParams {
    const unsigned short* slab;    // bf16 tiled slab
    long long field_stride;        // elements
    const float* q;                // [Q, E] row-major
    float* scores;                 // [CHAIN_QB][n_pad]
    const int* flags;              // [F] or nullptr (= every field)
    int field, q0, nq;             // this launch: queries q0 .. q0 + nq of field `field`
    int n_steps, E;
    long long n_blk, n_pad;        // 64-row blocks of the field; n_pad = n_blk * 64
};
// grid = ceil(n_blk / 4), block 256 (wave w <-> block 4 * blockIdx.x + w), dynamic LDS = QT * E * 4
template <int QT>
__global__ void __launch_bounds__(256) mfar_chain_scan_bf16_kernel(const ChainScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.flags && !p.flags[p.field]) return;                 // workgroup-uniform
    float* qs = (float*)smem;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long blk = (long long)blockIdx.x * 4 + w;
    const bool live = blk < p.n_blk;
    const char* row = (const char*)(p.slab + (size_t)p.field * p.field_stride) + ((size_t)(live ? blk : 0) * p.n_steps * 1024) * 2 + lane * 32;
    const int swb = (lane >> 3) & 1;
    for (int qb = 0; qb < p.nq; qb += QT) {
        const int nq = min(QT, p.nq - qb);
        __syncthreads();
        for (int i = threadIdx.x; i < QT * p.E; i += blockDim.x) {
            const int qi = i / p.E;
            qs[i] = qi < nq ? p.q[(size_t)(p.q0 + qb + qi) * p.E + (i - qi * p.E)] : 0.0f;
        }
        __syncthreads();
        float acc[QT];
#pragma unroll
        for (int i = 0; i < QT; ++i) acc[i] = 0.0f;
        if (live) {
            for (int s = 0; s < p.n_steps; ++s) {
                const char* t = row + (size_t)s * 2048;
                const bf16x8 c0 = *(const bf16x8*)(t + ((0 ^ swb) << 4));
                const bf16x8 c1 = *(const bf16x8*)(t + ((1 ^ swb) << 4));
                float d[16];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    d[x] = bf2f((unsigned short)c0[x]);
                    d[8 + x] = bf2f((unsigned short)c1[x]);
                }
#pragma unroll
                for (int i = 0; i < QT; ++i) {
                    const float* qq = qs + (size_t)i * p.E + s * 16;
#pragma unroll
                    for (int x = 0; x < 16; ++x) acc[i] = __builtin_fmaf(qq[x], d[x], acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < QT; ++i)
                if (i < nq) p.scores[(size_t)(qb + i) * p.n_pad + blk * 64 + lane] = acc[i];
        }
    }
}

struct ChainSelectParams {
    const float* scores;           // [CHAIN_QB][n_pad]
    const int* flags;              // [F] or nullptr
    long long* out_ids;            // [Q, nf, k]
    float* out_scores;
    long long n_rows, n_pad, row_offset;
    int field, fo, nf, q0, k, sentinel;
};
// grid = queries of the block, block 256
__global__ void __launch_bounds__(256) mfar_chain_select_kernel(const ChainSelectParams p) {
    __shared__ int hist[256];
    __shared__ u64 keys[256], sel[SEL_MAX_K], sorted[SEL_MAX_K];
    __shared__ int red[36];
    __shared__ u32 prefix_s;
    __shared__ int need_s, n_out_s, tie_take_s, wave_tie[4];
    if (p.flags && !p.flags[p.field]) return;                 // workgroup-uniform
    const int ql = blockIdx.x;
    const float* sc = p.scores + (size_t)ql * p.n_pad;
    const float tau0 = p.sentinel ? 0.0f : -__builtin_inff();
    // 1. the k-th largest key among the eligible scores (score > tau0; NaN never is): four rounds of 8 bits
    u32 prefix = 0;
    int need = p.k;                  // entries still wanted among the keys that match `prefix` on the bits fixed so far
    bool short_list = false;
    for (int round = 0; round < 4; ++round) {
        const int shift = 24 - 8 * round;
        hist[threadIdx.x] = 0;
        __syncthreads();
        const u32 mask_hi = round == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (long long r = threadIdx.x; r < p.n_rows; r += blockDim.x) {
            const float s = sc[r];
            if (!(s > tau0)) continue;
            const u32 o = f2ord(s);
            if ((o & mask_hi) == (prefix & mask_hi)) atomicAdd(&hist[(o >> shift) & 255u], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0, d = 255;
            for (; d >= 0; --d) {
                if (run + hist[d] >= need) break;
                run += hist[d];
            }
            if (d < 0) {             // fewer than `need` eligible entries in all: every one of them is taken
                need_s = -1;
            } else {
                prefix_s = prefix | ((u32)d << shift);
                need_s = need - run;
            }
        }
        __syncthreads();
        if (need_s < 0) {
            short_list = true;
            break;
        }
        prefix = prefix_s;
        need = need_s;
        __syncthreads();
    }
    // 2. collect: every eligible entry above the k-th key, and `need` of the entries that equal it -- those with the smallest ids
    const u32 kth = short_list ? 0u : prefix;
    if (threadIdx.x == 0) {
        n_out_s = 0;
        tie_take_s = 0;
    }
    __syncthreads();
    // (ties at the cut are rare: the ordered part below runs only in chunks that hold one)
    for (long long r0 = 0; r0 < p.n_rows; r0 += blockDim.x) {
        const long long r = r0 + threadIdx.x;
        float s = 0.0f;
        bool gt = false, eq = false;
        if (r < p.n_rows) {
            s = sc[r];
            if (s > tau0) {
                const u32 o = f2ord(s);
                gt = short_list || o > kth;
                eq = !short_list && o == kth;
            }
        }
        if (gt) sel[atomicAdd(&n_out_s, 1)] = make_key(s, (u32)(p.row_offset + r));
        // ties, in ascending id order: wave ballots + a running count (uniform branch: a chunk without ties skips the barriers)
        const u64 bal = __ballot(eq);
        if ((threadIdx.x & 63) == 0) wave_tie[threadIdx.x >> 6] = __popcll(bal);
        if (__syncthreads_or(eq ? 1 : 0)) {
            int before = tie_take_s;
            for (int ww = 0; ww < (int)(threadIdx.x >> 6); ++ww) before += wave_tie[ww];
            const int rank = before + mbcnt(bal);
            if (eq && rank < need) sel[atomicAdd(&n_out_s, 1)] = make_key(s, (u32)(p.row_offset + r));
            __syncthreads();
            if (threadIdx.x == 0) tie_take_s += wave_tie[0] + wave_tie[1] + wave_tie[2] + wave_tie[3];
            __syncthreads();
        }
    }
    __syncthreads();
    const int n = n_out_s;           // <= k by construction
    for (int i = threadIdx.x; i < 256; i += blockDim.x) keys[i] = i < n ? sel[i] : 0ull;
    __syncthreads();
    const int m = block_topk_sorted<1>(keys, n, p.k, sel, sorted, red);
    const size_t ob = ((size_t)(p.q0 + ql) * p.nf + p.fo) * p.k;
    for (int i = threadIdx.x; i < p.k; i += blockDim.x) {
        if (i < m) {
            p.out_ids[ob + i] = (long long)key_id(sorted[i]);
            p.out_scores[ob + i] = key_score(sorted[i]);
        } else {
            p.out_ids[ob + i] = p.sentinel ? 0 : -1;
            p.out_scores[ob + i] = p.sentinel ? 0.0f : -__builtin_inff();
        }
    }
}
