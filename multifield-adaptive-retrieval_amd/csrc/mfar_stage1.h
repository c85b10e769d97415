// mfar_stage1.h -- stage 1 of the scorer: per-(query, field) exhaustive top-k over the shard's rows.
//
// Replaces DenseFlatIndex.retrieve_batch (reference mfar/data/index.py:181-222): matmul(query, vectors^T) over
// 1 048 576-row chunks + cat + topk.  Here the [64 queries] x [rows] scores never leave registers:
//
//   * one workgroup = 4 waves = one contiguous chunk of 256-row tiles of ONE field, all 64 queries of the batch;
//     each wave owns one 64-row block per tile and streams it from the tiled slab (mfar_device.h) with linear
//     global_load_lds_dwordx4 (LDS-DMA) into a private LDS ring; only the query tile of a k-step is shared;
//   * fp32 slab: v_mfma_f32_32x32x2_f32, A = doc fragment (32 rows), B = query fragment (32 queries), 2x2 blocks per
//     wave, K = 16 per step; one fma chain per score in the order the oracle documents (bit-exact parity);
//     bf16 slab: v_mfma_f32_32x32x16_bf16 against the query split EXACTLY into three bf16 terms (q = hi + mid + lo),
//     so every product is exact in fp32 and only the fp32 accumulation order differs from the oracle;
//   * epilogue: every lane owns ONE query column, its threshold tau (current k-th best of this workgroup's chunk)
//     sits in a register; survivors (rare after warm-up) are appended to the workgroup's candidate list in HBM
//     through one LDS slot reservation per lane; a list is compacted by a wave-level radix select when it could
//     overflow;
//   * the per-workgroup lists are merged by mfar_merge_lists_kernel (mfar_select.h).
//
// fp32: bound by the fp32 MFMA rate (2*D*F*E flops per query, 157 TFLOP/s peak; at Q = 64 the slab read needs
// ~4.9 TB/s at full MFMA rate, so both roofs are close).  bf16: bound by HBM (D*F*E*2 bytes per batch).
#pragma once
#include "mfar_device.h"
#include "mfar_tables.h"

#ifdef MFAR_TRACE   // experiment builds only: one record per workgroup (kind, block, xcc << 8 | cu, units, start, end in wall-clock ticks)
struct TraceRec { int kind, blk, where, units; unsigned long long t0, t1; };
__device__ TraceRec g_trace[1 << 17];
__device__ int g_trace_n;
__device__ __forceinline__ void trace_put(int kind, int blk, int units, unsigned long long t0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int i = atomicAdd(&g_trace_n, 1);
    if (i < (1 << 17)) {
        TraceRec r = {kind, blk, (int)(((xcc & 15u) << 8) | ((hw >> 8) & 0xffu)), units, t0, (unsigned long long)wall_clock64()};
        g_trace[i] = r;
    }
}
#endif
#define S1_THREADS 256
#define S1_TILE_ROWS 256                     // rows per workgroup tile (4 waves x 64)
#define S1_CAP 512                           // list capacity per (workgroup, query)
#define S1_TRIG (S1_CAP - S1_TILE_ROWS)      // compact when more than this many entries are held
// selection state: tau[64] | cnt[64] | compaction flag (16 B) | tg[64] | staged counts[64] | staging area [64][SCAP] x 8 B
#define S1_STATE_BYTES_(SCAP) (1040 + 512 * (SCAP))
#define S1_SCAP_F32 0                        // fp32 pass: survivors are stored straight to the lists (MFMA-bound, LDS is tight)
#define S1_SCAP_LDS 8                        // 16-bit LDS-ring passes: what fits beside two 74 KB rings
#ifndef S1_SCAP_REG
#define S1_SCAP_REG 64                       // 16-bit register-ring passes (measured: 16 -> 2.06, 32 -> 2.00, 64 -> 1.95 ms per pass)
#endif
#ifndef S1_DRAIN_NUM
#define S1_DRAIN_NUM 2                        // drain a query once it holds SCAP * S1_DRAIN_NUM / 4 staged entries
#endif
#define S1_STATE_BYTES S1_STATE_BYTES_(S1_SCAP_F32)
// A list is compacted once it holds more than S1_TRIG_(SCAP) entries: before a tile it then holds at most that many, fewer
// than SCAP/2 more are staged, and the tile adds at most 256 -- never more than S1_CAP.  The list depth k must not exceed it.
#define S1_TRIG_(SCAP) (S1_TRIG - (SCAP))
#define S1_MAX_DEPTH S1_TRIG_(S1_SCAP_REG)   // 192: deepest list any pass supports (the screen keeps k + 64 <= 192)
// fp32 slab: 4 KB doc tile per (wave, k-step of 16 dims), 4 KB query tile per k-step shared by the workgroup
#define S1_STAGES 3                          // LDS ring depth: loads run two k-steps ahead of the MFMAs
#define S1_D_BYTES (4 * S1_STAGES * 4096)
#define S1_Q_BYTES (S1_STAGES * 4096)
#define S1_LDS_BYTES (S1_D_BYTES + S1_Q_BYTES + S1_STATE_BYTES)
// bf16 slab: 2 KB doc tile per (wave, k-step of 16 dims); query tile = 3 exact bf16 terms x 2 KB = 6 KB in LDS
// (8 KB per k-step in memory, the 4th quarter is padding): 72 KB per workgroup -> two workgroups per CU
#define S1B_STAGES 5                         // HBM-bound: keep four k-steps of loads in flight per wave
#define S1B_LDS_BYTES (4 * S1B_STAGES * 2048 + S1B_STAGES * 6144 + S1_STATE_BYTES_(S1_SCAP_LDS))
// fp16 screen slab (mfar_screen.h): 2 KB doc tile, query tile = 2 fp16 terms x 2 KB
#ifndef S1H_STAGES
#define S1H_STAGES 6                         // 74.5 KB per workgroup, two per CU; measured 7 % faster than 5
#endif
#define S1H_LDS_BYTES (4 * S1H_STAGES * 2048 + S1H_STAGES * 4096 + S1_STATE_BYTES_(S1_SCAP_LDS))

// One workgroup of a stage-1 pass scans one CHUNK: a contiguous run of 256-row tiles of one field.  The chunk table is built
// on the host from the fields' row counts (fields differ when the scanned slab holds each field's UNIQUE rows,
// mfar_screen.h): every field gets a share of the grid proportional to its tiles, chunks of a field are consecutive.
// struct S1Chunk: mfar_tables.h (plain C++, shared with the host-side table builder and its CPU sanitizer build)

struct S1Params {
    const void* slab;       // tiled slab (fp32, bf16, or the fp16 screen slab)
    const void* qt;         // tiled queries: fp32 [n_steps][64][16]; bf16 [n_steps][4][64][16] (terms hi, mid, lo, zero pad)
    uint2* lists;           // [n_chunks * 64][S1_CAP]  (score bits, local row)
    int* list_cnt;          // [n_chunks * 64]
    const S1Chunk* chunks;  // [n_chunks] chunk table of the scanned slab
    int chunk0;             // first chunk of this launch (grid = a contiguous chunk range: all fields, or one field)
    int n_launch;           // chunks of this launch: workgroup b scans chunks chunk0 + b, chunk0 + b + gridDim.x, ... (grid == n_launch
                            // except for a repair pass, whose finely cut table is walked by one wave of workgroups)
    int n_steps;            // E / 16
    int Q;                  // valid queries (<= qw)
    int qw;                 // query columns of the pass: 64, or 128 for the wide fp16 screen pass (mfar_stage1_f16w_kernel);
                            // sizes every per-query table: lists [n_chunks * qw][S1_CAP], gtau [F, qw], samp_out [..][qw][2]
    int k;                  // list depth (<= S1_MAX_DEPTH)
    float tau0;             // 0 (zero sentinel, index.py:192-193) or -inf
    const float* gtau;      // [F, qw] non-strict lower bounds from the sample pass, or nullptr
    int sample;             // 1/2: threshold-estimation pass, every workgroup scans only the first tile of its chunk;
                            //      2 = light form: no lists, every wave publishes the 2 best scores per query of its 64 rows
    int sample_tiles;       // tiles per workgroup scanned by the list-building sample pass (sample == 1); the light pass uses S1Chunk::ns
    float* samp_out;        // [F][samp_stride wave blocks][qw][2] (sample == 2): wave block = (sampled tile of the field, wave)
    int samp_stride;        // wave blocks reserved per field
    const int* only_failed; // [F] or nullptr: workgroups of fields whose flag is 0 exit at once (screen fall-back pass)
    int dbg;                // profiling only (MFAR_S1_DEBUG): 1 = skip the selection epilogue (results invalid; note that
                            // the downstream kernels then have no candidates, so they no longer compete with stage 1)
    const u64* rep_bits;    // [F][rep_stride] or nullptr: bit r of word b = row 64 b + r of the field is a real row AND the
                            // representative (lowest member) of its group of bit-identical rows (mfar_screen.h).  The certified
                            // passes over a bf16 slab scan every document but rank UNIQUE rows: the accumulators of all other
                            // rows start at -inf (s1_acc_init), so they reach neither the sample nor the lists.
    long long rep_stride;   // words per field
    const float* arow;      // [F, qw] or nullptr.  ROW MODE of the certified screen (mfar_screen.h): fields whose bit is set in row_mask rank rows
    const float* rnorm;     // by approx + arow[f, query] * rnorm[row] -- an upper bound of the exact score up to a row-independent rest -- where
    u32 row_mask;           // rnorm [rows of the scanned slab] is the row's centred 2-norm (field f's rows start at dump_base[f])
    void* dump;             // or nullptr.  SCORE DUMP of the wide fp16 screen pass (mfar_stage1_f16w_kernel): every approximate score of the
                            // launch as a 16-bit signed-normalised code, a = code / 32767 * B(query, field) with B >= |a| (dump_inv = 1 / B:
                            // mfar_screen_queries_kernel), laid out [row pair][128 query columns][2 rows] -- 256 bytes per scanned row; field
                            // f's rows start at dump_base[f] (in rows, even).  Stage 2 of a many-field / small-corpus index reads its
                            // approximate level from here instead of gathering 16-bit rows (mfar_select.h mfar_s2_lookup_kernel); the
                            // quantisation (<= B / 65534) is part of that level's error bound.
    const float* dump_inv;  // [F, 128] 1 / B
    const long long* dump_base;   // [F]
    const uint2* cvt;       // [F] (CV passes over a bf16 slab) per field: x = smallest bf16 magnitude that is a NORMAL fp16 number after the
                            // field's power-of-two scale, y = the exponent rebias, both replicated in the two halves of a dword
    int* unit_ctr;          // [F] zeroed before the launch, or nullptr.  DYNAMIC WORK DISTRIBUTION of a full pass (wide kernels): a
                            // workgroup is still bound to one field and one list per query (chunk_id), but instead of the fixed
                            // tile range of its chunk it claims UNITS of unit_tiles consecutive tiles from the field's counter
                            // until the field is exhausted (s1_unit_* below).  nullptr: the chunk's own range [t0, t1).
    int unit_tiles;         // >= 2
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
            if (pos < S1_CAP) store_untracked_b64(&list[pos], ((u64)(0xFFFFFFFFu - lo[j]) << 32) | __float_as_uint(ord2f(hi[j])));
        }
        base += __popcll(m);
    }
    return ord2f(T);
}

// The chunk descriptor of a workgroup, in scalar registers (inside the chunk loop of a repair pass hipcc fetches it with vector
// loads; the tile loop bounds and the slab base must not end up in VGPRs)
__device__ __forceinline__ S1Chunk s1_load_chunk(const S1Params& p, int chunk_id) {
    S1Chunk c = p.chunks[chunk_id];
    c.f = __builtin_amdgcn_readfirstlane(c.f);
    c.t0 = __builtin_amdgcn_readfirstlane(c.t0);
    c.t1 = __builtin_amdgcn_readfirstlane(c.t1);
    c.n_rows = __builtin_amdgcn_readfirstlane(c.n_rows);
    c.tl0 = __builtin_amdgcn_readfirstlane(c.tl0);
    c.ns = __builtin_amdgcn_readfirstlane(c.ns);
    const unsigned long long b = (unsigned long long)c.base;
    c.base = (long long)(((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(b >> 32)) << 32) |
                         (u32)__builtin_amdgcn_readfirstlane((int)(u32)b));
    return c;
}

// A score survives when it beats the workgroup's own running k-th best (strict: later rows lose ties to earlier
// ones) AND is not below the global lower bound from the sample pass (non-strict: ties with other chunks are
// decided by the merge).
// the same test as ONE compare: v > tq  <=>  v >= nextup(tq) for non-NaN v, so thr = max(nextup(tq), tg)
__device__ __forceinline__ float s1_nextup(float x) {
    if (!(x < __builtin_inff())) return x;                    // +inf (and NaN) stay
    if (x == 0.0f) return __uint_as_float(1u);                // +-0 -> smallest positive subnormal
    const u32 b = __float_as_uint(x);
    return __uint_as_float((b & 0x80000000u) ? b - 1u : b + 1u);
}
#define S1_PASS1(v, thr) ((v) >= (thr))

struct S1State {       // LDS-resident selection state of one workgroup
    float* tau;        // [64] strict local thresholds
    int* cnt;          // [64] list fill counts
    int* flag;         // some list needs compaction this tile
    float* tg;         // [64] non-strict global lower bounds
    int* scnt;         // [64] entries reserved in the staging area since the last drain (may exceed SCAP: the excess went direct)
    u32 stage;         // LDS byte address of the staging area [64][SCAP] x (score bits, row)
};
__device__ __forceinline__ S1State s1_state(char* base) {
    S1State s;
    s.tau = (float*)base;
    s.cnt = (int*)(base + 256);
    s.flag = (int*)(base + 512);
    s.tg = (float*)(base + 528);
    s.scnt = (int*)(base + 784);
    s.stage = (u32)(uintptr_t)(base + 1040);
    return s;
}
// qoff: first query column this state serves (0; the wide pass keeps a second state for columns 64 .. 127)
__device__ __forceinline__ void s1_state_init(const S1State& st, const S1Params& p, int f, int qoff = 0) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        st.tau[tid] = qoff + tid < p.Q ? p.tau0 : __builtin_inff();
        st.cnt[tid] = 0;
        st.tg[tid] = p.gtau ? p.gtau[f * p.qw + qoff + tid] : -__builtin_inff();
        st.scnt[tid] = 0;
    }
    if (tid == 0) *st.flag = 0;
    __syncthreads();
}

// Rows that must never qualify, for both epilogues: the padding rows behind the field's last row.  acc layout: see s1_epilogue.
__device__ __forceinline__ void s1_mask_rows(int n_rows, int t, int w, f32x16& acc00, f32x16& acc01, f32x16& acc10, f32x16& acc11) {
    const int h = (threadIdx.x & 63) >> 5;
    if (t * S1_TILE_ROWS + S1_TILE_ROWS > n_rows) {  // last tile of the field
        const int row_w = t * S1_TILE_ROWS + w * 64 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_w + (r & 3) + 8 * (r >> 2);
            if (row >= n_rows) acc00[r] = acc01[r] = -__builtin_inff();
            if (row + 32 >= n_rows) acc10[r] = acc11[r] = -__builtin_inff();
        }
    }
}

// Starting values of the accumulators of one tile: 0, or -inf for the rows the pass must not rank (S1Params::rep_bits).  The
// 64 flags of the wave's row block arrive by a SCALAR load (its own counter: a vector load here would make hipcc drain the doc
// ring, which it does not know about, with vmcnt(0)); lane (j, h) holds rows 32 db + (r & 3) + 8 (r >> 2) + 4 h of doc block db.
__device__ __forceinline__ u64 s1_sload_u64(const u64* ptr) {
    u64 v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void s1_acc_init(const S1Params& p, const S1Chunk& ck, int t, int w, f32x16& i0, f32x16& i1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) i0[r] = i1[r] = 0.0f;
    if (p.rep_bits) {   // kernel-uniform
        const u64 bm = s1_sload_u64(p.rep_bits + (size_t)ck.f * p.rep_stride + (size_t)(4 * t + w));
        const int sh = 4 * (int)((threadIdx.x & 63) >> 5);
        const u32 lo = (u32)bm >> sh, hi = (u32)(bm >> 32) >> sh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = (r & 3) + 8 * (r >> 2);
            i0[r] = ((lo >> b) & 1u) ? 0.0f : -__builtin_inff();
            i1[r] = ((hi >> b) & 1u) ? 0.0f : -__builtin_inff();
        }
    }
}

// LDS accesses of the staging area through inline asm (hipcc would drain the in-flight LDS-DMA in front of LDS accesses it
// can see next to it)
__device__ __forceinline__ void lds_write_b64(u32 addr, u64 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ u64 lds_read_b64(u32 addr) {
    u64 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

// Move the staged survivors of this wave's 16 queries (16w .. 16w+15) to their lists in HBM: one coalesced store per
// drained query.  Queries with fewer than `min_staged` staged entries keep them.  Called by all lanes of every wave.
template <int SCAP>
__device__ __forceinline__ void s1_drain(const S1Params& p, const S1State& st, int w, size_t wgq0, int min_staged) {
    const int lane = threadIdx.x & 63;
    const int sc = lane < 16 ? min(st.scnt[16 * w + lane], SCAP) : 0;
    u64 todo = __ballot(sc >= min_staged && sc > 0);
    while (todo) {
        const int b = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int qq = 16 * w + b;
        const int n = __builtin_amdgcn_readlane(sc, b);
        const int g = __builtin_amdgcn_readfirstlane(st.cnt[qq]);
        if (lane < n && g + lane < S1_CAP)
            store_untracked_b64(p.lists + (wgq0 + qq) * S1_CAP + g + lane, lds_read_b64(st.stage + (u32)(qq * SCAP + lane) * 8u));
        if (lane == 0) {
            st.cnt[qq] = g + n;
            st.scnt[qq] = 0;
            if (g + n > S1_TRIG_(SCAP)) *st.flag = 1;
        }
    }
}

// Selection epilogue of one 256-row tile.  acc[doc block][query block]: lane (j = lane & 31, h = lane >> 5) holds, for
// query 32*qb + j, the scores of doc rows 32*db + (r & 3) + 8*(r >> 2) + 4*h (the MFMA 32x32 accumulator layout).
//
// SCAP > 0: survivors are STAGED in LDS ([64 queries][SCAP] entries) and moved to the lists in HBM by s1_drain once a
// query holds SCAP/2 of them.  Every vector store issued here sits in the same vmcnt queue as the k-loop's prefetch, and a
// counted wait behind a pending store has to drain deeper than it needs (measured: the ~20 scattered 8-byte stores per
// wave and tile of the direct path cost 14 % of the 16-bit pass); staging turns them into about one coalesced store per
// wave and tile.  Entries that do not fit the staging area (a lane with many survivors in one tile) go straight to the list.
// The epilogue has two halves, so that the wide pass can run the first half for BOTH of its query blocks before anything
// heavy happens: s1_epilogue_append consumes the accumulators (compare, reserve, stage / store the survivors);
// s1_epilogue_finish (behind barrier B) drains the staging area and compacts lists -- register-hungry radix selects that
// must not run while another block's 64 accumulators are still live.
template <int SCAP>
__device__ __forceinline__ void s1_epilogue_append(const S1Params& p, const S1State& st, int n_rows, int t, int w, size_t wgq0, f32x16& acc00,
                                                   f32x16& acc01, f32x16& acc10, f32x16& acc11) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
    s1_mask_rows(n_rows, t, w, acc00, acc01, acc10, acc11);
    // (barrier A -- the compactions of the previous tile are complete, every wave has finished the tile's last k-step --
    //  is executed by the caller, which uses it to refill the ring slot that just became free before this epilogue runs)
    const float th0 = fmaxf(s1_nextup(st.tau[j]), st.tg[j]);
    const float th1 = fmaxf(s1_nextup(st.tau[32 + j]), st.tg[32 + j]);
    int n0 = 0, n1 = 0;  // survivors of this lane for query j / 32 + j
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        n0 += (S1_PASS1(acc00[r], th0) ? 1 : 0) + (S1_PASS1(acc10[r], th0) ? 1 : 0);
        n1 += (S1_PASS1(acc01[r], th1) ? 1 : 0) + (S1_PASS1(acc11[r], th1) ? 1 : 0);
    }
    if (__any((n0 | n1) != 0)) {
        // one LDS slot reservation per (lane, query block); inline asm keeps hipcc from draining the
        // LDS-DMA prefetch (it would wait vmcnt(0) before an LDS atomic it can see)
        int b0 = 0, b1 = 0;
        if (n0) b0 = lds_add_rtn(SCAP ? &st.scnt[j] : &st.cnt[j], n0);
        if (n1) b1 = lds_add_rtn(SCAP ? &st.scnt[32 + j] : &st.cnt[32 + j], n1);
        if (!SCAP && ((n0 && b0 + n0 > S1_TRIG) || (n1 && b1 + n1 > S1_TRIG))) *st.flag = 1;
        const int row_w = t * S1_TILE_ROWS + w * 64 + 4 * h;
        uint2* l0 = p.lists + (wgq0 + j) * S1_CAP;
        uint2* l1 = p.lists + (wgq0 + 32 + j) * S1_CAP;
        const u32 s0 = st.stage + (u32)(j * SCAP) * 8u, s1 = st.stage + (u32)((32 + j) * SCAP) * 8u;
#define S1_APPEND(ACC, DB, TH, L, B, SB, QI)                                                                 \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                         \
        const float v = ACC[r];                                                                              \
        if (S1_PASS1(v, TH)) {                                                                               \
            const u64 e_ = ((u64)(u32)(row_w + 32 * DB + (r & 3) + 8 * (r >> 2)) << 32) | __float_as_uint(v); \
            if (SCAP) {                                                                                      \
                if (B < SCAP) lds_write_b64(SB + (u32)B * 8u, e_);                                           \
                else {                                                                                       \
                    const int g_ = lds_add_rtn(&st.cnt[QI], 1);                                              \
                    if (g_ + 1 > S1_TRIG_(SCAP)) *st.flag = 1;                                                      \
                    if (g_ < S1_CAP) store_untracked_b64(&L[g_], e_);                                        \
                }                                                                                            \
            } else if (B < S1_CAP) store_untracked_b64(&L[B], e_);                                           \
            ++B;                                                                                             \
        }                                                                                                    \
    }
        S1_APPEND(acc00, 0, th0, l0, b0, s0, j)
        S1_APPEND(acc10, 1, th0, l0, b0, s0, j)
        S1_APPEND(acc01, 0, th1, l1, b1, s1, 32 + j)
        S1_APPEND(acc11, 1, th1, l1, b1, s1, 32 + j)
#undef S1_APPEND
    }
}
// second half; the caller has executed barrier B (slot counters and, on the direct path, the compaction flag of this tile
// are final) after the appends
template <int SCAP>
__device__ __forceinline__ void s1_epilogue_finish(const S1Params& p, const S1State& st, int w, size_t wgq0) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (SCAP) {
        s1_drain<SCAP>(p, st, w, wgq0, SCAP * S1_DRAIN_NUM / 4);
        // barrier C: list counts and the compaction flag are final
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (__builtin_amdgcn_readfirstlane(*st.flag)) {  // workgroup-uniform, rare after warm-up
        // every wave's appended entries must be in memory before another wave compacts a list
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0) *st.flag = 0;
        // wave w serves queries 16w .. 16w+15
        const int nv = lane < 16 ? min(st.cnt[16 * w + lane], S1_CAP) : 0;
        u64 todo = __ballot(nv > S1_TRIG_(SCAP));
        while (todo) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int qq = 16 * w + b;
            const int n = __builtin_amdgcn_readlane(nv, b);
            const float nt = s1_compact(p.lists + (wgq0 + qq) * S1_CAP, n, p.k);
            if (lane == 0) {
                st.tau[qq] = nt;
                st.cnt[qq] = p.k;
            }
        }
    }
}
template <int SCAP>
__device__ __forceinline__ void s1_epilogue(const S1Params& p, const S1State& st, int n_rows, int t, int w, size_t wgq0, f32x16& acc00,
                                            f32x16& acc01, f32x16& acc10, f32x16& acc11) {
    s1_epilogue_append<SCAP>(p, st, n_rows, t, w, wgq0, acc00, acc01, acc10, acc11);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier B
    s1_epilogue_finish<SCAP>(p, st, w, wgq0);
}

// Light threshold-estimation epilogue (sample == 2): per (wave, query) the two best scores among the wave's 64 rows.
// They are scores of distinct real rows, so the k-th largest of all published values is a valid (non-strict) lower
// bound of the final k-th best (mfar_sample_tau_kernel).
__device__ __forceinline__ void s1_sample_top2(const S1Params& p, const S1Chunk& c, int tl, int t, int w, f32x16& acc00, f32x16& acc01,
                                               f32x16& acc10, f32x16& acc11, int qoff = 0) {
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    s1_mask_rows(c.n_rows, t, w, acc00, acc01, acc10, acc11);
    float a1 = -__builtin_inff(), a2 = a1, b1 = a1, b2 = a1;   // query j: (a1 >= a2), query 32 + j: (b1 >= b2)
#define S1_TOP2(V, M1, M2)                    \
    {                                         \
        const float v_ = (V);                 \
        const float lo_ = fminf(v_, M1);      \
        M1 = fmaxf(v_, M1);                   \
        M2 = fmaxf(M2, lo_);                  \
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        S1_TOP2(acc00[r], a1, a2)
        S1_TOP2(acc10[r], a1, a2)
        S1_TOP2(acc01[r], b1, b2)
        S1_TOP2(acc11[r], b1, b2)
    }
#undef S1_TOP2
    // the other half-wave holds the other 32 rows of the same queries
    const float oa1 = __shfl_xor(a1, 32), oa2 = __shfl_xor(a2, 32), ob1 = __shfl_xor(b1, 32), ob2 = __shfl_xor(b2, 32);
    const float ra1 = fmaxf(a1, oa1), ra2 = fmaxf(fminf(a1, oa1), fmaxf(a2, oa2));
    const float rb1 = fmaxf(b1, ob1), rb2 = fmaxf(fminf(b1, ob1), fmaxf(b2, ob2));
    if (h == 0) {
        float* o = p.samp_out + (((size_t)c.f * p.samp_stride) + (size_t)(c.tl0 + tl) * 4 + w) * (size_t)(2 * p.qw) + 2 * qoff;
        store_untracked_b64(&o[j * 2], ((u64)__float_as_uint(ra2) << 32) | __float_as_uint(ra1));
        store_untracked_b64(&o[(32 + j) * 2], ((u64)__float_as_uint(rb2) << 32) | __float_as_uint(rb1));
    }
}

// leave at most k entries per query and publish the counts
template <int SCAP>
__device__ __forceinline__ void s1_flush(const S1Params& p, const S1State& st, int w, size_t wgq0, int qoff = 0) {
    const int lane = threadIdx.x & 63;
    __syncthreads();
    if (SCAP) {   // everything still staged goes to the lists (this wave drains exactly the queries it flushes below)
        s1_drain<SCAP>(p, st, w, wgq0, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    for (int qq = 16 * w; qq < 16 * w + 16; ++qq) {
        int n = __builtin_amdgcn_readfirstlane(min(st.cnt[qq], S1_CAP));
        if (n > p.k) {
            s1_compact(p.lists + (wgq0 + qq) * S1_CAP, n, p.k);
            n = p.k;
        }
        if (lane == 0) p.list_cnt[wgq0 + qq] = n;
    }
}

#define S1_GLDS(SRC, DST, AUX)                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),        \
                                     (__attribute__((address_space(3))) void*)(DST), 16, 0, AUX)

// ---------------------------------------------------------------------------------------------------------------------
// fp32 slab
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s1_body_f32(const S1Params& p, const int chunk_id) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem + S1_D_BYTES;
    const S1State st = s1_state(smem + S1_D_BYTES + S1_Q_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform
    const int f = ck.f;
    if (p.only_failed && !p.only_failed[f]) return;   // workgroup-uniform
    const int t0 = ck.t0;
    int t1 = ck.t1;
    if (p.sample) t1 = min(t1, t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    const size_t wgq0 = (size_t)chunk_id * p.qw;
    s1_state_init(st, p, f);

    // fragment read offsets inside a 4 KB tile: row (32*blk + j), dims 8g + 4h .. +3  (chunk c = 2g + h)
    const int sw = (j >> 2) & 3;
    const int off_g0 = j * 64 + (((0 + h) ^ sw) << 4);
    const int off_g1 = j * 64 + (((2 + h) ^ sw) << 4);

    char* const dring = smem + w * (S1_STAGES * 4096);
    const size_t step_bytes = 4096;
    // The slab stores a block as k-step PAIRS of [64 rows][32 floats] (mfar_device.h).  This lane's granule of the four 1 KB
    // DMA pieces of a k-step: piece pc covers rows 16 pc .. 16 pc + 15 of the wave's block, and LDS slot (row, position p)
    // must receive dims chunk c = p ^ ((row >> 2) & 3) -- the bank swizzle lives in this mapping, not in memory.
    int loff[4];
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
        const int rr = 16 * pc + (lane >> 2), c = (lane & 3) ^ ((rr >> 2) & 3);
        loff[pc] = rr * 128 + c * 16;
    }
    // load cursor (runs two k-steps ahead of the compute cursor): base of the wave's block of the current tile
    const char* dblk = (const char*)p.slab + (size_t)ck.base * 4 + ((size_t)(4 * t0 + w) * p.n_steps) * step_bytes;
    const size_t tile_jump = (size_t)4 * p.n_steps * step_bytes;  // to the wave's block of the next tile
    const char* const qbase = (const char*)p.qt + w * 1024 + lane * 16;
    int s_next = 0, st_next = 0;
    const int total = (t1 - t0) * p.n_steps;
    int issued = 0;
    // one k-step of loads: the wave's own 4 KB of docs (cached: the other k-step of the pair reads the same lines) + its
    // quarter of the shared query tile
#define S1_ISSUE_NEXT()                                                                                   \
    do {                                                                                                  \
        char* db_ = dring + st_next * 4096;                                                               \
        const char* src_ = dblk + (size_t)(s_next >> 1) * 8192 + (s_next & 1) * 64;                       \
        _Pragma("unroll") for (int pc = 0; pc < 4; ++pc) S1_GLDS(src_ + loff[pc], db_ + pc * 1024, 0);     \
        S1_GLDS(qbase + (size_t)s_next * step_bytes, qring + st_next * 4096 + w * 1024, 0);               \
        if (++s_next == p.n_steps) {                                                                      \
            s_next = 0;                                                                                   \
            dblk += tile_jump;                                                                            \
        }                                                                                                 \
        st_next = (st_next == S1_STAGES - 1) ? 0 : st_next + 1;                                           \
        ++issued;                                                                                         \
    } while (0)
    if (total > 0) S1_ISSUE_NEXT();
    if (total > 1) S1_ISSUE_NEXT();

    int it = 0, st_cur = 0;
    for (int t = t0; t < t1; ++t) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};  // [doc block][query block]
        for (int s = 0; s < p.n_steps; ++s, ++it) {
            // stage `it` must have landed (own doc tile: counted vmcnt; other waves' query quarters: the barrier).
            // The 5 newest loads (stage it+1) may stay in flight.
            const int ahead = issued - it - 1;   // stages issued after stage `it` (2 right after a tile boundary, see below)
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const char* cur = dring + st_cur * 4096;
            const char* curq = qring + st_cur * 4096;
            st_cur = (st_cur == S1_STAGES - 1) ? 0 : st_cur + 1;
            const f32x4 d00 = *(const f32x4*)(cur + off_g0);
            const f32x4 d01 = *(const f32x4*)(cur + off_g1);
            const f32x4 d10 = *(const f32x4*)(cur + 2048 + off_g0);
            const f32x4 d11 = *(const f32x4*)(cur + 2048 + off_g1);
            const f32x4 q00 = *(const f32x4*)(curq + off_g0);
            const f32x4 q01 = *(const f32x4*)(curq + off_g1);
            const f32x4 q10 = *(const f32x4*)(curq + 2048 + off_g0);
            const f32x4 q11 = *(const f32x4*)(curq + 2048 + off_g1);
            // every wave has passed the barrier, so stage it-1 (== ring slot of it+2) is no longer being read
            if (issued < total && issued < it + S1_STAGES) S1_ISSUE_NEXT();
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
        }
        if (p.dbg & 1) {
            asm volatile("" ::"v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11));
            continue;
        }
        if (p.sample == 2) {
            s1_sample_top2(p, ck, t - t0, t, w, acc00, acc01, acc10, acc11);
            continue;
        }
        // barrier A.  Every wave is past the last k-step of the tile, so that step's ring slot is free: refill it NOW,
        // the HBM stream then keeps its depth through the epilogue instead of draining (the next step skips its issue).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (issued < total) S1_ISSUE_NEXT();
        s1_epilogue<S1_SCAP_F32>(p, st, ck.n_rows, t, w, wgq0, acc00, acc01, acc10, acc11);
    }
#undef S1_ISSUE_NEXT
    if (p.sample != 2) s1_flush<S1_SCAP_F32>(p, st, w, wgq0);
}

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit slabs.  Tile = [64 rows][16 dims] x 2 bytes = 2 KB; the 16-byte granule (8 dims) at (row rr, position p) holds dims
// 16*step + 8c .. +7 with c = p ^ ((rr >> 3) & 1).  v_mfma_f32_32x32x16_{bf16,f16}: lane (r = lane & 31, h = lane >> 5)
// supplies A[row r][k = 8h .. 8h+7] and B[k = 8h .. 8h+7][col r], i.e. exactly one granule per operand per k-step.
//   MODE 0: bf16 slab (MFAR_DTYPE_BF16), queries split exactly into three bf16 terms;
//   MODE 1: fp16 SCREEN slab of an fp32 index (mfar_screen.h), queries split into two fp16 terms.
// ---------------------------------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
struct S1X {
    static constexpr int TERMS = MODE ? 2 : 3;
    static constexpr int STAGES = MODE ? S1H_STAGES : S1B_STAGES;
    static constexpr int Q_STAGE = TERMS * 2048;           // bytes of query tile per ring stage in LDS
    static constexpr int Q_STEP_MEM = MODE ? 4096 : 8192;  // bytes of query tile per k-step in memory
    static constexpr int LOADS = MODE ? 3 : 4;             // LDS-DMA instructions per wave per stage
    static constexpr int D_BYTES = 4 * STAGES * 2048;
    static constexpr int LDS_BYTES = D_BYTES + STAGES * Q_STAGE + S1_STATE_BYTES_(S1_SCAP_LDS);
};

template <int N>
__device__ __forceinline__ void s1_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int MODE>
__device__ __forceinline__ void s1_body_x16(const S1Params& p, const int chunk_id) {
    typedef S1X<MODE> X;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem + X::D_BYTES;
    const S1State st = s1_state(smem + X::D_BYTES + X::STAGES * X::Q_STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform
    const int f = ck.f;
    if (p.only_failed && !p.only_failed[f]) return;   // workgroup-uniform
    const int t0 = ck.t0;
    int t1 = ck.t1;
    if (p.sample) t1 = min(t1, t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    const size_t wgq0 = (size_t)chunk_id * p.qw;
    s1_state_init(st, p, f);

    // granule of row (32*blk + j), k-half h inside a 2 KB tile (rows are 32 B)
    const int off = j * 32 + ((h ^ ((j >> 3) & 1)) << 4);

    char* const dring = smem + w * (X::STAGES * 2048);
    const size_t step_bytes = 2048;
    const size_t tile_jump = (size_t)3 * p.n_steps * step_bytes;
    const char* dnext = (const char*)p.slab + (size_t)ck.base * 2 + ((size_t)(4 * t0 + w) * p.n_steps) * step_bytes + lane * 16;
    // bf16: the 6 KB query tile is 6 pieces of 1 KB: waves 0-1 load pieces {2w, 2w+1}, waves 2-3 load piece 2+w (twice, so
    // that every wave issues the same number of loads per stage and the counted vmcnt waits stay uniform).
    // fp16: 4 pieces, wave w loads piece w.
    const int qp0 = MODE ? w : (w < 2 ? 2 * w : 2 + w), qp1 = w < 2 ? 2 * w + 1 : 2 + w;
    const char* const qbase = (const char*)p.qt + lane * 16;
    int s_next = 0, st_next = 0;
    const int total = (t1 - t0) * p.n_steps;
    int issued = 0;
#define S1X_ISSUE_NEXT()                                                                                  \
    do {                                                                                                  \
        char* db_ = dring + st_next * 2048;                                                               \
        S1_GLDS(dnext, db_, 2);                                                                           \
        S1_GLDS(dnext + 1024, db_ + 1024, 2);                                                             \
        const char* qs_ = qbase + (size_t)s_next * X::Q_STEP_MEM;                                         \
        char* qd_ = qring + st_next * X::Q_STAGE;                                                         \
        S1_GLDS(qs_ + qp0 * 1024, qd_ + qp0 * 1024, 0);                                                   \
        if (MODE == 0) S1_GLDS(qs_ + qp1 * 1024, qd_ + qp1 * 1024, 0);                                    \
        dnext += step_bytes;                                                                              \
        if (++s_next == p.n_steps) {                                                                      \
            s_next = 0;                                                                                   \
            dnext += tile_jump;                                                                           \
        }                                                                                                 \
        st_next = (st_next == X::STAGES - 1) ? 0 : st_next + 1;                                           \
        ++issued;                                                                                         \
    } while (0)
    for (int i = 0; i < X::STAGES - 1; ++i)
        if (total > i) S1X_ISSUE_NEXT();

    int it = 0, st_cur = 0;
    for (int t = t0; t < t1; ++t) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
        for (int s = 0; s < p.n_steps; ++s, ++it) {
            // stage `it` landed; up to STAGES - 1 newer stages (LOADS loads each) may stay in flight
            const int ahead = issued - it - 1;
            static_assert(X::STAGES <= 6, "extend the wait ladder");
            if (ahead >= 5) s1_wait_barrier<5 * X::LOADS>();
            else if (ahead == 4) s1_wait_barrier<4 * X::LOADS>();
            else if (ahead == 3) s1_wait_barrier<3 * X::LOADS>();
            else if (ahead == 2) s1_wait_barrier<2 * X::LOADS>();
            else if (ahead == 1) s1_wait_barrier<1 * X::LOADS>();
            else s1_wait_barrier<0>();
            const char* cur = dring + st_cur * 2048;
            const char* curq = qring + st_cur * X::Q_STAGE;
            st_cur = (st_cur == X::STAGES - 1) ? 0 : st_cur + 1;
            if (MODE == 0) {
                const bf16x8 d0 = *(const bf16x8*)(cur + off);
                const bf16x8 d1 = *(const bf16x8*)(cur + 1024 + off);
                const bf16x8 qh0 = *(const bf16x8*)(curq + off), qh1 = *(const bf16x8*)(curq + 1024 + off);
                const bf16x8 qm0 = *(const bf16x8*)(curq + 2048 + off), qm1 = *(const bf16x8*)(curq + 3072 + off);
                const bf16x8 ql0 = *(const bf16x8*)(curq + 4096 + off), ql1 = *(const bf16x8*)(curq + 5120 + off);
                if (issued < total && issued < it + X::STAGES) S1X_ISSUE_NEXT();
                // smallest terms first: lo, mid, hi (all products are exact; this keeps the fp32 accumulation tight)
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql1, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qm0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qm1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qm0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qm1, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh1, acc11, 0, 0, 0);
            } else {
                const f16x8 d0 = *(const f16x8*)(cur + off);
                const f16x8 d1 = *(const f16x8*)(cur + 1024 + off);
                const f16x8 qh0 = *(const f16x8*)(curq + off), qh1 = *(const f16x8*)(curq + 1024 + off);
                const f16x8 ql0 = *(const f16x8*)(curq + 2048 + off), ql1 = *(const f16x8*)(curq + 3072 + off);
                if (issued < total && issued < it + X::STAGES) S1X_ISSUE_NEXT();
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, ql0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, ql1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, ql0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, ql1, acc11, 0, 0, 0);
                acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, qh0, acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d0, qh1, acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, qh0, acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(d1, qh1, acc11, 0, 0, 0);
            }
        }
        if (p.dbg & 1) {
            asm volatile("" ::"v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11));
            continue;
        }
        if (p.sample == 2) {
            s1_sample_top2(p, ck, t - t0, t, w, acc00, acc01, acc10, acc11);
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier A + early refill (see the fp32 body)
        if (issued < total) S1X_ISSUE_NEXT();
        s1_epilogue<S1_SCAP_LDS>(p, st, ck.n_rows, t, w, wgq0, acc00, acc01, acc10, acc11);
    }
#undef S1X_ISSUE_NEXT
    if (p.sample != 2) s1_flush<S1_SCAP_LDS>(p, st, w, wgq0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-ring variant of the 16-bit pass ("x16r").  A doc row is consumed by exactly one lane pair, and the MFMA A operand
// lives in VGPRs anyway, so the doc tiles do not need LDS at all: every lane loads ITS 16-byte granule of the 2 KB tile
// straight into registers (the wave's 64 granules of a 32-row half tile are one contiguous KB: fully coalesced), R k-steps
// ahead.  Only the query tile -- shared by the four waves -- still goes through an LDS ring.  LDS per workgroup drops from
// 74.5 KB to R * 4 KB + state (25 KB for fp16, R = 6), which leaves > 100 KB per CU to the small kernels of the previous /
// next batch that run beside this pass.  Needs n_steps % R == 0 (register slots are compile-time indices); the host
// falls back to the LDS-ring kernel otherwise.
// ---------------------------------------------------------------------------------------------------------------------
//   MODE 2: bf16 slab scanned by the CERTIFIED pass (mfar_screen.h "bf16 indexes"): queries split into TWO bf16 terms (16 of
//           their 24 significant bits; the docs are the index's own rows, exact), 64 columns; tiles as in MODE 1.
template <int MODE, int R>
struct S1XR {
    static constexpr int TERMS = MODE ? 2 : 3;
    static constexpr int Q_STAGE = TERMS * 2048;
    static constexpr int Q_STEP_MEM = MODE ? 4096 : 8192;
    static constexpr int LOADS = MODE ? 3 : 4;             // vm instructions per wave per stage: 2 doc loads + query pieces
    static constexpr int SCAP = MODE ? S1_SCAP_REG : S1_SCAP_REG / 2;   // bf16: 6 KB query stages, keep 50 KB per CU free
    static constexpr int LDS_BYTES = R * Q_STAGE + S1_STATE_BYTES_(SCAP);
};

template <int MODE, int R>
__device__ __forceinline__ void s1_body_x16r(const S1Params& p, const int chunk_id) {
    typedef S1XR<MODE, R> X;
    typedef short vec8 __attribute__((ext_vector_type(8)));   // 16 bytes of bf16 / fp16 bits
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem;
    const S1State st = s1_state(smem + R * X::Q_STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform
    const int f = ck.f;
    if (p.only_failed && !p.only_failed[f]) return;   // workgroup-uniform
    const int t0 = ck.t0;
    int t1 = ck.t1;
    if (p.sample) t1 = min(t1, t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    const size_t wgq0 = (size_t)chunk_id * p.qw;
    s1_state_init(st, p, f);

    // granule of row (32*blk + j), k-half h inside a 2 KB tile: the same offset in memory (docs) and in LDS (queries)
    const int off = j * 32 + ((h ^ ((j >> 3) & 1)) << 4);
    const size_t step_bytes = 2048;
    const size_t tile_jump = (size_t)3 * p.n_steps * step_bytes;
    const char* dnext = (const char*)p.slab + (size_t)ck.base * 2 + ((size_t)(4 * t0 + w) * p.n_steps) * step_bytes + off;
    const int qp0 = MODE ? w : (w < 2 ? 2 * w : 2 + w), qp1 = w < 2 ? 2 * w + 1 : 2 + w;
    const char* const qbase = (const char*)p.qt + lane * 16;
    int s_next = 0;
    // Every k-step issues exactly one stage, unconditionally: the R - 1 stages past the end of the chunk re-read the
    // chunk's last stage (clamped address) and are never consumed.  A constant issue pattern keeps the compiler's own
    // waitcnt insertion for the doc registers precise (with conditional issues it has to assume the worst path and
    // drains the queue once per R steps).
    const char* const dlast = (const char*)p.slab + (size_t)ck.base * 2 +
                              ((size_t)(4 * (t1 - 1) + w) * p.n_steps + (p.n_steps - 1)) * step_bytes + off;
    vec8 dr0[R], dr1[R];
    // the query-piece LDS-DMA goes through inline asm: hipcc treats a visible global_load_lds and plain global loads as
    // two event kinds on one counter and answers the mix with vmcnt(0) in front of every use of a loaded register
#define S1R_QDMA(S, D)                                                                                           \
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(S), "s"(__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(D))) : "memory")   // (m0 is reserved: hipcc re-loads it in front of every instruction of its own that reads it)
#define S1R_ISSUE(SLOT)                                                                                   \
    do {                                                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dr0[SLOT]) : "v"(dnext) : "memory");    \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(dr1[SLOT]) : "v"(dnext) : "memory"); \
        const char* qs_ = qbase + (size_t)s_next * X::Q_STEP_MEM;                                         \
        char* qd_ = qring + (SLOT) * X::Q_STAGE;                                                          \
        S1R_QDMA(qs_ + qp0 * 1024, qd_ + qp0 * 1024);                                                     \
        if (MODE == 0) S1R_QDMA(qs_ + qp1 * 1024, qd_ + qp1 * 1024);                                      \
        const char* nx_ = dnext + step_bytes;                                                             \
        if (++s_next == p.n_steps) {                                                                      \
            s_next = 0;                                                                                   \
            nx_ += tile_jump;                                                                             \
        }                                                                                                 \
        dnext = (unsigned long long)nx_ <= (unsigned long long)dlast ? nx_ : dlast;                       \
    } while (0)
    // prologue: R - 1 stages in flight (slots 0 .. R-2)
#pragma unroll
    for (int i = 0; i < R - 1; ++i) S1R_ISSUE(i);

    for (int t = t0; t < t1; ++t) {
        f32x16 acc00, acc01, acc10, acc11;
        s1_acc_init(p, ck, t, w, acc00, acc10);
        acc01 = acc00;
        acc11 = acc10;
        for (int s0 = 0; s0 < p.n_steps; s0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                // query tile of this stage landed (own piece: counted vmcnt -- R - 2 younger stages may stay in flight;
                // other waves' pieces: the barrier); the stage's doc registers were loaded before that piece
                // (the doc registers are in/out operands of the wait: nothing that reads them can be scheduled above it)
                asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(dr0[u]), "+v"(dr1[u]) : "n"((R - 2) * X::LOADS) : "memory");
                const char* curq = qring + u * X::Q_STAGE;
                const vec8 d0 = dr0[u], d1 = dr1[u];
                if (MODE == 0) {
                    const bf16x8 qh0 = *(const bf16x8*)(curq + off), qh1 = *(const bf16x8*)(curq + 1024 + off);
                    const bf16x8 qm0 = *(const bf16x8*)(curq + 2048 + off), qm1 = *(const bf16x8*)(curq + 3072 + off);
                    const bf16x8 ql0 = *(const bf16x8*)(curq + 4096 + off), ql1 = *(const bf16x8*)(curq + 5120 + off);
                    // every wave is past the barrier: slot (u + R - 1) % R (the previous stage) is free
                    S1R_ISSUE((u + R - 1) % R);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql1, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qm0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qm1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qm0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qm1, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh1, acc11, 0, 0, 0);
                } else if (MODE == 2) {
                    const bf16x8 qh0 = *(const bf16x8*)(curq + off), qh1 = *(const bf16x8*)(curq + 1024 + off);
                    const bf16x8 ql0 = *(const bf16x8*)(curq + 2048 + off), ql1 = *(const bf16x8*)(curq + 3072 + off);
                    S1R_ISSUE((u + R - 1) % R);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ql1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ql1, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, qh1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, qh1, acc11, 0, 0, 0);
                } else {
                    const f16x8 qh0 = *(const f16x8*)(curq + off), qh1 = *(const f16x8*)(curq + 1024 + off);
                    const f16x8 ql0 = *(const f16x8*)(curq + 2048 + off), ql1 = *(const f16x8*)(curq + 3072 + off);
                    const f16x8 e0 = __builtin_bit_cast(f16x8, d0), e1 = __builtin_bit_cast(f16x8, d1);
                    S1R_ISSUE((u + R - 1) % R);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, ql0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, ql1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, ql0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, ql1, acc11, 0, 0, 0);
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qh0, acc00, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qh1, acc01, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qh0, acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qh1, acc11, 0, 0, 0);
                }
            }
        }
        if (p.dbg & 1) {
            asm volatile("" ::"v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11));
            continue;
        }
        if (p.sample == 2) {
            s1_sample_top2(p, ck, t - t0, t, w, acc00, acc01, acc10, acc11);
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier A (see the fp32 body)
        s1_epilogue<X::SCAP>(p, st, ck.n_rows, t, w, wgq0, acc00, acc01, acc10, acc11);
    }
#undef S1R_ISSUE
    // the stages issued past the end are still in flight: no LDS-DMA write may land after the workgroup has left
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.sample != 2) s1_flush<X::SCAP>(p, st, w, wgq0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-ring variant of the EXACT fp32 pass ("f32r").  v_mfma_f32_32x32x2_f32 takes its A operand from one VGPR per lane: lane
// (j, h) supplies row 32 db + j, dim k = h, and the four MFMAs x = 0 .. 3 of a fragment consume dims 8 g + 4 h + x -- 16 contiguous
// bytes of the row's 128-byte line in the fp32 slab (mfar_device.h).  So the docs do not need LDS here either: every lane loads its
// four 16-byte pieces of a k-step (two doc blocks x two fragments) straight into one of R register slots, R - 1 k-steps ahead; only
// the shared 4 KB query tile still goes through an LDS ring.  The chain order is the contract's (fragment 0 then 1, x ascending):
// bits unchanged.  Why: the LDS-DMA doc stream of s1_body_f32 costs MFMA issue slots -- the same MFMA sequence sustains 150-155
// TFLOP/s from registers and 119-130 beside an LDS-DMA stream at this kernel's ratio (profiles/r03_e_mfma_f32_clock_probe.txt).
// Addresses: a wave-uniform base in SGPRs + per-lane 32-bit offsets (j * 128 + h * 16, and + 4096 for the second doc block: the
// 13-bit immediate cannot carry it); the two k-steps of a pair read the same lines, the second time from L2.  Needs n_steps % R == 0.
// (A variant that also took the QUERY fragments straight from memory -- no LDS, no barrier in the loop -- hung on the GPU and was
// dropped unexplained: round 4.)
// ---------------------------------------------------------------------------------------------------------------------
#define S1_SCAP_F32R 32
template <int R>
struct S1FR {
    static constexpr int LOADS = 5;                          // 4 doc pieces + the wave's quarter of the query tile
    static constexpr int SCAP = S1_SCAP_F32R;
    static constexpr int LDS_BYTES = R * 4096 + S1_STATE_BYTES_(S1_SCAP_F32R);
};
template <int R>
__device__ __forceinline__ void s1_body_f32r(const S1Params& p, const int chunk_id) {
    typedef S1FR<R> X;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem;
    const S1State st = s1_state(smem + R * 4096);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform
    const int f = ck.f;
    if (p.only_failed && !p.only_failed[f]) return;   // workgroup-uniform
    const int t0 = ck.t0;
    int t1 = ck.t1;
    if (p.sample) t1 = min(t1, t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    const size_t wgq0 = (size_t)chunk_id * p.qw;
    s1_state_init(st, p, f);

    // query fragments in the 4 KB LDS image of a k-step: row (32 blk + j), dims 8 g + 4 h .. + 3 (chunk c = 2 g + h, swizzled)
    const int sw = (j >> 2) & 3;
    const int off_g0 = j * 64 + (((0 + h) ^ sw) << 4);
    const int off_g1 = j * 64 + (((2 + h) ^ sw) << 4);
    // docs: this lane's piece of row (32 db + j) inside the 8 KB tile of a k-step pair
    const u32 voff0 = (u32)(j * 128 + h * 16), voff1 = voff0 + 4096u;
    const size_t step_bytes = 4096;
    const size_t tile_jump = (size_t)3 * p.n_steps * step_bytes;       // from the end of the wave's block to its block of the next tile
    const char* dblk = (const char*)p.slab + (size_t)ck.base * 4 + ((size_t)(4 * t0 + w) * p.n_steps) * step_bytes;        // uniform
    const char* const dlast_blk = (const char*)p.slab + (size_t)ck.base * 4 + ((size_t)(4 * (t1 - 1) + w) * p.n_steps) * step_bytes;
    const char* const qbase = (const char*)p.qt + w * 1024;                                                              // uniform
    const u32 ldsq = (u32)(uintptr_t)qring + (u32)w * 1024u;
    const u32 l16 = (u32)lane * 16u;
    int s_next = 0, pf_clamped = 0;
    f32x4 dr[R][4];
#define S1FR_ISSUE(SLOT)                                                                                   \
    do {                                                                                                   \
        const char* src_ = dblk + (size_t)(s_next >> 1) * 8192 + (size_t)(s_next & 1) * 64;                \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dr[SLOT][0]) : "v"(voff0), "s"(src_) : "memory");            \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=&v"(dr[SLOT][1]) : "v"(voff0), "s"(src_) : "memory");  \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(dr[SLOT][2]) : "v"(voff1), "s"(src_) : "memory");            \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=&v"(dr[SLOT][3]) : "v"(voff1), "s"(src_) : "memory");  \
        const char* qs_ = qbase + (size_t)s_next * step_bytes;                                             \
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(l16), "s"(qs_), "s"(ldsq + (u32)((SLOT) * 4096)) : "memory"); \
        if (!pf_clamped && ++s_next == p.n_steps) {                                                        \
            if (dblk == dlast_blk) {            /* past the chunk: keep re-reading its last stage (never consumed) */ \
                s_next = p.n_steps - 1;                                                                    \
                pf_clamped = 1;                                                                            \
            } else {                                                                                       \
                s_next = 0;                                                                                \
                dblk += (size_t)p.n_steps * step_bytes + tile_jump;                                        \
            }                                                                                              \
        }                                                                                                  \
    } while (0)
#pragma unroll
    for (int i = 0; i < R - 1; ++i) S1FR_ISSUE(i);

    for (int t = t0; t < t1; ++t) {
        f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};  // [doc block][query block]
        for (int s0 = 0; s0 < p.n_steps; s0 += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier"
                             : "+v"(dr[u][0]), "+v"(dr[u][1]), "+v"(dr[u][2]), "+v"(dr[u][3])
                             : "n"((R - 2) * X::LOADS)
                             : "memory");
                const char* curq = qring + u * 4096;
                const f32x4 q00 = *(const f32x4*)(curq + off_g0);
                const f32x4 q01 = *(const f32x4*)(curq + off_g1);
                const f32x4 q10 = *(const f32x4*)(curq + 2048 + off_g0);
                const f32x4 q11 = *(const f32x4*)(curq + 2048 + off_g1);
                const f32x4 d00 = dr[u][0], d01 = dr[u][1], d10 = dr[u][2], d11 = dr[u][3];
                // every wave is past the barrier: slot (u + R - 1) % R (the previous stage) is free
                S1FR_ISSUE((u + R - 1) % R);
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
            }
        }
        if (p.dbg & 1) {
            asm volatile("" ::"v"(acc00), "v"(acc01), "v"(acc10), "v"(acc11));
            continue;
        }
        if (p.sample == 2) {
            s1_sample_top2(p, ck, t - t0, t, w, acc00, acc01, acc10, acc11);
            continue;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier A (see the fp32 body)
        s1_epilogue<X::SCAP>(p, st, ck.n_rows, t, w, wgq0, acc00, acc01, acc10, acc11);
    }
#undef S1FR_ISSUE
    // the stages issued past the end are still in flight: no LDS-DMA write may land after the workgroup has left
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.sample != 2) s1_flush<X::SCAP>(p, st, w, wgq0);
}

// ROW MODE (S1Params::arow): acc[db][x][r] += A_x * |c_row| for the wave's 64 rows.  The norms arrive one per lane (lane = row of the
// block, loaded at the start of the tile), are parked in a 256-byte LDS area of the wave and read back in the accumulator layout: lane
// (j, h) holds rows 32 db + 8 g + 4 h + i (i = 0 .. 3) in elements r = 4 g + i -- four consecutive floats per (db, g).
__device__ __forceinline__ void lds_write_b32(u32 addr, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 lds_read_b128(u32 addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void s1_row_norms_park(u32 area, float nr) {
    lds_write_b32(area + (u32)(threadIdx.x & 63) * 4u, nr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void s1_row_bound(u32 area, float A0, float A1, f32x16& x00, f32x16& x01, f32x16& x10, f32x16& x11) {
    const u32 hb = (u32)((threadIdx.x & 63) >> 5) * 16u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 n0 = lds_read_b128(area + (u32)(8 * g) * 4u + hb);
        const f32x4 n1 = lds_read_b128(area + (u32)(32 + 8 * g) * 4u + hb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x00[4 * g + i] = __builtin_fmaf(A0, n0[i], x00[4 * g + i]);
            x01[4 * g + i] = __builtin_fmaf(A1, n0[i], x01[4 * g + i]);
            x10[4 * g + i] = __builtin_fmaf(A0, n1[i], x10[4 * g + i]);
            x11[4 * g + i] = __builtin_fmaf(A1, n1[i], x11[4 * g + i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE passes: 128 query columns per scan (s1_body_wide below; kernels mfar_stage1_f16w_* over the fp16 screen slab of an fp32 index,
// mfar_stage1_bf16c_* / bf16w_* over a bf16 slab).  The 64-column screened pass is HBM-bound with the MFMA pipe at ~40 % (two fp16
// query terms x 64 queries = 8 MFMAs per 2 KB of docs per wave).  Spending the same 8 MFMAs on ONE fp16 term of 128 queries reads the
// slab once per 128 queries instead of once per 64: half the scan bytes per query at the same MFMA load.  The price is the query
// rounding error (u16 |q_i| per element instead of u16^2), which the certificate's bound accounts for (mfar_screen.h: eps roughly
// doubles; the re-scored margin k' - k is as wide as before).  4 waves, two workgroups per CU, docs in a register ring, the query
// stage (two 64-query blocks x 2 KB) in an LDS ring, 8 accumulators (128 VGPRs), one selection state per 64-query block; the append
// half of the epilogue runs for both blocks before anything register-hungry (drain, compaction) does.
// Shapes that were measured and dropped (1 M x 8 x 768, 2.0-2.3 ms per launch for this one): ONE workgroup per CU -- 8 waves of 32
// rows x 128 queries (184 VGPRs, no spills: 3.2 ms) or these 4 waves with a 12-slot ring and 512 VGPRs (3.6 ms): with a single barrier
// domain per CU every late load stalls the whole CU.  The price of this shape is its register file: 256 VGPRs x 2 waves per SIMD
// leave no room for the small kernels of the neighbouring launches, which therefore run between the scans, not beside them.
// Round 4: wave-uniform address bases in SGPRs + one per-lane 32-bit offset (no 64-bit VALU address arithmetic in the loop, fewer
// spills around the epilogue) and dynamic work distribution: 2.10-2.13 -> 1.99-2.02 ms per launch at 1 M x 8 x 768 (0.67 -> 0.71 of
// 8 TB/s).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef S1_SCAP_WIDE
#define S1_SCAP_WIDE 32
#endif
// ---------------------------------------------------------------------------------------------------------------------
// Dynamic work distribution (S1Params::unit_ctr).  A scan grid is ONE wave of workgroups that fill the register file, and a
// workgroup with a fixed chunk runs for the whole kernel: any workgroup the dispatcher places late -- because the small
// kernels of the previous launch's tail hold its CU when the scan starts -- ends late by as much, while the CUs of the
// punctual ones sit half empty (measured: 1.25 M x 16 bf16, scan 7.2 ms alone, 11.5 ms behind a 1.3 ms tail; 129 k x 22: 0.8 ->
// 1.8 ms).  With units, a late workgroup simply claims fewer of them.  Lists stay per (workgroup, query): the merge does
// not change, and neither do the results (a list is a set; thresholds only ever drop rows that k' better rows of the same
// list beat).
//   * the first unit is claimed synchronously, every further one a whole unit AHEAD by one lane's returning atomic issued in
//     inline asm (no compiler wait), picked up at the epilogue of the current unit's first tile and published through LDS;
//   * the prefetch cursor runs R - 1 k-steps ahead of the MFMAs and moves into the published unit when it leaves the current
//     one; units hold >= 2 tiles except the last of a field, behind which nothing can follow, so a unit is always published
//     before the cursor needs it.  With no unit to go to the cursor keeps re-reading its last stage (never consumed).
// ---------------------------------------------------------------------------------------------------------------------
struct S1Unit {           // workgroup-uniform
    int t0, t1;           // tiles [t0, t1) of one field; t0 < 0: none
};
__device__ __forceinline__ S1Unit s1_unit_of(int u, int U, int n_tiles) {
    S1Unit r;
    r.t0 = u * U;
    r.t1 = min(r.t0 + U, n_tiles);
    if (u < 0 || r.t0 >= n_tiles) r.t0 = r.t1 = -1;
    return r;
}
// one lane's returning atomic, invisible to hipcc's waitcnt insertion; `raw` is preset to -2 ("not back yet")
__device__ __forceinline__ void s1_unit_claim_async(int* ctr, int& raw) {
    const int one = 1;
    asm volatile("v_mov_b32 %0, -2\n\tglobal_atomic_add %0, %1, %2, off sc0" : "=&v"(raw) : "v"(ctr), "v"(one) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE certified pass over a bf16 slab ("bf16w"): 128 query columns per scan, straight off the index's own bf16 rows -- no fp16
// copy.  Two bf16 query terms (hi + mid = 16 significant bits; every product exact in fp32) x 128 columns = 16 MFMAs per 2 KB of
// docs per wave: the slab is read once per 128 queries (the exact bf16 pass: three terms x 64 columns, once per 64), and the
// only approximation left is the query's third term -- the certificate's eps is ~6x tighter than the fp16 screen's
// (mfar_screen.h).  The query stage is 8 KB per k-step ([term][query block][64][16]), two LDS-DMA
// pieces per wave.  The pass scans every document and ranks unique rows (s1_acc_init).
// ---------------------------------------------------------------------------------------------------------------------
//
// CV = 1 ("bf16c"): the same scan at HALF the MFMAs.  The bf16 granules are turned into fp16 IN REGISTERS -- a bf16 value has 8
// significant bits, so under the field's power-of-two scale (largest |value| in [2^13, 2^14)) every value that is a normal fp16
// number converts exactly; smaller magnitudes are clamped UP to 2^-14 (error <= 2^-14 scaled units per element, in the certificate's
// eps) -- with five packed-integer VALU ops per dword that issue in the shadow of the MFMAs, and the query is ONE fp16 term of 128
// columns as in the wide pass of an fp32 index: 8 MFMAs per k-step instead of 16, a 4 KB query stage instead of 8 KB.  eps carries
// the query rounding (u16) but no doc rounding and no centring (mfar_screen.h).
template <int R, int SCAP_, int CV>
struct S1BW {
    static constexpr int Q_STAGE = CV ? 4096 : 8192;
    static constexpr int LOADS = CV ? 3 : 4;
    static constexpr int SCAP = SCAP_;
    static constexpr int LDS_BYTES = R * Q_STAGE + 2 * S1_STATE_BYTES_(SCAP_);             // (ROW MODE kernels: + 1 KB, the four waves' row-norm areas)
};
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// one dword = two bf16 values -> two fp16 values: magnitude clamped up to the smallest normal, exponent rebiased, mantissa moved
__device__ __forceinline__ u32 s1_bf16x2_to_f16x2(u32 x, u32 tmin2, u32 bias2) {
    u32 mag = x & 0x7FFF7FFFu;
    mag = __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, mag), __builtin_bit_cast(u16x2, tmin2)));
    mag = __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, mag) - __builtin_bit_cast(u16x2, bias2)));
    mag = __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, mag) << (u16x2){3, 3}));
    return (mag & 0x7FFF7FFFu) | (x & ~0x7FFF7FFFu);       // (one v_bfi_b32: the sign bits of x over the converted magnitudes)
}
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 s1_cvt_granule(u32x4 g, u32 tmin2, u32 bias2) {
#ifdef MFAR_EXP_NOCVT      // timing experiment only (wrong results): what the conversion costs
    return __builtin_bit_cast(f16x8, g);
#endif
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = s1_bf16x2_to_f16x2(g[i], tmin2, bias2);
    return __builtin_bit_cast(f16x8, o);
}

// ROWM = 1: the ROW MODE instantiation (its own kernels, mfar_stage1_f16w*_rm_kernel): the extra registers of the row-norm code would
// otherwise cost the common path 70 more spilled VGPRs.
template <int R, int SCAP_, int CV, int ROWM = 0>
__device__ __forceinline__ void s1_body_wide(const S1Params& p, const int chunk_id) {
    typedef S1BW<R, SCAP_, CV> X;
    typedef short vec8 __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const qring = smem;
    const S1State stA = s1_state(smem + R * X::Q_STAGE);
    const S1State stB = s1_state(smem + R * X::Q_STAGE + S1_STATE_BYTES_(X::SCAP));
    int* const unit_slot = stA.flag + 2;              // spare (8-byte aligned) words behind the compaction flag

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const S1Chunk ck = s1_load_chunk(p, chunk_id);    // workgroup-uniform
    const int f = ck.f;
    if (p.only_failed && !p.only_failed[f]) return;
    const size_t wgq0 = (size_t)chunk_id * p.qw;      // qw == 128
    s1_state_init(stA, p, f, 0);
    s1_state_init(stB, p, f, 64);
#ifdef MFAR_TRACE
    const unsigned long long tr_t0 = wall_clock64();
    int tr_units = 0;
#endif

    // the first unit: the chunk's own tile range, or -- dynamic distribution, full passes only -- claimed from the field's counter
    const bool dyn = p.unit_ctr != nullptr && !p.sample;
    const int n_tiles_f = max(1, (ck.n_rows + S1_TILE_ROWS - 1) / S1_TILE_ROWS);
    int* const ctr = dyn ? p.unit_ctr + f : nullptr;
    S1Unit cur;
    cur.t0 = ck.t0;
    cur.t1 = ck.t1;
    if (p.sample) cur.t1 = min(cur.t1, cur.t0 + (p.sample == 2 ? ck.ns : p.sample_tiles));
    if (dyn) {
        if (tid == 0) *unit_slot = atomicAdd(ctr, 1);
        __syncthreads();
        cur = s1_unit_of(__builtin_amdgcn_readfirstlane(*unit_slot), p.unit_tiles, n_tiles_f);
        __syncthreads();
    }
    S1Unit nxt;                                       // the published unit behind `cur`
    nxt.t0 = nxt.t1 = -1;
    int raw = -1;                                     // wave 0, lane 0: the pending claim

    // Addresses: a wave-uniform 64-bit base in SGPRs + ONE per-lane 32-bit offset.  The epilogue below needs every VGPR it can
    // get (8 accumulators + the ring in flight); per-lane 64-bit cursors were spilled around it, and hipcc answers the reload of a
    // value the k-loop uses with a vmcnt wait INSIDE the loop that drains the doc ring (it does not count the asm loads).
    const size_t step_bytes = 2048;
    const size_t tile_jump = (size_t)3 * p.n_steps * step_bytes;
    const char* const fbase = (const char*)p.slab + (size_t)ck.base * 2 + (size_t)w * p.n_steps * step_bytes;   // block w of tile 0      // uniform
    const size_t tile_bytes = (size_t)4 * p.n_steps * step_bytes;
    const char* const qbase = (const char*)p.qt + w * 1024;   // wave w loads pieces w and 4 + w of the 8 KB stage (CV: piece w of 4 KB)   // uniform
    const u32 ldsq = (u32)(uintptr_t)qring + (u32)w * 1024u;                                                            // uniform
    // ROW MODE: this field ranks by upper bounds (S1Params::arow); the lane's four query columns' factors, fetched before the ring starts
    const bool row_on = ROWM && p.arow != nullptr && ((p.row_mask >> f) & 1u) != 0u;
    float rA0 = 0.0f, rA1 = 0.0f, rB0 = 0.0f, rB1 = 0.0f, nr = 0.0f;
    const u32 rarea = (u32)(uintptr_t)(smem + R * X::Q_STAGE + 2 * S1_STATE_BYTES_(X::SCAP)) + (u32)w * 256u;
    const float* rn_field = nullptr;
    if (row_on) {
        const unsigned long long rb_ = (unsigned long long)(p.rnorm + p.dump_base[f]);
        rn_field = (const float*)(((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(rb_ >> 32)) << 32) |
                                  (u32)__builtin_amdgcn_readfirstlane((int)(u32)rb_));
    }
    if (row_on) {
        const float* ar = p.arow + (size_t)f * p.qw + (tid & 31);
        rA0 = ar[0];
        rA1 = ar[32];
        rB0 = ar[64];
        rB1 = ar[96];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rA0), "+v"(rA1), "+v"(rB0), "+v"(rB1)::"memory");
    }
    // SCORE DUMP: the field's 128 quantisation factors, parked in LDS (the epilogue has no VGPR to keep them in)
    float* const dinv_s = (float*)(smem + R * X::Q_STAGE + 2 * S1_STATE_BYTES_(X::SCAP) + (ROWM ? 1024 : 0));
    if (CV == 2 && p.dump && !p.sample) {              // kernel-uniform
        if (tid < 128) dinv_s[tid] = p.dump_inv[(size_t)f * 128 + tid];
        __syncthreads();
    }
    u32 tmin2 = 0, bias2 = 0;
    if (CV == 1) {
        const uint2 cv = p.cvt[f];
        tmin2 = __builtin_amdgcn_readfirstlane(cv.x);
        bias2 = __builtin_amdgcn_readfirstlane(cv.y);
    }
    // prefetch cursor (uniform): next stage to issue; tiles of its unit not completely issued yet; whether it has moved into `nxt`
    const char* dnext = fbase + (size_t)max(cur.t0, 0) * tile_bytes;
    int s_next = 0, pf_left = cur.t1 - cur.t0, pf_in_next = 0, pf_clamped = cur.t0 < 0 ? 1 : 0;
    vec8 dr0[R], dr1[R];
    // per-lane offsets, recomputed from the lane id at every tile (two VALU ops; never worth a spill slot)
    int lane = tid & 63;
    int off = (lane & 31) * 32 + ((((lane >> 5) ^ ((lane >> 3) & 1)) & 1) << 4);   // granule of row (32 blk + j), k-half h in a 2 KB tile
    int l16 = lane * 16;
#define S1BW_QDMA(VO, SB, M0)                                                                                      \
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VO), "s"(SB), "s"(M0) : "memory")
#define S1BW_ISSUE(SLOT)                                                                                  \
    do {                                                                                                  \
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(dr0[SLOT]) : "v"(off), "s"(dnext) : "memory");               \
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=&v"(dr1[SLOT]) : "v"(off), "s"(dnext) : "memory");   \
        const char* qs_ = qbase + (size_t)s_next * X::Q_STAGE;                                            \
        S1BW_QDMA(l16, qs_, ldsq + (u32)((SLOT) * X::Q_STAGE));                                           \
        if (!CV) S1BW_QDMA(l16, qs_ + 4096, ldsq + (u32)((SLOT) * X::Q_STAGE + 4096));                    \
        if (!pf_clamped) {                                                                                \
            dnext += step_bytes;                                                                          \
            if (++s_next == p.n_steps) {                                                                  \
                s_next = 0;                                                                               \
                if (--pf_left > 0) dnext += tile_jump;                                                    \
                else if (nxt.t0 >= 0 && !pf_in_next) {   /* into the published unit */                    \
                    dnext = fbase + (size_t)nxt.t0 * tile_bytes;                                          \
                    pf_left = nxt.t1 - nxt.t0;                                                            \
                    pf_in_next = 1;                                                                       \
                } else {                                 /* nothing follows: keep re-reading the last stage */ \
                    dnext -= step_bytes;                                                                  \
                    s_next = p.n_steps - 1;                                                               \
                    pf_clamped = 1;                                                                       \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
    } while (0)
#pragma unroll
    for (int i = 0; i < R - 1; ++i) S1BW_ISSUE(i);

    while (cur.t0 >= 0) {
        if (dyn && w == 0 && (tid & 63) == 0) s1_unit_claim_async(ctr, raw);      // the unit after this one
        for (int t = cur.t0; t < cur.t1; ++t) {
            lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            asm volatile("" : "+v"(lane));
            off = (lane & 31) * 32 + ((((lane >> 5) ^ ((lane >> 3) & 1)) & 1) << 4);
            l16 = lane * 16;
            if (row_on) {      // this lane's row of the wave's block: its norm (arrives long before the epilogue: loads return in order)
                const float* rs_ = rn_field + ((size_t)t * S1_TILE_ROWS + (size_t)w * 64);
                asm volatile("global_load_dword %0, %1, %2" : "=&v"(nr) : "v"(lane * 4), "s"(rs_) : "memory");
            }
            // [query block A/B][doc block][query half]
            f32x16 a00, a01, a10, a11, b00, b01, b10, b11;
            s1_acc_init(p, ck, t, w, a00, a10);
            a01 = b00 = b01 = a00;
            a11 = b10 = b11 = a10;
            for (int s0 = 0; s0 < p.n_steps; s0 += R) {
#pragma unroll
                for (int u = 0; u < R; ++u) {
                    asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(dr0[u]), "+v"(dr1[u]) : "n"((R - 2) * X::LOADS) : "memory");
                    const char* curq = qring + u * X::Q_STAGE;
                    const vec8 d0 = dr0[u], d1 = dr1[u];
                    if (CV) {
                        // tiles of the stage: query block A, query block B (one fp16 term)
                        const f16x8 qa0 = *(const f16x8*)(curq + off), qa1 = *(const f16x8*)(curq + 1024 + off);
                        const f16x8 qb0 = *(const f16x8*)(curq + 2048 + off), qb1 = *(const f16x8*)(curq + 3072 + off);
                        S1BW_ISSUE((u + R - 1) % R);
                        const f16x8 e0 = CV == 2 ? __builtin_bit_cast(f16x8, d0) : s1_cvt_granule(__builtin_bit_cast(u32x4, d0), tmin2, bias2);
                        a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa0, a00, 0, 0, 0);
                        a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa1, a01, 0, 0, 0);
                        const f16x8 e1 = CV == 2 ? __builtin_bit_cast(f16x8, d1) : s1_cvt_granule(__builtin_bit_cast(u32x4, d1), tmin2, bias2);
                        b00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb0, b00, 0, 0, 0);
                        b01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb1, b01, 0, 0, 0);
                        a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa0, a10, 0, 0, 0);
                        a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa1, a11, 0, 0, 0);
                        b10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb0, b10, 0, 0, 0);
                        b11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb1, b11, 0, 0, 0);
                        continue;
                    }
                    // tiles of the stage: hi / block A, hi / block B, mid / block A, mid / block B
                    const bf16x8 ma0 = *(const bf16x8*)(curq + 4096 + off), ma1 = *(const bf16x8*)(curq + 5120 + off);
                    const bf16x8 mb0 = *(const bf16x8*)(curq + 6144 + off), mb1 = *(const bf16x8*)(curq + 7168 + off);
                    const bf16x8 ha0 = *(const bf16x8*)(curq + off), ha1 = *(const bf16x8*)(curq + 1024 + off);
                    const bf16x8 hb0 = *(const bf16x8*)(curq + 2048 + off), hb1 = *(const bf16x8*)(curq + 3072 + off);
                    S1BW_ISSUE((u + R - 1) % R);
                    // smallest term first (all products are exact; this keeps the fp32 accumulation tight)
                    a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ma0, a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ma1, a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ma0, a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ma1, a11, 0, 0, 0);
                    b00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, mb0, b00, 0, 0, 0);
                    b01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, mb1, b01, 0, 0, 0);
                    b10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, mb0, b10, 0, 0, 0);
                    b11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, mb1, b11, 0, 0, 0);
                    a00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ha0, a00, 0, 0, 0);
                    a01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, ha1, a01, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ha0, a10, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, ha1, a11, 0, 0, 0);
                    b00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, hb0, b00, 0, 0, 0);
                    b01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, hb1, b01, 0, 0, 0);
                    b10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, hb0, b10, 0, 0, 0);
                    b11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, hb1, b11, 0, 0, 0);
                }
            }
            if (CV == 2 && p.dump && !p.sample) {
                // score dump (S1Params::dump): row (32 db + (r & 3) + 8 (r >> 2) + 4 h) of the wave's block, query column 64 blk + 32 x + j.
                // Two rows of one query column share a dword (registers r, r + 1 of an accumulator: rows 2 m, 2 m + 1), as 16-bit
                // signed-normalised codes of a / B (v_cvt_pknorm_i16_f32).  One SGPR base per (doc block, r >> 2), one per-lane offset, the
                // rest in the immediate; a store instruction covers two row PAIRS x 32 consecutive queries (two full 128-byte lines).
                // Issued BEFORE the selection epilogue so that the stores have left the vmcnt queue by the time the next tile's counted
                // waits look at it.
                const u32 voff = (u32)((lane >> 5) * 2 * 512 + (lane & 31) * 4);
                const char* const dtile = (const char*)p.dump + ((((size_t)p.dump_base[f] + (size_t)t * S1_TILE_ROWS + (size_t)w * 64)) >> 1) * 512;
#define S1W_DUMP4(ACC, DB, Q0, INV)                                                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                                          \
        const char* sb_ = dtile + (size_t)(16 * (DB) + 4 * g_) * 512;                                                           \
        const u32 c0_ = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pknorm_i16(ACC[4 * g_ + 0] * (INV), ACC[4 * g_ + 1] * (INV))); \
        const u32 c1_ = __builtin_bit_cast(u32, __builtin_amdgcn_cvt_pknorm_i16(ACC[4 * g_ + 2] * (INV), ACC[4 * g_ + 3] * (INV))); \
        asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(c0_), "s"(sb_), "n"((Q0)) : "memory");           \
        asm volatile("global_store_dword %0, %1, %2 offset:%3" ::"v"(voff), "v"(c1_), "s"(sb_), "n"((Q0) + 512) : "memory");     \
    }
                {
                    const float i0_ = dinv_s[lane & 31];
                    S1W_DUMP4(a00, 0, 0, i0_)
                    S1W_DUMP4(a10, 1, 0, i0_)
                }
                {
                    const float i1_ = dinv_s[32 + (lane & 31)];
                    S1W_DUMP4(a01, 0, 128, i1_)
                    S1W_DUMP4(a11, 1, 128, i1_)
                }
                {
                    const float i2_ = dinv_s[64 + (lane & 31)];
                    S1W_DUMP4(b00, 0, 256, i2_)
                    S1W_DUMP4(b10, 1, 256, i2_)
                }
                {
                    const float i3_ = dinv_s[96 + (lane & 31)];
                    S1W_DUMP4(b01, 0, 384, i3_)
                    S1W_DUMP4(b11, 1, 384, i3_)
                }
#undef S1W_DUMP4
            }
            if (row_on) {      // (after the dump, which holds the plain approximate scores)
                asm volatile("" : "+v"(nr)::"memory");
                s1_row_norms_park(rarea, nr);
                s1_row_bound(rarea, rA0, rA1, a00, a01, a10, a11);
                s1_row_bound(rarea, rB0, rB1, b00, b01, b10, b11);
            }
            if (p.dbg & 1) {
                asm volatile("" ::"v"(a00), "v"(a01), "v"(a10), "v"(a11), "v"(b00), "v"(b01), "v"(b10), "v"(b11));
                if (!dyn) continue;
            } else if (p.sample == 2) {
                s1_sample_top2(p, ck, t - cur.t0, t, w, a00, a01, a10, a11, 0);
                s1_sample_top2(p, ck, t - cur.t0, t, w, b00, b01, b10, b11, 64);
                continue;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier A (see the fp32 body)
            if (dyn && t == cur.t0 && tid == 0) {         // the claim issued a whole tile ago is back: publish it (read after barrier B)
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(raw) : "n"((R - 1) * X::LOADS) : "memory");
                if (raw == -2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw)::"memory");
                lds_write_b64((u32)(uintptr_t)unit_slot, (u64)(u32)raw);
            }
            if (!(p.dbg & 1)) {
                s1_epilogue_append<X::SCAP>(p, stA, ck.n_rows, t, w, wgq0, a00, a01, a10, a11);
                s1_epilogue_append<X::SCAP>(p, stB, ck.n_rows, t, w, wgq0 + 64, b00, b01, b10, b11);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // barrier B for both blocks; the accumulators are dead from here
            if (dyn && t == cur.t0) nxt = s1_unit_of(__builtin_amdgcn_readfirstlane((int)(u32)lds_read_b64((u32)(uintptr_t)unit_slot)), p.unit_tiles, n_tiles_f);
            if (!(p.dbg & 1)) {
                s1_epilogue_finish<X::SCAP>(p, stA, w, wgq0);
                s1_epilogue_finish<X::SCAP>(p, stB, w, wgq0 + 64);
            }
        }
        cur = nxt;
        nxt.t0 = nxt.t1 = -1;
        pf_in_next = 0;
#ifdef MFAR_TRACE
        ++tr_units;
#endif
    }
#ifdef MFAR_TRACE
    if (tid == 0 && !p.sample) trace_put(1, (int)blockIdx.x, tr_units, tr_t0);
#endif
#undef S1BW_ISSUE
#undef S1BW_QDMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.sample != 2) {
        s1_flush<X::SCAP>(p, stA, w, wgq0);
        s1_flush<X::SCAP>(p, stB, w, wgq0 + 64);
    }
}

// The exact passes (fp32, bf16) double as REPAIR passes of the certified screen (only_failed): their table is then cut finely
// (every field into up to a whole wave of chunks, so that a single failed field is scanned by the whole GPU), but launched as
// ONE wave of workgroups that walk it -- the workgroups of fields that did not fail would otherwise cost more to dispatch
// than the repair itself (idle repair launches follow every screened batch of the non-pipelined entry points).
#define S1_COMMA ,
#define S1_CHUNK_LOOP(BODY)                                                                   \
    for (int c_ = (int)blockIdx.x; c_ < p.n_launch; c_ += (int)gridDim.x) {                   \
        BODY(p, __builtin_amdgcn_readfirstlane(p.chunk0 + c_));                               \
        __syncthreads();                                                                      \
    }
// The full pass and the threshold-estimation pass are the same code under two kernel names, so that profiles list them
// separately (the sample pass scans 1 tile per workgroup and is ~30x shorter).
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f32r_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32r<6>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f32r_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32r<6>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f32r4_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32r<4>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f32r4_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_f32r<4>) }
#define S1FR_LDS_BYTES (6 * 4096 + S1_STATE_BYTES_(S1_SCAP_F32R))
#define S1FR4_LDS_BYTES (4 * 4096 + S1_STATE_BYTES_(S1_SCAP_F32R))
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16<0>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16<0>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16_kernel(const S1Params p) { s1_body_x16<1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16_sample_kernel(const S1Params p) { s1_body_x16<1>(p, p.chunk0 + (int)blockIdx.x); }
#ifndef S1HR_R
#define S1HR_R 6
#endif
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16r_kernel(const S1Params p) { s1_body_x16r<1, S1HR_R>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16r_sample_kernel(const S1Params p) { s1_body_x16r<1, S1HR_R>(p, p.chunk0 + (int)blockIdx.x); }
#define S1HR_LDS_BYTES (S1HR_R * 4096 + S1_STATE_BYTES_(S1_SCAP_REG))
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16r_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16r<0 S1_COMMA  6>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16r_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16r<0 S1_COMMA  6>) }
#define S1BR_LDS_BYTES (6 * 6144 + S1_STATE_BYTES_(S1_SCAP_REG / 2))
// 4-slot twins for dims whose k-steps divide by 4 but not by 6 (64, 128, 256, 512, 1024, ...): measured equal to the 6-slot
// ring; without them those dims would fall to the LDS ring, beside which nothing else fits on a CU
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16r4_kernel(const S1Params p) { s1_body_x16r<1, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16r4_sample_kernel(const S1Params p) { s1_body_x16r<1, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16r4_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16r<0 S1_COMMA  4>) }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16r4_sample_kernel(const S1Params p) { S1_CHUNK_LOOP(s1_body_x16r<0 S1_COMMA  4>) }
#define S1HR4_LDS_BYTES (4 * 4096 + S1_STATE_BYTES_(S1_SCAP_REG))
#define S1BR4_LDS_BYTES (4 * 6144 + S1_STATE_BYTES_(S1_SCAP_REG / 2))
// wide fp16 screen pass (128 queries, one fp16 term; CV = 2: the granules are fp16 already): 6-slot ring for k-steps divisible by 6,
// 4-slot twin otherwise
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w_kernel(const S1Params p) { s1_body_wide<6, S1_SCAP_WIDE, 2>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w_sample_kernel(const S1Params p) { s1_body_wide<6, S1_SCAP_WIDE, 2>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w4_kernel(const S1Params p) { s1_body_wide<4, S1_SCAP_WIDE, 2>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w4_sample_kernel(const S1Params p) { s1_body_wide<4, S1_SCAP_WIDE, 2>(p, p.chunk0 + (int)blockIdx.x); }
// ... and their ROW MODE twins (indexes with a heavy-tailed field)
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w_rm_kernel(const S1Params p) { s1_body_wide<6, S1_SCAP_WIDE, 2, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w_rm_sample_kernel(const S1Params p) { s1_body_wide<6, S1_SCAP_WIDE, 2, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w4_rm_kernel(const S1Params p) { s1_body_wide<4, S1_SCAP_WIDE, 2, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_f16w4_rm_sample_kernel(const S1Params p) { s1_body_wide<4, S1_SCAP_WIDE, 2, 1>(p, p.chunk0 + (int)blockIdx.x); }
#define S1HW_LDS_BYTES (6 * 4096 + 2 * S1_STATE_BYTES_(S1_SCAP_WIDE) + 512)      // + the score dump's 128 factors
// certified passes over a bf16 slab: 64 columns x two bf16 terms (register ring of 6 / 4 slots), 128 columns x two terms
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16s_kernel(const S1Params p) { s1_body_x16r<2, 6>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16s_sample_kernel(const S1Params p) { s1_body_x16r<2, 6>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16s4_kernel(const S1Params p) { s1_body_x16r<2, 4>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16s4_sample_kernel(const S1Params p) { s1_body_x16r<2, 4>(p, p.chunk0 + (int)blockIdx.x); }
// LDS per workgroup decides more than occupancy here.  LDS is allocated first-fit and contiguously: when a scan workgroup is placed
// while a small workgroup of the previous launch's tail still sits at the bottom of the CU's LDS (a row-gather workgroup holds
// 35 KB), the free space behind it is split in two once that workgroup leaves, and the CU's SECOND scan workgroup only fits if one
// fragment holds it: 160 - 35 - 2 x LDS >= 0.  At 66.6 KB (6-slot ring, 16-entry staging) 40 % of the CUs ran ONE scan workgroup
// for a whole scan (1.25 M x 16: 7.2 ms alone, 11.5 ms in the pipeline; workgroup trace in profiles/r04_a_lds_fragmentation.txt);
// at 50.5 KB (4-slot ring, measured equal alone) the pipeline runs the scan in 7.5 ms.  Keep every scan kernel under ~60 KB.
#ifndef S1BW_SCAP6
#define S1BW_SCAP6 8                         // 6-slot query ring (48 KB): dims whose k-steps do not divide by 4
#endif
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16w_kernel(const S1Params p) { s1_body_wide<6, S1BW_SCAP6, 0>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16w_sample_kernel(const S1Params p) { s1_body_wide<6, S1BW_SCAP6, 0>(p, p.chunk0 + (int)blockIdx.x); }
#ifndef S1BW_SCAP4
#define S1BW_SCAP4 16
#endif
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16w4_kernel(const S1Params p) { s1_body_wide<4, S1BW_SCAP4, 0>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16w4_sample_kernel(const S1Params p) { s1_body_wide<4, S1BW_SCAP4, 0>(p, p.chunk0 + (int)blockIdx.x); }
// converted docs x one fp16 query term (CV = 1): 6-slot ring (24 KB) or 4-slot twin, 16-entry staging areas: 42.5 / 34.5 KB of LDS
#define S1BC_SCAP 16
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16c_kernel(const S1Params p) { s1_body_wide<6, S1BC_SCAP, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16c_sample_kernel(const S1Params p) { s1_body_wide<6, S1BC_SCAP, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16c4_kernel(const S1Params p) { s1_body_wide<4, S1BC_SCAP, 1>(p, p.chunk0 + (int)blockIdx.x); }
__global__ void __launch_bounds__(S1_THREADS, 2) mfar_stage1_bf16c4_sample_kernel(const S1Params p) { s1_body_wide<4, S1BC_SCAP, 1>(p, p.chunk0 + (int)blockIdx.x); }
#define S1BC_LDS_BYTES (6 * 4096 + 2 * S1_STATE_BYTES_(S1BC_SCAP))
#define S1BC4_LDS_BYTES (4 * 4096 + 2 * S1_STATE_BYTES_(S1BC_SCAP))
#define S1BW_LDS_BYTES (6 * 8192 + 2 * S1_STATE_BYTES_(S1BW_SCAP6))
#define S1BW4_LDS_BYTES (4 * 8192 + 2 * S1_STATE_BYTES_(S1BW_SCAP4))
#define S1HW4_LDS_BYTES (4 * 4096 + 2 * S1_STATE_BYTES_(S1_SCAP_WIDE) + 512)
#define S1HW_RM_LDS_BYTES (S1HW_LDS_BYTES + 1024)
#define S1HW4_RM_LDS_BYTES (S1HW4_LDS_BYTES + 1024)
