// mfar_hip.hip -- host side of libmfar_hip.so: the C ABI declared in include/mfar_hip.h.
// gfx950 only.  Every entry point returns an error code; nothing throws across the ABI.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>   // device radix sort + prefix sums of the unique-row build (index construction, not the hot path)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "mfar_hip.h"
#include "mfar_policy.h"
#include "mfar_select.h"
#include "mfar_screen.h"
#include "mfar_exact16.h"

#define MFAR_VERSION MFAR_ABI_VERSION   /* include/mfar_hip.h: bumped on EVERY signature change; mfar/_native.py refuses any other value */
#define PAYLOAD_MAGIC 0x6d464152 /* "mFAR" */

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (void)hipGetLastError(); /* reported HERE: not left behind for the next call's launch check */ \
            return fail(MFAR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
        }                                                                                             \
    } while (0)
#define RETCHK(expr)            \
    do {                        \
        int rc_ = (expr);       \
        if (rc_ != MFAR_OK) return rc_; \
    } while (0)

// Test hook (mfar_debug_fail_allocations_above): DevBuf allocations of at least this many bytes fail exactly like a hipMalloc that ran out of
// device memory -- the NOMEM paths can then be driven deterministically (a really full HBM is a moving target: the HIP runtime gives
// cached resources back when an allocation fails, so whether a given call fits depends on the process's history).  0 = off.
static std::atomic<long long> g_fail_alloc_above{0};
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // exact: no growth slack (the big resident buffers of an index -- screen slab, gather slab, unique-row tables -- are sized once)
    int ensure(size_t bytes, bool exact = false) {
        if (bytes <= cap) return MFAR_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = exact ? ((bytes + 255) & ~(size_t)255) : bytes + (bytes >> 3) + 256;
        const long long lim = g_fail_alloc_above.load(std::memory_order_relaxed);
        hipError_t e = (lim > 0 && (long long)want >= lim) ? hipErrorOutOfMemory : hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();      // the failed allocation is REPORTED here; left sticky it would fail the caller's next launch check
            return fail(MFAR_ERR_NOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
        }
        cap = want;
        return MFAR_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return (T*)p; }
};

// scratch shared by the handle-less entry points (mfar_mix_topk, mfar_merge_payloads): one per device
struct DevCtx {
    std::mutex mu;
    DevBuf in[8], out[4], lists_ids, lists_sc, cand, ncand, x;
};
static DevCtx g_ctx[16];
static bool g_attr_done[16] = {false};

// Geometry of one scanned slab + its chunk tables (mfar_stage1.h: S1Chunk).  `all`: every field gets a share of the grid
// proportional to its tiles (one launch scans all fields); `solo`: every field is cut as finely as the list merge allows
// (a launch scans one field, mfar_retrieve_field).
struct S1Table : S1TableHost {       // mfar_tables.h + the device copies
    DevBuf d_chunks, d_fchunk, d_samp_n;
    DevBuf d_gchunk, d_fgroup, d_gfield;
    u32 skip = 0;                    // built without these fields (AUTO-OFF: they get no chunks)
};
struct S1Geom : S1GeomHost {
    S1Table all, solo;
    S1Table all_w, solo_w;   // the wide pass keeps its own tables (same rule today; its list depth k' may differ from a 64-column pass in flight)
    S1Table all_skip, all_w_skip;   // the all-fields tables WITHOUT the fields that are switched off (rebuilt when that set changes; the
                                    // full tables stay for the probe launches)
    void reset(int F) {
        n_rows.assign(F, 0);
        base.assign(F, 0);
        n_tiles.assign(F, 0);
        all.k = solo.k = all_w.k = solo_w.k = all_skip.k = all_w_skip.k = -1;
    }
};

// Pipeline slots: independent sets of per-launch scratch (stage-1 lists, thresholds, candidate tables, ...).  A caller keeps as many
// launches in flight as it uses slots (mfar.data.pipeline: 2 by default, MFAR_PIPE_DEPTH up to MFAR_SLOTS).
#define MFAR_SLOTS 4
struct mfar_index {
    int device = 0;
    int64_t n_rows = 0, row_offset = 0;
    int F = 0, E = 0, dtype = 0;
    int64_t n_blk = 0;  // 64-row blocks per field, multiple of 4
    int n_steps = 0;
    void* slab = nullptr;         // fp32 or bf16 tiled slab
    int esize = 4;                // bytes per stored element
    size_t slab_bytes = 0;
    long long field_stride = 0;   // elements between fields
    int n_cu = 256;
    int wgs_per_cu = 2;
    // stage-1 scratch, two slots: the pipelined caller finishes batch i (merge, re-score, certify) on one stream while
    // batch i+1 scans on another (mfar_stage1_begin / mfar_stage1_finish)
    struct S1Slot {
        DevBuf qt, lists, list_cnt, gtau, samp, lists2, list_cnt2;      // any pass (lists2: group lists of a two-level merge)
        DevBuf unit_ctr;                                                // per-field unit counters of a dynamically distributed scan (mfar_stage1.h)
        DevBuf dump;                                                    // score dump of the wide screened pass (mfar_select.h mfar_s2_lookup_kernel)
        bool row_mode = false;                                          // ROW MODE decided by the begin phase of the batch, with the fields it covers
        u32 row_mask = 0;                                               // (latched: the index's mask may change while the batch is in flight)
        bool dump_on = false;                                           // this batch's scan writes it
        bool dump_ready = false;                                        // ... and has been launched: stage 2 of (dump_q, dump_Q) may read it, once
        const float* dump_q = nullptr;
        int dump_Q = 0;
        DevBuf qt16, qinfo, eps, base, fail, sids, ssc, scnt, sx;        // fp16 screen
        DevBuf arow, eps_cert;                                          // ... ROW MODE: per (field, query) factor of the row norm / what is left of eps
        DevBuf dinv, dstep, eps_dump, darel;                            // ... SCORE DUMP: 1 / B, B / 32767, the approximate level's bound per (field, query), its row-norm share
        bool screened = false;                                          // decided by the begin phase of the batch
        int qw = 64;                                                    // query columns of the batch's pass (128: wide screen pass)
        // AUTO-OFF, latched by the begin phase: fields whose lists the exact pass writes in this batch / fields its screen leaves out
        // (equal, except in a probe launch, which screens everything)
        u32 exact_mask = 0, skip_mask = 0;
        DevBuf off_flags;                                               // [MFAR_MAX_FIELDS] device copy of exact_mask, one int per field
        DevBuf chain;                                                   // bf16 index: scores of the exhaustive chain pass (mfar_exact16.h), [CHAIN_QB][rows]
        DevBuf tau2, lfail, t2cand, t2cnt, t2sx;                        // TIER 2 (mfar_screen.h): thresholds [F, qw], per-list flags, candidate sets [qw, F, T2_CAP]
        bool t2 = false;                                                // this batch runs tier 2 behind its certificate (latched by the begin phase)
        bool t2_rescan = false;                                         // ... with the rescan enqueued behind collect pass A (latched too)
        const float* scan_tau = nullptr;                                // thresholds [F, qw] the batch's screened scan ran with (nullptr: none) -- tier 2's
                                                                        // collect kernel checks them against its own before it trusts that scan's chunk lists
        u32 deep_mask = 0;                                              // DEEP SCAN fields of this batch (latched by the begin phase; mfar_screen.h)
        u32 fb_deep = 0;                                                // ... as reported with the batch's feedback
        DevBuf deepinfo;                                                // [F, qw] float4 {band, eps, exact 0} in scan units (mfar_sample_tau_kernel)
        // feedback: the batch's certificate flags, copied to pinned host memory behind the certify kernel and read by a LATER call once
        // the event has completed (never waited for)
        int* fb_host = nullptr;                                         // [SCREEN_FLAGS] pinned
        hipEvent_t fb_ev = nullptr;
        bool fb_pending = false;
        u32 fb_screened = 0, fb_probed = 0;                             // fields screened for real / as a probe in that batch
    } s1[MFAR_SLOTS];
    DevBuf fid, fsc, cand[MFAR_SLOTS], ncand[MFAR_SLOTS], x[MFAR_SLOTS], own[MFAR_SLOTS], in[8], out[8];
    // 16-bit GATHER slab (mfar_select.h mfar_score_rows_kernel): row-major [F][n_rows] rows of g_row_bytes.
    //   fp32 index: fp16 of the centred + scaled rows (built with the screen: same mean / scale) -> the approximate level of the
    //               certified two-level stage 2;   bf16 index: the slab's values bit for bit -> every row gather reads whole lines
    DevBuf gslab;
    size_t g_row_bytes = 0;
    bool gslab_ok = false;        // the gather slab describes the rows as they are now
    bool gslab_nomem = false;     // it could not be allocated: gathers stay on the scan-ordered slab
    bool rows16_dirty = true;     // bf16 index: rows were written since the companion was filled
    int stage2_mode = 1;          // 0 = gather every (candidate, field) row from the fp32 slab; 1 = certified two-level stage 2 when available
    bool s2_fused = true;         // the tail as the round-6 kernels (gate / front / bounds / select: mfar_select.h); false = the kernels of rounds 3-5
    DevBuf xa[MFAR_SLOTS], cand2[MFAR_SLOTS], ncand2[MFAR_SLOTS], s2qm[MFAR_SLOTS], s2eps[MFAR_SLOTS], s2stats;   // two-level stage 2 scratch (per pipeline slot) + counters
    DevBuf kmask[MFAR_SLOTS], src2[MFAR_SLOTS];                      // ... known pairs (stage-1 scores reused), survivor -> candidate index
    DevBuf xe[MFAR_SLOTS];                                           // ... per-pair bounds of the score dump's level (row norms)
    DevBuf s2wgt[MFAR_SLOTS], lbub[MFAR_SLOTS];                      // field weights of the batch (mfar_s2_gate_kernel), interval ends (mfar_s2_bounds_kernel)
    struct S2Pre {                                                   // what the gate kernel last computed for the slot (run_stage2_pre)
        const float* q = nullptr;
        const float* W = nullptr;
        int Q = 0;
        bool approx = false;                                         // ... with q . mean and eps of the approximate level
    } s2pre[MFAR_SLOTS];
    // certified fp16 screen of an fp32 index (mfar_screen.h)
    int screen_mode = 1;          // 0 off, 1 auto, 2 always (when the shapes allow)
    float screen_eps_mult = 1.0f; // test knob: scales the certificate's error bound
    DevBuf screen;                // fp16 tiled slab of the fields' UNIQUE rows (per-field bases: geom_screen)
    size_t screen_used = 0;       // bytes of it in use
    bool screen_dirty = true;     // rows were written since the screen was built
    bool screen_nomem = false;    // the screen slab could not be allocated: stay on the exact pass
    long long screen_checked = 0; // (query, field) lists certified so far
    DevBuf s_stats, s_field, s_mean;
    DevBuf dump_base;             // [F] first row of every field in the screen slab (= in a score dump, = in s_rnorm)
    DevBuf s_rnorm, s_nsum;       // ROW MODE (mfar_screen.h): centred 2-norm of every row of the screen slab; per-field sum of the norms
    u32 row_mask = 0;             // fields whose scans run in row mode now
    u32 row_eligible = 0;         // fields with heavy-tailed row norms (host copy of ScreenField::row_mode)
    int row_mode_setting = 1;     // 0 never, 1 auto: eligible fields are activated once a certificate has failed (mfar_row_mode_activate; the
                                  // pipelined searcher calls it), 2 always.  MFAR_SCREEN_ROW_MODE
    int deep_mode = 0;            // DEEP SCAN of fields whose first certificates keep failing (mfar_screen.h): 0 never, 1 auto (policy), 2 every
                                  // field always (tests / experiments).  MFAR_SCREEN_DEEP
    bool t2_force_rescan = false; // diagnostic (mfar_set_tier2 mode + 4): tier 2 never trusts the launch's own chunk lists
    int tier2_mode = 1;           // TIER 2 of the certified screen (mfar_screen.h "threshold rescan"): 0 never, 1 auto (its kernels are
                                  // enqueued while the policy has seen a failed certificate recently), 2 always.  MFAR_SCREEN_TIER2
    int dump_mode = 1;            // 0 never, 1 when it moves fewer bytes than the row gathers (dump_wanted), 2 whenever possible
    long long dump_launches = 0;
    DevBuf s_field1, s_cvt;       // bf16 index: ScreenField of the two-term passes (scale 1); conversion constants of the converted-docs pass
    // unique rows of every field (mfar_screen.h), each table [F][n_rows] (stride n_rows): representative document of a
    // unique row, start / length of its member run in `members` (local rows grouped by unique row, ascending inside a group)
    DevBuf u_rep, u_start, u_count, u_members, u_n;
    DevBuf u_repof;               // [F][n_rows] representative of every row's group (stage 2 gathers it in the row's place); optional
    // bf16 index: the certified pass scans the slab itself (no screen slab) and ranks unique rows through these (mfar_screen.h)
    DevBuf rep_bits;              // [F][n_blk] u64: row is real and the representative of its group
    DevBuf u_of;                  // [F][n_rows] u32: unique number + 1 of the row's group (every row, when u_repof exists; else representatives only)
    bool uof_packed = false;      // fp32 index: the entries carry a 10-bit row-norm code in their top bits (mfar_uof_norm_code_kernel)
    DevBuf u_of_t;                // fp32 index, optional: the same entries transposed, [n_rows][F] (mfar_s2_lookup_bounds_kernel: a document's F
    bool uof_t_ok = false;        // entries in two sectors instead of F)
    bool screen_built = false;    // statistics + unique-row tables (+ the fp16 screen slab of an fp32 index) were built at least once
    std::vector<int> n_unique, largest_group;   // per field (host copies)
    bool screen_dedup = true;     // MFAR_SCREEN_DEDUP=0: every document is its own unique row (diagnostic)
    bool wide = true;             // blocks of 65 .. 128 queries go through the wide screened pass (MFAR_WIDE=0: always 64 per pass)
    bool repair_sample = false;   // mfar_set_repair_mode: repairs run their own sample pass (see stage1_pass)
    ScreenPolicy pol;             // AUTO-OFF, inline repair (mfar_policy.h): fed with the certificate flags of finished launches
    S1Geom geom_docs, geom_screen;
    // fused mode (mfar_search_fused): a one-field companion index of dim F * E over the same rows, built on first use
    mfar_index* fused = nullptr;
    bool fused_dirty = true;      // rows were written since the companion was filled
    DevBuf fused_q;               // folded queries [Q, F * E]
    hipEvent_t mid_ev = nullptr;  // recorded right before the full stage-1 kernel is launched
    const char* last_s1_kernel = "";   // the scan kernel the last timed stage-1 launch ran (mfar_last_stage1_kernel)
    bool timing = false;
    std::vector<hipEvent_t> ev;  // start/stop pairs
    int ev_n = 0;
    // launches the pipelines over this index have enqueued on the library's streams / how many of them the last slab write was ordered
    // behind (order_after_pipelines)
    long long pipe_launches = 0, pipe_ordered = 0;
    hipEvent_t order_ev[3] = {nullptr, nullptr, nullptr};
    // ... and the other direction: behind the last slab write (on the caller's stream), waited for by the next reader on any other stream
    hipEvent_t write_ev = nullptr;
    hipStream_t write_stream = nullptr;
    bool write_pending = false;
};

static bool s2_fused_default();
static std::mutex g_attr_mu;               // (handles may be created from different host threads)
static int set_kernel_attrs(int device) {
    if (device < 0 || device >= 16) return fail(MFAR_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (g_attr_done[device]) return MFAR_OK;
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f32r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1FR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f32r_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1FR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f32r4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1FR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f32r4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1FR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_lists_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_lists_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_lists_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_topk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_topk_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_shards_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_merge_shards_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_score_rows_kernel<SRC_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_score_rows_kernel<SRC_F16G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_score_rows_kernel<SRC_BF16G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_s2_prune_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_t2_collect_deep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // (these two also hold a little static LDS: the dynamic part must leave room for it)
    HIPCHK(hipFuncSetAttribute((const void*)mfar_s2_gate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S2_DYN_LDS_MAX));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_s2_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S2_DYN_LDS_MAX));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_s2_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_score_candidates_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1B_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1B_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_mix_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_fold_queries_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1H_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1H_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16r_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16r_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16r_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16r4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16r4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16r4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16r4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w_rm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW_RM_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w_rm_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW_RM_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w4_rm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW4_RM_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_f16w4_rm_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HW4_RM_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16s_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16s4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16s4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1HR4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BW_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16w_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BW_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16c_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BC_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16c_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BC_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16c4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BC4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16c4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BC4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BW4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_stage1_bf16w4_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S1BW4_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_chain_scan_bf16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_chain_scan_bf16_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)mfar_chain_scan_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    g_attr_done[device] = true;
    return MFAR_OK;
}

#ifdef MFAR_TRACE
extern "C" int mfar_trace_dump(void* host, int max_rec) {   // experiment builds only
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_trace_n), 4) != hipSuccess) return -1;
    n = std::min(std::min(n, max_rec), 1 << 17);
    if (n > 0 && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), (size_t)n * sizeof(TraceRec)) != hipSuccess) return -1;
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_n), &zero, 4);
    return n;
}
#endif
extern "C" int mfar_version(void) { return MFAR_VERSION; }
extern "C" int mfar_debug_fail_allocations_above(int64_t bytes) {
    if (bytes < 0) return fail(MFAR_ERR_INVALID, "bytes must be >= 0 (0 = off)");
    g_fail_alloc_above.store((long long)bytes);
    return MFAR_OK;
}
extern "C" const char* mfar_last_error(void) { return g_err.c_str(); }
extern "C" int mfar_device_count(int* n_out) {
    if (!n_out) return fail(MFAR_ERR_INVALID, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *n_out = 0;
        return fail(MFAR_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *n_out = n;
    return MFAR_OK;
}

extern "C" int mfar_index_create(mfar_index** out, int device, int64_t n_rows_local, int64_t row_offset, int n_fields,
                                 int dim, int dtype) {
    if (!out) return fail(MFAR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (n_rows_local < 0 || row_offset < 0) return fail(MFAR_ERR_INVALID, "negative row count / offset");
    if (row_offset + n_rows_local >= 0xFFFFFFFELL) return fail(MFAR_ERR_INVALID, "doc ids must fit in 32 bits");
    if (n_rows_local > 0x7FFFFC00LL) return fail(MFAR_ERR_INVALID, "at most 2^31 - 1024 rows per shard");
    if (n_fields <= 0 || n_fields > MFAR_MAX_FIELDS) return fail(MFAR_ERR_INVALID, "n_fields must be in [1, 32]");
    if (dim <= 0 || (dim & 31)) return fail(MFAR_ERR_INVALID, "dim must be a positive multiple of 32");
    if (dtype != MFAR_DTYPE_F32 && dtype != MFAR_DTYPE_BF16) return fail(MFAR_ERR_INVALID, "unknown dtype");
    int ndev = 0;
    RETCHK(mfar_device_count(&ndev));
    if (device < 0 || device >= ndev) return fail(MFAR_ERR_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    RETCHK(set_kernel_attrs(device));
    mfar_index* idx = new (std::nothrow) mfar_index();
    if (!idx) return fail(MFAR_ERR_NOMEM, "host allocation failed");
    idx->device = device;
    idx->n_rows = n_rows_local;
    idx->row_offset = row_offset;
    idx->F = n_fields;
    idx->E = dim;
    idx->dtype = dtype;
    idx->n_steps = dim / 16;
    int64_t n_blk = (n_rows_local + 63) / 64;
    n_blk = ((n_blk + 3) / 4) * 4;
    if (n_blk == 0) n_blk = 4;
    idx->n_blk = n_blk;
    idx->field_stride = (long long)n_blk * 64 * dim;
    idx->esize = dtype == MFAR_DTYPE_BF16 ? 2 : 4;
    idx->slab_bytes = (size_t)idx->field_stride * idx->esize * n_fields;
    if (const char* e = getenv("MFAR_SCREEN")) idx->screen_mode = atoi(e);
    if (const char* e = getenv("MFAR_SCREEN_EPS_MULT")) idx->screen_eps_mult = (float)atof(e);
    if (const char* e = getenv("MFAR_SCREEN_DEDUP")) idx->screen_dedup = atoi(e) != 0;
    if (const char* e = getenv("MFAR_WIDE")) idx->wide = atoi(e) != 0;
    if (const char* e = getenv("MFAR_STAGE2_PRUNE")) idx->stage2_mode = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("MFAR_S2_DUMP")) idx->dump_mode = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("MFAR_SCREEN_ROW_MODE")) idx->row_mode_setting = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("MFAR_SCREEN_AUTO_OFF")) idx->pol.set_mode(atoi(e) != 0 ? 1 : 0);
    if (const char* e = getenv("MFAR_SCREEN_TIER2")) idx->tier2_mode = std::max(0, std::min(2, atoi(e)));
    if (const char* e = getenv("MFAR_SCREEN_DEEP")) idx->deep_mode = std::max(0, std::min(2, atoi(e)));
    idx->pol.deep_mode = idx->deep_mode ? 1 : 0;
    idx->s2_fused = s2_fused_default();
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) idx->n_cu = prop.multiProcessorCount;
    hipError_t e = hipMalloc(&idx->slab, idx->slab_bytes);
    if (e != hipSuccess) {
        const size_t wanted = idx->slab_bytes;
        delete idx;
        (void)hipGetLastError();
        return fail(MFAR_ERR_NOMEM, std::string("hipMalloc(slab ") + std::to_string(wanted) + " B): " + hipGetErrorString(e));
    }
    e = hipMemset(idx->slab, 0, idx->slab_bytes);
    if (e != hipSuccess) {
        (void)hipFree(idx->slab);
        delete idx;
        return fail(MFAR_ERR_HIP, std::string("hipMemset(slab): ") + hipGetErrorString(e));
    }
    idx->geom_docs.reset(n_fields);
    for (int f = 0; f < n_fields; ++f) {
        idx->geom_docs.n_rows[f] = n_rows_local;
        idx->geom_docs.base[f] = (long long)f * idx->field_stride;
        idx->geom_docs.n_tiles[f] = (int)(n_blk / 4);
    }
    idx->n_unique.assign(n_fields, 0);
    idx->largest_group.assign(n_fields, 0);
    *out = idx;
    return MFAR_OK;
}

extern "C" void mfar_index_destroy(mfar_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    (void)hipDeviceSynchronize();
    if (idx->fused) mfar_index_destroy(idx->fused);
    idx->fused_q.release();
    for (hipEvent_t e : idx->ev) (void)hipEventDestroy(e);
    if (idx->mid_ev) (void)hipEventDestroy(idx->mid_ev);
    for (hipEvent_t e : idx->order_ev)
        if (e) (void)hipEventDestroy(e);
    if (idx->write_ev) (void)hipEventDestroy(idx->write_ev);
    DevBuf* bufs[] = {&idx->fid, &idx->fsc, &idx->s_stats, &idx->s_field, &idx->s_mean, &idx->screen, &idx->u_rep, &idx->u_start,
                      &idx->u_count, &idx->u_members, &idx->u_n, &idx->u_repof, &idx->gslab, &idx->s2stats, &idx->rep_bits, &idx->u_of, &idx->u_of_t, &idx->s_field1, &idx->s_cvt, &idx->dump_base, &idx->s_rnorm, &idx->s_nsum};
    for (int i = 0; i < MFAR_SLOTS; ++i)
        for (DevBuf* b : {&idx->cand[i], &idx->ncand[i], &idx->x[i], &idx->own[i], &idx->xa[i], &idx->cand2[i], &idx->ncand2[i], &idx->s2qm[i],
                          &idx->s2eps[i], &idx->kmask[i], &idx->src2[i], &idx->xe[i], &idx->s2wgt[i], &idx->lbub[i]})
            b->release();
    for (S1Geom* g : {&idx->geom_docs, &idx->geom_screen})
        for (S1Table* t : {&g->all, &g->solo, &g->all_w, &g->solo_w, &g->all_skip, &g->all_w_skip}) {
            t->d_chunks.release();
            t->d_fchunk.release();
            t->d_samp_n.release();
            t->d_gchunk.release();
            t->d_fgroup.release();
            t->d_gfield.release();
        }
    for (DevBuf* b : bufs) b->release();
    for (auto& sl : idx->s1) {
        if (sl.fb_host) (void)hipHostFree(sl.fb_host);
        if (sl.fb_ev) (void)hipEventDestroy(sl.fb_ev);
        sl.off_flags.release();
        sl.chain.release();
        for (DevBuf* b : {&sl.tau2, &sl.lfail, &sl.t2cand, &sl.t2cnt, &sl.t2sx, &sl.deepinfo}) b->release();
        DevBuf* sb[] = {&sl.qt, &sl.lists, &sl.list_cnt, &sl.gtau, &sl.samp, &sl.lists2, &sl.list_cnt2, &sl.unit_ctr, &sl.dump, &sl.arow, &sl.eps_cert, &sl.dinv, &sl.dstep, &sl.eps_dump, &sl.darel, &sl.qt16, &sl.qinfo, &sl.eps, &sl.base, &sl.fail, &sl.sids,
                        &sl.ssc, &sl.scnt, &sl.sx};
        for (DevBuf* b : sb) b->release();
    }
    for (auto& b : idx->in) b.release();
    for (auto& b : idx->out) b.release();
    if (idx->slab) (void)hipFree(idx->slab);
    delete idx;
}

extern "C" int mfar_index_info(const mfar_index* idx, int64_t* n_rows_local, int64_t* row_offset, int* n_fields, int* dim,
                               int* dtype, int64_t* slab_bytes) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (n_rows_local) *n_rows_local = idx->n_rows;
    if (row_offset) *row_offset = idx->row_offset;
    if (n_fields) *n_fields = idx->F;
    if (dim) *dim = idx->E;
    if (dtype) *dtype = idx->dtype;
    if (slab_bytes) *slab_bytes = (int64_t)idx->slab_bytes;
    return MFAR_OK;
}

extern "C" int mfar_index_resident_bytes(const mfar_index* idx, int64_t* rows, int64_t* screen, int64_t* gather, int64_t* tables, int64_t* dumps) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (rows) *rows = (int64_t)idx->slab_bytes;
    if (screen) *screen = (int64_t)idx->screen.cap;
    if (gather) *gather = (int64_t)idx->gslab.cap;
    if (tables) {
        size_t t = 0;
        for (const DevBuf* b : {&idx->u_rep, &idx->u_start, &idx->u_count, &idx->u_members, &idx->u_n, &idx->u_repof, &idx->rep_bits, &idx->u_of, &idx->u_of_t, &idx->s_stats,
                                &idx->s_field, &idx->s_mean, &idx->s_field1, &idx->s_cvt, &idx->s_rnorm, &idx->s_nsum, &idx->dump_base})
            t += b->cap;
        *tables = (int64_t)t;
    }
    if (dumps) {      // the score dumps of the pipeline slots that have used one (released when the shape stops wanting them)
        size_t t = 0;
        for (const auto& sl : idx->s1) t += sl.dump.cap + sl.chain.cap;
        *dumps = (int64_t)t;
    }
    return MFAR_OK;
}

extern "C" int mfar_set_wgs_per_cu(mfar_index* idx, int wgs) {
    if (!idx || wgs < 1 || wgs > 8) return fail(MFAR_ERR_INVALID, "wgs must be in [1, 8]");
    idx->wgs_per_cu = wgs;
    return MFAR_OK;
}

extern "C" int mfar_stream_wait_stage1_start(mfar_index* idx, void* stream) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (!idx->mid_ev) return MFAR_OK;  // no stage-1 launch yet: nothing to wait for
    HIPCHK(hipSetDevice(idx->device));
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, idx->mid_ev, 0));
    return MFAR_OK;
}

extern "C" int mfar_set_timing(mfar_index* idx, int enable) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    idx->timing = enable != 0;
    idx->ev_n = 0;
    return MFAR_OK;
}

extern "C" const char* mfar_last_stage1_kernel(const mfar_index* idx) { return idx ? idx->last_s1_kernel : ""; }
extern "C" int mfar_stage1_timing(mfar_index* idx, double* total_ms_out, int* n_launches_out) {
    if (!idx || !total_ms_out || !n_launches_out) return fail(MFAR_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(idx->device));
    double tot = 0;
    for (int i = 0; i < idx->ev_n; ++i) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(idx->ev[2 * i + 1]));
        HIPCHK(hipEventElapsedTime(&ms, idx->ev[2 * i], idx->ev[2 * i + 1]));
        tot += ms;
    }
    *total_ms_out = tot;
    *n_launches_out = idx->ev_n;
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ the pipelines' streams
struct PipeStreams {            // one set per device and process, shared by every pipeline on it (HIP maps streams onto a few hardware queues
    hipStream_t main = nullptr, side[2] = {nullptr, nullptr}, copy = nullptr;   // round-robin: a second set would share queues with the first)
    bool ok = false;
};
static PipeStreams g_pipe_streams[16];
static std::mutex g_pipe_streams_mu;      // (handles may be created from different host threads)
static int pipe_streams(int device, PipeStreams** out) {
    std::lock_guard<std::mutex> lk(g_pipe_streams_mu);
    PipeStreams& s = g_pipe_streams[device];
    if (!s.ok) {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        hipStream_t made[4] = {nullptr, nullptr, nullptr, nullptr};
        const int prio[4] = {greatest, least, least, least};                              // the scans are dispatched ahead of the small kernels
        for (int i = 0; i < 4; ++i) {
            const hipError_t e = hipStreamCreateWithPriority(&made[i], hipStreamNonBlocking, prio[i]);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                for (int j = 0; j < i; ++j) (void)hipStreamDestroy(made[j]);              // nothing half-made is kept (or leaked)
                return fail(MFAR_ERR_HIP, std::string("hipStreamCreateWithPriority: ") + hipGetErrorString(e));
            }
        }
        s.main = made[0];
        s.side[0] = made[1];
        s.side[1] = made[2];
        s.copy = made[3];
        s.ok = true;
    }
    *out = &s;
    return MFAR_OK;
}
// A slab mutator (mfar_index_write_rows) runs on the CALLER's stream; the launches of a pipeline over this index run on the library's
// non-blocking streams.  Before the first write after such launches the caller's stream is made to wait for everything those streams
// hold: launches submitted before the write read the old rows to the end (scan, exact re-scoring, stage-2 gathers), the write lands
// behind them.  (Launches submitted AFTER the write are ordered by the rebuild of the screen -- a device synchronisation -- or, without
// a screen, by the caller: submit's query copy orders the scan stream behind the caller's stream.)
static int order_after_pipelines(mfar_index* idx, hipStream_t st) {
    if (idx->pipe_launches == idx->pipe_ordered) return MFAR_OK;
    PipeStreams& s = g_pipe_streams[idx->device];
    if (s.ok) {
        hipStream_t src[3] = {s.main, s.side[0], s.side[1]};
        for (int i = 0; i < 3; ++i) {
            if (!idx->order_ev[i]) HIPCHK(hipEventCreateWithFlags(&idx->order_ev[i], hipEventDisableTiming));
            HIPCHK(hipEventRecord(idx->order_ev[i], src[i]));
            HIPCHK(hipStreamWaitEvent(st, idx->order_ev[i], 0));
        }
    }
    idx->pipe_ordered = idx->pipe_launches;
    return MFAR_OK;
}
// ... and readers behind writers: an asynchronous write (device pointers) leaves an event behind it on the writer's stream; whatever reads
// the slab next on ANOTHER stream (a launch on the library's streams, a search on a stream of the caller's) waits for it first.  One event:
// a write on a second stream is ordered behind the earlier one, so that the latest record stands for all of them.
static int mark_written(mfar_index* idx, hipStream_t st) {
    if (!idx->write_ev) HIPCHK(hipEventCreateWithFlags(&idx->write_ev, hipEventDisableTiming));
    if (idx->write_pending && idx->write_stream != st) HIPCHK(hipStreamWaitEvent(st, idx->write_ev, 0));
    HIPCHK(hipEventRecord(idx->write_ev, st));
    idx->write_stream = st;
    idx->write_pending = true;
    return MFAR_OK;
}
static int order_after_writes(mfar_index* idx, hipStream_t st) {
    if (!idx->write_pending) return MFAR_OK;
    if (hipEventQuery(idx->write_ev) == hipSuccess) idx->write_pending = false;           // (landed: later readers skip the wait)
    else if (st != idx->write_stream) HIPCHK(hipStreamWaitEvent(st, idx->write_ev, 0));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ rows in / out
static int check_rows(const mfar_index* idx, int field, int64_t row0, int64_t n, const void* ptr) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (field < 0 || field >= idx->F) return fail(MFAR_ERR_INVALID, "field out of range");
    if (row0 < 0 || n < 0 || row0 + n > idx->n_rows) return fail(MFAR_ERR_INVALID, "row range outside the shard");
    if (n > 0 && !ptr) return fail(MFAR_ERR_INVALID, "NULL data pointer");
    return MFAR_OK;
}

extern "C" int mfar_index_write_rows(mfar_index* idx, int field, int64_t local_row0, int64_t n, const float* src,
                                     int on_device, void* stream) {
    RETCHK(check_rows(idx, field, local_row0, n, src));
    if (n == 0) return MFAR_OK;
    HIPCHK(hipSetDevice(idx->device));
    idx->screen_dirty = true;
    idx->fused_dirty = true;
    idx->rows16_dirty = true;
    idx->gslab_ok = false;
    hipStream_t st = (hipStream_t)stream;
    RETCHK(order_after_pipelines(idx, st));      // launches in flight read the old rows to the end
    char* fbase = (char*)idx->slab + (size_t)field * idx->field_stride * idx->esize;
    const int64_t chunk = on_device ? n : std::min<int64_t>(n, (int64_t)(256u << 20) / (idx->E * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t m = std::min(chunk, n - r0);
        const float* s = src + (size_t)r0 * idx->E;
        if (!on_device) {
            RETCHK(idx->in[0].ensure((size_t)m * idx->E * 4));
            HIPCHK(hipMemcpyAsync(idx->in[0].p, s, (size_t)m * idx->E * 4, hipMemcpyHostToDevice, st));
            s = idx->in[0].as<float>();
        }
        if (idx->dtype == MFAR_DTYPE_BF16) {
            const long long total = m * (idx->E / 8);
            mfar_tile_rows_bf16_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(s, (unsigned short*)fbase,
                                                                                                     local_row0 + r0, m, idx->E);
        } else {
            const long long total = m * (idx->E / 4);
            mfar_tile_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(s, (float*)fbase, local_row0 + r0, m, idx->E);
        }
        HIPCHK(hipGetLastError());
        if (!on_device) HIPCHK(hipStreamSynchronize(st));
    }
    if (on_device) RETCHK(mark_written(idx, st));
    else if (idx->write_pending && idx->write_stream == st) idx->write_pending = false;      // (that stream was just synchronised)
    return MFAR_OK;
}

extern "C" int mfar_index_read_rows(mfar_index* idx, int field, int64_t local_row0, int64_t n, float* dst, int on_device,
                                    void* stream) {
    RETCHK(check_rows(idx, field, local_row0, n, dst));
    if (n == 0) return MFAR_OK;
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    RETCHK(order_after_writes(idx, st));
    const char* fbase = (const char*)idx->slab + (size_t)field * idx->field_stride * idx->esize;
    const int64_t chunk = on_device ? n : std::min<int64_t>(n, (int64_t)(256u << 20) / (idx->E * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t m = std::min(chunk, n - r0);
        float* d = dst + (size_t)r0 * idx->E;
        float* dd = d;
        if (!on_device) {
            RETCHK(idx->out[0].ensure((size_t)m * idx->E * 4));
            dd = idx->out[0].as<float>();
        }
        if (idx->dtype == MFAR_DTYPE_BF16) {
            const long long total = m * (idx->E / 8);
            mfar_untile_rows_bf16_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>((const unsigned short*)fbase, dd,
                                                                                                       local_row0 + r0, m, idx->E);
        } else {
            const long long total = m * (idx->E / 4);
            mfar_untile_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>((const float*)fbase, dd, local_row0 + r0, m, idx->E);
        }
        HIPCHK(hipGetLastError());
        if (!on_device) {
            HIPCHK(hipMemcpyAsync(d, dd, (size_t)m * idx->E * 4, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ staging helpers
// Bring a host input to the device (or pass a device pointer through).
template <typename T>
static int stage_in(DevBuf& buf, const T* src, size_t count, int on_device, hipStream_t st, const T** out) {
    if (!src || count == 0) {
        *out = src;
        return MFAR_OK;
    }
    if (on_device) {
        *out = src;
        return MFAR_OK;
    }
    RETCHK(buf.ensure(count * sizeof(T)));
    HIPCHK(hipMemcpyAsync(buf.p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
    *out = buf.as<T>();
    return MFAR_OK;
}
template <typename T>
static int stage_out(DevBuf& buf, T* dst, size_t count, int on_device, T** out) {
    if (!dst || count == 0) {
        *out = dst;
        return MFAR_OK;
    }
    if (on_device) {
        *out = dst;
        return MFAR_OK;
    }
    RETCHK(buf.ensure(count * sizeof(T)));
    *out = buf.as<T>();
    return MFAR_OK;
}
template <typename T>
static int copy_back(T* host_dst, const T* dev_src, size_t count, int on_device, hipStream_t st) {
    if (on_device || !host_dst || count == 0) return MFAR_OK;
    HIPCHK(hipMemcpyAsync(host_dst, dev_src, count * sizeof(T), hipMemcpyDeviceToHost, st));
    return MFAR_OK;
}

static int check_search_common(const mfar_index* idx, const float* q, int Q, int k) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (Q < 0) return fail(MFAR_ERR_INVALID, "Q < 0");
    if (Q > 0 && !q) return fail(MFAR_ERR_INVALID, "q is NULL");
    if (k <= 0 || k > MFAR_MAX_K) return fail(MFAR_ERR_INVALID, "k must be in [1, 128]");
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ stage 1
// Chunk table of a scanned slab for list depth k (mfar_stage1.h).  A field's share of the grid follows its tiles; the list
// merge holds n_chunks * k keys of one field, which caps the chunks of a field.
//   waves   waves per workgroup of the pass (4; 8 for the wide pass): wave blocks published per sampled tile
//   wgs     workgroups per CU the grid is sized for
//   skip    fields the table leaves out (no chunks: mfar_tables.h)
static int build_table(mfar_index* idx, const S1Geom& g_in, S1Table& t, int k, bool solo, int sample_tiles_max, bool sample_forced, int waves, int wgs, hipStream_t st,
                       u32 skip = 0) {
    if (t.k == k && t.wgs == wgs && t.skip == skip) return MFAR_OK;
    S1GeomHost g = g_in;
    for (int f = 0; f < idx->F; ++f)
        if ((skip >> f) & 1u) g.n_tiles[f] = 0;
    if (t.k >= 0) HIPCHK(hipDeviceSynchronize());   // a launch in flight may still read the old table
    const int F = idx->F;
    static const int sample_div = getenv("MFAR_SAMPLE_DIV") ? std::max(1, atoi(getenv("MFAR_SAMPLE_DIV"))) : 12;
    static const int append_target = getenv("MFAR_APPEND_TARGET") ? std::max(1, atoi(getenv("MFAR_APPEND_TARGET"))) : 130;
    t.k = -1;                                       // invalid until everything below succeeded
    static const int group_chunks = getenv("MFAR_MERGE_GROUP") ? std::max(0, atoi(getenv("MFAR_MERGE_GROUP"))) : 0;
    s1_build_table(g, F, idx->n_cu, k, solo, sample_tiles_max, sample_forced, waves, wgs, sample_div, append_target, t,
                   group_chunks);   // mfar_tables.h
    t.k = -1;
    if (t.two_level) {
        RETCHK(t.d_gchunk.ensure(t.gchunk.size() * sizeof(int)));
        RETCHK(t.d_fgroup.ensure((F + 1) * sizeof(int)));
        RETCHK(t.d_gfield.ensure(t.gfield.size() * sizeof(int)));
        HIPCHK(hipMemcpyAsync(t.d_gchunk.p, t.gchunk.data(), t.gchunk.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(t.d_fgroup.p, t.fgroup.data(), (F + 1) * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(t.d_gfield.p, t.gfield.data(), t.gfield.size() * sizeof(int), hipMemcpyHostToDevice, st));
    }
    RETCHK(t.d_chunks.ensure(t.chunks.size() * sizeof(S1Chunk)));
    RETCHK(t.d_fchunk.ensure((F + 1) * sizeof(int)));
    RETCHK(t.d_samp_n.ensure(F * sizeof(int)));
    HIPCHK(hipMemcpyAsync(t.d_chunks.p, t.chunks.data(), t.chunks.size() * sizeof(S1Chunk), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(t.d_fchunk.p, t.fchunk.data(), (F + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(t.d_samp_n.p, t.samp_n.data(), F * sizeof(int), hipMemcpyHostToDevice, st));
    t.k = k;
    t.wgs = wgs;
    t.skip = skip;
    return MFAR_OK;
}

// One stage-1 pass over one slab for one block of <= 64 queries: [sample pass] -> full pass -> list merge.
struct S1Out {
    long long* ids;   // [., nf, k]
    float* sc;
    int* cnt;         // [qt_n * nf] or nullptr
    int q0;           // first output query row
    int sentinel;     // padding convention of the output lists
    long long row_offset;   // added to the local rows of the lists (0: the lists hold unique-row numbers)
};
enum { S1_F32 = 0, S1_BF16 = 1, S1_F16 = 2, S1_F16W = 3, S1_BF16S = 4, S1_BF16W = 5, S1_BF16C = 6 };
// S1_F16W: the wide (128-query, one fp16 term) screen pass of an fp32 index; S1_BF16S / S1_BF16W: the certified passes over a bf16
// slab (two bf16 query terms; 64 / 128 columns); S1_BF16C: 128 columns over the same slab with the docs converted to fp16 in
// registers and one fp16 query term (half the MFMAs; the default wide pass of a bf16 index)
static bool s1_is_wide(int kind) { return kind == S1_F16W || kind == S1_BF16W || kind == S1_BF16C; }
static int launch_s1(int kind, bool sample, unsigned n_chunks, unsigned wave, hipStream_t st, const S1Params& p_in, const char** name_out = nullptr) {
    S1Params p = p_in;
    p.n_launch = (int)n_chunks;
    // a repair pass walks its finely cut table with one wave of workgroups (mfar_stage1.h S1_CHUNK_LOOP); any other launch: one
    // workgroup per chunk
    const unsigned grid = p.only_failed ? std::min(n_chunks, wave) : n_chunks;
    const dim3 g(grid), b(S1_THREADS);
    const char* name = "";
    if (kind == S1_F32) {
        // register-ring variant (docs straight into VGPRs) when the k-steps divide into 6 or 4 register slots; MFAR_S1_REGRING=0: LDS ring
        static const bool regring32 = !(getenv("MFAR_S1_REGRING") && atoi(getenv("MFAR_S1_REGRING")) == 0);
        static const int f32_ring = getenv("MFAR_F32_RING") ? atoi(getenv("MFAR_F32_RING")) : 0;       // diagnostic: 6 forces the 6-slot ring
        // (4 slots: 237 VGPRs and no spills; 6 slots: 256 and 9 spilled around the epilogue -- measured equal, 6.91 / 6.93 ms at 1 M x 8 x 768)
        if (regring32 && p.n_steps % 4 == 0 && !(f32_ring == 6 && p.n_steps % 6 == 0)) {
            if (sample) { name = "mfar_stage1_f32r4_sample_kernel"; mfar_stage1_f32r4_sample_kernel<<<g, b, S1FR4_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f32r4_kernel"; mfar_stage1_f32r4_kernel<<<g, b, S1FR4_LDS_BYTES, st>>>(p); }
        } else if (regring32 && p.n_steps % 6 == 0) {
            if (sample) { name = "mfar_stage1_f32r_sample_kernel"; mfar_stage1_f32r_sample_kernel<<<g, b, S1FR_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f32r_kernel"; mfar_stage1_f32r_kernel<<<g, b, S1FR_LDS_BYTES, st>>>(p); }
        } else if (sample) { name = "mfar_stage1_sample_kernel"; mfar_stage1_sample_kernel<<<g, b, S1_LDS_BYTES, st>>>(p); }
        else { name = "mfar_stage1_kernel"; mfar_stage1_kernel<<<g, b, S1_LDS_BYTES, st>>>(p); }
    } else if (kind == S1_F16W) {
        // 4-slot ring whenever the k-steps divide by 4 (round 4, with the SGPR-addressed body: 14 spilled VGPRs instead of 27, 50.8 KB of LDS
        // instead of 58.8; same run, interleaved: 1 M x 8 52.0-52.3 k against 51.6-52.2 k q/s sustained, 129 375 x 22 68.3 k against 66.0 k)
        static const int w_ring = getenv("MFAR_WIDE_RING") ? atoi(getenv("MFAR_WIDE_RING")) : 0;   // diagnostic: 6 forces the 6-slot ring
        const bool r4 = p.n_steps % 4 == 0 && !(w_ring == 6 && p.n_steps % 6 == 0);
        if (p.arow) {              // ROW MODE twins
            if (r4 && sample) { name = "mfar_stage1_f16w4_rm_sample_kernel"; mfar_stage1_f16w4_rm_sample_kernel<<<g, b, S1HW4_RM_LDS_BYTES, st>>>(p); }
            else if (r4) { name = "mfar_stage1_f16w4_rm_kernel"; mfar_stage1_f16w4_rm_kernel<<<g, b, S1HW4_RM_LDS_BYTES, st>>>(p); }
            else if (sample) { name = "mfar_stage1_f16w_rm_sample_kernel"; mfar_stage1_f16w_rm_sample_kernel<<<g, b, S1HW_RM_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f16w_rm_kernel"; mfar_stage1_f16w_rm_kernel<<<g, b, S1HW_RM_LDS_BYTES, st>>>(p); }
        } else if (r4) {
            if (sample) { name = "mfar_stage1_f16w4_sample_kernel"; mfar_stage1_f16w4_sample_kernel<<<g, b, S1HW4_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f16w4_kernel"; mfar_stage1_f16w4_kernel<<<g, b, S1HW4_LDS_BYTES, st>>>(p); }
        } else {
            if (sample) { name = "mfar_stage1_f16w_sample_kernel"; mfar_stage1_f16w_sample_kernel<<<g, b, S1HW_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f16w_kernel"; mfar_stage1_f16w_kernel<<<g, b, S1HW_LDS_BYTES, st>>>(p); }
        }
    } else if (kind == S1_BF16W) {
        // the 4-slot ring (50.5 KB of LDS) whenever the k-steps divide by 4: see mfar_stage1.h on LDS fragmentation
        static const int bw_ring = getenv("MFAR_BF16W_RING") ? atoi(getenv("MFAR_BF16W_RING")) : 0;   // diagnostic: 6 forces the 6-slot ring
        if (p.n_steps % 4 == 0 && !(bw_ring == 6 && p.n_steps % 6 == 0)) {
            if (sample) { name = "mfar_stage1_bf16w4_sample_kernel"; mfar_stage1_bf16w4_sample_kernel<<<g, b, S1BW4_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16w4_kernel"; mfar_stage1_bf16w4_kernel<<<g, b, S1BW4_LDS_BYTES, st>>>(p); }
        } else {
            if (sample) { name = "mfar_stage1_bf16w_sample_kernel"; mfar_stage1_bf16w_sample_kernel<<<g, b, S1BW_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16w_kernel"; mfar_stage1_bf16w_kernel<<<g, b, S1BW_LDS_BYTES, st>>>(p); }
        }
    } else if (kind == S1_BF16C) {
        if (p.n_steps % 6 == 0) {
            if (sample) { name = "mfar_stage1_bf16c_sample_kernel"; mfar_stage1_bf16c_sample_kernel<<<g, b, S1BC_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16c_kernel"; mfar_stage1_bf16c_kernel<<<g, b, S1BC_LDS_BYTES, st>>>(p); }
        } else {
            if (sample) { name = "mfar_stage1_bf16c4_sample_kernel"; mfar_stage1_bf16c4_sample_kernel<<<g, b, S1BC4_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16c4_kernel"; mfar_stage1_bf16c4_kernel<<<g, b, S1BC4_LDS_BYTES, st>>>(p); }
        }
    } else if (kind == S1_BF16S) {
        if (p.n_steps % 6 == 0) {
            if (sample) { name = "mfar_stage1_bf16s_sample_kernel"; mfar_stage1_bf16s_sample_kernel<<<g, b, S1HR_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16s_kernel"; mfar_stage1_bf16s_kernel<<<g, b, S1HR_LDS_BYTES, st>>>(p); }
        } else {
            if (sample) { name = "mfar_stage1_bf16s4_sample_kernel"; mfar_stage1_bf16s4_sample_kernel<<<g, b, S1HR4_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16s4_kernel"; mfar_stage1_bf16s4_kernel<<<g, b, S1HR4_LDS_BYTES, st>>>(p); }
        }
    } else {
        // register-ring variants (docs straight into VGPRs, 25 / 37 KB of LDS) when the k-steps divide into the 6 register slots
        static const bool regring = !(getenv("MFAR_S1_REGRING") && atoi(getenv("MFAR_S1_REGRING")) == 0);
        const int R = !regring ? 0 : (p.n_steps % 6 == 0 && (kind != S1_F16 || S1HR_R == 6) ? 6 : (p.n_steps % 4 == 0 ? 4 : 0));
        if (kind == S1_BF16) {
            if (R == 6 && sample) { name = "mfar_stage1_bf16r_sample_kernel"; mfar_stage1_bf16r_sample_kernel<<<g, b, S1BR_LDS_BYTES, st>>>(p); }
            else if (R == 6) { name = "mfar_stage1_bf16r_kernel"; mfar_stage1_bf16r_kernel<<<g, b, S1BR_LDS_BYTES, st>>>(p); }
            else if (R == 4 && sample) { name = "mfar_stage1_bf16r4_sample_kernel"; mfar_stage1_bf16r4_sample_kernel<<<g, b, S1BR4_LDS_BYTES, st>>>(p); }
            else if (R == 4) { name = "mfar_stage1_bf16r4_kernel"; mfar_stage1_bf16r4_kernel<<<g, b, S1BR4_LDS_BYTES, st>>>(p); }
            else if (sample) { name = "mfar_stage1_bf16_sample_kernel"; mfar_stage1_bf16_sample_kernel<<<g, b, S1B_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_bf16_kernel"; mfar_stage1_bf16_kernel<<<g, b, S1B_LDS_BYTES, st>>>(p); }
        } else {
            if (R == 6 && sample) { name = "mfar_stage1_f16r_sample_kernel"; mfar_stage1_f16r_sample_kernel<<<g, b, S1HR_LDS_BYTES, st>>>(p); }
            else if (R == 6) { name = "mfar_stage1_f16r_kernel"; mfar_stage1_f16r_kernel<<<g, b, S1HR_LDS_BYTES, st>>>(p); }
            else if (R == 4 && sample) { name = "mfar_stage1_f16r4_sample_kernel"; mfar_stage1_f16r4_sample_kernel<<<g, b, S1HR4_LDS_BYTES, st>>>(p); }
            else if (R == 4) { name = "mfar_stage1_f16r4_kernel"; mfar_stage1_f16r4_kernel<<<g, b, S1HR4_LDS_BYTES, st>>>(p); }
            else if (sample) { name = "mfar_stage1_f16_sample_kernel"; mfar_stage1_f16_sample_kernel<<<g, b, S1H_LDS_BYTES, st>>>(p); }
            else { name = "mfar_stage1_f16_kernel"; mfar_stage1_f16_kernel<<<g, b, S1H_LDS_BYTES, st>>>(p); }
        }
    }
    if (name_out) *name_out = name;
    HIPCHK(hipGetLastError());
    return MFAR_OK;
}

//   geom       geometry (rows / bases per field) of `slab`; f0, nf: the pass covers fields [f0, f0 + nf) (all, or one)
//   tau0       strict starting threshold (0 = zero sentinel of index.py:192-193, -inf = none)
//   tau_base   [F, 64] non-strict starting thresholds or nullptr (screened pass)
//   only_failed  [F] device flags or nullptr: restrict the pass to flagged fields (screen fall-back)
//   record     this is the pass the pipelining event and the timing events bracket
//   phases     S1_PREPARE (sample pass + thresholds) | S1_SCAN (the full pass) | S1_FINISH (list merge); the three may be
//              issued by separate calls on different streams (ordered by the caller), all with the same arguments
enum { S1_PREPARE = 1, S1_SCAN = 2, S1_FINISH = 4, S1_CERTIFY = 8, S1_ALL = 15 };   // S1_CERTIFY: stage1_block only
//   skip       fields (of an all-fields pass) that are left out: no chunks, empty lists (AUTO-OFF)
//   force_sample  a restricted pass (only_failed) that is certain to scan its fields: always with its own sample pass
//   t2         the TIER 2 rescan of the certified screen (mfar_screen.h): the flagged fields (only_failed) again, on the table of the
//              screened pass itself (same chunk ids, same list scratch: nothing to allocate), S1_SCAN only, no sample pass -- tau_base
//              holds the fixed thresholds -- and no merge: mfar_t2_collect_kernel reads the chunk lists
static int stage1_pass(mfar_index* idx, mfar_index::S1Slot& sl, S1Geom& geom, int f0, int nf, int phases, int kind, const void* slab,
                       const void* qt, int qt_n, int k, float tau0, const float* tau_base, const int* only_failed, bool record,
                       const S1Out& o, hipStream_t st, u32 skip = 0, bool force_sample = false, bool t2 = false, const S1DeepDev* deep = nullptr) {
    const int qw = s1_is_wide(kind) ? 128 : 64;   // query columns of the pass: stride of every per-query table below
    // a repair pass (only_failed) uses the finely cut table as well: the workgroups of the fields that did not fail exit at once,
    // and a failed field is then scanned by the whole GPU instead of by its share of one wave (one failed field of eight at 1 M
    // rows: 64 workgroups x 61 tiles at the MFMA-bound rate = several ms; cut into 512 chunks: under 1 ms)
    const bool repair = only_failed != nullptr;
    const bool solo = nf != idx->F || (repair && !t2), wide = s1_is_wide(kind);
    if (solo) skip = 0;
    S1Table& tb = wide ? (solo ? geom.solo_w : (skip ? geom.all_w_skip : geom.all_w)) : (solo ? geom.solo : (skip ? geom.all_skip : geom.all));
    static const int sample_tiles_env = getenv("MFAR_SAMPLE_TILES") ? atoi(getenv("MFAR_SAMPLE_TILES")) : 0;
    RETCHK(build_table(idx, geom, tb, k, solo, sample_tiles_env > 0 ? sample_tiles_env : (kind == S1_F32 ? 1 : 2), sample_tiles_env > 0, 4, idx->wgs_per_cu, st,
                       skip));
    const int c_lo = tb.fchunk[f0], c_hi = tb.fchunk[f0 + nf];
    RETCHK(sl.lists.ensure((size_t)tb.n_chunks * qw * S1_CAP * sizeof(uint2)));
    RETCHK(sl.list_cnt.ensure((size_t)tb.n_chunks * qw * sizeof(int)));
    RETCHK(sl.gtau.ensure((size_t)idx->F * qw * sizeof(float)));
    S1Params p = {};
    p.slab = slab;
    p.qt = qt;
    p.lists = sl.lists.as<uint2>();
    p.list_cnt = sl.list_cnt.as<int>();
    p.chunks = tb.d_chunks.as<S1Chunk>();
    p.chunk0 = c_lo;
    p.n_steps = idx->n_steps;
    p.Q = qt_n;
    p.qw = qw;
    p.k = k;
    p.tau0 = tau0;
    p.gtau = tau_base;
    p.sample = 0;
    p.samp_out = nullptr;
    p.samp_stride = tb.samp_stride;
    p.only_failed = only_failed;
    if (kind == S1_F16W && sl.row_mode && !repair && sl.arow.p) {     // ROW MODE (mfar_screen.h); the 64-column pass keeps the field-wide bound
        p.arow = sl.arow.as<float>();
        p.rnorm = idx->s_rnorm.as<float>();
        p.row_mask = sl.row_mask;
        p.dump_base = idx->dump_base.as<long long>();
    }
    if (kind == S1_F16W && sl.dump_on && nf == idx->F && !repair) {
        p.dump = sl.dump.p;
        p.dump_inv = sl.dinv.as<float>();
        p.dump_base = idx->dump_base.as<long long>();
    }
    if (kind == S1_BF16S || kind == S1_BF16W || kind == S1_BF16C) {     // documents are scanned, unique rows are ranked (mfar_stage1.h s1_acc_init)
        p.rep_bits = idx->rep_bits.as<u64>();
        p.rep_stride = idx->n_blk;
        p.cvt = idx->s_cvt.as<uint2>();
    }
    {
        const char* dbg = getenv("MFAR_S1_DEBUG");
        p.dbg = dbg ? atoi(dbg) : 0;
    }
    MergeParams m = {};
    m.lists = p.lists;
    m.list_cnt = p.list_cnt;
    m.fchunk = tb.d_fchunk.as<int>();
    m.row_offset = o.row_offset;
    m.f0 = f0;
    m.nf = nf;
    m.max_chunks = tb.max_chunks;
    m.k = k;
    m.q0 = o.q0;
    m.sentinel = o.sentinel;
    m.qw = qw;
    m.cnt_out = nullptr;
    m.only_failed = only_failed;
    m.skip_mask = skip;      // (+ the DEEP SCAN fields below: their chunk lists go to tier 2's collect kernel unmerged)
    // one merge launch: `n_lists` = most lists a workgroup of it merges, grid = (query, list owner) pairs
    auto launch_merge = [&](const MergeParams& mp, int n_lists, int owners) -> int {
        const int n_keys = n_lists * k;
        const dim3 grid(qt_n * owners), block(256);
        if (k <= SEL_MAX_K && n_lists <= 128 && n_keys <= 48 * 256) {   // keys stay in registers: no LDS staging
            if (n_keys <= 8 * 256) mfar_merge_lists_regs_kernel<8><<<grid, block, 0, st>>>(mp);
            else if (n_keys <= 16 * 256) mfar_merge_lists_regs_kernel<16><<<grid, block, 0, st>>>(mp);
            else if (n_keys <= 32 * 256) mfar_merge_lists_regs_kernel<32><<<grid, block, 0, st>>>(mp);
            else mfar_merge_lists_regs_kernel<48><<<grid, block, 0, st>>>(mp);
            HIPCHK(hipGetLastError());
            return MFAR_OK;
        }
        const size_t lds = SEL_LDS_BYTES(n_keys);
        if (n_keys <= 8 * 256) mfar_merge_lists_kernel<8><<<grid, block, lds, st>>>(mp);
        else if (n_keys <= 32 * 256) mfar_merge_lists_kernel<32><<<grid, block, lds, st>>>(mp);
        else mfar_merge_lists_kernel<64><<<grid, block, lds, st>>>(mp);
        HIPCHK(hipGetLastError());
        return MFAR_OK;
    };
    const unsigned grid = (unsigned)(c_hi - c_lo);
    if (grid == 0) return fail(MFAR_ERR_INVALID, "empty chunk range");
    // Sample pass: every workgroup scans only the first tile(s) of its chunk; the k-th best score of that sample is a valid
    // (non-strict) lower bound of the final k-th best, so the full pass starts with a tight threshold and appends /
    // compacts almost nothing.  Worth it once the chunks are much longer than one tile.
    static const int sample_min_tiles = getenv("MFAR_SAMPLE_MIN_TILES") ? atoi(getenv("MFAR_SAMPLE_MIN_TILES")) : 3;
    // ... or, with short chunks, whenever it yields thresholds for most of the rows: a tile without one costs 8 x a normal tile in
    // the full pass (90 % empty 129 k x 22: 2.4 tiles per chunk, scan 0.85 ms without the sample pass)
    const bool use_sample = (tb.total_tiles >= (long long)sample_min_tiles * tb.n_chunks || 2 * tb.thresholded_tiles >= tb.total_tiles) &&
                            !(p.dbg & 2) && (!repair || idx->repair_sample || force_sample) && !t2;
    const bool light_sample = use_sample && 2 * tb.samp_stride <= 4096;
    p.sample_tiles = light_sample ? tb.sample_tiles : 1;
    // DEEP SCAN needs the light sample pass (its thresholds come from the published sample values): without one the batch has no deep field
    // (the same decision in every phase: it follows from the table alone)
    S1DeepDev dd = {};
    if (deep && light_sample && !repair) dd = *deep;
    else if (deep) sl.deep_mask = 0;
    m.skip_mask |= dd.mask;
    if (light_sample) {
        // every wave publishes the 2 best scores per query of its 64 sampled rows; tau = k-th largest of those
        if (phases & S1_PREPARE) {
            S1Params ps = p;
            ps.sample = 2;
            RETCHK(sl.samp.ensure((size_t)idx->F * tb.samp_stride * 2 * qw * sizeof(float)));
            ps.samp_out = sl.samp.as<float>();
            RETCHK(launch_s1(kind, true, grid, (unsigned)(idx->wgs_per_cu * idx->n_cu), st, ps));
            const dim3 tg((qw * nf + 3) / 4), tb_(256);
            const int* sn = tb.d_samp_n.as<int>();
            if (2 * tb.samp_stride <= 512) mfar_sample_tau_kernel<8><<<tg, tb_, 0, st>>>(ps.samp_out, sn, tb.samp_stride, f0, nf, k, p.tau0, tau_base, sl.gtau.as<float>(), qw, dd);
            else if (2 * tb.samp_stride <= 1024) mfar_sample_tau_kernel<16><<<tg, tb_, 0, st>>>(ps.samp_out, sn, tb.samp_stride, f0, nf, k, p.tau0, tau_base, sl.gtau.as<float>(), qw, dd);
            else if (2 * tb.samp_stride <= 2048) mfar_sample_tau_kernel<32><<<tg, tb_, 0, st>>>(ps.samp_out, sn, tb.samp_stride, f0, nf, k, p.tau0, tau_base, sl.gtau.as<float>(), qw, dd);
            else mfar_sample_tau_kernel<64><<<tg, tb_, 0, st>>>(ps.samp_out, sn, tb.samp_stride, f0, nf, k, p.tau0, tau_base, sl.gtau.as<float>(), qw, dd);
            HIPCHK(hipGetLastError());
        }
        p.gtau = sl.gtau.as<float>();
    } else if (use_sample && !tau_base && !tb.two_level) {
        if (phases & S1_PREPARE) {
            S1Params ps = p;
            ps.sample = 1;
            RETCHK(launch_s1(kind, true, grid, (unsigned)(idx->wgs_per_cu * idx->n_cu), st, ps));
            MergeParams ms = m;
            ms.out_ids = nullptr;
            ms.out_scores = nullptr;
            ms.tau_out = sl.gtau.as<float>();
            RETCHK(launch_merge(ms, tb.max_chunks, nf));
        }
        p.gtau = sl.gtau.as<float>();
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (record && (phases & S1_SCAN)) {
        if (idx->timing && idx->ev_n < 4096) {
            if ((int)idx->ev.size() < 2 * (idx->ev_n + 1)) {
                hipEvent_t a, b;
                HIPCHK(hipEventCreate(&a));
                HIPCHK(hipEventCreate(&b));
                idx->ev.push_back(a);
                idx->ev.push_back(b);
            }
            e0 = idx->ev[2 * idx->ev_n];
            e1 = idx->ev[2 * idx->ev_n + 1];
            idx->ev_n++;
            HIPCHK(hipEventRecord(e0, st));
        }
        if (!idx->mid_ev) HIPCHK(hipEventCreateWithFlags(&idx->mid_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(idx->mid_ev, st));
    }
    if (phases & S1_SCAN) {
        // dynamic work distribution (mfar_stage1.h s1_unit_*): the kernels that have it, full passes that are not repairs
        static const bool dyn_on = !(getenv("MFAR_S1_DYN") && atoi(getenv("MFAR_S1_DYN")) == 0);
        static const int unit_tiles = getenv("MFAR_UNIT_TILES") ? std::max(2, atoi(getenv("MFAR_UNIT_TILES"))) : 2;
        if (dyn_on && !repair && s1_is_wide(kind)) {
            RETCHK(sl.unit_ctr.ensure((size_t)MFAR_MAX_FIELDS * sizeof(int)));
            HIPCHK(hipMemsetAsync(sl.unit_ctr.p, 0, (size_t)MFAR_MAX_FIELDS * sizeof(int), st));
            p.unit_ctr = sl.unit_ctr.as<int>();
            p.unit_tiles = unit_tiles;
        }
        if (!t2 && !repair) sl.scan_tau = p.gtau;       // (the screened pass of the batch; the exact passes write no chunk lists tier 2 reads)
        // (t2: the 16-bit kernels take one chunk per workgroup, flagged or not -- the grid is the whole chunk range)
        RETCHK(launch_s1(kind, false, grid, t2 ? grid : (unsigned)(idx->wgs_per_cu * idx->n_cu), st, p, record ? &idx->last_s1_kernel : nullptr));
    }
    if (e1) HIPCHK(hipEventRecord(e1, st));
    if (phases & S1_FINISH) {
        m.out_ids = o.ids;
        m.out_scores = o.sc;
        m.tau_out = nullptr;
        m.cnt_out = o.cnt;
        if (tb.two_level) {
            // level 1: every group of chunks -> one list in chunk format; level 2: the groups of a field -> the final list
            RETCHK(sl.lists2.ensure((size_t)tb.n_groups * qw * S1_CAP * sizeof(uint2)));
            RETCHK(sl.list_cnt2.ensure((size_t)tb.n_groups * qw * sizeof(int)));
            MergeParams m1 = m;
            m1.fchunk = tb.d_gchunk.as<int>();
            m1.gfield = tb.d_gfield.as<int>();
            m1.f0 = tb.fgroup[f0];
            m1.nf = tb.fgroup[f0 + nf] - tb.fgroup[f0];
            m1.out_ids = nullptr;
            m1.out_scores = nullptr;
            m1.cnt_out = nullptr;
            m1.out_lists = sl.lists2.as<uint2>();
            m1.out_cnt = sl.list_cnt2.as<int>();
            m1.max_chunks = tb.max_group_chunks;
            RETCHK(launch_merge(m1, tb.max_group_chunks, m1.nf));
            MergeParams m2 = m;
            m2.lists = sl.lists2.as<uint2>();
            m2.list_cnt = sl.list_cnt2.as<int>();
            m2.fchunk = tb.d_fgroup.as<int>();
            m2.max_chunks = tb.max_groups;
            RETCHK(launch_merge(m2, tb.max_groups, nf));
        } else {
            RETCHK(launch_merge(m, tb.max_chunks, nf));
        }
    }
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ fp16 screen
static bool screen_wanted(const mfar_index* idx, int k) {
    if (idx->screen_mode == 0 || idx->screen_nomem) return false;
    // (bf16 indexes: the certified pass scans the bf16 slab itself -- nothing to allocate but the unique-row tables; it needs a
    //  register-ring kernel for the dim)
    if (idx->dtype == MFAR_DTYPE_BF16 && idx->n_steps % 4 != 0 && idx->n_steps % 6 != 0) return false;
    if (k + SCREEN_EXTRA_MIN > SCREEN_MAX_KP) return false;
    if (idx->E * 4 > 60 * 1024) return false;   // the re-scoring kernel stages a query row in LDS
    // auto mode stays off for very wide rows: the certificate's fp32-accumulation term (4 E + 66) u32 grows with E while the
    // relative gap between the k-th and the k'-th score shrinks like 1 / sqrt(E).  Isotropic 300 k x 8 rows: no failed list up
    // to E = 2048, 2 of 11 264 at 2560, 380 at 3072 -- and ONE failed list sends its whole field through the exact pass, so at
    // 3072 every launch pays the screen AND the exact pass.  mfar_set_screen(2) still forces the screen.
    if (idx->screen_mode < 2 && idx->E > 2560) return false;
    // a bf16 index takes the certified pass at ANY size: its lists then carry the chain's bits whatever the shard size (the plain MFMA pass
    // sums in another order), and it needs no second copy of the rows
    if (idx->dtype == MFAR_DTYPE_BF16 && idx->n_rows >= 1) return true;
    return idx->screen_mode >= 2 || idx->n_rows >= 16384;
}

// uof[row] = unique number + 1 of the row's group, for every row (in: valid for representatives only; a representative is its own repof,
// so its entry is never written while others read it)
__global__ void mfar_uof_all_kernel(long long n, const int* __restrict__ repof, u32* __restrict__ uof) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int r = repof[i];
        if (r != (int)i) uof[i] = uof[r];
    }
}
__global__ void mfar_iota_kernel(int* __restrict__ a, int* __restrict__ b, int* __restrict__ c, int* __restrict__ d, int* __restrict__ e, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        a[i] = (int)i;   // urep
        b[i] = (int)i;   // ustart
        c[i] = 1;        // ucount
        d[i] = (int)i;   // members
        if (e) e[i] = (int)i;   // repof
    }
}

// Unique rows of one field -> idx->u_* tables of that field; returns the number of unique rows and the largest group.
//   bits_out / uof_out   (bf16 index) the field's "real row and representative" bits [n_blk words] and unique number + 1 per row, or nullptr
static int build_unique_rows(mfar_index* idx, int f, const float* field, DevBuf* tmp, hipStream_t st, int* n_unique_out, int* largest_out,
                             u64* bits_out, u32* uof_out) {
    const long long n = idx->n_rows;
    int* urep = idx->u_rep.as<int>() + (size_t)f * n;
    int* ustart = idx->u_start.as<int>() + (size_t)f * n;
    int* ucount = idx->u_count.as<int>() + (size_t)f * n;
    int* members = idx->u_members.as<int>() + (size_t)f * n;
    int* repof = idx->u_repof.p ? idx->u_repof.as<int>() + (size_t)f * n : nullptr;
    if (!idx->screen_dedup || n < 2) {
        if (n > 0) {
            mfar_iota_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(urep, ustart, ucount, members, repof, n);
            HIPCHK(hipGetLastError());
            if (uof_out) mfar_iota1_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(uof_out, n);
        }
        if (bits_out) mfar_rep_bits_kernel<<<dim3((unsigned)idx->n_blk), dim3(64), 0, st>>>(nullptr, n, bits_out);
        HIPCHK(hipGetLastError());
        *n_unique_out = (int)n;
        *largest_out = n > 0 ? 1 : 0;
        return MFAR_OK;
    }
    // tmp: 0 keys in, 1 keys out, 2 vals in, 3 head, 4 gid, 5 is_rep, 6 urank, 7 gstart, 8 cub scratch
    RETCHK(tmp[0].ensure(n * 8));
    RETCHK(tmp[1].ensure(n * 8));
    for (int i = 2; i <= 7; ++i) RETCHK(tmp[i].ensure(n * 4));
    u64* k_in = tmp[0].as<u64>();
    u64* k_out = tmp[1].as<u64>();
    u32* v_in = tmp[2].as<u32>();
    u32* v_out = (u32*)members;      // the sorted row numbers ARE the member table
    u32 *head = tmp[3].as<u32>(), *gid = tmp[4].as<u32>(), *is_rep = tmp[5].as<u32>(), *urank = uof_out ? uof_out : tmp[6].as<u32>();
    int* gstart = tmp[7].as<int>();
    mfar_row_hash_kernel<<<dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st>>>(field, idx->n_steps, n, k_in, v_in);
    HIPCHK(hipGetLastError());
    size_t need = 0, need2 = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, k_in, k_out, v_in, v_out, (int)n, 0, 64, st));
    HIPCHK(hipcub::DeviceScan::InclusiveSum(nullptr, need2, head, gid, (int)n, st));
    RETCHK(tmp[8].ensure(std::max(need, need2) + 256));
    size_t cap = tmp[8].cap;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(tmp[8].p, cap, k_in, k_out, v_in, v_out, (int)n, 0, 64, st));   // stable: rows ascend inside a hash
    mfar_group_heads_kernel<<<dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, st>>>(field, idx->n_steps, n, k_out, v_out, head);
    HIPCHK(hipGetLastError());
    cap = tmp[8].cap;
    HIPCHK(hipcub::DeviceScan::InclusiveSum(tmp[8].p, cap, head, gid, (int)n, st));
    HIPCHK(hipMemsetAsync(is_rep, 0, n * 4, st));
    mfar_group_starts_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, head, gid, v_out, gstart, is_rep);
    HIPCHK(hipGetLastError());
    cap = tmp[8].cap;
    HIPCHK(hipcub::DeviceScan::InclusiveSum(tmp[8].p, cap, is_rep, urank, (int)n, st));
    if (bits_out) {
        mfar_rep_bits_kernel<<<dim3((unsigned)idx->n_blk), dim3(64), 0, st>>>(is_rep, n, bits_out);
        HIPCHK(hipGetLastError());
    }
    u32 n_groups = 0;
    HIPCHK(hipMemcpyAsync(&n_groups, gid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    mfar_unique_table_kernel<<<dim3((n_groups + 255) / 256), dim3(256), 0, st>>>(n, (int)n_groups, gstart, v_out, urank, urep, ustart, ucount);
    HIPCHK(hipGetLastError());
    if (repof) {
        mfar_rep_of_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, gid, gstart, v_out, repof);
        HIPCHK(hipGetLastError());
        if (uof_out) {      // every row gets its GROUP's unique number (a representative keeps its own): one table look-up per stage-2 pair
            mfar_uof_all_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(n, repof, uof_out);
            HIPCHK(hipGetLastError());
        }
    }
    // largest group (statistics only)
    cap = tmp[8].cap;
    int* d_max = (int*)tmp[3].p;
    HIPCHK(hipcub::DeviceReduce::Max(nullptr, need, ucount, d_max, (int)n_groups, st));
    RETCHK(tmp[8].ensure(need + 256));
    cap = tmp[8].cap;
    HIPCHK(hipcub::DeviceReduce::Max(tmp[8].p, cap, ucount, d_max, (int)n_groups, st));
    int largest = 0;
    HIPCHK(hipMemcpyAsync(&largest, d_max, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    *n_unique_out = (int)n_groups;
    *largest_out = largest;
    return MFAR_OK;
}

// 16-bit gather slab (mfar_select.h).  Allocation failure is not an error: gathers stay on the scan-ordered slab.
static bool gslab_alloc(mfar_index* idx) {
    if (idx->gslab_nomem) return false;
    idx->g_row_bytes = GSLAB_ROW_BYTES(idx->E);
    const size_t need = (size_t)idx->F * (size_t)std::max<int64_t>(idx->n_rows, 1) * idx->g_row_bytes;
    if (idx->gslab.ensure(need, true) != MFAR_OK) {
        (void)hipGetLastError();
        g_err.clear();
        idx->gslab.release();
        idx->gslab_nomem = true;
        return false;
    }
    return true;
}
// bf16 index: the row-major companion (the slab's values bit for bit), filled lazily by the first gather after rows were written
static int ensure_rows16(mfar_index* idx, hipStream_t st, bool* ok) {
    *ok = false;
    if (idx->dtype != MFAR_DTYPE_BF16 || idx->n_rows == 0) return MFAR_OK;
    static const bool enabled = !(getenv("MFAR_GATHER_SLAB") && atoi(getenv("MFAR_GATHER_SLAB")) == 0);   // diagnostic: 0 = gather from the scan-ordered slab
    if (!enabled) return MFAR_OK;
    if (idx->gslab_ok && !idx->rows16_dirty) {
        *ok = true;
        return MFAR_OK;
    }
    HIPCHK(hipDeviceSynchronize());   // a gather in flight on another stream may still read the companion
    if (!gslab_alloc(idx)) return MFAR_OK;
    const long long total = idx->n_rows * (long long)(idx->g_row_bytes / 16);
    for (int f = 0; f < idx->F; ++f) {
        mfar_gslab_build_bf16_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(
            (const unsigned short*)idx->slab + (size_t)f * idx->field_stride, idx->gslab.as<char>() + (size_t)f * idx->n_rows * idx->g_row_bytes,
            idx->n_rows, idx->n_steps, (int)idx->g_row_bytes);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(st));
    idx->rows16_dirty = false;
    idx->gslab_ok = true;
    *ok = true;
    return MFAR_OK;
}

// (re)build what the certified stage 1 needs when rows changed.
//   fp32 index: per-field statistics, the unique rows of every field, the fp16 screen slab of those rows, the fp16 gather slab;
//   bf16 index: the largest row norm per field, the unique rows (through an fp32 staging copy of one field at a time, so that the
//               hash / grouping kernels exist once), their "representative" bits and unique numbers -- the scan reads the slab itself.
// *ok = false: not available (allocation failed) -> the caller stays on the exact pass.
static int ensure_screen(mfar_index* idx, hipStream_t st, bool* ok) {
    *ok = false;
    if (!idx->screen_dirty && idx->screen_built) {
        *ok = true;
        return MFAR_OK;
    }
    HIPCHK(hipDeviceSynchronize());   // a rebuild replaces tables that launches in flight may still read
    const int F = idx->F;
    const long long n = idx->n_rows;
    const bool bf16 = idx->dtype == MFAR_DTYPE_BF16;
    auto nomem = [&]() {
        (void)hipGetLastError();
        g_err.clear();
        idx->screen_nomem = true;
        return MFAR_OK;
    };
    if (idx->u_rep.ensure((size_t)F * n * 4, true) != MFAR_OK || idx->u_start.ensure((size_t)F * n * 4, true) != MFAR_OK ||
        idx->u_count.ensure((size_t)F * n * 4, true) != MFAR_OK || idx->u_members.ensure((size_t)F * n * 4, true) != MFAR_OK ||
        idx->u_n.ensure((size_t)F * 4) != MFAR_OK)
        return nomem();
    if (bf16 && idx->rep_bits.ensure((size_t)F * idx->n_blk * 8, true) != MFAR_OK) return nomem();
    if (idx->u_of.ensure((size_t)F * std::max<long long>(n, 1) * 4, true) != MFAR_OK) {
        if (bf16) return nomem();
        (void)hipGetLastError();      // fp32 index: optional (without it stage 2 cannot read the scan's score dump)
        g_err.clear();
        idx->u_of.release();
    }
    if (idx->u_repof.ensure((size_t)F * n * 4, true) != MFAR_OK) {   // optional: without it stage 2 gathers every row itself
        (void)hipGetLastError();
        g_err.clear();
        idx->u_repof.release();
    }
    RETCHK(idx->s_stats.ensure((size_t)F * 2 * sizeof(u32)));
    RETCHK(idx->s_field.ensure((size_t)F * sizeof(ScreenField)));
    RETCHK(idx->s_mean.ensure((size_t)F * idx->E * sizeof(float)));
    HIPCHK(hipMemsetAsync(idx->s_stats.p, 0, (size_t)F * 2 * sizeof(u32), st));
    HIPCHK(hipMemsetAsync(idx->s_mean.p, 0, (size_t)F * idx->E * sizeof(float), st));   // (bf16: the "mean" stays the zero vector)
    DevBuf stage_rows, stage_field, rnorm_doc;
    if (bf16) idx->row_mask = idx->row_eligible = 0;
    if (bf16 && (stage_rows.ensure((size_t)std::max<long long>(n, 1) * idx->E * 4) != MFAR_OK ||
                 stage_field.ensure((size_t)idx->field_stride * 4) != MFAR_OK)) {
        stage_rows.release();
        stage_field.release();
        return nomem();
    }
    // field_src(f) = the fp32 tiled rows of field f (a bf16 field: widened exactly into the staging copy)
    auto field_src = [&](int f, const float** out) -> int {
        if (!bf16) {
            *out = (const float*)idx->slab + (size_t)f * idx->field_stride;
            return MFAR_OK;
        }
        HIPCHK(hipMemsetAsync(stage_field.p, 0, (size_t)idx->field_stride * 4, st));
        if (n > 0) {
            const long long t8 = n * (idx->E / 8), t4 = n * (idx->E / 4);
            mfar_untile_rows_bf16_kernel<<<dim3((unsigned)((t8 + 255) / 256)), dim3(256), 0, st>>>(
                (const unsigned short*)idx->slab + (size_t)f * idx->field_stride, stage_rows.as<float>(), 0, n, idx->E);
            mfar_tile_rows_kernel<<<dim3((unsigned)((t4 + 255) / 256)), dim3(256), 0, st>>>(stage_rows.as<float>(), stage_field.as<float>(), 0, n, idx->E);
            HIPCHK(hipGetLastError());
        }
        *out = stage_field.as<float>();
        return MFAR_OK;
    };
    // pass 1: per-field statistics
    if (bf16) {
        for (int f = 0; f < F; ++f) {
            mfar_direct_stats_kernel<<<dim3((unsigned)idx->n_blk), dim3(128), 0, st>>>((const unsigned short*)idx->slab + (size_t)f * idx->field_stride,
                                                                                     idx->n_steps, idx->n_rows, idx->s_stats.as<u32>() + 2 * f);
            HIPCHK(hipGetLastError());
        }
        RETCHK(idx->s_field1.ensure((size_t)F * sizeof(ScreenField)));
        RETCHK(idx->s_cvt.ensure((size_t)F * sizeof(uint2)));
        mfar_direct_fields_kernel<<<dim3(1), dim3(64), 0, st>>>(idx->s_stats.as<u32>(), F, idx->s_field1.as<ScreenField>(), idx->s_field.as<ScreenField>(),
                                                                idx->s_cvt.as<uint2>());
        HIPCHK(hipGetLastError());
    } else {
        // mean vector and statistics of the centred rows (+ every row's norm: ROW MODE)
        RETCHK(idx->s_nsum.ensure((size_t)F * sizeof(float)));
        HIPCHK(hipMemsetAsync(idx->s_nsum.p, 0, (size_t)F * sizeof(float), st));
        const bool rows_ok = idx->row_mode_setting != 0 && (idx->n_steps % 4 == 0 || idx->n_steps % 6 == 0) &&
                             rnorm_doc.ensure((size_t)F * std::max<long long>(n, 1) * sizeof(float)) == MFAR_OK;
        if (!rows_ok) {
            (void)hipGetLastError();
            g_err.clear();
            rnorm_doc.release();
        }
        for (int f = 0; f < F; ++f) {
            const float* src = (const float*)idx->slab + (size_t)f * idx->field_stride;
            float* mean_f = idx->s_mean.as<float>() + (size_t)f * idx->E;
            mfar_screen_mean_kernel<<<dim3((unsigned)((idx->n_blk + 7) / 8), 1), dim3(256), 0, st>>>(src, idx->field_stride, idx->n_steps, idx->n_blk,
                                                                                                    idx->n_rows, mean_f);
            HIPCHK(hipGetLastError());
            mfar_screen_mean_finish_kernel<<<dim3((idx->E + 255) / 256), dim3(256), 0, st>>>(mean_f, idx->E, idx->n_rows);
            HIPCHK(hipGetLastError());
            mfar_screen_stats_kernel<<<dim3((unsigned)idx->n_blk, 1), dim3(256), 0, st>>>(src, idx->field_stride, idx->n_steps, idx->n_rows, mean_f,
                                                                                         idx->s_stats.as<u32>() + 2 * f,
                                                                                         rows_ok ? rnorm_doc.as<float>() + (size_t)f * n : nullptr,
                                                                                         idx->s_nsum.as<float>() + f);
            HIPCHK(hipGetLastError());
        }
        mfar_screen_scale_kernel<<<dim3(1), dim3(64), 0, st>>>(idx->s_stats.as<u32>(), idx->s_mean.as<float>(), F, idx->E,
                                                             idx->s_field.as<ScreenField>(), idx->s_nsum.as<float>(), idx->n_rows, rows_ok ? 1 : 0);
        HIPCHK(hipGetLastError());
        {
            std::vector<ScreenField> hf(F);
            HIPCHK(hipMemcpyAsync(hf.data(), idx->s_field.p, (size_t)F * sizeof(ScreenField), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            const u32 was_active = idx->row_mask;
            idx->row_eligible = 0;
            for (int f = 0; f < F; ++f)
                if (hf[f].row_mode != 0.0f) idx->row_eligible |= 1u << f;
            // a rebuild keeps what was activated; mode 2 activates every eligible field at once
            idx->row_mask = idx->row_mode_setting == 2 ? idx->row_eligible : (was_active ? idx->row_eligible : 0u);
        }
    }
    // pass 2: unique rows, field by field (scratch shared); an fp32 index reads its slab in place
    DevBuf tmp[9];
    int rc = MFAR_OK;
    for (int f = 0; f < F && rc == MFAR_OK; ++f) {
        const float* src = nullptr;
        rc = field_src(f, &src);
        if (rc == MFAR_OK)
            rc = build_unique_rows(idx, f, src, tmp, st, &idx->n_unique[f], &idx->largest_group[f],
                                   bf16 ? idx->rep_bits.as<u64>() + (size_t)f * idx->n_blk : nullptr, idx->u_of.p ? idx->u_of.as<u32>() + (size_t)f * n : nullptr);
        if (bf16) HIPCHK(hipStreamSynchronize(st));   // the staging copy is reused by the next field
    }
    HIPCHK(hipStreamSynchronize(st));
    for (auto& b : tmp) b.release();
    stage_rows.release();
    stage_field.release();
    if (rc == MFAR_ERR_NOMEM) return nomem();
    RETCHK(rc);
    HIPCHK(hipMemcpyAsync(idx->u_n.p, idx->n_unique.data(), (size_t)F * 4, hipMemcpyHostToDevice, st));
    if (bf16) {
        idx->screen_used = 0;
        idx->screen_dirty = false;
        idx->screen_built = true;
        *ok = true;
        return MFAR_OK;
    }
    // geometry of the screen slab: every field holds its unique rows, padded to whole 256-row tiles
    S1Geom& g = idx->geom_screen;
    g.reset(F);
    long long total = 0;
    for (int f = 0; f < F; ++f) {
        long long blk = ((long long)idx->n_unique[f] + 63) / 64;
        blk = std::max(4LL, ((blk + 3) / 4) * 4);
        g.n_rows[f] = idx->n_unique[f];
        g.base[f] = total;
        g.n_tiles[f] = (int)(blk / 4);
        total += blk * 64 * idx->E;
    }
    if (idx->screen.ensure((size_t)total * 2, true) != MFAR_OK) return nomem();
    idx->screen_used = (size_t)total * 2;
    {
        std::vector<long long> db(F);
        for (int f = 0; f < F; ++f) db[f] = g.base[f] / idx->E;
        RETCHK(idx->dump_base.ensure((size_t)F * sizeof(long long)));
        HIPCHK(hipMemcpyAsync(idx->dump_base.p, db.data(), (size_t)F * sizeof(long long), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        // ROW MODE: the norm of every row of the screen slab (unique rows, padded to whole tiles), only when some field needs it
        if (idx->row_eligible && rnorm_doc.p && idx->s_rnorm.ensure((size_t)(total / idx->E) * sizeof(float), true) == MFAR_OK) {
            for (int f = 0; f < F; ++f) {
                const long long n_pad = (long long)g.n_tiles[f] * 256;
                mfar_rownorm_gather_kernel<<<dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st>>>(
                    rnorm_doc.as<float>() + (size_t)f * n, idx->u_rep.as<int>() + (size_t)f * n, idx->n_unique[f], n_pad, idx->s_rnorm.as<float>() + db[f]);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipStreamSynchronize(st));
        } else if (idx->row_eligible) {
            (void)hipGetLastError();
            g_err.clear();
            idx->row_mask = idx->row_eligible = 0;     // no table: every field keeps its field-wide bound (ScreenField::row_mode is ignored without arow)
        }
        // every row's norm as a 10-bit code inside its u_of entry (the score dump's look-up reads the entry anyway: per-row bounds for free)
        idx->uof_packed = false;
        if (idx->u_of.p && *std::max_element(idx->n_unique.begin(), idx->n_unique.end()) < (int)UOF_INDEX_MASK - 1) {
            for (int f = 0; f < F; ++f) {
                mfar_uof_norm_code_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(
                    n, rnorm_doc.p ? rnorm_doc.as<float>() + (size_t)f * n : nullptr, idx->s_field.as<ScreenField>() + f, idx->u_of.as<u32>() + (size_t)f * n);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipStreamSynchronize(st));
            idx->uof_packed = true;
        }
        // ... and transposed, for the fused look-up of the score dump's level (only shapes that may dump: many fields / few rows)
        idx->uof_t_ok = false;
        const double dump_b = (double)(total / idx->E) * 256.0 + 128.0 * F * 100.0 * F * 64.0, gather_b = 128.0 * F * 100.0 * F * (double)GSLAB_ROW_BYTES(idx->E);
        if (idx->u_of.p && n > 0 && (idx->dump_mode == 2 || (idx->dump_mode == 1 && dump_b * 3.0 < gather_b)) &&      // (dump_wanted's rule at k1 = 100)
            idx->u_of_t.ensure((size_t)n * F * 4, true) == MFAR_OK) {
            const long long tot = n * F;
            mfar_uof_transpose_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st>>>(idx->u_of.as<u32>(), n, F, idx->u_of_t.as<u32>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            idx->uof_t_ok = true;
        } else if (idx->u_of.p) {
            (void)hipGetLastError();
            g_err.clear();
            idx->u_of_t.release();
        }
        rnorm_doc.release();
    }
    // pass 3: the fp16 rows
    for (int f = 0; f < F; ++f) {
        const float* src = (const float*)idx->slab + (size_t)f * idx->field_stride;
        const long long n_gran = (long long)g.n_tiles[f] * 4 * idx->n_steps * 128;
        mfar_screen_build_kernel<<<dim3((unsigned)((n_gran + 255) / 256)), dim3(256), 0, st>>>(
            src, (_Float16*)idx->screen.p + g.base[f], n_gran, idx->n_steps, idx->n_unique[f], idx->u_rep.as<int>() + (size_t)f * n,
            idx->s_mean.as<float>() + (size_t)f * idx->E, idx->s_field.as<ScreenField>() + f);
        HIPCHK(hipGetLastError());
    }
    // the fp16 gather slab of an fp32 index: the same centred + scaled values, every document's row, row-major (the approximate
    // level of the certified two-level stage 2; +50 % of the fp32 slab, optional)
    idx->gslab_ok = false;
    static const bool gs_enabled = !(getenv("MFAR_GATHER_SLAB") && atoi(getenv("MFAR_GATHER_SLAB")) == 0);
    if (gs_enabled && n > 0 && gslab_alloc(idx)) {
        const long long total = n * (long long)(idx->g_row_bytes / 16);
        for (int f = 0; f < F; ++f) {
            mfar_gslab_build_f16_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(
                (const float*)idx->slab + (size_t)f * idx->field_stride, idx->gslab.as<char>() + (size_t)f * n * idx->g_row_bytes, n, idx->n_steps,
                (int)idx->g_row_bytes, idx->s_mean.as<float>() + (size_t)f * idx->E, idx->s_field.as<ScreenField>() + f);
            HIPCHK(hipGetLastError());
        }
        idx->gslab_ok = true;
    }
    idx->screen_dirty = false;
    idx->screen_built = true;
    *ok = true;
    return MFAR_OK;
}

// the wide pass exists as 6-slot and 4-slot register-ring kernels: the k-steps must divide into one of them
static bool wide_ok(const mfar_index* idx) { return idx->wide && (idx->n_steps % 6 == 0 || idx->n_steps % 4 == 0); }

// Should the wide screened pass of this index write its score dump for stage 2 (mfar_select.h mfar_s2_lookup_kernel)?  The dump costs
// rows x 256 bytes of writes per launch (16-bit codes; + one 64-byte sector per looked-up pair); the row gathers it replaces cost 128 queries x
// (F k1 candidates) x F fields x E x 2 bytes.  Auto: when the dump moves less than a third of that -- measured: 129 375 x 22 (1.8 GB against
// 9.5 GB) 48.3 k -> 64.2 k queries/s; the 125 k x 8 row shard (0.56 against 1.26 GB) 185 k -> 182 k: the dump's looser bound keeps ~15 %
// more survivors and its stores sit in the scan; 1 M x 8 (4.1 against 1.26 GB) 51.6 k -> 39.8 k.
static bool dump_wanted(const mfar_index* idx, int k1) {
    if (idx->dump_mode == 0 || idx->dtype != MFAR_DTYPE_F32 || !idx->u_of.p || !idx->u_repof.p || !idx->screen_built || idx->screen_dirty ||
        idx->stage2_mode < 1 || !idx->gslab_ok)
        return false;
    if (idx->dump_mode == 2) return true;
    const double dump = (double)(idx->screen_used / 2 / (size_t)idx->E) * 256.0 + 128.0 * idx->F * k1 * idx->F * 64.0;
    const double gather = 128.0 * idx->F * k1 * idx->F * (double)idx->g_row_bytes;
    return dump * 3.0 < gather;
}

// ------------------------------------------------------------------------------------------------ bf16 index: the exhaustive chain pass
// (mfar_exact16.h) for queries q0 .. q0 + nq of fields [f0, f0 + nf) whose flag is set (flags == nullptr: every field) -> final lists.
static int exact16_pass(mfar_index* idx, mfar_index::S1Slot& sl, const float* q, int q0, int nq, int k, int sentinel, int f0, int nf, long long* fid,
                        float* fsc, const int* flags, hipStream_t st) {
    const long long n_pad = idx->n_blk * 64;
    // scratch: one score per (query of the block, row).  The block is CHAIN_QB queries when that fits (256 MB per slot at 1 M rows) and is
    // halved until the allocation succeeds: a repair must not turn a search into MFAR_ERR_NOMEM while a smaller block still runs
    // (every halving re-reads the field once more; one query per block is the floor)
    int qblock = CHAIN_QB;
    if (sl.chain.cap >= (size_t)n_pad * sizeof(float)) qblock = (int)std::min<size_t>(CHAIN_QB, sl.chain.cap / ((size_t)n_pad * sizeof(float)));
    else
        for (;; qblock >>= 1) {
            if (sl.chain.ensure((size_t)qblock * n_pad * sizeof(float), true) == MFAR_OK) break;
            if (qblock == 1) return MFAR_ERR_NOMEM;          // (g_err holds the failed size)
            (void)hipGetLastError();
            g_err.clear();
        }
    ChainScanParams sp = {};
    sp.slab = (const unsigned short*)idx->slab;
    sp.field_stride = idx->field_stride;
    sp.q = q;
    sp.scores = sl.chain.as<float>();
    sp.flags = flags;
    sp.n_steps = idx->n_steps;
    sp.E = idx->E;
    sp.n_blk = idx->n_blk;
    sp.n_pad = n_pad;
    ChainSelectParams cp = {};
    cp.scores = sl.chain.as<float>();
    cp.flags = flags;
    cp.out_ids = fid;
    cp.out_scores = fsc;
    cp.n_rows = idx->n_rows;
    cp.n_pad = n_pad;
    cp.row_offset = idx->row_offset;
    cp.nf = nf;
    cp.k = k;
    cp.sentinel = sentinel;
    const int QT = idx->E <= 1024 ? 32 : (idx->E <= 4096 ? 8 : 2);
    const dim3 sg((unsigned)((idx->n_blk + 3) / 4)), sb(256);
    const size_t lds = (size_t)QT * idx->E * 4;
    for (int f = f0; f < f0 + nf; ++f)
        for (int b0 = 0; b0 < nq; b0 += qblock) {
            sp.field = cp.field = f;
            sp.q0 = cp.q0 = q0 + b0;
            sp.nq = std::min(qblock, nq - b0);
            cp.fo = f - f0;
            if (QT == 32) mfar_chain_scan_bf16_kernel<32><<<sg, sb, lds, st>>>(sp);
            else if (QT == 8) mfar_chain_scan_bf16_kernel<8><<<sg, sb, lds, st>>>(sp);
            else mfar_chain_scan_bf16_kernel<2><<<sg, sb, lds, st>>>(sp);
            HIPCHK(hipGetLastError());
            mfar_chain_select_kernel<<<dim3(sp.nq), dim3(256), 0, st>>>(cp);
            HIPCHK(hipGetLastError());
        }
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ adaptive policy of the certified screen
// The certificate is data dependent.  A list fails when more than k' - k of its field's unique rows sit within ~2 eps of the k-th best
// score -- near-duplicate rows (product variants, templated texts, one text encoded in different batches: not bit-identical, so the
// unique-row build cannot merge them) do exactly that, list after list.  Every failure costs the exact pass of that field ON TOP of the
// screen, and a caller that only asked for a report pays a pipeline drain and a second launch.  Three decisions keep the worst case at the
// price of the exact pass, all taken from the certificate flags of launches that have FINISHED (copied to pinned host memory behind the
// certify kernel, picked up by a later call once their event has completed; nothing waits):
//   * AUTO-OFF, per field: a field that failed in >= off_fails of its last 16 screened launches is switched off -- the screen's chunk
//     table leaves it out (the other fields share the whole grid), and the exact fp32 pass writes its lists straight from the begin phase,
//     on the scan stream.  Every probe_every-th launch screens it anyway (a PROBE: certificate evaluated, lists discarded); on_clean clean
//     probes in a row switch it back on.  With every field off a launch is the exact pass and nothing else.
//   * inline repair: when >= 4 of the last 16 launches had a failure among the fields that are ON, mfar_stage1_finish repairs on the
//     device even when asked to report only (no drain, nothing done twice); back to reporting when at most one of the last 16 had.
//   * ROW MODE (mfar_screen.h) is activated for eligible fields by the first failure.
// Results do not depend on any of this: a list is either certified or written by the exact pass.
__global__ void mfar_mask_flags_kernel(int* __restrict__ flags, u32 mask) {
    if (threadIdx.x < MFAR_MAX_FIELDS) flags[threadIdx.x] = (int)((mask >> threadIdx.x) & 1u);
}
static void consume_feedback(mfar_index* idx) {
    for (auto& sl : idx->s1) {
        if (!sl.fb_pending || hipEventQuery(sl.fb_ev) != hipSuccess) continue;
        sl.fb_pending = false;
        const int* h = sl.fb_host;
        const bool failed = idx->pol.feed(idx->F, h, h + MFAR_MAX_FIELDS + 2, h[MFAR_MAX_FIELDS], sl.fb_screened, sl.fb_probed, idx->dtype == MFAR_DTYPE_BF16,
                                          h[SCREEN_FLAG_T1]);
        if ((failed || h[SCREEN_FLAG_T1]) && idx->row_mode_setting == 1) idx->row_mask = idx->row_eligible;       // ROW MODE for heavy-tailed fields (mfar_screen.h)
        idx->pol.feed_deep(idx->F, h + SCREEN_T2_FIELDS, h, sl.fb_screened & ~sl.fb_deep, sl.fb_deep);
        if (sl.fb_screened) {
            bool want = h[SCREEN_FLAG_T2_WANT] != 0;
            for (int f = 0; f < idx->F; ++f) want = want || h[SCREEN_T2_RESCAN_FIELDS + f] != 0;
            idx->pol.feed_rescan(want);
        }
    }
}
// behind the certify kernel of a batch: its flags -> pinned host memory, event behind the copy
static int post_feedback(mfar_index* idx, mfar_index::S1Slot& sl, u32 screened, u32 probed, hipStream_t st) {
    if (!sl.fb_host) {
        HIPCHK(hipHostMalloc((void**)&sl.fb_host, SCREEN_FLAGS * sizeof(int), hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&sl.fb_ev, hipEventDisableTiming));
    }
    HIPCHK(hipMemcpyAsync(sl.fb_host, sl.fail.p, SCREEN_FLAGS * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(sl.fb_ev, st));
    sl.fb_pending = true;
    sl.fb_screened = screened;
    sl.fb_probed = probed;
    return MFAR_OK;
}

// One block of queries (rows q0 .. of q) through stage 1 for fields [f0, f0 + nf): 64 per block, or up to 128 when the
// wide screened pass applies (more than 64 queries left, screen available).  all pointers are device pointers; fid/fsc are
// [Q, nf, k].  *n_done (may be nullptr) = queries of this block.
//   any_fail_out  nullptr: a failed certificate is repaired here by the exact pass (always launched, idle when nothing
//                 failed); non-null (device int): only report -- the caller re-runs the batch exactly when it reads != 0
static int stage1_block(mfar_index* idx, int slot, int phases, const float* q, int Q, int q0, int k, int sentinel, int f0, int nf,
                        long long* fid, float* fsc, int* any_fail_out, hipStream_t st, int* n_done = nullptr) {
    mfar_index::S1Slot& sl = idx->s1[slot];
    const float tau0 = sentinel ? 0.0f : -INFINITY;
    const int F = idx->F, kp = std::min(k + SCREEN_EXTRA, SCREEN_MAX_KP);
    const bool bf16 = idx->dtype == MFAR_DTYPE_BF16;
    if (phases & S1_PREPARE) {
        consume_feedback(idx);
        RETCHK(order_after_writes(idx, st));      // rows written asynchronously on another stream land before this batch reads any
        bool screened = false;
        if (screen_wanted(idx, k)) RETCHK(ensure_screen(idx, st, &screened));
        sl.screened = screened;
        sl.qw = (screened && wide_ok(idx) && Q - q0 > 64) ? 128 : 64;
        // AUTO-OFF (all-fields passes of an fp32 index): which fields the exact pass writes / the screen leaves out in this batch
        sl.exact_mask = sl.skip_mask = 0;
        if (screened && f0 == 0 && nf == idx->F) idx->pol.plan(F, &sl.exact_mask, &sl.skip_mask);
        // the wide pass of an fp32 index over all fields may leave its scores behind for stage 2 (one block of queries: the dump holds
        // the launch that wrote it last; a launch that leaves fields out has no scores for them)
        sl.dump_on = sl.qw == 128 && !bf16 && f0 == 0 && nf == idx->F && q0 == 0 && Q <= 128 && sl.skip_mask == 0 && dump_wanted(idx, k);
        sl.dump_ready = false;
        if (sl.dump_on && (sl.dump.ensure(idx->screen_used / 2 / (size_t)idx->E * 256, true) != MFAR_OK || sl.dinv.ensure((size_t)F * 128 * 4) != MFAR_OK ||
                           sl.dstep.ensure((size_t)F * 128 * 4) != MFAR_OK || sl.eps_dump.ensure((size_t)F * 128 * 4) != MFAR_OK || sl.darel.ensure((size_t)F * 128 * 4) != MFAR_OK)) {
            (void)hipGetLastError();
            g_err.clear();
            sl.dump_on = false;
        }
        if (!sl.dump_on && sl.dump.p && !dump_wanted(idx, k)) sl.dump.release();      // the shape no longer wants it (rows rewritten, mode changed)
        // TIER 2 behind this batch's certificate (all-fields passes; armed by the policy: mfar_policy.h).  A bf16 index rescans its own slab
        // and re-scores from the row-major companion: the lists tier 2 finishes carry the natural-order chain's bits like the certified ones
        sl.t2 = screened && f0 == 0 && nf == idx->F && (idx->tier2_mode == 2 || (idx->tier2_mode == 1 && idx->pol.t2_armed) || (idx->tier2_mode && idx->deep_mode == 2));
        if (sl.t2 && (sl.tau2.ensure((size_t)F * 128 * 4) != MFAR_OK || sl.lfail.ensure((size_t)128 * F * 4) != MFAR_OK || sl.t2cnt.ensure((size_t)128 * F * 4) != MFAR_OK ||
                      sl.t2cand.ensure((size_t)128 * F * T2_CAP * 8, true) != MFAR_OK || sl.t2sx.ensure((size_t)128 * F * T2_CAP * 4, true) != MFAR_OK)) {
            (void)hipGetLastError();      // optional: without its scratch the batch keeps the exact pass as its only fall-back
            g_err.clear();
            sl.t2 = false;
        }
        // the rescan behind collect pass A: always (tests: mode 2, the forced fallback), or while the policy has seen a list ask for it
        sl.t2_rescan = sl.t2 && (idx->tier2_mode == 2 || idx->t2_force_rescan || idx->pol.t2_rescan_armed);
        // DEEP SCAN fields of the batch (fp32 index): no first attempt, the scan collects the complete candidate set (stage1_pass drops
        // the mask again when this shape runs no light sample pass)
        sl.deep_mask = 0;
        if (sl.t2 && !bf16 && idx->deep_mode) {
            const u32 allm = F >= 32 ? 0xFFFFFFFFu : ((1u << F) - 1u);
            sl.deep_mask = (idx->deep_mode == 2 ? allm : idx->pol.deep_mask) & allm & ~sl.exact_mask & ~sl.skip_mask;
            if (sl.deep_mask && sl.deepinfo.ensure((size_t)F * 128 * sizeof(float4)) != MFAR_OK) {
                (void)hipGetLastError();
                g_err.clear();
                sl.deep_mask = 0;
            }
        }
    }
    const int qw = sl.qw;
    const u32 all_mask = F >= 32 ? 0xFFFFFFFFu : ((1u << F) - 1u);
    const bool none_screened = sl.screened && sl.skip_mask == all_mask;          // every field is switched off: the exact pass is the launch
    const int qt_n = std::min(qw, Q - q0);
    if (n_done) *n_done = qt_n;
    RETCHK(sl.qt.ensure((size_t)idx->n_steps * (bf16 ? 8192 : 4096)));
    if (!sl.screened) {
        if (phases & S1_PREPARE) {
            if (bf16) {
                const int total = 64 * (idx->E / 8);
                mfar_tile_queries_bf16_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(q, sl.qt.as<unsigned short>(), q0, Q, idx->E);
            } else {
                const int total = 64 * (idx->E / 4);
                mfar_tile_queries_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(q, sl.qt.as<float>(), q0, Q, idx->E);
            }
            HIPCHK(hipGetLastError());
        }
        const S1Out o = {fid, fsc, nullptr, q0, sentinel, idx->row_offset};
        RETCHK(stage1_pass(idx, sl, idx->geom_docs, f0, nf, phases, bf16 ? S1_BF16 : S1_F32, idx->slab, sl.qt.p, qt_n, k, tau0, nullptr,
                           nullptr, true, o, st));
        if ((phases & S1_CERTIFY) && any_fail_out) HIPCHK(hipMemsetAsync(any_fail_out, 0, 4, st));
        return MFAR_OK;
    }
    RETCHK(sl.qt16.ensure((size_t)idx->n_steps * 8192));
    RETCHK(sl.qinfo.ensure(128 * sizeof(ScreenQuery)));
    RETCHK(sl.eps.ensure((size_t)F * 128 * 4));
    RETCHK(sl.base.ensure((size_t)F * 128 * 4));
    if (!sl.fail.p) {
        RETCHK(sl.fail.ensure((size_t)SCREEN_FLAGS * 4));
        HIPCHK(hipMemsetAsync(sl.fail.p, 0, (size_t)SCREEN_FLAGS * 4, st));
    }
    RETCHK(sl.sids.ensure((size_t)128 * F * kp * 8));
    RETCHK(sl.ssc.ensure((size_t)128 * F * kp * 4));
    RETCHK(sl.sx.ensure((size_t)128 * F * kp * 4));
    RETCHK(sl.scnt.ensure((size_t)128 * F * 4));
    int* fflags = sl.fail.as<int>();
    // ROW MODE: some field of this fp32 index ranks its rows by upper bounds (heavy-tailed row norms)
    if (phases & S1_PREPARE) {
        sl.row_mode = !bf16 && qw == 128 && idx->row_mask != 0 && idx->s_rnorm.p != nullptr;     // (the wide pass has the row-norm code)
        sl.row_mask = sl.row_mode ? idx->row_mask : 0u;
    }
    const bool row_mode = sl.row_mode;
    if (row_mode) {
        RETCHK(sl.arow.ensure((size_t)F * 128 * 4));
        RETCHK(sl.eps_cert.ensure((size_t)F * 128 * 4));
    } else if (sl.arow.p) {
        sl.arow.release();          // (the flag stage1_pass looks at)
        sl.eps_cert.release();
    }
    // bf16 index, which certified pass: 64 columns = two bf16 query terms over the raw rows (scale 1); 128 columns = the rows converted to
    // fp16 in registers against one fp16 query term (the field's power-of-two scale), or -- MFAR_BF16_WIDE_TERMS=2, diagnostic -- two
    // bf16 terms at twice the MFMAs
    static const bool wide_terms2 = getenv("MFAR_BF16_WIDE_TERMS") && atoi(getenv("MFAR_BF16_WIDE_TERMS")) == 2;
    const int bkind = qw == 128 ? (wide_terms2 ? S1_BF16W : S1_BF16C) : S1_BF16S;
    const ScreenField* sfield = (bf16 && bkind != S1_BF16C) ? idx->s_field1.as<ScreenField>() : idx->s_field.as<ScreenField>();
    // 1. screened pass: the k' best approximate scores per (query, field)
    if (phases & S1_PREPARE) {
        if (bf16 && bkind != S1_BF16C)
            mfar_direct_queries_kernel<<<dim3(qw), dim3(256), 0, st>>>(q, (unsigned short*)sl.qt16.p, sl.qinfo.as<ScreenQuery>(), sfield,
                                                                       sl.eps.as<float>(), sl.base.as<float>(), fflags, q0, Q, idx->E, F,
                                                                       idx->screen_eps_mult, qw);
        else
            mfar_screen_queries_kernel<<<dim3(qw), dim3(256), 0, st>>>(q, (_Float16*)sl.qt16.p, sl.qinfo.as<ScreenQuery>(), sfield,
                                                                       sl.eps.as<float>(), sl.base.as<float>(), fflags, q0, Q, idx->E, F,
                                                                       idx->screen_eps_mult, qw, bf16 ? 1 : 0, row_mode ? sl.arow.as<float>() : nullptr,
                                                                       row_mode ? sl.eps_cert.as<float>() : nullptr, sl.row_mask,
                                                                       sl.dump_on ? sl.dinv.as<float>() : nullptr, sl.dstep.as<float>(), sl.eps_dump.as<float>(), sl.darel.as<float>());
        HIPCHK(hipGetLastError());
    }
    // lists of unique-row numbers (fp32 index: rows of the screen slab) / of the local rows of group representatives (bf16 index:
    // the pass scans the documents themselves)
    const S1Out so = {sl.sids.as<long long>(), sl.ssc.as<float>(), sl.scnt.as<int>(), 0, 0, 0};
    // AUTO-OFF: the exact fp32 pass writes the final lists of the switched-off fields, here, on the scan stream, ahead of the screened scan
    // of the others (it shares the slot's chunk-list scratch with it: same stream, so one after the other).  A restricted pass walks the
    // finely cut table with the whole GPU, like a repair; with every field off it is the plain all-fields pass.
    if (sl.exact_mask && (phases & S1_SCAN)) {
        const bool all_off = sl.exact_mask == all_mask;
        if (!all_off) {
            RETCHK(sl.off_flags.ensure(MFAR_MAX_FIELDS * sizeof(int)));
            mfar_mask_flags_kernel<<<dim3(1), dim3(64), 0, st>>>(sl.off_flags.as<int>(), sl.exact_mask);
            HIPCHK(hipGetLastError());
        }
        if (bf16) RETCHK(exact16_pass(idx, sl, q, q0, qt_n, k, sentinel, f0, nf, fid, fsc, all_off ? nullptr : sl.off_flags.as<int>(), st));
        for (int b0 = 0; b0 < qt_n && !bf16; b0 += 64) {
            const int total = 64 * (idx->E / 4);
            mfar_tile_queries_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(q, sl.qt.as<float>(), q0 + b0, Q, idx->E);
            HIPCHK(hipGetLastError());
            const S1Out o = {fid, fsc, nullptr, q0 + b0, sentinel, idx->row_offset};
            RETCHK(stage1_pass(idx, sl, idx->geom_docs, f0, nf, S1_ALL, S1_F32, idx->slab, sl.qt.p, std::min(64, qt_n - b0), k, tau0, nullptr,
                               all_off ? nullptr : sl.off_flags.as<int>(), none_screened, o, st, 0, true));
        }
    }
    if (none_screened) {
        // nothing to screen, re-score or certify (the flags were cleared by the query kernel: the feedback of this batch reports no field)
        if ((phases & S1_CERTIFY) && any_fail_out) HIPCHK(hipMemsetAsync(any_fail_out, 0, 4, st));
        return MFAR_OK;
    }
    if (bf16)
        RETCHK(stage1_pass(idx, sl, idx->geom_docs, f0, nf, phases, bkind, idx->slab, sl.qt16.p, qt_n, kp, -INFINITY,
                           sl.base.as<float>(), nullptr, true, so, st, sl.skip_mask));
    else {
        S1DeepDev dd = {};
        if (sl.deep_mask) {
            dd.mask = sl.deep_mask;
            dd.row_mask = sl.row_mode ? sl.row_mask : 0u;
            dd.k = k;
            dd.sentinel = sentinel;
            dd.E = idx->E;
            dd.Q = qt_n;
            dd.eps = sl.eps.as<float>();
            dd.qinfo = sl.qinfo.as<ScreenQuery>();
            dd.sf = sfield;
            dd.q = q + (size_t)q0 * idx->E;
            dd.mean = idx->s_mean.as<float>();
            dd.info = sl.deepinfo.as<float4>();
        }
        RETCHK(stage1_pass(idx, sl, idx->geom_screen, f0, nf, phases, qw == 128 ? S1_F16W : S1_F16, idx->screen.p, sl.qt16.p, qt_n, kp, -INFINITY,
                           sl.base.as<float>(), nullptr, true, so, st, sl.skip_mask, false, false, sl.deep_mask ? &dd : nullptr));
    }
    if ((phases & S1_SCAN) && sl.dump_on) {
        sl.dump_ready = true;
        sl.dump_q = q + (size_t)q0 * idx->E;
        sl.dump_Q = qt_n;
    }
    if (!(phases & S1_CERTIFY)) return MFAR_OK;
    // 2. exact scores of those unique rows' representatives (the contract's fma chain over the fp32 slab)
    ScoreParams sp = {};
    sp.slab = idx->slab;
    sp.field_stride = idx->field_stride;
    sp.q = q + (size_t)q0 * idx->E;
    sp.cand = sl.sids.as<long long>();
    sp.n_cand = nullptr;
    sp.out = sl.sx.as<float>();
    sp.row_offset = idx->row_offset;
    sp.n_rows = (int)idx->n_rows;
    sp.n_steps = idx->n_steps;
    sp.E = idx->E;
    sp.F = nf;
    sp.C = kp;
    sp.per_field = 1;
    if (bf16) sp.row_offset = 0;            // the lists hold local rows
    else {
        sp.urep = idx->u_rep.as<int>();     // ... unique-row numbers: the representative document is gathered
        sp.nuniq = idx->u_n.as<int>();
    }
    sp.ustride = idx->n_rows;
    sp.f0 = f0;
    bool rows16 = false;
    if (bf16) RETCHK(ensure_rows16(idx, st, &rows16));
    static const bool prefix = !(getenv("MFAR_RESCORE_PREFIX") && atoi(getenv("MFAR_RESCORE_PREFIX")) == 0);   // diagnostic: 0 = re-score all k' rows
    if (prefix) {      // rows that cannot reach the exact top-k of their list are not gathered (mfar_select.h ScoreParams::pre_sc)
        sp.pre_sc = sl.ssc.as<float>();
        sp.pre_eps = sl.eps.as<float>();
        sp.pre_qinfo = sl.qinfo.as<ScreenQuery>();
        sp.pre_k = k;
        sp.pre_qw = qw;
        sp.sfld = sfield;
    }
    sp.gslab = idx->gslab.p;
    sp.g_row_bytes = (long long)idx->g_row_bytes;
    const dim3 sgrid((unsigned)((kp * nf + SCF_THREADS - 1) / SCF_THREADS), qt_n);
    if (bf16 && rows16) mfar_score_rows_kernel<SRC_BF16G><<<sgrid, dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(sp);
    else if (bf16) mfar_score_candidates_kernel<1><<<dim3((unsigned)((kp * nf + 255) / 256), qt_n), dim3(256), SCORE_LDS_BYTES(idx->E), st>>>(sp);
    else mfar_score_rows_kernel<SRC_F32><<<sgrid, dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(sp);
    HIPCHK(hipGetLastError());
    // 3. exact top-k documents + certificate
    CertifyParams cp = {};
    cp.sid = sl.sids.as<long long>();
    cp.ssc = sl.ssc.as<float>();
    cp.scnt = sl.scnt.as<int>();
    cp.sx = sl.sx.as<float>();
    cp.sf = sfield;
    cp.qinfo = sl.qinfo.as<ScreenQuery>();
    cp.eps = sl.eps.as<float>();
    cp.eps_cert = row_mode ? sl.eps_cert.as<float>() : nullptr;
    cp.q = q + (size_t)q0 * idx->E;
    cp.mean = idx->s_mean.as<float>();
    cp.E = idx->E;
    cp.out_ids = fid;
    cp.out_scores = fsc;
    cp.fail = fflags;
    cp.ustart = idx->u_start.as<int>();
    cp.ucount = idx->u_count.as<int>();
    cp.members = idx->u_members.as<int>();
    cp.uof = bf16 ? idx->u_of.as<u32>() : nullptr;
    cp.ustride = idx->n_rows;
    cp.row_offset = idx->row_offset;
    cp.f0 = f0;
    cp.nf = nf;
    cp.k = k;
    cp.kp = kp;
    cp.q0 = q0;
    cp.sentinel = sentinel;
    cp.qw = qw;
    cp.skip_mask = sl.skip_mask;
    cp.quiet_mask = sl.exact_mask & ~sl.skip_mask;        // a probe launch: switched-off fields that were screened anyway
    const bool t2_run = sl.t2 && (!bf16 || rows16);       // (a bf16 index re-scores tier 2's candidates from its row-major companion)
    if (t2_run) {
        cp.tau2 = sl.tau2.as<float>();
        cp.lfail = sl.lfail.as<int>();
        cp.deep_mask = sl.deep_mask;
    }
    static const bool cert_debug = getenv("MFAR_CERT_DEBUG") != nullptr;
    DevBuf dbg;
    if (cert_debug) {
        RETCHK(dbg.ensure((size_t)qt_n * nf * 8 * 4));
        cp.dbg = dbg.as<float>();
    }
    mfar_screen_certify_kernel<<<dim3(qt_n * nf), dim3(256), 0, st>>>(cp);
    HIPCHK(hipGetLastError());
    if (cert_debug) {   // diagnostics: failed certificates with their numbers (synchronises)
        std::vector<float> h((size_t)qt_n * nf * 8);
        HIPCHK(hipMemcpyAsync(h.data(), dbg.p, h.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int i = 0; i < qt_n * nf; ++i)
            if (h[8 * i] == 0.0f)
                fprintf(stderr, "[mfar cert] q=%d f=%d bound=%.7g T_k=%.7g a_real=%.7g eps=%.4g cnt=%g m_out=%g (ovf + 10 n + 1e4 total)=%g\n", q0 + i / nf,
                        f0 + i % nf, h[8 * i + 1], h[8 * i + 2], h[8 * i + 3], h[8 * i + 4], h[8 * i + 5], h[8 * i + 6], h[8 * i + 7]);
        dbg.release();
    }
    if (t2_run) {
        // TIER 2 (mfar_screen.h): the complete candidate sets of the failed lists -- from the chunk lists the batch's own scan wrote where
        // those provably hold them (collect pass A), else by rescanning the fields that hold such lists with their fixed thresholds (pass B)
        // -> exact scores -> the certify kernel again, on those lists only.  Every kernel is idle when nothing failed.
        const S1Geom& g2 = bf16 ? idx->geom_docs : idx->geom_screen;
        const S1Table& tb2 = qw == 128 ? (sl.skip_mask ? g2.all_w_skip : g2.all_w) : (sl.skip_mask ? g2.all_skip : g2.all);
        T2CollectParams tc = {};
        tc.lists = sl.lists.as<uint2>();
        tc.list_cnt = sl.list_cnt.as<int>();
        tc.fchunk = tb2.d_fchunk.as<int>();
        tc.lfail = tc.lfail_out = sl.lfail.as<int>();
        tc.cand = sl.t2cand.as<long long>();
        tc.cnt = sl.t2cnt.as<int>();
        tc.f0 = f0;
        tc.nf = nf;
        tc.qw = qw;
        tc.kp = kp;
        tc.stats = fflags;
        tc.tau2 = sl.tau2.as<float>();
        tc.scan_tau = sl.scan_tau;
        tc.deep_mask = sl.deep_mask;
        tc.k = k;
        tc.sentinel = sentinel;
        tc.info = sl.deepinfo.as<float4>();
        static const bool t2_first = !(getenv("MFAR_T2_FIRST_SCAN") && atoi(getenv("MFAR_T2_FIRST_SCAN")) == 0);   // diagnostic: 0 = always rescan
        tc.no_first = t2_first && !idx->t2_force_rescan ? 0 : 1;
        tc.rescan_on = sl.t2_rescan ? 1 : 0;
        mfar_t2_collect_kernel<<<dim3(qt_n * nf), dim3(T2_COLLECT_THREADS), 0, st>>>(tc);
        HIPCHK(hipGetLastError());
        if (sl.deep_mask) {
            mfar_t2_collect_deep_kernel<<<dim3(qt_n * nf), dim3(256), T2_COLLECT_LDS_BYTES, st>>>(tc);
            HIPCHK(hipGetLastError());
        }
        if (sl.t2_rescan) {
            // (a full-width scan kernel even when no field is flagged -- its workgroups exit at once, but they cannot START before the next
            //  launch's scan lets go of the register file: enqueued only while the policy has seen a list ask for it)
            if (bf16)
                RETCHK(stage1_pass(idx, sl, idx->geom_docs, f0, nf, S1_SCAN, bkind, idx->slab, sl.qt16.p, qt_n, kp, -INFINITY, sl.tau2.as<float>(),
                                   fflags + SCREEN_T2_RESCAN_FIELDS, false, so, st, sl.skip_mask, false, true));
            else
                RETCHK(stage1_pass(idx, sl, idx->geom_screen, f0, nf, S1_SCAN, qw == 128 ? S1_F16W : S1_F16, idx->screen.p, sl.qt16.p, qt_n, kp, -INFINITY,
                                   sl.tau2.as<float>(), fflags + SCREEN_T2_RESCAN_FIELDS, false, so, st, sl.skip_mask, false, true));
            tc.pass_b = 1;
            mfar_t2_collect_kernel<<<dim3(qt_n * nf), dim3(T2_COLLECT_THREADS), 0, st>>>(tc);
            HIPCHK(hipGetLastError());
        }
        ScoreParams s2 = sp;
        s2.cand = sl.t2cand.as<long long>();
        s2.out = sl.t2sx.as<float>();
        s2.C = T2_CAP;
        s2.n_cand_pf = sl.t2cnt.as<int>();
        s2.pre_sc = nullptr;
        if (bf16) mfar_score_rows_kernel<SRC_BF16G><<<dim3((unsigned)((T2_CAP * nf) / SCF_THREADS), qt_n), dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(s2);
        else mfar_score_rows_kernel<SRC_F32><<<dim3((unsigned)((T2_CAP * nf) / SCF_THREADS), qt_n), dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(s2);
        HIPCHK(hipGetLastError());
        T2SelectParams ts = {};
        ts.cand = sl.t2cand.as<long long>();
        ts.sx2 = sl.t2sx.as<float>();
        ts.cnt = sl.t2cnt.as<int>();
        ts.lfail = sl.lfail.as<int>();
        ts.sid = sl.sids.as<long long>();
        ts.sx = sl.sx.as<float>();
        ts.scnt = sl.scnt.as<int>();
        ts.stats = fflags;
        ts.nf = nf;
        ts.kp = kp;
        ts.k = k;
        mfar_t2_select_kernel<<<dim3(qt_n * nf), dim3(256), T2_SELECT_LDS_BYTES, st>>>(ts);
        HIPCHK(hipGetLastError());
        CertifyParams cp2 = cp;
        cp2.pass2 = 1;
        cp2.dbg = nullptr;
        mfar_screen_certify_kernel<<<dim3(qt_n * nf), dim3(256), 0, st>>>(cp2);
        HIPCHK(hipGetLastError());
    }
    {
        u32 fields = 0;
        for (int f = f0; f < f0 + nf; ++f) fields |= 1u << f;
        idx->screen_checked += (long long)qt_n * __builtin_popcount(fields & ~sl.exact_mask);
        sl.fb_deep = t2_run ? sl.deep_mask : 0u;
        RETCHK(post_feedback(idx, sl, fields & ~sl.exact_mask, fields & cp.quiet_mask, st));
    }
    if (any_fail_out) {   // report only: the caller repairs
        HIPCHK(hipMemcpyAsync(any_fail_out, fflags + MFAR_MAX_FIELDS, 4, hipMemcpyDeviceToDevice, st));
        return MFAR_OK;
    }
    // 4. fall-back: the exact fp32 pass over the DOCUMENTS of the fields whose certificate failed (workgroups of other
    //    fields exit at once)
    // (bf16 index: the exhaustive CHAIN pass -- the plain MFMA pass would leave other bits in the repaired lists, mfar_exact16.h)
    if (bf16) return exact16_pass(idx, sl, q, q0, qt_n, k, sentinel, f0, nf, fid, fsc, fflags, st);
    for (int b0 = 0; b0 < qt_n; b0 += 64) {   // the exact pass takes 64 queries at a time
        const int total = 64 * (idx->E / 4);
        mfar_tile_queries_kernel<<<dim3((total + 255) / 256), dim3(256), 0, st>>>(q, sl.qt.as<float>(), q0 + b0, Q, idx->E);
        HIPCHK(hipGetLastError());
        const S1Out o = {fid, fsc, nullptr, q0 + b0, sentinel, idx->row_offset};
        RETCHK(stage1_pass(idx, sl, idx->geom_docs, f0, nf, S1_ALL, S1_F32, idx->slab, sl.qt.p, std::min(64, qt_n - b0), k, tau0,
                           nullptr, fflags, false, o, st));
    }
    return MFAR_OK;
}

static int run_stage1(mfar_index* idx, const float* q, int Q, int k, int sentinel, long long* fid, float* fsc, hipStream_t st) {
    for (int q0 = 0, n = 0; q0 < Q; q0 += n) RETCHK(stage1_block(idx, 0, S1_ALL, q, Q, q0, k, sentinel, 0, idx->F, fid, fsc, nullptr, st, &n));
    return MFAR_OK;
}

// queries one split-phase batch may hold: 128 when the wide screened pass serves blocks of more than 64 queries, else 64
static int max_split_batch(const mfar_index* idx, int k) { return (screen_wanted(idx, k) && wide_ok(idx)) ? 128 : 64; }
static int ensure_screen(mfar_index* idx, hipStream_t st, bool* ok);
extern "C" int mfar_max_split_batch(mfar_index* idx, int k) {
    if (!idx || k <= 0 || k > MFAR_MAX_K) return 0;
    if (max_split_batch(idx, k) == 64) return 64;
    // the wide pass needs the screen slab: build it now (synchronously), so that the answer holds for the batches to come
    bool ok = false;
    if (hipSetDevice(idx->device) != hipSuccess || ensure_screen(idx, nullptr, &ok) != MFAR_OK || !ok) return 64;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return 64;
    return 128;
}
extern "C" int mfar_set_repair_mode(mfar_index* idx, int thorough) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    idx->repair_sample = thorough != 0;
    return MFAR_OK;
}
extern "C" int mfar_set_wide(mfar_index* idx, int enable) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    idx->wide = enable != 0;
    return MFAR_OK;
}
// Where the list merge of a split-phase batch runs: with the scan (begin, default) or with the tail (finish: the scan stream then
// carries nothing but query prep, sample pass and scans).  MFAR_MERGE_IN_FINISH=1; measured, see DESIGN 4.1c.
static bool merge_in_finish() {
    static const bool v = getenv("MFAR_MERGE_IN_FINISH") && atoi(getenv("MFAR_MERGE_IN_FINISH")) != 0;
    return v;
}
static int check_split(const mfar_index* idx, const float* q, int Q, int k, int slot) {
    RETCHK(check_search_common(idx, q, Q, k));
    if (Q > max_split_batch(idx, k))
        return fail(MFAR_ERR_INVALID, "the split-phase entry points take at most mfar_max_split_batch() queries (64; 128 with the wide screened pass)");
    if (slot < 0 || slot >= MFAR_SLOTS) return fail(MFAR_ERR_INVALID, "slot must be in [0, 4)");
    return MFAR_OK;
}
extern "C" int mfar_stage1_begin(mfar_index* idx, const float* q, int Q, int k, int sentinel, int slot, int64_t* field_ids,
                                 float* field_scores, void* stream) {
    RETCHK(check_split(idx, q, Q, k, slot));
    if (Q == 0) return MFAR_OK;
    HIPCHK(hipSetDevice(idx->device));
    if (!field_ids || !field_scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    RETCHK(stage1_block(idx, slot, S1_PREPARE | S1_SCAN | (merge_in_finish() ? 0 : S1_FINISH), q, Q, 0, k, sentinel, 0, idx->F, (long long*)field_ids,
                        field_scores, nullptr, (hipStream_t)stream));
    if (Q > idx->s1[slot].qw)   // the screen slab became unavailable (rows rewritten, rebuild out of memory): only 64 queries were begun
        return fail(MFAR_ERR_UNSUPPORTED, "a split-phase batch of more than 64 queries needs the screen slab, which could not be (re)built");
    return MFAR_OK;
}
extern "C" int mfar_stage1_finish(mfar_index* idx, const float* q, int Q, int k, int sentinel, int slot, int64_t* field_ids,
                                  float* field_scores, int32_t* any_fail, void* stream) {
    RETCHK(check_split(idx, q, Q, k, slot));
    if (Q == 0) return MFAR_OK;
    if (!field_ids || !field_scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    HIPCHK(hipSetDevice(idx->device));
    // failures are frequent on this data (adaptive policy above): repair on the device, report a clean batch
    const bool inline_rep = any_fail && idx->pol.inline_repair;
    RETCHK(stage1_block(idx, slot, S1_CERTIFY | (merge_in_finish() ? S1_FINISH : 0), q, Q, 0, k, sentinel, 0, idx->F, (long long*)field_ids,
                        field_scores, inline_rep ? nullptr : (int*)any_fail, (hipStream_t)stream));
    if (inline_rep) HIPCHK(hipMemsetAsync(any_fail, 0, 4, (hipStream_t)stream));
    return MFAR_OK;
}

extern "C" int mfar_set_auto_off(mfar_index* idx, int mode, int off_fails, int probe_every) {
    if (!idx || mode < 0 || mode > 1 || off_fails < 0 || off_fails > 16 || probe_every < 0)
        return fail(MFAR_ERR_INVALID, "mode must be 0 or 1, off_fails in [0, 16] (0 = default), probe_every >= 0 (0 = default)");
    idx->pol.set_mode(mode);
    if (off_fails) idx->pol.off_fails = off_fails;
    if (probe_every) idx->pol.probe_every = std::max(2, probe_every);
    return MFAR_OK;
}
extern "C" int mfar_set_tier2(mfar_index* idx, int mode) {
    if (!idx || mode < 0 || (mode & 3) > 2 || mode > 6)
        return fail(MFAR_ERR_INVALID, "mode must be 0 (never), 1 (auto: armed by failed certificates) or 2 (always); + 4: always take the rescan (diagnostic)");
    idx->tier2_mode = mode & 3;
    idx->t2_force_rescan = (mode & 4) != 0;
    return MFAR_OK;
}
extern "C" int mfar_set_deep_scan(mfar_index* idx, int mode) {
    if (!idx || mode < 0 || mode > 2) return fail(MFAR_ERR_INVALID, "mode must be 0 (never), 1 (auto: fields whose first certificates keep failing) or 2 (every field, always)");
    idx->deep_mode = mode;
    idx->pol.deep_mode = mode ? 1 : 0;
    if (!mode) idx->pol.deep_mask = 0;
    return MFAR_OK;
}
extern "C" int mfar_deep_scan_info(mfar_index* idx, uint32_t* deep_fields, int64_t* n_switched) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    consume_feedback(idx);
    const u32 allm = idx->F >= 32 ? 0xFFFFFFFFu : ((1u << idx->F) - 1u);
    if (deep_fields) *deep_fields = idx->deep_mode == 2 ? allm : (idx->deep_mode ? idx->pol.deep_mask : 0u);
    if (n_switched) *n_switched = idx->pol.n_deep_on;
    return MFAR_OK;
}
extern "C" int mfar_tier2_stats(mfar_index* idx, int* armed, int64_t* n_lists, int64_t* n_passed_on, int64_t* causes) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    consume_feedback(idx);
    if (armed) *armed = idx->tier2_mode == 2 || (idx->tier2_mode == 1 && idx->pol.t2_armed) ? 1 : 0;
    int64_t a = 0, b = 0, c[4] = {0, 0, 0, 0};
    for (auto& sl : idx->s1)
        if (sl.fail.p) {
            HIPCHK(hipDeviceSynchronize());
            int v[2] = {0, 0}, w[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpy(v, sl.fail.as<int>() + SCREEN_STAT_T2_LISTS, 8, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(w, sl.fail.as<int>() + SCREEN_STAT_T2_OVF, 16, hipMemcpyDeviceToHost));
            a += v[0];
            b += v[1];
            for (int i = 0; i < 4; ++i) c[i] += w[i];
        }
    if (n_lists) *n_lists = a;
    if (n_passed_on) *n_passed_on = b;
    if (causes)
        for (int i = 0; i < 4; ++i) causes[i] = c[i];
    return MFAR_OK;
}
extern "C" int mfar_tier2_rescan_stats(mfar_index* idx, int64_t* n_from_scan, int64_t* n_rescanned) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    int64_t a = 0, b = 0;
    for (auto& sl : idx->s1)
        if (sl.fail.p) {
            HIPCHK(hipDeviceSynchronize());
            int v[2] = {0, 0};        // {SCREEN_STAT_T2_RESCAN, SCREEN_STAT_T2_FIRST}
            HIPCHK(hipMemcpy(v, sl.fail.as<int>() + SCREEN_STAT_T2_RESCAN, 8, hipMemcpyDeviceToHost));
            b += v[0];
            a += v[1];
        }
    if (n_from_scan) *n_from_scan = a;
    if (n_rescanned) *n_rescanned = b;
    return MFAR_OK;
}
extern "C" int mfar_auto_off_info(mfar_index* idx, uint32_t* off_fields, int64_t* n_switched_off, int64_t* n_switched_on, int64_t* n_probes,
                                  int* inline_repair) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    consume_feedback(idx);
    if (off_fields) *off_fields = idx->pol.off_mask;
    if (n_switched_off) *n_switched_off = idx->pol.n_off;
    if (n_switched_on) *n_switched_on = idx->pol.n_on;
    if (n_probes) *n_probes = idx->pol.n_probes;
    if (inline_repair) *inline_repair = idx->pol.inline_repair ? 1 : 0;
    return MFAR_OK;
}

static bool two_level_ok(const mfar_index* idx, int C, int k2, int query_cond, int n_masks);
extern "C" int mfar_set_stage2_mode(mfar_index* idx, int mode) {
    if (!idx || mode < 0 || mode > 2)
        return fail(MFAR_ERR_INVALID, "mode must be 0 (gather every row), 1 (certified two-level stage 2) or 2 (also for sweeps of many masks)");
    idx->stage2_mode = mode;
    return MFAR_OK;
}

extern "C" int mfar_set_stage2_kernels(mfar_index* idx, int family) {
    if (!idx || family < 0 || family > 1) return fail(MFAR_ERR_INVALID, "family must be 0 (the kernels of rounds 3-5) or 1 (round 6: gate / front / bounds / select)");
    idx->s2_fused = family != 0;
    for (auto& pre : idx->s2pre) pre.q = nullptr;
    return MFAR_OK;
}
extern "C" int mfar_set_row_mode(mfar_index* idx, int mode) {
    if (!idx || mode < 0 || mode > 2) return fail(MFAR_ERR_INVALID, "mode must be 0 (never), 1 (auto: after a failed certificate) or 2 (always)");
    idx->row_mode_setting = mode;
    if (mode == 0) idx->row_mask = 0;
    if (mode == 2) idx->row_mask = idx->row_eligible;
    return MFAR_OK;
}
extern "C" int mfar_row_mode_activate(mfar_index* idx) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (idx->row_mode_setting != 0) idx->row_mask = idx->row_eligible;
    return MFAR_OK;
}
extern "C" int mfar_row_mode_info(const mfar_index* idx, uint32_t* eligible_fields, uint32_t* active_fields) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (eligible_fields) *eligible_fields = idx->row_eligible;
    if (active_fields) *active_fields = idx->row_mask;
    return MFAR_OK;
}

extern "C" int mfar_set_stage2_dump(mfar_index* idx, int mode) {
    if (!idx || mode < 0 || mode > 2) return fail(MFAR_ERR_INVALID, "mode must be 0 (never), 1 (when it moves fewer bytes) or 2 (whenever possible)");
    idx->dump_mode = mode;
    return MFAR_OK;
}
extern "C" int mfar_stage2_dump_info(mfar_index* idx, int k1, int* wanted, int64_t* bytes_per_launch, int64_t* n_launches) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (wanted) *wanted = dump_wanted(idx, k1) ? 1 : 0;
    if (bytes_per_launch) *bytes_per_launch = idx->screen_built && idx->dtype == MFAR_DTYPE_F32 ? (int64_t)(idx->screen_used / 2 / (size_t)idx->E * 256) : 0;
    if (n_launches) *n_launches = idx->dump_launches;
    return MFAR_OK;
}

extern "C" int mfar_stage2_stats(mfar_index* idx, int* two_level_available, int64_t* gather_slab_bytes, int64_t* n_candidates,
                                 int64_t* n_survivors) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    if (two_level_available) *two_level_available = two_level_ok(idx, MFAR_MAX_K + 1, 1, 0, 1) ? 1 : 0;
    if (gather_slab_bytes) *gather_slab_bytes = idx->gslab_ok ? (int64_t)((size_t)idx->F * idx->n_rows * idx->g_row_bytes) : 0;
    unsigned long long h[2] = {0, 0};
    if (idx->s2stats.p) {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(h, idx->s2stats.p, sizeof(h), hipMemcpyDeviceToHost));
    }
    if (n_candidates) *n_candidates = (int64_t)h[0];
    if (n_survivors) *n_survivors = (int64_t)h[1];
    return MFAR_OK;
}

extern "C" int mfar_set_screen(mfar_index* idx, int mode, float eps_mult) {
    if (!idx || mode < 0 || mode > 2 || !(eps_mult >= 0.0f)) return fail(MFAR_ERR_INVALID, "mode must be 0, 1 or 2 and eps_mult >= 0");
    idx->screen_mode = mode;
    idx->screen_eps_mult = eps_mult;
    return MFAR_OK;
}

extern "C" int mfar_get_screen(const mfar_index* idx, int* mode, float* eps_mult) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (mode) *mode = idx->screen_mode;
    if (eps_mult) *eps_mult = idx->screen_eps_mult;
    return MFAR_OK;
}

extern "C" int mfar_screen_field_info(mfar_index* idx, int field, int64_t* n_unique_rows, int64_t* largest_group) {
    if (!idx || field < 0 || field >= idx->F) return fail(MFAR_ERR_INVALID, "bad idx / field");
    const bool built = idx->screen_built && !idx->screen_dirty;
    if (n_unique_rows) *n_unique_rows = built ? idx->n_unique[field] : -1;
    if (largest_group) *largest_group = built ? idx->largest_group[field] : -1;
    return MFAR_OK;
}

extern "C" int mfar_screen_stats(mfar_index* idx, int* built, int64_t* screen_bytes, int64_t* n_checked, int64_t* n_failed) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    HIPCHK(hipSetDevice(idx->device));
    const bool have = idx->screen_built && !idx->screen_dirty;
    if (built) *built = have ? 1 : 0;
    if (screen_bytes) *screen_bytes = have ? (int64_t)idx->screen_used : 0;
    if (n_checked) *n_checked = idx->screen_checked;
    if (n_failed) {
        *n_failed = 0;
        for (auto& sl : idx->s1)
            if (sl.fail.p) {
                HIPCHK(hipDeviceSynchronize());
                int v = 0;
                HIPCHK(hipMemcpy(&v, sl.fail.as<int>() + MFAR_MAX_FIELDS + 1, 4, hipMemcpyDeviceToHost));
                *n_failed += v;
            }
    }
    return MFAR_OK;
}

extern "C" int mfar_retrieve_fields(mfar_index* idx, const float* q, int Q, int k, int sentinel, int64_t* field_ids,
                                    float* field_scores, int on_device, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k));
    if (Q == 0) return MFAR_OK;
    if (!field_ids || !field_scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nl = (size_t)Q * idx->F * k;
    const float* qd;
    long long* fid;
    float* fsc;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * idx->E, on_device, st, &qd));
    RETCHK(stage_out(idx->out[0], (long long*)field_ids, nl, on_device, &fid));
    RETCHK(stage_out(idx->out[1], field_scores, nl, on_device, &fsc));
    RETCHK(run_stage1(idx, qd, Q, k, sentinel, fid, fsc, st));
    RETCHK(copy_back((long long*)field_ids, fid, nl, on_device, st));
    RETCHK(copy_back(field_scores, fsc, nl, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

extern "C" int mfar_retrieve_field(mfar_index* idx, int field, const float* q, int Q, int k, int sentinel, int64_t* ids, float* scores,
                                   int on_device, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k));
    if (field < 0 || field >= idx->F) return fail(MFAR_ERR_INVALID, "field out of range");
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nl = (size_t)Q * k;
    const float* qd;
    long long* fid;
    float* fsc;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * idx->E, on_device, st, &qd));
    RETCHK(stage_out(idx->out[0], (long long*)ids, nl, on_device, &fid));
    RETCHK(stage_out(idx->out[1], scores, nl, on_device, &fsc));
    for (int q0 = 0, n = 0; q0 < Q; q0 += n) RETCHK(stage1_block(idx, 0, S1_ALL, qd, Q, q0, k, sentinel, field, 1, fid, fsc, nullptr, st, &n));
    RETCHK(copy_back((long long*)ids, fid, nl, on_device, st));
    RETCHK(copy_back(scores, fsc, nl, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ stage 2
// approx != nullptr: the APPROXIMATE level of the two-level stage 2 (fp16 gather slab of an fp32 index; the caller checked
// two_level_ok): approx->qm / the field scales are applied by the kernel
struct ApproxArgs {
    const float* qm;   // [Q, MFAR_MAX_FIELDS]
};
struct KnownArgs {     // pairs stage 1 already scored exactly (mfar_select.h ScoreParams::kmask)
    const u32* kmask;
    const int* ksrc;
    const float* kval;
};
static int run_score(mfar_index* idx, const float* q, int Q, const long long* cand, const int* ncand, int C, float* x,
                     hipStream_t st, const ApproxArgs* approx = nullptr, const KnownArgs* known = nullptr) {
    ScoreParams p = {};
    p.slab = idx->slab;
    p.field_stride = idx->field_stride;
    p.q = q;
    p.cand = cand;
    p.n_cand = ncand;
    p.out = x;
    p.row_offset = idx->row_offset;
    p.n_rows = (int)idx->n_rows;
    p.n_steps = idx->n_steps;
    p.E = idx->E;
    p.F = idx->F;
    p.C = C;
    static const bool use_rep = !(getenv("MFAR_STAGE2_REP") && atoi(getenv("MFAR_STAGE2_REP")) == 0);   // diagnostic: 0 = gather every row itself
    if (use_rep && idx->screen_built && !idx->screen_dirty && idx->u_repof.p) {   // the unique-row tables describe the rows as they are now
        p.repof = idx->u_repof.as<int>();
        p.ustride = idx->n_rows;
    }
    const unsigned gx = (unsigned)(((size_t)C * idx->F + 255) / 256), gf = (unsigned)(((size_t)C * idx->F + SCF_THREADS - 1) / SCF_THREADS);
    if (gx == 0 || Q == 0) return MFAR_OK;
    RETCHK(order_after_writes(idx, st));
    p.gslab = idx->gslab.p;
    p.g_row_bytes = (long long)idx->g_row_bytes;
    if (known) {
        p.kmask = known->kmask;
        p.ksrc = known->ksrc;
        p.kval = known->kval;
    }
    if (approx) {
        p.sfld = idx->s_field.as<ScreenField>();
        p.qm = approx->qm;
        p.qm_stride = MFAR_MAX_FIELDS;
        mfar_score_rows_kernel<SRC_F16G><<<dim3(gf, Q), dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(p);
    } else if (idx->dtype == MFAR_DTYPE_BF16) {
        bool rows16 = false;
        RETCHK(ensure_rows16(idx, st, &rows16));
        p.gslab = idx->gslab.p;
        p.g_row_bytes = (long long)idx->g_row_bytes;
        if (rows16) mfar_score_rows_kernel<SRC_BF16G><<<dim3(gf, Q), dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(p);
        else mfar_score_candidates_kernel<1><<<dim3(gx, Q), dim3(256), SCORE_LDS_BYTES(idx->E), st>>>(p);
    } else {
        mfar_score_rows_kernel<SRC_F32><<<dim3(gf, Q), dim3(SCF_THREADS), SCORE_F32_LDS_BYTES(idx->E), st>>>(p);
    }
    HIPCHK(hipGetLastError());
    return MFAR_OK;
}

extern "C" int mfar_score_candidates(mfar_index* idx, const float* q, int Q, const int64_t* cand, int C, float* out,
                                     int on_device, void* stream) {
    if (!idx) return fail(MFAR_ERR_INVALID, "idx is NULL");
    if (Q < 0 || C < 0) return fail(MFAR_ERR_INVALID, "negative size");
    if (Q == 0 || C == 0) return MFAR_OK;
    if (!q || !cand || !out) return fail(MFAR_ERR_INVALID, "NULL pointer");
    if (idx->E * 4 > 60 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim too large for the stage-2 kernel");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const float* qd;
    const long long* cd;
    float* od;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * idx->E, on_device, st, &qd));
    RETCHK(stage_in(idx->in[1], (const long long*)cand, (size_t)Q * C, on_device, st, &cd));
    RETCHK(stage_out(idx->out[0], out, (size_t)Q * C * idx->F, on_device, &od));
    RETCHK(run_score(idx, qd, Q, cd, nullptr, C, od, st));
    RETCHK(copy_back(out, od, (size_t)Q * C * idx->F, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ mixer
//   wgt   [Q, MFAR_MAX_FIELDS] field weights from mfar_s2_gate_kernel, or nullptr (the mixer computes them itself: same code, same bits)
static int run_mix(const float* x, const long long* cand, const int* ncand, const float* q, const float* W, int query_cond,
                   const float* mask, int Q, int C, int F, int E, int k, long long* ids, float* scores, int* n_valid,
                   hipStream_t st, const float* wgt = nullptr) {
    MixParams p = {};
    p.wgt = wgt;
    p.x = x;
    p.cand = cand;
    p.n_cand = ncand;
    p.q = q;
    p.W = W;
    p.mask = mask;
    p.ids = ids;
    p.scores = scores;
    p.n_valid = n_valid;
    p.C = C;
    p.F = F;
    p.E = E;
    p.k = k;
    p.query_cond = query_cond;
    if (Q == 0) return MFAR_OK;
    const size_t lds = MIX_LDS_BYTES(C, (query_cond && !wgt) ? E : 0, F);
    if (lds > 160 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim * n_fields too large for the mixer kernel's LDS staging");
    mfar_mix_topk_kernel<<<dim3(Q), dim3(256), lds, st>>>(p);
    HIPCHK(hipGetLastError());
    return MFAR_OK;
}

static int check_mix(int Q, int C, int F, int E, int k, const float* q, const float* W, int query_cond) {
    if (Q < 0 || C < 0) return fail(MFAR_ERR_INVALID, "negative size");
    if (C > 4096) return fail(MFAR_ERR_INVALID, "at most 4096 candidates per query");
    if (F <= 0 || F > MFAR_MAX_FIELDS) return fail(MFAR_ERR_INVALID, "n_fields must be in [1, 32]");
    if (k <= 0 || k > MFAR_MAX_K) return fail(MFAR_ERR_INVALID, "k must be in [1, 128]");
    if (!W) return fail(MFAR_ERR_INVALID, "W is NULL");
    if (query_cond && (!q || E <= 0)) return fail(MFAR_ERR_INVALID, "query-conditioned weights need q and E");
    return MFAR_OK;
}

extern "C" int mfar_mix_topk(int device, const float* cand_scores, const int64_t* cand_ids, const int32_t* n_cand,
                             const float* q, const float* W, int query_cond, const float* mask, int Q, int C, int F, int E,
                             int k, int64_t* ids, float* scores, int32_t* n_valid, int on_device, void* stream) {
    RETCHK(check_mix(Q, C, F, E, k, q, W, query_cond));
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores || (C > 0 && (!cand_scores || !cand_ids))) return fail(MFAR_ERR_INVALID, "NULL pointer");
    int ndev = 0;
    RETCHK(mfar_device_count(&ndev));
    if (device < 0 || device >= ndev || device >= 16) return fail(MFAR_ERR_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    RETCHK(set_kernel_attrs(device));
    hipStream_t st = (hipStream_t)stream;
    DevCtx& cx = g_ctx[device];
    std::unique_lock<std::mutex> lk(cx.mu, std::defer_lock);
    if (!on_device) lk.lock();
    const float *xd, *qd, *Wd, *md;
    const long long* cd;
    const int* nd;
    long long* idd;
    float* scd;
    int* nvd;
    RETCHK(stage_in(cx.in[0], cand_scores, (size_t)Q * C * F, on_device, st, &xd));
    RETCHK(stage_in(cx.in[1], (const long long*)cand_ids, (size_t)Q * C, on_device, st, &cd));
    RETCHK(stage_in(cx.in[2], (const int*)n_cand, (size_t)Q, on_device, st, &nd));
    RETCHK(stage_in(cx.in[3], q, query_cond ? (size_t)Q * E : 0, on_device, st, &qd));
    RETCHK(stage_in(cx.in[4], W, query_cond ? (size_t)E * F : (size_t)F, on_device, st, &Wd));
    RETCHK(stage_in(cx.in[5], mask, (size_t)F, on_device, st, &md));
    RETCHK(stage_out(cx.out[0], (long long*)ids, (size_t)Q * k, on_device, &idd));
    RETCHK(stage_out(cx.out[1], scores, (size_t)Q * k, on_device, &scd));
    RETCHK(stage_out(cx.out[2], (int*)n_valid, (size_t)Q, on_device, &nvd));
    RETCHK(run_mix(xd, cd, nd, qd, Wd, query_cond, md, Q, C, F, E, k, idd, scd, nvd, st));
    RETCHK(copy_back((long long*)ids, idd, (size_t)Q * k, on_device, st));
    RETCHK(copy_back(scores, scd, (size_t)Q * k, on_device, st));
    RETCHK(copy_back((int*)n_valid, nvd, (size_t)Q, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ full scorer
// The certified two-level stage 2 (mfar_select.h): approximate scores of every (candidate, field) pair from the fp16 gather slab ->
// interval bounds on the mixed score -> the survivors' rows from the fp32 slab.  Available for fp32 indexes whose screen (mean,
// scale, norms) and gather slab are current; otherwise every row is gathered from the fp32 / bf16 slab as before.
//   n_masks   masks of the call.  The prune kernel runs one k2-th-lower-bound selection per mask (~0.08 ms per 128 queries), so a
//             sweep of 2 F + 2 masks (mask_fields.py:143-170) would spend more there than the full gather costs (measured: the
//             18-mask sweep at 1 M x 8 68.9 ms against 53.7 ms, 46 masks at 129 k x 22 193.9 against 113.4 ms,
//             profiles/r03_b_mask_sweep_two_level_every_mask.txt): mode 1 prunes for at most S2_MAX_MASKS masks, mode 2 always (tests)
#define S2_MAX_MASKS 2
static bool two_level_ok(const mfar_index* idx, int C, int k2, int query_cond, int n_masks) {
    if (idx->stage2_mode == 1 && n_masks > S2_MAX_MASKS) return false;
    return idx->stage2_mode >= 1 && idx->dtype == MFAR_DTYPE_F32 && idx->gslab_ok && idx->screen_built && !idx->screen_dirty && C > k2 &&
           k2 <= SEL_MAX_K && PRUNE_LDS_BYTES(C, query_cond ? idx->E : 0, idx->F) <= 160 * 1024;
}
// Round 6: the tail as fewer, wider kernels (mfar_select.h "Round 6").  MFAR_S2_FUSED=0: the kernels of rounds 3-5 (diagnostic; the tests
// run both families against each other).
static bool s2_fused_default() {
    static const bool v = !(getenv("MFAR_S2_FUSED") && atoi(getenv("MFAR_S2_FUSED")) == 0);
    return v;
}
static bool approx_level_ok(const mfar_index* idx) {
    return idx->dtype == MFAR_DTYPE_F32 && idx->stage2_mode >= 1 && idx->gslab_ok && idx->screen_built && !idx->screen_dirty;
}
// Field weights of the batch (+ q . mean and eps of the approximate level when the index has one) -> idx->s2wgt / s2qm / s2eps of the slot.
// Depends on q and W only: a pipelined caller enqueues it before its tail waits for the scan (pipe_launch); run_stage2_mix calls it itself
// when nobody has.
static int run_stage2_pre(mfar_index* idx, const float* qd, int Q, const float* Wd, int query_cond, int slot, hipStream_t st) {
    const int F = idx->F, E = idx->E;
    mfar_index::S2Pre& pre = idx->s2pre[slot];
    pre.q = nullptr;
    if (!idx->s2_fused || Q == 0) return MFAR_OK;
    const size_t lds = GATE_LDS_BYTES(query_cond ? E : 0, F);
    if (lds > S2_DYN_LDS_MAX) return MFAR_OK;                  // (the mixer's own staging decides whether the shape is supported)
    RETCHK(idx->s2wgt[slot].ensure((size_t)Q * MFAR_MAX_FIELDS * 4));
    RETCHK(idx->s2qm[slot].ensure((size_t)Q * MFAR_MAX_FIELDS * 4));
    RETCHK(idx->s2eps[slot].ensure((size_t)Q * MFAR_MAX_FIELDS * 4));
    GateParams g = {};
    g.q = qd;
    g.W = Wd;
    g.wgt = idx->s2wgt[slot].as<float>();
    g.E = E;
    g.F = F;
    g.query_cond = query_cond;
    const bool approx = approx_level_ok(idx);
    if (approx) {
        g.mean = idx->s_mean.as<float>();
        g.sf = idx->s_field.as<ScreenField>();
        g.qm = idx->s2qm[slot].as<float>();
        g.eps = idx->s2eps[slot].as<float>();
        g.eps_mult = idx->screen_eps_mult;
    }
    mfar_s2_gate_kernel<<<dim3(Q), dim3(256), lds, st>>>(g);
    HIPCHK(hipGetLastError());
    pre.q = qd;
    pre.W = Wd;
    pre.Q = Q;
    pre.approx = approx;
    return MFAR_OK;
}
static const float* s2_weights(const mfar_index* idx, const float* qd, int Q, const float* Wd, int slot) {
    const mfar_index::S2Pre& pre = idx->s2pre[slot];
    return (pre.q == qd && pre.Q == Q && pre.W == Wd && qd) ? idx->s2wgt[slot].as<float>() : nullptr;
}

//   cand / ncand [Q, C] / [Q]: the candidates to score (sorted unique ids);  masks [n_masks, F] or nullptr (ones, n_masks = 1)
//   x [Q, C, F]: exact score vectors of the SURVIVORS, row c of x belongs to (*cand_out)[q, c]
//   fid / fsc  the stage-1 lists [Q, F, k1] with their exact scores, or fsc == nullptr: no known pairs (every pair is gathered)
//   known_done  mfar_s2_front_kernel already wrote the known pairs into xa / kmask of the slot
static int run_two_level(mfar_index* idx, const float* qd, int Q, const float* Wd, int query_cond, const float* masks, int n_masks, int k2,
                         const long long* cand, const int* ncand, int C, int slot, float* x, const long long** cand_out,
                         const int** ncand_out, const long long* fid, const float* fsc, int k1, int sentinel, hipStream_t st, bool known_done = false) {
    const int F = idx->F, E = idx->E;
    RETCHK(idx->xa[slot].ensure((size_t)Q * C * F * 4));
    RETCHK(idx->cand2[slot].ensure((size_t)Q * C * 8));
    RETCHK(idx->ncand2[slot].ensure((size_t)Q * 4));
    RETCHK(idx->s2qm[slot].ensure((size_t)Q * MFAR_MAX_FIELDS * 4));
    RETCHK(idx->s2eps[slot].ensure((size_t)Q * MFAR_MAX_FIELDS * 4));
    if (!idx->s2stats.p) {
        RETCHK(idx->s2stats.ensure(2 * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(idx->s2stats.p, 0, 2 * sizeof(unsigned long long), st));
    }
    // the approximate level: this slot's scan left every score it computed behind (same queries, used once), or 16-bit row gathers
    mfar_index::S1Slot& sl = idx->s1[slot];
    const bool from_dump = sl.dump_ready && sl.dump_q == qd && sl.dump_Q == Q && Q <= 128 && idx->u_of.p && idx->u_repof.p;
    sl.dump_ready = false;
    const float* wgt = s2_weights(idx, qd, Q, Wd, slot);       // the gate kernel ran for this batch: weights, q . mean and eps are in place
    const bool pre_ok = wgt != nullptr && idx->s2pre[slot].approx;
    float* qm = idx->s2qm[slot].as<float>();
    float* eps = idx->s2eps[slot].as<float>();
    if (!pre_ok) {
        S2PrepParams pp = {};
        pp.q = qd;
        pp.mean = idx->s_mean.as<float>();
        pp.sf = idx->s_field.as<ScreenField>();
        pp.qm = qm;
        pp.eps = eps;
        pp.E = E;
        pp.F = F;
        pp.eps_mult = idx->screen_eps_mult;
        if (from_dump) {
            pp.eps_src = sl.eps_dump.as<float>();      // the screened pass's eps + the dump's quantisation step
            pp.eps_qw = 128;
        }
        mfar_s2_prep_kernel<<<dim3(Q), dim3(256), 0, st>>>(pp);
        HIPCHK(hipGetLastError());
    }
    // (from the dump with the gate kernel's eps: unused -- every pair of that level carries its own bound, xe below)
    const ApproxArgs ap = {qm};
    static const bool reuse = !(getenv("MFAR_STAGE2_KNOWN") && atoi(getenv("MFAR_STAGE2_KNOWN")) == 0);   // diagnostic: 0 = gather the known pairs too
    KnownArgs kn = {nullptr, nullptr, nullptr};
    if (fsc && fid && reuse) {     // a candidate's score in the field whose list it came from is exact already: not gathered, eps = 0
        RETCHK(idx->kmask[slot].ensure((size_t)Q * C * 4));
        RETCHK(idx->src2[slot].ensure((size_t)Q * C * 4));
        if (!known_done) {
            HIPCHK(hipMemsetAsync(idx->kmask[slot].p, 0, (size_t)Q * C * 4, st));
            KnownParams kp = {};
            kp.fid = fid;
            kp.fsc = fsc;
            kp.cand = cand;
            kp.n_cand = ncand;
            kp.xa = idx->xa[slot].as<float>();
            kp.kmask = idx->kmask[slot].as<u32>();
            kp.F = F;
            kp.k = k1;
            kp.C = C;
            kp.sentinel = sentinel;
            mfar_s2_known_kernel<<<dim3(Q), dim3(256), 0, st>>>(kp);
            HIPCHK(hipGetLastError());
        }
        kn.kmask = idx->kmask[slot].as<u32>();
    }
    // one mask, weights in place, transposed table: the dump's level and the interval ends in one kernel (nothing of xa / xe is written)
    const bool fused_lb = from_dump && wgt && (masks ? n_masks : 1) == 1 && idx->uof_t_ok && idx->lbub[slot].ensure((size_t)Q * C * 8) == MFAR_OK;
    if (from_dump) {
        S2LookupParams lp = {};
        lp.dump = sl.dump.as<unsigned short>();
        lp.dump_step = sl.dstep.as<float>();
        RETCHK(idx->xe[slot].ensure((size_t)Q * C * F * 4));
        lp.eps_dump = sl.eps_dump.as<float>();
        lp.dump_arel = sl.darel.as<float>();
        lp.xe = idx->xe[slot].as<float>();
        lp.uof_packed = idx->uof_packed ? 1 : 0;
        lp.dump_base = idx->dump_base.as<long long>();
        lp.cand = cand;
        lp.n_cand = ncand;
        lp.repof = idx->u_repof.as<int>();
        lp.uof = idx->u_of.as<u32>();
        lp.ustride = idx->n_rows;
        lp.row_offset = idx->row_offset;
        lp.sf = idx->s_field.as<ScreenField>();
        lp.qinfo = sl.qinfo.as<ScreenQuery>();
        lp.qm = qm;
        lp.kmask = kn.kmask;
        lp.xa = idx->xa[slot].as<float>();
        lp.n_rows = (int)idx->n_rows;
        lp.F = F;
        lp.C = C;
        if (fused_lb) {
            LookupBoundsParams lb = {};
            lb.lp = lp;
            lb.uof_t = idx->u_of_t.as<u32>();
            lb.wgt = wgt;
            lb.mask = masks;
            lb.lbub = idx->lbub[slot].as<float>();
            mfar_s2_lookup_bounds_kernel<<<dim3((unsigned)((C + 255) / 256), Q), dim3(256), 0, st>>>(lb);
        } else {
            mfar_s2_lookup_kernel<<<dim3((unsigned)(((size_t)C * F + 255) / 256), Q), dim3(256), 0, st>>>(lp);
        }
        HIPCHK(hipGetLastError());
        idx->dump_launches++;
    } else {
        RETCHK(run_score(idx, qd, Q, cand, ncand, C, idx->xa[slot].as<float>(), st, &ap, kn.kmask ? &kn : nullptr));
    }
    PruneParams pr = {};
    pr.xa = idx->xa[slot].as<float>();
    pr.cand = cand;
    pr.n_cand = ncand;
    pr.eps = eps;
    pr.xe = from_dump ? idx->xe[slot].as<float>() : nullptr;
    pr.q = qd;
    pr.W = Wd;
    pr.masks = masks;
    pr.cand2 = idx->cand2[slot].as<long long>();
    pr.n_cand2 = idx->ncand2[slot].as<int>();
    pr.kmask = kn.kmask;
    pr.src2 = kn.kmask ? idx->src2[slot].as<int>() : nullptr;
    pr.stats = idx->s2stats.as<unsigned long long>();
    pr.C = C;
    pr.F = F;
    pr.E = E;
    pr.k = k2;
    pr.query_cond = query_cond;
    pr.n_masks = masks ? n_masks : 1;
    if (wgt && (fused_lb || idx->lbub[slot].ensure((size_t)pr.n_masks * Q * C * 8) == MFAR_OK)) {
        // the weights exist: interval ends by several workgroups per query (or already by the fused look-up), then the per-query selection
        BoundsParams bp = {};
        bp.pr = pr;
        bp.wgt = wgt;
        bp.lbub = idx->lbub[slot].as<float>();
        bp.Q = Q;
        if (!fused_lb) {
            mfar_s2_bounds_kernel<<<dim3(S2_BOUND_SPLIT, Q), dim3(256), 0, st>>>(bp);
            HIPCHK(hipGetLastError());
        }
        mfar_s2_select_kernel<<<dim3(Q), dim3(256), S2_SELECT_LDS_BYTES(C), st>>>(bp);
    } else {
        if (wgt) {
            (void)hipGetLastError();
            g_err.clear();
        }
        mfar_s2_prune_kernel<<<dim3(Q), dim3(256), PRUNE_LDS_BYTES(C, query_cond ? E : 0, F), st>>>(pr);
    }
    HIPCHK(hipGetLastError());
    kn.ksrc = pr.src2;
    kn.kval = idx->xa[slot].as<float>();
    RETCHK(run_score(idx, qd, Q, pr.cand2, pr.n_cand2, C, x, st, nullptr, kn.kmask ? &kn : nullptr));
    *cand_out = pr.cand2;
    *ncand_out = pr.n_cand2;
    return MFAR_OK;
}

// bf16 index, full gather with KNOWN pairs: a candidate's score in the field whose stage-1 list it came from is the chain's bits already
// whenever the certified stage 1 wrote the lists (certified, or the exhaustive chain pass: include/mfar_hip.h "bf16 contract") -- stage 2
// walks the same chain, so the pair is copied from the list instead of being gathered again (-1/F of the stage-2 rows).
static bool bf16_lists_are_chain_exact(const mfar_index* idx, int k) {
    return idx->dtype == MFAR_DTYPE_BF16 && screen_wanted(idx, k) && idx->screen_built && !idx->screen_dirty;
}
static int run_score_known(mfar_index* idx, const float* qd, int Q, const long long* cand, const int* ncand, int C, int slot, float* x,
                           const long long* fid, const float* fsc, int k1, int sentinel, hipStream_t st) {
    static const bool reuse = !(getenv("MFAR_STAGE2_KNOWN") && atoi(getenv("MFAR_STAGE2_KNOWN")) == 0);
    if (!reuse || !fid || !fsc || !bf16_lists_are_chain_exact(idx, k1)) return run_score(idx, qd, Q, cand, ncand, C, x, st);
    RETCHK(idx->kmask[slot].ensure((size_t)Q * C * 4));
    HIPCHK(hipMemsetAsync(idx->kmask[slot].p, 0, (size_t)Q * C * 4, st));
    KnownParams kp = {};
    kp.fid = fid;
    kp.fsc = fsc;
    kp.cand = cand;
    kp.n_cand = ncand;
    kp.xa = x;                      // (the known pairs go straight into the final score table)
    kp.kmask = idx->kmask[slot].as<u32>();
    kp.F = idx->F;
    kp.k = k1;
    kp.C = C;
    kp.sentinel = sentinel;
    mfar_s2_known_kernel<<<dim3(Q), dim3(256), 0, st>>>(kp);
    HIPCHK(hipGetLastError());
    const KnownArgs kn = {kp.kmask, nullptr, nullptr};
    return run_score(idx, qd, Q, cand, ncand, C, x, st, nullptr, &kn);
}

// union -> stage 2 -> mixer (one launch per mask), all pointers on the device; `slot` selects one of two internal workspaces so
// that two batches can be in flight on different streams.  masks [n_masks, F] (nullptr = no mask, n_masks = 1); ids / scores
// [n_masks, Q, k2], nvd [n_masks, Q] or nullptr.
//   fsc / sentinel   the lists' exact scores and padding convention, or fsc == nullptr (then no stage-1 score is reused)
static int run_stage2_mix(mfar_index* idx, const float* qd, int Q, const float* Wd, int query_cond, const float* masks, int n_masks, int k1,
                          int k2, const long long* fid, const float* fsc, int sentinel, int slot, long long* idd, float* scd, int* nvd,
                          int* ncd_out, hipStream_t st) {
    const int F = idx->F, E = idx->E, C = F * k1;
    int* ncd = ncd_out;
    if (!ncd) {
        RETCHK(idx->ncand[slot].ensure((size_t)Q * 4));
        ncd = idx->ncand[slot].as<int>();
    }
    RETCHK(idx->cand[slot].ensure((size_t)Q * C * 8));
    RETCHK(idx->x[slot].ensure((size_t)Q * C * F * 4));
    if (!s2_weights(idx, qd, Q, Wd, slot)) RETCHK(run_stage2_pre(idx, qd, Q, Wd, query_cond, slot, st));      // (a pipelined caller did this beside the scan)
    const float* wgt = s2_weights(idx, qd, Q, Wd, slot);
    const bool two = two_level_ok(idx, C, k2, query_cond, n_masks);
    // candidate union (+ the known pairs of the two-level path) by bitmap when the ids span <= 2^20, else sort + unique
    static const bool reuse = !(getenv("MFAR_STAGE2_KNOWN") && atoi(getenv("MFAR_STAGE2_KNOWN")) == 0);
    const long long span = idx->row_offset + idx->n_rows;
    const bool front = idx->s2_fused && span <= S2_FRONT_MAX_SPAN && S2_FRONT_LDS_BYTES(span, C) <= S2_DYN_LDS_MAX;
    bool known_done = false;
    if (front) {
        FrontParams fp = {};
        fp.fid = fid;
        fp.cand = idx->cand[slot].as<long long>();
        fp.n_cand = ncd;
        fp.F = F;
        fp.k = k1;
        fp.sentinel = sentinel;
        fp.span = (int)span;
        if (two && fsc && reuse && idx->xa[slot].ensure((size_t)Q * C * F * 4) == MFAR_OK && idx->kmask[slot].ensure((size_t)Q * C * 4) == MFAR_OK) {
            fp.fsc = fsc;
            fp.xa = idx->xa[slot].as<float>();
            fp.kmask = idx->kmask[slot].as<u32>();
            known_done = true;
        }
        mfar_s2_front_kernel<<<dim3(Q), dim3(256), S2_FRONT_LDS_BYTES(span, C), st>>>(fp);
    } else {
        mfar_union_kernel<<<dim3(Q), dim3(256), 0, st>>>(fid, F, k1, idx->cand[slot].as<long long>(), ncd);
    }
    HIPCHK(hipGetLastError());
    const long long* cm = idx->cand[slot].as<long long>();
    const int* nm = ncd;
    if (two)
        RETCHK(run_two_level(idx, qd, Q, Wd, query_cond, masks, n_masks, k2, cm, nm, C, slot, idx->x[slot].as<float>(), &cm, &nm, fid, fsc, k1,
                             sentinel, st, known_done));
    else
        RETCHK(run_score_known(idx, qd, Q, cm, nm, C, slot, idx->x[slot].as<float>(), fid, fsc, k1, sentinel, st));
    for (int m = 0; m < n_masks; ++m)
        RETCHK(run_mix(idx->x[slot].as<float>(), cm, nm, qd, Wd, query_cond, masks ? masks + (size_t)m * F : nullptr, Q, C, F, E, k2,
                       idd + (size_t)m * Q * k2, scd + (size_t)m * Q * k2, nvd ? nvd + (size_t)m * Q : nullptr, st, wgt));
    idx->s2pre[slot].q = nullptr;                              // one batch, one use (the caller's q buffer is refilled in place)
    return MFAR_OK;
}

extern "C" int mfar_search_two_stage(mfar_index* idx, const float* q, int Q, const float* W, int query_cond,
                                     const float* mask, int k1, int k2, int sentinel, int64_t* ids, float* scores,
                                     int32_t* n_valid, int64_t* field_ids, float* field_scores, int32_t* n_cand,
                                     int on_device, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    const int F = idx->F, E = idx->E, C = F * k1;
    RETCHK(check_mix(Q, C, F, E, k2, q, W, query_cond));
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    if (E * 4 > 60 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim too large for the stage-2 kernel");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nl = (size_t)Q * F * k1;
    const float *qd, *Wd, *md;
    long long *idd, *fid;
    float *scd, *fsc;
    int *nvd, *ncd;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * E, on_device, st, &qd));
    RETCHK(stage_in(idx->in[1], W, query_cond ? (size_t)E * F : (size_t)F, on_device, st, &Wd));
    RETCHK(stage_in(idx->in[2], mask, (size_t)F, on_device, st, &md));
    RETCHK(stage_out(idx->out[0], (long long*)ids, (size_t)Q * k2, on_device, &idd));
    RETCHK(stage_out(idx->out[1], scores, (size_t)Q * k2, on_device, &scd));
    RETCHK(stage_out(idx->out[2], (int*)n_valid, (size_t)Q, on_device, &nvd));
    // stage-1 lists: caller's buffers when given on the device, else internal
    if (field_ids && on_device) fid = (long long*)field_ids;
    else {
        RETCHK(idx->fid.ensure(nl * 8));
        fid = idx->fid.as<long long>();
    }
    if (field_scores && on_device) fsc = field_scores;
    else {
        RETCHK(idx->fsc.ensure(nl * 4));
        fsc = idx->fsc.as<float>();
    }
    if (n_cand && on_device) ncd = (int*)n_cand;
    else {
        RETCHK(idx->ncand[0].ensure((size_t)Q * 4));
        ncd = idx->ncand[0].as<int>();
    }
    RETCHK(run_stage1(idx, qd, Q, k1, sentinel, fid, fsc, st));
    RETCHK(run_stage2_mix(idx, qd, Q, Wd, query_cond, md, 1, k1, k2, fid, fsc, sentinel, 0, idd, scd, nvd, ncd, st));
    RETCHK(copy_back((long long*)ids, idd, (size_t)Q * k2, on_device, st));
    RETCHK(copy_back(scores, scd, (size_t)Q * k2, on_device, st));
    RETCHK(copy_back((int*)n_valid, nvd, (size_t)Q, on_device, st));
    if (!on_device) {
        RETCHK(copy_back((long long*)field_ids, fid, nl, 0, st));
        RETCHK(copy_back(field_scores, fsc, nl, 0, st));
        RETCHK(copy_back((int*)n_cand, ncd, (size_t)Q, 0, st));
        HIPCHK(hipStreamSynchronize(st));
    }
    return MFAR_OK;
}

extern "C" int mfar_search_stage2(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* mask,
                                  int k1, int k2, const int64_t* field_ids, const float* field_scores, int sentinel, int slot, int64_t* ids,
                                  float* scores, int32_t* n_valid, int32_t* n_cand, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    const int F = idx->F, E = idx->E, C = F * k1;
    RETCHK(check_mix(Q, C, F, E, k2, q, W, query_cond));
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores || !field_ids) return fail(MFAR_ERR_INVALID, "NULL pointer");
    if (slot < 0 || slot >= MFAR_SLOTS) return fail(MFAR_ERR_INVALID, "slot must be in [0, 4)");
    HIPCHK(hipSetDevice(idx->device));
    return run_stage2_mix(idx, q, Q, W, query_cond, mask, 1, k1, k2, (const long long*)field_ids, field_scores, sentinel, slot, (long long*)ids,
                          scores, (int*)n_valid, (int*)n_cand, (hipStream_t)stream);
}

extern "C" int mfar_search_stage2_masks(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* masks,
                                        int n_masks, int k1, int k2, const int64_t* field_ids, const float* field_scores, int sentinel, int slot,
                                        int64_t* ids, float* scores, int32_t* n_valid, int32_t* n_cand, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    const int F = idx->F, E = idx->E, C = F * k1;
    RETCHK(check_mix(Q, C, F, E, k2, q, W, query_cond));
    if (n_masks <= 0 || !masks) return fail(MFAR_ERR_INVALID, "n_masks must be positive and masks non-NULL");
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores || !field_ids) return fail(MFAR_ERR_INVALID, "NULL pointer");
    if (slot < 0 || slot >= MFAR_SLOTS) return fail(MFAR_ERR_INVALID, "slot must be in [0, 4)");
    HIPCHK(hipSetDevice(idx->device));
    // candidate union and stage 2 once, then one mixer launch per mask over the same scores
    return run_stage2_mix(idx, q, Q, W, query_cond, masks, n_masks, k1, k2, (const long long*)field_ids, field_scores, sentinel, slot,
                          (long long*)ids, scores, (int*)n_valid, (int*)n_cand, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ fused mode
// The companion: rows of dim F * E -- for every document the concatenation of its F field vectors -- spread over
// FUSED_GROUPS interleaved row groups that are stored as the "fields" of an ordinary index (document r = group r % G, local
// row r / G).  Everything stage 1 offers (certified fp16 screen over unique rows, wide pass, sample pass, a full grid) then
// applies to the fused search unchanged; mfar_merge_groups_kernel takes the final top-k of the G group lists.
#define FUSED_GROUPS 8
static int ensure_fused(mfar_index* idx, hipStream_t st) {
    if (idx->dtype != MFAR_DTYPE_F32) return fail(MFAR_ERR_UNSUPPORTED, "the fused mode is built for fp32 indexes");
    const long long FE = (long long)idx->F * idx->E;
    const int G = FUSED_GROUPS;
    if (!idx->fused) {
        RETCHK(mfar_index_create(&idx->fused, idx->device, (idx->n_rows + G - 1) / G, idx->row_offset, G, (int)FE, MFAR_DTYPE_F32));
        idx->fused_dirty = true;
    }
    if (idx->fused_dirty) {
        HIPCHK(hipDeviceSynchronize());   // a fused search in flight may still read the companion
        const long long total = idx->n_rows * (idx->E / 4);
        for (int f = 0; f < idx->F && total > 0; ++f) {
            mfar_concat_rows_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st>>>(
                (const float*)idx->slab + (size_t)f * idx->field_stride, (float*)idx->fused->slab, idx->fused->field_stride, idx->n_rows, idx->E,
                (int)FE, f * idx->E, G);
            HIPCHK(hipGetLastError());
        }
        idx->fused->screen_dirty = true;
        idx->fused_dirty = false;
    }
    // The companion is scanned by the exact fp32 pass.  The certified screen does not pay here (measured at 1 M x 8 x 768: every
    // certificate fails and the exact pass runs anyway): its error bound follows |q|_2 |c|_2 of the 6144-dim vectors, which
    // does not shrink when F per-field scores are averaged, while the gap between the k-th and the k'-th mixed score does
    // (by sqrt(F)); MFAR_FUSED_SCREEN=1 turns it on for experiments.
    static const bool fused_screen = getenv("MFAR_FUSED_SCREEN") && atoi(getenv("MFAR_FUSED_SCREEN")) != 0;
    idx->fused->screen_mode = fused_screen ? 2 : 0;
    idx->fused->screen_eps_mult = idx->screen_eps_mult;
    idx->fused->wide = idx->wide;
    idx->fused->repair_sample = idx->repair_sample;
    return MFAR_OK;
}

extern "C" int mfar_search_fused(mfar_index* idx, const float* q, int Q, const float* W, int query_cond, const float* mask, int k,
                                 int64_t* ids, float* scores, int on_device, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k));
    if (k >= MFAR_MAX_K) return fail(MFAR_ERR_INVALID, "the fused mode takes k < 128 (its group lists are one entry deeper)");
    if (!W) return fail(MFAR_ERR_INVALID, "W is NULL");
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    const int F = idx->F, E = idx->E, G = FUSED_GROUPS, kg = k + 1;
    if (FOLD_LDS_BYTES(E, query_cond ? F : 0) > 160 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim * n_fields too large for the gate kernel's LDS staging");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    RETCHK(ensure_fused(idx, st));
    const float *qd, *Wd, *md;
    long long* idd;
    float* scd;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * E, on_device, st, &qd));
    RETCHK(stage_in(idx->in[1], W, query_cond ? (size_t)E * F : (size_t)F, on_device, st, &Wd));
    RETCHK(stage_in(idx->in[2], mask, (size_t)F, on_device, st, &md));
    RETCHK(stage_out(idx->out[0], (long long*)ids, (size_t)Q * k, on_device, &idd));
    RETCHK(stage_out(idx->out[1], scores, (size_t)Q * k, on_device, &scd));
    RETCHK(idx->fused_q.ensure((size_t)Q * F * E * 4));
    RETCHK(idx->fid.ensure((size_t)Q * G * kg * 8));
    RETCHK(idx->fsc.ensure((size_t)Q * G * kg * 4));
    mfar_fold_queries_kernel<<<dim3(Q), dim3(256), FOLD_LDS_BYTES(E, query_cond ? F : 0), st>>>(qd, Wd, md, query_cond, F, E,
                                                                                               idx->fused_q.as<float>());
    HIPCHK(hipGetLastError());
    // exhaustive top-(k + 1) of the folded inner product per row group: no zero sentinel (the reference's mixed scores have
    // none, contrastive.py:696)
    RETCHK(run_stage1(idx->fused, idx->fused_q.as<float>(), Q, kg, 0, idx->fid.as<long long>(), idx->fsc.as<float>(), st));
    GroupMergeParams gp = {};
    gp.gids = idx->fid.as<long long>();
    gp.gsc = idx->fsc.as<float>();
    gp.ids = idd;
    gp.scores = scd;
    gp.row_offset = idx->row_offset;
    gp.n_rows = idx->n_rows;
    gp.G = G;
    gp.kg = kg;
    gp.k = k;
    mfar_merge_groups_kernel<<<dim3(Q), dim3(256), SEL_LDS_BYTES(G * kg), st>>>(gp);
    HIPCHK(hipGetLastError());
    RETCHK(copy_back((long long*)ids, idd, (size_t)Q * k, on_device, st));
    RETCHK(copy_back(scores, scd, (size_t)Q * k, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ multi-GPU
extern "C" int64_t mfar_merge_workspace_bytes(int Q, int n_fields, int k1) {
    if (Q < 0 || n_fields <= 0 || k1 <= 0) return 0;
    return merge_ws_layout(Q, n_fields, k1).total;
}
__global__ void mfar_write_header_kernel(PayloadHeader* dst, const PayloadHeader h) { *dst = h; }

extern "C" int64_t mfar_payload_bytes(int Q, int n_fields, int k1) {
    if (Q < 0 || n_fields <= 0 || k1 <= 0) return 0;
    return payload_layout(Q, n_fields, k1).total;
}

extern "C" int mfar_search_local(mfar_index* idx, const float* q, int Q, int k1, int sentinel, void* payload, int phases,
                                 int on_device, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    if (!payload) return fail(MFAR_ERR_INVALID, "payload is NULL");
    if (phases < 1 || phases > 3) return fail(MFAR_ERR_INVALID, "phases must be 1, 2 or 3");
    if (phases != 3 && !on_device) return fail(MFAR_ERR_INVALID, "split phases need device buffers");
    const int F = idx->F, E = idx->E, C = F * k1;
    if (C > 4096) return fail(MFAR_ERR_INVALID, "n_fields * k1 must be <= 4096");
    if (E * 4 > 60 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim too large for the stage-2 kernel");
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const PayloadLayout L = payload_layout(Q, F, k1);
    const float* qd;
    char* pd;
    RETCHK(stage_in(idx->in[0], q, (size_t)Q * E, on_device, st, &qd));
    RETCHK(stage_out(idx->out[3], (char*)payload, (size_t)L.total, on_device, &pd));
    PayloadHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = PAYLOAD_MAGIC;
    h.Q = Q;
    h.F = F;
    h.k1 = k1;
    h.row_offset = idx->row_offset;
    h.n_rows = idx->n_rows;
    h.sentinel = sentinel;
    if (phases & 1) {
        mfar_write_header_kernel<<<dim3(1), dim3(1), 0, st>>>((PayloadHeader*)(pd + L.hdr), h);
        HIPCHK(hipGetLastError());
    }
    if (Q > 0) {
        long long* fid = (long long*)(pd + L.ids);
        float* fsc = (float*)(pd + L.scores);
        long long* cand = (long long*)(pd + L.cand);
        int* ncd = (int*)(pd + L.ncand);
        float* x = (float*)(pd + L.x);
        if (phases & 1) RETCHK(run_stage1(idx, qd, Q, k1, sentinel, fid, fsc, st));
        if (phases & 2) {
            mfar_union_kernel<<<dim3(Q), dim3(256), 0, st>>>(fid, F, k1, cand, ncd);
            HIPCHK(hipGetLastError());
            RETCHK(run_score(idx, qd, Q, cand, ncd, C, x, st));
        }
    }
    RETCHK(copy_back((char*)payload, pd, (size_t)L.total, on_device, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

extern "C" int mfar_merge_payloads(int device, const void* payloads, int n_shards, const float* q, int Q, int E,
                                   const float* W, int query_cond, const float* mask, int n_fields, int k1, int k2,
                                   int sentinel, int64_t* ids, float* scores, int32_t* n_valid, void* workspace,
                                   int64_t workspace_bytes, int on_device, void* stream) {
    const int F = n_fields, C = F * k1;
    if (!payloads || n_shards <= 0 || n_shards > 64) return fail(MFAR_ERR_INVALID, "bad payloads / n_shards");
    if (k1 <= 0 || k1 > MFAR_MAX_K) return fail(MFAR_ERR_INVALID, "k1 must be in [1, 128]");
    RETCHK(check_mix(Q, C, F, E, k2, q, W, query_cond));
    if (Q == 0) return MFAR_OK;
    if (!ids || !scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    int ndev = 0;
    RETCHK(mfar_device_count(&ndev));
    if (device < 0 || device >= ndev || device >= 16) return fail(MFAR_ERR_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    RETCHK(set_kernel_attrs(device));
    hipStream_t st = (hipStream_t)stream;
    DevCtx& cx = g_ctx[device];
    const bool own_ws = workspace != nullptr;
    if (own_ws && !on_device) return fail(MFAR_ERR_INVALID, "a caller workspace needs device buffers");
    if (own_ws && workspace_bytes < mfar_merge_workspace_bytes(Q, F, k1)) return fail(MFAR_ERR_INVALID, "workspace too small");
    std::unique_lock<std::mutex> lk(cx.mu, std::defer_lock);
    if (!own_ws) lk.lock();
    const PayloadLayout L = payload_layout(Q, F, k1);
    const char* pd;
    const float *qd, *Wd, *md;
    long long* idd;
    float* scd;
    int* nvd;
    RETCHK(stage_in(cx.in[6], (const char*)payloads, (size_t)L.total * n_shards, on_device, st, &pd));
    RETCHK(stage_in(cx.in[3], q, query_cond ? (size_t)Q * E : 0, on_device, st, &qd));
    RETCHK(stage_in(cx.in[4], W, query_cond ? (size_t)E * F : (size_t)F, on_device, st, &Wd));
    RETCHK(stage_in(cx.in[5], mask, (size_t)F, on_device, st, &md));
    RETCHK(stage_out(cx.out[0], (long long*)ids, (size_t)Q * k2, on_device, &idd));
    RETCHK(stage_out(cx.out[1], scores, (size_t)Q * k2, on_device, &scd));
    RETCHK(stage_out(cx.out[2], (int*)n_valid, (size_t)Q, on_device, &nvd));
    long long *w_lids, *w_cand;
    float *w_lsc, *w_x;
    int* w_ncand;
    if (own_ws) {
        const MergeWsLayout wl = merge_ws_layout(Q, F, k1);
        char* wb = (char*)workspace;
        w_lids = (long long*)(wb + wl.lids);
        w_lsc = (float*)(wb + wl.lsc);
        w_cand = (long long*)(wb + wl.cand);
        w_ncand = (int*)(wb + wl.ncand);
        w_x = (float*)(wb + wl.x);
    } else {
        RETCHK(cx.lists_ids.ensure((size_t)Q * F * k1 * 8));
        RETCHK(cx.lists_sc.ensure((size_t)Q * F * k1 * 4));
        RETCHK(cx.cand.ensure((size_t)Q * C * 8));
        RETCHK(cx.ncand.ensure((size_t)Q * 4));
        RETCHK(cx.x.ensure((size_t)Q * C * F * 4));
        w_lids = cx.lists_ids.as<long long>();
        w_lsc = cx.lists_sc.as<float>();
        w_cand = cx.cand.as<long long>();
        w_ncand = cx.ncand.as<int>();
        w_x = cx.x.as<float>();
    }
    ShardMergeParams sp = {};
    sp.payloads = pd;
    sp.payload_stride = L.total;
    sp.ids_off = L.ids;
    sp.scores_off = L.scores;
    sp.out_ids = w_lids;
    sp.out_scores = w_lsc;
    sp.S = n_shards;
    sp.F = F;
    sp.k = k1;
    sp.sentinel = sentinel;
    if (n_shards * k1 <= 8 * 256) mfar_merge_shards_kernel<8><<<dim3(Q * F), dim3(256), SEL_LDS_BYTES(n_shards * k1), st>>>(sp);
    else mfar_merge_shards_kernel<32><<<dim3(Q * F), dim3(256), SEL_LDS_BYTES(n_shards * k1), st>>>(sp);
    HIPCHK(hipGetLastError());
    mfar_union_kernel<<<dim3(Q), dim3(256), 0, st>>>(sp.out_ids, F, k1, w_cand, w_ncand);
    HIPCHK(hipGetLastError());
    LookupParams lp = {};
    lp.payloads = pd;
    lp.payload_stride = L.total;
    lp.hdr_off = L.hdr;
    lp.cand_off = L.cand;
    lp.ncand_off = L.ncand;
    lp.x_off = L.x;
    lp.cand = w_cand;
    lp.n_cand = w_ncand;
    lp.out = w_x;
    lp.S = n_shards;
    lp.F = F;
    lp.C = C;
    mfar_lookup_kernel<<<dim3((C + 255) / 256, Q), dim3(256), 0, st>>>(lp);
    HIPCHK(hipGetLastError());
    RETCHK(run_mix(w_x, w_cand, w_ncand, qd, Wd, query_cond, md, Q, C, F, E, k2, idd, scd, nvd, st));
    RETCHK(copy_back((long long*)ids, idd, (size_t)Q * k2, on_device, st));
    RETCHK(copy_back(scores, scd, (size_t)Q * k2, on_device, st));
    RETCHK(copy_back((int*)n_valid, nvd, (size_t)Q, on_device, st));
    // the shared scratch is reused by the next call: finish before releasing the lock
    if (!own_ws) HIPCHK(hipStreamSynchronize(st));
    return MFAR_OK;
}

// ------------------------------------------------------------------------------------------------ lists-first exchange
extern "C" int64_t mfar_lists_bytes(int Q, int n_fields, int k1) {
    if (Q < 0 || n_fields <= 0 || k1 <= 0) return 0;
    return lists_layout(Q, n_fields, k1).total;
}
extern "C" int64_t mfar_topk_bytes(int Q, int k2) {
    if (Q < 0 || k2 <= 0) return 0;
    return topk_layout(Q, k2).total;
}

extern "C" int mfar_retrieve_lists(mfar_index* idx, const float* q, int Q, int k1, int sentinel, void* lists, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    if (!lists) return fail(MFAR_ERR_INVALID, "lists is NULL");
    HIPCHK(hipSetDevice(idx->device));
    if (Q == 0) return MFAR_OK;
    const ListsLayout L = lists_layout(Q, idx->F, k1);
    return run_stage1(idx, q, Q, k1, sentinel, (long long*)((char*)lists + L.ids), (float*)((char*)lists + L.scores), (hipStream_t)stream);
}

static int search_owned(mfar_index* idx, const void* gathered_lists, int n_shards, const float* q, int Q, const float* W,
                        int query_cond, const float* mask, int n_masks, int k1, int k2, int sentinel, int slot, const int32_t* any_fail,
                        void* topk, void* stream) {
    RETCHK(check_search_common(idx, q, Q, k1));
    const int F = idx->F, E = idx->E, C = F * k1;
    RETCHK(check_mix(Q, C, F, E, k2, q, W, query_cond));
    if (!gathered_lists || !topk || n_shards <= 0 || n_shards > 64) return fail(MFAR_ERR_INVALID, "bad lists / topk / n_shards");
    if (slot < 0 || slot >= MFAR_SLOTS) return fail(MFAR_ERR_INVALID, "slot must be in [0, 4)");
    if (Q == 0) return MFAR_OK;
    HIPCHK(hipSetDevice(idx->device));
    hipStream_t st = (hipStream_t)stream;
    const ListsLayout LL = lists_layout(Q, F, k1);
    const TopkLayout TL = topk_layout(Q, k2);
    const size_t nl = (size_t)Q * F * k1;
    // per-slot scratch: merged lists (ids | scores), global union, owned run, counts, owned score vectors
    DevBuf& ws = idx->own[slot];
    const size_t off_lids = 0, off_lsc = off_lids + nl * 8, off_cand = off_lsc + ((nl * 4 + 255) & ~(size_t)255),
                 off_owned = off_cand + (size_t)Q * C * 8, off_ncand = off_owned + (size_t)Q * C * 8,
                 off_nowned = off_ncand + (((size_t)Q * 4 + 255) & ~(size_t)255), off_x = off_nowned + (((size_t)Q * 4 + 255) & ~(size_t)255);
    RETCHK(ws.ensure(off_x + (size_t)Q * C * F * 4));
    char* wb = ws.as<char>();
    long long* lids = (long long*)(wb + off_lids);
    float* lsc = (float*)(wb + off_lsc);
    long long* cand = (long long*)(wb + off_cand);
    long long* owned = (long long*)(wb + off_owned);
    int* ncand = (int*)(wb + off_ncand);
    int* nowned = (int*)(wb + off_nowned);
    float* x = (float*)(wb + off_x);
    ShardMergeParams sp = {};
    sp.payloads = (const char*)gathered_lists;
    sp.payload_stride = LL.total;
    sp.ids_off = LL.ids;
    sp.scores_off = LL.scores;
    sp.out_ids = lids;
    sp.out_scores = lsc;
    sp.S = n_shards;
    sp.F = F;
    sp.k = k1;
    sp.sentinel = sentinel;
    if (n_shards * k1 <= 8 * 256) mfar_merge_shards_kernel<8><<<dim3(Q * F), dim3(256), SEL_LDS_BYTES(n_shards * k1), st>>>(sp);
    else mfar_merge_shards_kernel<32><<<dim3(Q * F), dim3(256), SEL_LDS_BYTES(n_shards * k1), st>>>(sp);
    HIPCHK(hipGetLastError());
    mfar_union_kernel<<<dim3(Q), dim3(256), 0, st>>>(lids, F, k1, cand, ncand);
    HIPCHK(hipGetLastError());
    mfar_filter_owned_kernel<<<dim3(Q), dim3(64), 0, st>>>(cand, ncand, C, idx->row_offset, idx->row_offset + idx->n_rows, owned, nowned);
    HIPCHK(hipGetLastError());
    // the local top-k2 of the OWNED candidates: the two-level stage 2 prunes against the owned set's own k2-th lower bound
    const long long* cm = owned;
    const int* nm = nowned;
    if (two_level_ok(idx, C, k2, query_cond, n_masks))
        RETCHK(run_two_level(idx, q, Q, W, query_cond, mask, n_masks, k2, owned, nowned, C, slot, x, &cm, &nm, lids, lsc, k1, sentinel, st));
    else RETCHK(run_score(idx, q, Q, owned, nowned, C, x, st));      // (merged lists of ALL shards: another shard's list may come from its plain bf16 pass)
    // one top-k payload per mask (a sweep of field masks shares everything up to here: mfar_search_owned_masks)
    for (int m = 0; m < n_masks; ++m) {
        char* tb = (char*)topk + (size_t)m * TL.total;
        RETCHK(run_mix(x, cm, nm, q, W, query_cond, mask ? mask + (size_t)m * F : nullptr, Q, C, F, E, k2, (long long*)(tb + TL.ids),
                       (float*)(tb + TL.scores), nullptr, st));
        HIPCHK(hipMemcpyAsync(tb + TL.ncand, ncand, (size_t)Q * 4, hipMemcpyDeviceToDevice, st));
        if (any_fail) HIPCHK(hipMemcpyAsync(tb + TL.flag, any_fail, 4, hipMemcpyDeviceToDevice, st));
        else HIPCHK(hipMemsetAsync(tb + TL.flag, 0, 4, st));
    }
    return MFAR_OK;
}

extern "C" int mfar_search_owned(mfar_index* idx, const void* gathered_lists, int n_shards, const float* q, int Q, const float* W,
                                 int query_cond, const float* mask, int k1, int k2, int sentinel, int slot, const int32_t* any_fail,
                                 void* topk, void* stream) {
    return search_owned(idx, gathered_lists, n_shards, q, Q, W, query_cond, mask, 1, k1, k2, sentinel, slot, any_fail, topk, stream);
}

extern "C" int mfar_search_owned_masks(mfar_index* idx, const void* gathered_lists, int n_shards, const float* q, int Q, const float* W,
                                       int query_cond, const float* masks, int n_masks, int k1, int k2, int sentinel, int slot,
                                       const int32_t* any_fail, void* topk, void* stream) {
    if (n_masks <= 0 || !masks) return fail(MFAR_ERR_INVALID, "n_masks must be positive and masks non-NULL");
    return search_owned(idx, gathered_lists, n_shards, q, Q, W, query_cond, masks, n_masks, k1, k2, sentinel, slot, any_fail, topk, stream);
}

extern "C" int mfar_merge_topk(int device, const void* gathered_topk, int n_shards, int Q, int k2, int64_t* ids, float* scores,
                               int32_t* n_valid, int32_t* any_fail, void* stream) {
    if (!gathered_topk || !ids || !scores || n_shards <= 0 || n_shards > 64) return fail(MFAR_ERR_INVALID, "bad arguments");
    if (k2 <= 0 || k2 > MFAR_MAX_K || Q < 0) return fail(MFAR_ERR_INVALID, "bad k2 / Q");
    if (Q == 0) return MFAR_OK;
    int ndev = 0;
    RETCHK(mfar_device_count(&ndev));
    if (device < 0 || device >= ndev || device >= 16) return fail(MFAR_ERR_INVALID, "no such device");
    HIPCHK(hipSetDevice(device));
    RETCHK(set_kernel_attrs(device));
    const TopkLayout TL = topk_layout(Q, k2);
    TopkMergeParams p = {};
    p.payloads = (const char*)gathered_topk;
    p.stride = TL.total;
    p.ids_off = TL.ids;
    p.scores_off = TL.scores;
    p.ncand_off = TL.ncand;
    p.flag_off = TL.flag;
    p.any_fail = (int*)any_fail;
    p.ids = (long long*)ids;
    p.scores = scores;
    p.n_valid = (int*)n_valid;
    p.S = n_shards;
    p.k = k2;
    hipStream_t st = (hipStream_t)stream;
    if (n_shards * k2 <= 8 * 256) mfar_merge_topk_kernel<8><<<dim3(Q), dim3(256), SEL_LDS_BYTES(n_shards * k2), st>>>(p);
    else mfar_merge_topk_kernel<32><<<dim3(Q), dim3(256), SEL_LDS_BYTES(n_shards * k2), st>>>(p);
    HIPCHK(hipGetLastError());
    return MFAR_OK;
}

#include "mfar_pipeline.h"
