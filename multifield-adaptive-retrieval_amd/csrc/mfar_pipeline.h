// mfar_pipeline.h -- the batch pipeline behind the C ABI (mfar_pipeline_* in include/mfar_hip.h).  Included at the end of mfar_hip.hip:
// it drives the split-phase entry points of that file (stage1_block, run_stage1, run_stage2_mix) on its own HIP streams.
//
// The reference scores one batch of dev_batch_size = 64 queries at a time, synchronously (modeling/contrastive.py:559-563 -> 669-704).
// Here a LAUNCH is one long scan (stage 1) and a chain of short gather / selection kernels around it; the pipeline keeps the scans of
// consecutive launches back to back on a high-priority stream and runs everything that follows a scan (exact re-scoring + certificate,
// candidate union, stage 2, mixer) on one of two side streams beside the NEXT launch's scan, `depth` launches in flight (three slots of
// scratch).  When the index offers the wide pass (mfar_max_split_batch() == 128) two consecutive 64-query batches are COALESCED into one
// launch: the scan reads the slab once per 128 queries -- half the scan bytes per query; a query's results do not depend on its
// neighbours, so this is invisible except in throughput.  A launch whose certificate failed is redone exactly when its result is taken.
// Single shard (the row-sharded exchange keeps its collectives in the host language: mfar/data/pipeline.py).  Results: bit for bit those
// of mfar_search_two_stage.
#pragma once

#define PIPE_MAX_DEPTH MFAR_SLOTS
// (PipeStreams / pipe_streams(): mfar_hip.hip, next to the slab mutators that order themselves behind these streams)

struct mfar_pipeline {
    mfar_index* idx = nullptr;
    int device = 0;
    PipeStreams* st = nullptr;
    int k1 = 100, k2 = 100, sentinel = 1, query_cond = 1;
    int Qb = 64, coalesce = 1, depth = 3, Qmax = 64;
    DevBuf W, mask;                       // the pipeline's own copies (replaced by mfar_pipeline_set_weights)
    bool has_mask = false;
    struct Slot {
        DevBuf q, ids, scores, n_valid, fid, fsc, fail;
        int* fail_host = nullptr;         // pinned
        hipEvent_t stage1 = nullptr, done = nullptr, copied = nullptr;
        hipEvent_t qready = nullptr;      // on the scan stream where the launch's begin phase starts: its queries are in the slot
        hipEvent_t taken = nullptr;       // behind the last result / lists copy out of this slot on a CALLER's stream (device pointers)
        bool taken_pending = false;
        hipStream_t taken_stream = nullptr;
        int Q = 0;
        bool checked = true;
        bool failed = false;              // pipe_launch returned an error part-way (out of memory ...): nothing valid was enqueued for this launch
        long long launch = -1;
    } slots[PIPE_MAX_DEPTH];
    struct Where {
        long long ticket, launch;
        int off, Q;
    };
    std::vector<Where> where;             // batches held for coalescing + launches still in flight
    std::vector<Where> pending;
    long long n_submitted = 0, n_launched = 0, n_redone = 0;
};

// Copies out of a slot were enqueued on the caller's stream `st`: the slot's next launch waits for them (pipe_launch), whatever stream the
// caller submits on.  One event per slot: copies taken on a second stream are first ordered behind the earlier ones, so that the latest
// record stands for all of them.
static int pipe_mark_taken(mfar_pipeline::Slot& s, hipStream_t st) {
    if (s.taken_pending && s.taken_stream != st) HIPCHK(hipStreamWaitEvent(st, s.taken, 0));
    HIPCHK(hipEventRecord(s.taken, st));
    s.taken_pending = true;
    s.taken_stream = st;
    return MFAR_OK;
}

static int pipe_tail(mfar_pipeline* p, mfar_pipeline::Slot& s, int slot, hipStream_t st) {
    return run_stage2_mix(p->idx, s.q.as<float>(), s.Q, p->W.as<float>(), p->query_cond, p->has_mask ? p->mask.as<float>() : nullptr, 1, p->k1, p->k2,
                          s.fid.as<long long>(), s.fsc.as<float>(), p->sentinel, slot, s.ids.as<long long>(), s.scores.as<float>(), s.n_valid.as<int>(),
                          nullptr, st);
}

static int pipe_launch(mfar_pipeline* p) {
    mfar_index* idx = p->idx;
    const int slot = (int)(p->n_launched % p->depth);
    mfar_pipeline::Slot& s = p->slots[slot];
    hipStream_t side = p->st->side[p->depth > 2 ? (int)(p->n_launched % 2) : 0];
    int Q = 0;
    for (const auto& w : p->pending) Q += w.Q;
    p->pending.clear();
    // this launch overwrites the results of the slot's previous launch
    p->where.erase(std::remove_if(p->where.begin(), p->where.end(), [&](const mfar_pipeline::Where& w) { return w.launch <= p->n_launched - p->depth; }),
                   p->where.end());
    s.Q = Q;
    s.checked = false;
    s.launch = p->n_launched;
    s.failed = true;                 // until everything below is enqueued: a launch that errors out is redone by pipe_check, never read as is
    p->n_launched++;
    idx->pipe_launches++;            // (mfar_index_write_rows orders itself behind these: order_after_pipelines)
    if (s.taken_pending) {           // copies out of this slot enqueued on a caller's stream: this launch's begin phase rewrites the slot's lists,
        HIPCHK(hipStreamWaitEvent(p->st->main, s.taken, 0));       // its tail (ordered behind the begin phase by s.stage1) the results
        s.taken_pending = false;
    }
    HIPCHK(hipEventRecord(s.qready, p->st->main));
    RETCHK(stage1_block(idx, slot, S1_PREPARE | S1_SCAN | (merge_in_finish() ? 0 : S1_FINISH), s.q.as<float>(), Q, 0, p->k1, p->sentinel, 0, idx->F,
                        s.fid.as<long long>(), s.fsc.as<float>(), nullptr, p->st->main));
    if (Q > idx->s1[slot].qw) return fail(MFAR_ERR_UNSUPPORTED, "a coalesced launch needs the screen slab, which could not be (re)built");
    HIPCHK(hipEventRecord(s.stage1, p->st->main));
    // what the tail needs from q and W alone (field weights, q . mean, eps of stage 2's approximate level): beside the scan, not behind it
    // (enqueued after the begin phase on the host: a rebuild of the screen, if rows were written, has happened by now)
    HIPCHK(hipStreamWaitEvent(side, s.qready, 0));
    RETCHK(run_stage2_pre(idx, s.q.as<float>(), Q, p->W.as<float>(), p->query_cond, slot, side));
    HIPCHK(hipStreamWaitEvent(side, s.stage1, 0));
    // finish REPORTS a failed certificate (read in pipe_check); when failures are frequent the library repairs on the device instead and
    // switches fields that keep failing off (mfar_hip.hip "adaptive policy"): nothing here latches
    const bool inline_rep = idx->pol.inline_repair;
    RETCHK(stage1_block(idx, slot, S1_CERTIFY | (merge_in_finish() ? S1_FINISH : 0), s.q.as<float>(), Q, 0, p->k1, p->sentinel, 0, idx->F,
                        s.fid.as<long long>(), s.fsc.as<float>(), inline_rep ? nullptr : s.fail.as<int>(), side));
    if (inline_rep) HIPCHK(hipMemsetAsync(s.fail.p, 0, 4, side));
    RETCHK(pipe_tail(p, s, slot, side));
    HIPCHK(hipMemcpyAsync(s.fail_host, s.fail.p, 4, hipMemcpyDeviceToHost, side));
    HIPCHK(hipEventRecord(s.done, side));
    s.failed = false;
    return MFAR_OK;
}

// host side of the certificate: wait for the launch, redo it exactly if its screen could not be proven
static int pipe_check(mfar_pipeline* p, long long launch) {
    if (launch < 0) return MFAR_OK;
    const int slot = (int)(launch % p->depth);
    mfar_pipeline::Slot& s = p->slots[slot];
    if (s.checked || s.launch != launch) return MFAR_OK;
    if (s.failed) {                  // the launch never made it onto the streams (its error went to the caller of submit / flush): the batch's
        HIPCHK(hipDeviceSynchronize());      // queries are in the slot, so it is run now, synchronously -- or the error repeats
        RETCHK(run_stage1(p->idx, s.q.as<float>(), s.Q, p->k1, p->sentinel, s.fid.as<long long>(), s.fsc.as<float>(), p->st->main));
        RETCHK(pipe_tail(p, s, slot, p->st->main));
        HIPCHK(hipStreamSynchronize(p->st->main));
        s.failed = false;
        s.checked = true;
        p->n_redone++;
        return MFAR_OK;
    }
    HIPCHK(hipEventSynchronize(s.done));
    if (s.fail_host[0] == 0) {
        s.checked = true;
        return MFAR_OK;
    }
    p->n_redone++;
    if (p->idx->row_mode_setting != 0) p->idx->row_mask = p->idx->row_eligible;
    HIPCHK(hipDeviceSynchronize());                 // the redo uses the index's slot-0 stage-1 scratch: nothing else may be in flight
    // the non-split stage 1 repairs a failed certificate itself: screened pass again, then the exact pass for the failed fields only
    RETCHK(run_stage1(p->idx, s.q.as<float>(), s.Q, p->k1, p->sentinel, s.fid.as<long long>(), s.fsc.as<float>(), p->st->main));
    RETCHK(pipe_tail(p, s, slot, p->st->main));
    HIPCHK(hipStreamSynchronize(p->st->main));
    s.fail_host[0] = 0;
    s.checked = true;                // (only now: a redo that errors out is attempted again, its half-written slot never returned)
    return MFAR_OK;
}

extern "C" void mfar_pipeline_destroy(mfar_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);        // (never through p->idx: the index may be gone already)
    (void)hipDeviceSynchronize();
    for (auto& s : p->slots) {
        for (DevBuf* b : {&s.q, &s.ids, &s.scores, &s.n_valid, &s.fid, &s.fsc, &s.fail}) b->release();
        if (s.fail_host) (void)hipHostFree(s.fail_host);
        for (hipEvent_t e : {s.stage1, s.done, s.copied, s.taken, s.qready})
            if (e) (void)hipEventDestroy(e);
    }
    p->W.release();
    p->mask.release();
    delete p;
}

static int pipe_upload_weights(mfar_pipeline* p, const float* W, const float* mask, int on_device) {
    mfar_index* idx = p->idx;
    const size_t nW = p->query_cond ? (size_t)idx->E * idx->F : (size_t)idx->F;
    RETCHK(p->W.ensure(nW * 4));
    RETCHK(p->mask.ensure((size_t)idx->F * 4));
    HIPCHK(hipMemcpy(p->W.p, W, nW * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    p->has_mask = mask != nullptr;
    if (mask) HIPCHK(hipMemcpy(p->mask.p, mask, (size_t)idx->F * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    return MFAR_OK;
}

extern "C" int mfar_pipeline_create(mfar_pipeline** out, mfar_index* idx, const float* W, int query_cond, const float* mask, int k1, int k2,
                                    int sentinel, int max_batch, int depth, int coalesce, int on_device) {
    if (!out) return fail(MFAR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!idx || !W) return fail(MFAR_ERR_INVALID, "idx / W is NULL");
    if (k1 <= 0 || k1 > MFAR_MAX_K || k2 <= 0 || k2 > MFAR_MAX_K) return fail(MFAR_ERR_INVALID, "k1, k2 must be in [1, 128]");
    if ((long long)idx->F * k1 > 4096) return fail(MFAR_ERR_INVALID, "n_fields * k1 must be <= 4096");
    if (idx->E * 4 > 60 * 1024) return fail(MFAR_ERR_UNSUPPORTED, "dim too large for the stage-2 kernel");
    if (depth == 0) depth = 3;
    if (depth < 2 || depth > PIPE_MAX_DEPTH) return fail(MFAR_ERR_INVALID, "depth must be 2, 3 or 4 (0 = 3)");
    if (max_batch <= 0) return fail(MFAR_ERR_INVALID, "max_batch must be positive");
    HIPCHK(hipSetDevice(idx->device));
    const int cap = mfar_max_split_batch(idx, k1);                  // 128 with the wide screened pass (builds the screen), else 64
    if (max_batch > cap) return fail(MFAR_ERR_INVALID, "max_batch exceeds mfar_max_split_batch() of this index");
    if (coalesce == 0) coalesce = std::max(1, std::min(2, cap / max_batch));
    if (coalesce < 1 || coalesce * max_batch > cap) return fail(MFAR_ERR_INVALID, "coalesce x max_batch must not exceed mfar_max_split_batch()");
    mfar_pipeline* p = new (std::nothrow) mfar_pipeline();
    if (!p) return fail(MFAR_ERR_NOMEM, "host allocation failed");
    p->idx = idx;
    p->device = idx->device;
    p->k1 = k1;
    p->k2 = k2;
    p->sentinel = sentinel != 0;
    p->query_cond = query_cond != 0;
    p->Qb = max_batch;
    p->coalesce = coalesce;
    p->depth = depth;
    p->Qmax = max_batch * coalesce;
    int rc = pipe_streams(idx->device, &p->st);
    if (rc == MFAR_OK) rc = pipe_upload_weights(p, W, mask, on_device);
    const size_t Qm = (size_t)p->Qmax, F = (size_t)idx->F;
    for (int i = 0; i < depth && rc == MFAR_OK; ++i) {
        mfar_pipeline::Slot& s = p->slots[i];
        DevBuf* bufs[] = {&s.q, &s.ids, &s.scores, &s.n_valid, &s.fid, &s.fsc, &s.fail};
        const size_t bytes[] = {Qm * idx->E * 4, Qm * k2 * 8, Qm * k2 * 4, Qm * 4, Qm * F * k1 * 8, Qm * F * k1 * 4, 4};
        for (int j = 0; j < 7 && rc == MFAR_OK; ++j) rc = bufs[j]->ensure(bytes[j], true);
        if (rc != MFAR_OK) break;
        if (hipMemset(s.fail.p, 0, 4) != hipSuccess || hipHostMalloc((void**)&s.fail_host, 4, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s.stage1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s.taken, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.qready, hipEventDisableTiming) != hipSuccess)
            rc = fail(MFAR_ERR_HIP, "pipeline slot: event / pinned allocation failed");
        else {
            s.fail_host[0] = 0;
            if (hipEventRecord(s.done, p->st->side[0]) != hipSuccess) rc = fail(MFAR_ERR_HIP, "hipEventRecord failed");
        }
    }
    if (rc != MFAR_OK) {
        const std::string keep = g_err;
        mfar_pipeline_destroy(p);
        g_err = keep;
        return rc;
    }
    idx->repair_sample = true;      // repairs are launched here only after a failure was reported (or when they are frequent)
    *out = p;
    return MFAR_OK;
}

extern "C" int mfar_pipeline_info(const mfar_pipeline* p, int* depth, int* coalesce, int* queries_per_launch, int* lag, int64_t* n_redone) {
    if (!p) return fail(MFAR_ERR_INVALID, "pipeline is NULL");
    if (depth) *depth = p->depth;
    if (coalesce) *coalesce = p->coalesce;
    if (queries_per_launch) *queries_per_launch = p->Qmax;
    if (lag) *lag = p->depth * p->coalesce - 1;
    if (n_redone) *n_redone = p->n_redone;
    return MFAR_OK;
}

extern "C" int mfar_pipeline_flush(mfar_pipeline* p) {
    if (!p) return fail(MFAR_ERR_INVALID, "pipeline is NULL");
    HIPCHK(hipSetDevice(p->idx->device));
    if (!p->pending.empty()) RETCHK(pipe_launch(p));
    return MFAR_OK;
}

extern "C" int mfar_pipeline_set_weights(mfar_pipeline* p, const float* W, const float* mask, int on_device) {
    if (!p || !W) return fail(MFAR_ERR_INVALID, "pipeline / W is NULL");
    RETCHK(mfar_pipeline_flush(p));
    for (long long L = std::max(0LL, p->n_launched - p->depth); L < p->n_launched; ++L) RETCHK(pipe_check(p, L));      // nothing in flight reads the old ones
    return pipe_upload_weights(p, W, mask, on_device);
}

extern "C" int mfar_pipeline_submit(mfar_pipeline* p, const float* q, int Q, int on_device, void* stream, int64_t* ticket) {
    if (!p || !ticket) return fail(MFAR_ERR_INVALID, "pipeline / ticket is NULL");
    if (Q <= 0 || Q > p->Qb || !q) return fail(MFAR_ERR_INVALID, "a batch holds 1 .. max_batch queries");
    mfar_index* idx = p->idx;
    HIPCHK(hipSetDevice(idx->device));
    const int slot = (int)(p->n_launched % p->depth);
    mfar_pipeline::Slot& s = p->slots[slot];
    hipStream_t cs = on_device ? (hipStream_t)stream : p->st->copy;
    if (p->pending.empty() && !s.checked) RETCHK(pipe_check(p, p->n_launched - p->depth));   // the slot's previous launch must be verified before its buffers go
    HIPCHK(hipStreamWaitEvent(cs, s.done, 0));                                                // ... and its tail has finished with them
    int off = 0;
    for (const auto& w : p->pending) off += w.Q;
    // the rows are copied on the CALLER's stream (device pointers) or on the copy stream (host pointers; pageable memory is staged by the
    // runtime): the scan stream only waits for the event, so the copy runs in the first gap it finds instead of queueing behind the
    // previous launch's list merge
    HIPCHK(hipMemcpyAsync(s.q.as<float>() + (size_t)off * idx->E, q, (size_t)Q * idx->E * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs));
    HIPCHK(hipEventRecord(s.copied, cs));
    HIPCHK(hipStreamWaitEvent(p->st->main, s.copied, 0));                     // the scan stream takes this batch's rows from here on
    if (!on_device) HIPCHK(hipEventSynchronize(s.copied));                    // the caller may reuse its host buffer when submit returns
    const mfar_pipeline::Where w = {p->n_submitted, p->n_launched, off, Q};
    p->pending.push_back(w);
    p->where.push_back(w);
    *ticket = p->n_submitted++;
    if ((int)p->pending.size() == p->coalesce) RETCHK(pipe_launch(p));
    return MFAR_OK;
}

extern "C" int mfar_pipeline_result(mfar_pipeline* p, int64_t ticket, int64_t* ids, float* scores, int32_t* n_valid, int on_device, void* stream) {
    if (!p) return fail(MFAR_ERR_INVALID, "pipeline is NULL");
    if (!ids || !scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    HIPCHK(hipSetDevice(p->idx->device));
    const mfar_pipeline::Where* w = nullptr;
    for (const auto& x : p->where)
        if (x.ticket == ticket) w = &x;
    if (!w) return fail(MFAR_ERR_INVALID, "ticket is no longer (or not yet) in flight");
    const mfar_pipeline::Where ww = *w;                  // (pipe_launch edits the table)
    if (ww.launch == p->n_launched) RETCHK(pipe_launch(p));       // still held for coalescing: launch it alone
    if (ww.launch < p->n_launched - p->depth) return fail(MFAR_ERR_INVALID, "ticket is no longer in flight");
    RETCHK(pipe_check(p, ww.launch));
    mfar_pipeline::Slot& s = p->slots[ww.launch % p->depth];
    hipStream_t st = on_device ? (hipStream_t)stream : p->st->copy;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t k2 = (size_t)p->k2;
    HIPCHK(hipMemcpyAsync(ids, s.ids.as<long long>() + (size_t)ww.off * k2, (size_t)ww.Q * k2 * 8, kind, st));
    HIPCHK(hipMemcpyAsync(scores, s.scores.as<float>() + (size_t)ww.off * k2, (size_t)ww.Q * k2 * 4, kind, st));
    if (n_valid) HIPCHK(hipMemcpyAsync(n_valid, s.n_valid.as<int>() + ww.off, (size_t)ww.Q * 4, kind, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    else RETCHK(pipe_mark_taken(s, st));
    return MFAR_OK;
}

// Device views of a batch's results and of its stage-1 lists inside the launch's slot (no copy): valid until depth * coalesce more
// batches were submitted.  Any pointer may be NULL.
extern "C" int mfar_pipeline_result_view(mfar_pipeline* p, int64_t ticket, const int64_t** ids, const float** scores, const int32_t** n_valid,
                                         const int64_t** field_ids, const float** field_scores) {
    if (!p) return fail(MFAR_ERR_INVALID, "pipeline is NULL");
    HIPCHK(hipSetDevice(p->idx->device));
    const mfar_pipeline::Where* w = nullptr;
    for (const auto& x : p->where)
        if (x.ticket == ticket) w = &x;
    if (!w) return fail(MFAR_ERR_INVALID, "ticket is no longer (or not yet) in flight");
    const mfar_pipeline::Where ww = *w;
    if (ww.launch == p->n_launched) RETCHK(pipe_launch(p));
    if (ww.launch < p->n_launched - p->depth) return fail(MFAR_ERR_INVALID, "ticket is no longer in flight");
    RETCHK(pipe_check(p, ww.launch));
    mfar_pipeline::Slot& s = p->slots[ww.launch % p->depth];
    const size_t k2 = (size_t)p->k2, fk = (size_t)p->idx->F * p->k1;
    if (ids) *ids = (const int64_t*)(s.ids.as<long long>() + (size_t)ww.off * k2);
    if (scores) *scores = s.scores.as<float>() + (size_t)ww.off * k2;
    if (n_valid) *n_valid = s.n_valid.as<int>() + ww.off;
    if (field_ids) *field_ids = (const int64_t*)(s.fid.as<long long>() + (size_t)ww.off * fk);
    if (field_scores) *field_scores = s.fsc.as<float>() + (size_t)ww.off * fk;
    return MFAR_OK;
}

// The batch's stage-1 lists [Q, F, k1] copied out (host: synchronous; device: on `stream`), same validity as its result.
extern "C" int mfar_pipeline_lists(mfar_pipeline* p, int64_t ticket, int64_t* field_ids, float* field_scores, int on_device, void* stream) {
    if (!field_ids || !field_scores) return fail(MFAR_ERR_INVALID, "output pointer is NULL");
    const int64_t* fi = nullptr;
    const float* fs = nullptr;
    RETCHK(mfar_pipeline_result_view(p, ticket, nullptr, nullptr, nullptr, &fi, &fs));
    int Q = 0;
    for (const auto& x : p->where)
        if (x.ticket == ticket) Q = x.Q;
    hipStream_t st = on_device ? (hipStream_t)stream : p->st->copy;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t n = (size_t)Q * p->idx->F * p->k1;
    HIPCHK(hipMemcpyAsync(field_ids, fi, n * 8, kind, st));
    HIPCHK(hipMemcpyAsync(field_scores, fs, n * 4, kind, st));
    if (!on_device) HIPCHK(hipStreamSynchronize(st));
    else
        for (auto& sl : p->slots)
            if (sl.fid.p && fi >= (const int64_t*)sl.fid.p && fi < (const int64_t*)((const char*)sl.fid.p + sl.fid.cap)) RETCHK(pipe_mark_taken(sl, st));
    return MFAR_OK;
}
