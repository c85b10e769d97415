#!/usr/bin/env python3
"""bench.py -- queries/sec of the mFAR dense multi-field scorer on MI355X (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W)

Workload (config.workload): the corpus BASELINE.json's metric is quoted on -- 1,000,000 docs x 8 dense fields x 768-d
fp32, synthetic and STaRK-amazon shaped (mfar/synth.py) -- resident in HBM as the tiled slab, row-sharded over the N
ranks exactly like the reference shards its corpus encode (reference mfar/modeling/contrastive.py:470).  One step =
one pass of the hot path over one batch of 64 query embeddings (reference default dev_batch_size, train.py:45):
per-field exhaustive top-100 -> candidate union -> re-score -> mask -> query-conditioned field-weight softmax ->
top-100, i.e. RetrievalTrainingModule.trec_eval_step (contrastive.py:669-704) with the encoder forward excluded
(query embeddings are inputs, already in HBM).  N > 1: lists-first exchange (two small RCCL all-gathers per batch, every rank
re-scores only the candidates it owns).  The corpus is fixed while N grows: strong scaling.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel from its HIP-event duration measured on the
launch stream.  Default (fp32 index, certified fp16 screen on): `mfar_stage1_f16_kernel`, HBM-bound, algorithmic bytes =
D_local * F * E * 2 per launch (the fp16 screen slab is read once per batch).  `--screen off`: `mfar_stage1_kernel`,
fp32-MFMA-bound, algorithmic flops = 2 * D_local * F * E * 64 per launch.  The results are bit-identical in both modes.
`cpu_baseline` times the oracle's torch port of the reference algorithm (same torch ops as the reference's CPU path)
on a bounded row sample of the same corpus, on this box's host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--fields", type=int, default=8)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--wgs-per-cu", type=int, default=0)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="slab storage (default: the exact fp32 path)")
    ap.add_argument("--screen", choices=["auto", "off"], default="auto",
                    help="f32 only: certified fp16 screening of stage 1 (bit-identical results; csrc/mfar_screen.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-docs", type=int, default=100_000)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    N = world
    dist = None
    if N > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # MFAR_BENCH_BACKEND=gloo + MFAR_BENCH_SHARE_GPU=1: dry-run of the N > 1 control flow on a one-GPU box (all ranks
        # on cuda:0, host-staged collectives).  The driver's scaling runs use the default: nccl (= RCCL), one GPU per rank.
        backend = os.environ.get("MFAR_BENCH_BACKEND", "nccl")
        if os.environ.get("MFAR_BENCH_SHARE_GPU") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=N, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=N)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from mfar import synth
    from mfar.data import index as idxmod

    D, F, E, Q = args.docs, args.fields, args.dim, args.batch
    K1 = K2 = 100
    # distinct synthetic queries (each plants up to 5 relevant documents); long runs cycle through them
    n_q_total = max(Q, min(max(4096, (args.steps + args.warmup) * Q), 65536, D // 16))
    t_build = time.time()
    corpus = synth.SyntheticCorpus(D, F, E, n_queries=n_q_total, seed=0xDEADBEEF, device=str(dev))
    row0, row1 = D * rank // N, D * (rank + 1) // N          # contrastive.py:470
    ix = corpus.build_index(idxmod, row0=row0, n=row1 - row0, dtype=args.dtype)
    if args.wgs_per_cu:
        ix.set_wgs_per_cu(args.wgs_per_cu)
    if args.screen == "off":
        ix.set_screen(0)
    t_build = time.time() - t_build
    W = corpus.W
    mask = torch.ones(F, device=dev)

    from mfar.data.pipeline import PipelinedSearcher
    # Two-deep pipeline: stage 1 of batch i+1 (main stream) overlaps the tail of batch i (side stream).  Every batch is
    # still processed completely inside the timed region (the region ends with a full device synchronisation).
    ps = PipelinedSearcher(ix, W, mask, k1=K1, k2=K2, sentinel=True, query_cond=True, max_batch=Q)

    def run(first, n, keep):
        prev = None
        for i in range(n):
            t = ps.submit(corpus.queries((first + i) * Q, Q))
            if prev is not None and keep is not None:
                r = ps.result(prev)
                keep.append((r["ids"].clone(), r["n_valid"].clone()))
            prev = t
        if prev is not None and keep is not None:
            r = ps.result(prev)
            keep.append((r["ids"].clone(), r["n_valid"].clone()))

    # Setup, not warm-up: the first two batches allocate the pipeline's scratch (both slots) and build the fp16 screen slab
    # of an fp32 index (one pass over the corpus, part of index construction).  --warmup steps follow as asked.
    t_prime = time.time()
    run(0, 2, None)
    torch.cuda.synchronize()
    t_build += time.time() - t_prime
    results = []
    run(0, args.warmup, None)
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ix.set_timing(True)
    t0 = time.perf_counter()
    run(args.warmup, args.steps, results)
    torch.cuda.synchronize()
    if N > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1_ms, s1_n = ix.stage1_timing()
    ix.set_timing(False)
    if N > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        # Recall@20 against the synthetic qrels (quality gate named by the metric)
        rec = []
        for i, (ids, nv) in enumerate(results):
            ids = ids.cpu().numpy()
            rel = corpus.qrels((args.warmup + i) * Q, Q)
            for j in range(Q):
                rec.append(len(set(ids[j, :20].tolist()) & rel[j]) / len(rel[j]))
        recall20 = float(np.mean(rec))
        checksum = None
        if os.environ.get("MFAR_BENCH_DUMP_IDS") == "1":      # used by tests/test_gpu_multirank.py
            import hashlib
            h = hashlib.sha256()
            for ids, _ in results:
                h.update(ids.cpu().numpy().tobytes())
            checksum = h.hexdigest()
        qps = args.steps * Q / dt
        s1_avg_ms = s1_ms / max(1, s1_n)
        flops_per_launch = 2.0 * (row1 - row0) * F * E * 64      # algorithmic: 2*D*F*E per query x 64 queries
        scr = ix.screen_stats()
        screened = args.dtype == "f32" and scr["built"]          # stage 1 ran on the fp16 screen slab of the fp32 index
        esize = 2 if (args.dtype == "bf16" or screened) else 4
        bytes_per_launch = float(row1 - row0) * F * E * esize    # the scanned slab is read once per batch
        # 16-bit passes: the register-ring kernels ("...r") run when dim / 16 divides into their 6 register slots
        rr = "" if os.environ.get("MFAR_S1_REGRING", "1") == "0" else ("r" if (E // 16) % 6 == 0 else ("r4" if (E // 16) % 4 == 0 else ""))
        s1_kernel = f"mfar_stage1_bf16{rr}_kernel" if args.dtype == "bf16" else (f"mfar_stage1_f16{rr}_kernel" if screened else "mfar_stage1_kernel")
        achieved_tf = flops_per_launch / (s1_avg_ms * 1e-3) / 1e12 if s1_avg_ms > 0 else 0.0
        traffic = None          # HBM bytes per stage-1 launch from the committed PMC pass of this same workload
        tj = os.path.join(ROOT, "profiles", "r01_stage1_bf16_traffic.json" if args.dtype == "bf16" else
                          ("r01_stage1_f16_traffic.json" if screened else "r01_stage1_traffic.json"))
        if N == 1 and (D, F, E, Q) == (1_000_000, 8, 768, 64) and os.path.exists(tj):
            t_ = json.load(open(tj))
            if t_.get("kernel") == s1_kernel:
                traffic = t_["hbm_read_bytes_per_launch"] + t_["hbm_write_bytes_per_launch"]
        line = {
            "metric": "queries/sec (whole node) at Recall@20 parity, 1M-doc x 8-field x 768d corpus",
            "value": qps, "unit": "queries/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 docs x fp32 queries (3 exact bf16 terms), fp32 accumulate", "data": "synthetic",
            "config": {"workload": f"synthetic STaRK-amazon-shaped corpus, {D} docs x {F} dense fields x {E}d {args.dtype}, "
                                   f"row-sharded over {N} GPU(s); two-stage scorer k1=k2=100, zero-sentinel mode",
                       "docs": D, "fields": F, "dim": E, "query_batch": Q, "k1": K1, "k2": K2,
                       "parallelism": (f"row-shard x{N}, lists-first exchange over RCCL (per batch: all-gather of the stage-1 lists, "
                                       f"all-gather of the local top-k, all-reduce of the certificate flag)") if N > 1 else "single shard",
                       "pipeline": "2 batches in flight (stage 1 of batch i+1 overlaps the tail of batch i)"},
            "stage1": ("certified fp16 screen of the fp32 slab (k+64 rows per list re-scored with the exact fp32 chain, top-k proven "
                       "or redone by the exact fp32 pass per field): outputs bit-identical to the plain fp32 pass" if screened else
                       ("exact fp32 MFMA pass" if args.dtype == "f32" else "bf16 slab pass")),
            "screen": ({"lists_certified": scr["n_checked"] - scr["n_failed"], "lists_redone_exactly": scr["n_failed"],
                        "screen_slab_bytes": scr["screen_bytes"]} if screened else None),
            "recall_at_20": recall20, "ids_checksum": checksum,
            "index_build_s": t_build,
            "roofline": ({"bound": "mfma", "kernel": s1_kernel, "achieved": achieved_tf,
                          "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS}
                         if esize == 4 else
                         {"bound": "hbm", "kernel": s1_kernel, "achieved": bytes_per_launch / (s1_avg_ms * 1e-3) / 1e9,
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bytes_per_launch / (s1_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}) | {
                         "traffic": traffic, "traffic_source": os.path.relpath(tj, ROOT) + " (rocprofv3 PMC pass)" if traffic else None, "avg_launch_ms": s1_avg_ms, "launches": s1_n,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "hbm_GBps_algorithmic": bytes_per_launch / (s1_avg_ms * 1e-3) / 1e9 if s1_avg_ms > 0 else 0.0,
                         "hbm_frac_of_8TBps": bytes_per_launch / (s1_avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if s1_avg_ms > 0 else 0.0},
        }
        if N == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(corpus, ix, idxmod, args, np, torch)
        print(json.dumps(line), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(corpus, ix, idxmod, args, np, torch):
    """The reference CPU path, restated (oracle torch port = same torch ops as index.py:181-232 + weighting.py:17-29 +
    contrastive.py:669-704), on the first `cpu_sample_docs` rows of the same corpus, all host cores.  The oracle is
    only the thing timed/checked here, never part of the GPU path."""
    from oracle import mfar_oracle as O
    Ds = min(args.cpu_sample_docs, ix.n_rows)
    F, E, Q = ix.n_fields, ix.dim, args.batch
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    slab = np.empty((F, Ds, E), dtype=np.float32)
    for f in range(F):
        ix.read_rows(f, 0, Ds, out=slab[f])
    W = corpus.W.cpu().numpy()
    mask = np.ones(F, dtype=np.float32)
    # GPU result on the same sample, for the parity gate
    sub = idxmod.MultiFieldIndex(Ds, F, E, device=ix.device)
    for f in range(F):
        sub.write_rows(f, 0, slab[f])
    # the reference lets torch pick its thread count (= all cores); on many-core hosts the per-query python loop of
    # small ops runs faster with fewer threads, so time one batch with 32 threads and one with all and keep the faster
    q0 = corpus.queries(0, Q).cpu().numpy()
    best_threads, best_t = cores, None
    for th in sorted({min(32, cores), cores}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        O.ref_two_stage(slab, q0, W, mask)
        dt0 = time.perf_counter() - t0
        if best_t is None or dt0 < best_t:
            best_threads, best_t = th, dt0
    torch.set_num_threads(best_threads)
    cores = best_threads
    n_batches, t_used, match, match_tol = 0, 0.0, [], []
    while t_used < 12.0 and n_batches < 50:
        q = corpus.queries(n_batches * Q, Q).cpu().numpy()
        t0 = time.perf_counter()
        ci, cs = O.ref_two_stage(slab, q, W, mask)
        t_used += time.perf_counter() - t0
        g = sub.search(q, W, mask)
        match.append(float(np.mean([np.array_equal(g["ids"][i, :20], ci[i, :20]) for i in range(Q)])))
        ok = 0
        for i in range(Q):
            try:
                O.assert_topk_equivalent(g["ids"][i], g["scores"][i], ci[i], cs[i], tol=1e-4)
                ok += 1
            except AssertionError:
                pass
        match_tol.append(ok / Q)
        n_batches += 1
    sub.close()
    qps_sample = n_batches * Q / t_used
    return {"value": qps_sample * Ds / corpus.D, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{n_batches} batches of {Q} queries over the first {Ds} docs x {F} fields x {E}d of the same corpus "
                      f"({qps_sample:.1f} q/s on the sample, scaled by {Ds}/{corpus.D} to the full corpus; work is linear in docs)",
            "top20_ids_identical_to_gpu": float(np.mean(match)),
            "top100_equivalent_to_gpu_within_1e-4": float(np.mean(match_tol))}


if __name__ == "__main__":
    main()
