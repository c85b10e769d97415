"""The N > 1 path of bench.py end to end on ONE GPU: two ranks share cuda:0 and exchange payloads through gloo
(RCCL refuses two ranks on one device).  Exercises row sharding, split-phase search_local, the all-gather, the merge with
a caller workspace and the two-stream pipeline; the result must match the single-rank run of the same corpus."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(n, extra_env, dtype="f32", launcher=True, extra_args=()):
    env = dict(os.environ, **extra_env)
    env.pop("WORLD_SIZE", None)
    args = ["--docs", "70000", "--fields", "4", "--dim", "128", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs",
            "--dtype", dtype] + list(extra_args)
    if n == 1 or not launcher:      # plain `python bench.py --gpus N`: the script starts its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world,dtype", [(2, "f32"), (4, "f32"), (2, "bf16")])
def test_ranks_on_one_gpu_match_single_rank(world, dtype):
    """The PRODUCT multi-rank path (PipelinedSearcher, lists-first exchange: two all-gathers per batch, the certificate flag
    travelling in the second) with `world` ranks, fp32 (screened) and bf16 slabs."""
    one = _bench(1, {"MFAR_BENCH_DUMP_IDS": "1"}, dtype)
    two = _bench(world, {"MFAR_BENCH_BACKEND": "gloo", "MFAR_BENCH_SHARE_GPU": "1", "MFAR_BENCH_DUMP_IDS": "1"}, dtype)
    assert one["n_gpus"] == 1 and two["n_gpus"] == world and two["scaling"] == "strong"
    assert one["recall_at_20"] > 0.3 and two["recall_at_20"] == one["recall_at_20"]   # (weak planted signal at dim 128)
    assert two["ids_checksum"] == one["ids_checksum"]          # same top-100 ids for every query of every step


def test_tier2_in_the_row_sharded_exchange():
    """Round 6's tier 2 (threshold rescan of lists whose first certificate failed) inside the multi-rank product path: a CLUSTERED corpus
    over two row shards -- every rank's certificates fail, every rank's tier 2 finishes its lists before the lists-first exchange -- returns
    the single-rank run's ids, and rank 0's line names the tier-2 work and both ranks."""
    extra = ["--corpus", "clustered", "--cluster-noise", "1e-2", "--steps", "12"]
    env = {"MFAR_BENCH_DUMP_IDS": "1", "MFAR_SCREEN_TIER2": "2"}
    one = _bench(1, env, extra_args=extra)
    two = _bench(2, dict(env, MFAR_BENCH_BACKEND="gloo", MFAR_BENCH_SHARE_GPU="1"), launcher=False, extra_args=extra)
    assert two["n_gpus"] == 2 and two["rccl"]["world_size"] == 2 and two["config"]["row_shards"] == 2
    assert two["ids_checksum"] == one["ids_checksum"]
    for d in (one, two):
        t2 = d["adaptive"]["tier2"]
        assert t2["lists"] > 0 and t2["passed_on_to_exact"] <= t2["lists"] // 10, t2
        assert d["adaptive"]["off"] == [], d["adaptive"]


def test_bench_starts_its_own_ranks_and_proves_them():
    """`python bench.py --gpus 2` with NO launcher (the driver's command shape): the script spawns its two ranks itself, rank 0's
    line is relayed, it names both ranks (all-reduce of ones == 2, per-rank rows) and the ids equal the single-rank run's."""
    one = _bench(1, {"MFAR_BENCH_DUMP_IDS": "1"})
    two = _bench(2, {"MFAR_BENCH_BACKEND": "gloo", "MFAR_BENCH_SHARE_GPU": "1", "MFAR_BENCH_DUMP_IDS": "1"}, launcher=False)
    assert one["n_gpus"] == 1 and one["rccl"] is None
    assert two["n_gpus"] == 2 and two["rccl"]["world_size"] == 2 and two["rccl"]["allreduce_of_ones"] == 2.0
    assert [r["rows"] for r in two["rccl"]["ranks"]] == [[0, 35000], [35000, 70000]] and two["config"]["row_shards"] == 2
    assert len({r["pid"] for r in two["rccl"]["ranks"]}) == 2
    assert two["ids_checksum"] == one["ids_checksum"] and two["value"] > 0


def test_bench_extra_legs_with_two_ranks():
    """The default N > 1 run also measures the sustained leg and the other corner of the layout (N full replicas, no exchange)."""
    env = dict(os.environ, MFAR_BENCH_BACKEND="gloo", MFAR_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--docs", "70000", "--fields", "4", "--dim", "128",
                          "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--sustain-s", "0.2"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["sustained"]["seconds"] >= 0.2 and d["sustained"]["queries_per_s"] > 0
    rl = d["replica_layout"]
    assert rl["row_shards"] == 1 and rl["replica_groups"] == 2 and rl["batches_per_rank"] == [2, 2] and rl["queries_per_s"] > 0
    assert d["config"]["row_shards"] == 2 and d["roofline_exact_fp32"] is None


def test_bench_fails_when_a_rank_fails():
    """A rank that dies takes the whole run down with a non-zero exit code (no line is printed, nothing hangs)."""
    env = dict(os.environ, MFAR_BENCH_BACKEND="gloo", MFAR_BENCH_SHARE_GPU="1", MFAR_SCREEN_EPS_MULT="7")   # refused by every rank
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--docs", "20000", "--fields", "2", "--dim", "64",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("R", [1, 2])
def test_replica_group_layouts_match_single_rank(R):
    """4 ranks = G replica groups x R row shards (`--row-shards R`; all on cuda:0 over gloo): R = 1 is the reference's
    query-sharded search over full replicas (no exchange), R = 2 two groups of two row shards (exchange inside a group only).
    Batches are dealt round-robin to the groups; the ids of every step equal the single-rank run's."""
    one = _bench(1, {"MFAR_BENCH_DUMP_IDS": "1"})
    four = _bench(4, {"MFAR_BENCH_BACKEND": "gloo", "MFAR_BENCH_SHARE_GPU": "1", "MFAR_BENCH_DUMP_IDS": "1"}, launcher=False,
                  extra_args=["--row-shards", str(R)])
    assert four["config"]["row_shards"] == R and four["config"]["replica_groups"] == 4 // R
    assert [r["replica_group"] for r in four["rccl"]["ranks"]] == [i // R for i in range(4)]
    assert four["ids_checksum"] == one["ids_checksum"] and four["recall_at_20"] == one["recall_at_20"]


@pytest.mark.parametrize("R", [5, 1])
def test_five_ranks_on_one_gpu_odd_world_and_padded_launch(R):
    """The largest world this pool admits on one card (six processes may hold the GPU: five ranks + this test process) -- the
    eight-rank layouts run on the CPU in tests/test_dist_gloo.py.  Five ranks, an ODD world: row shards of unequal size (70 000 / 5
    is even, so 5 steps x 64 queries leave every group a short last launch with --row-shards 1, and the single group of --row-shards 5
    pads its last coalesced launch), every rank named in the proof block, ids of every step equal to the single-rank run's."""
    one = _bench(1, {"MFAR_BENCH_DUMP_IDS": "1"}, extra_args=["--steps", "5"])
    five = _bench(5, {"MFAR_BENCH_BACKEND": "gloo", "MFAR_BENCH_SHARE_GPU": "1", "MFAR_BENCH_DUMP_IDS": "1"}, launcher=False,
                  extra_args=["--steps", "5", "--row-shards", str(R)])
    assert five["n_gpus"] == 5 and five["rccl"]["world_size"] == 5 and five["rccl"]["allreduce_of_ones"] == 5.0
    assert len({r["pid"] for r in five["rccl"]["ranks"]}) == 5
    assert five["config"]["row_shards"] == R and five["config"]["replica_groups"] == 5 // R
    rows = [r["rows"] for r in five["rccl"]["ranks"]]
    assert rows == ([[70000 * i // 5, 70000 * (i + 1) // 5] for i in range(5)] if R == 5 else [[0, 70000]] * 5)
    assert five["ids_checksum"] == one["ids_checksum"] and five["recall_at_20"] == one["recall_at_20"]


def test_bench_fails_when_one_of_four_ranks_dies():
    """One rank of four exits while the others are inside the run (a rank-specific failure, not a shared refusal): the launcher
    terminates the rest and returns non-zero, no line is printed, nothing hangs."""
    env = dict(os.environ, MFAR_BENCH_BACKEND="gloo", MFAR_BENCH_SHARE_GPU="1", MFAR_BENCH_KILL_RANK="2")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--docs", "20000", "--fields", "2", "--dim", "64",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "rank 2 exited" in out.stderr


def test_exchange_path_over_rccl_with_one_rank():
    """RCCL itself: a one-rank nccl process group on cuda:0 carries the two all-gathers of every launch of the pipelined
    searcher's exchange path (fp32 index with the wide screened pass, bf16 index); results equal the plain search bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_one_rank.py"), "29577"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["f32"]["same"] and res["bf16"]["same"], res
    assert res["f32"]["sweep_same"] and res["bf16"]["sweep_same"], res      # PipelinedSearcher(masks=...) over the exchange
    assert res["f32"]["coalesce"] == 2 and res["f32"]["n"] == 7 and res["f32"]["redone"] == 0, res


def test_mask_sweep_over_two_row_shards():
    """PipelinedSearcher(masks=...) with two ranks (gloo, sharing cuda:0): per mask the sharded sweep equals the unsharded search."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "helpers", "two_rank_sweep.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["same"] and res["n"] == 5 and res["world"] == 2, res


def test_mask_fields_cli_over_two_row_shards(tmp_path, monkeypatch):
    """The product CLI with two ranks (gloo, sharing cuda:0): each rank encodes and holds half of the corpus, `test_sweep` runs
    the whole mask sweep through the lists-first exchange; the files equal a single process's, byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_cli as T
    from mfar.commands import mask_fields, train
    data = str(tmp_path / "data")
    T._write_dataset(data)
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    train.main(dataset_name="amazon", lexical_index="unused", out=out, temp_dir=tmp, data=data, model_name="random-init:64x2",
               field_names="title_dense,brand_dense,feature_dense", weights_lr=1e-2, encoder_lr=1e-4, train_batch_size=8,
               dev_batch_size=16, max_epochs=1, precision="32", additional_partition="test")
    monkeypatch.setenv("MFAR_ENCODE_TOKEN_BUDGET", "0")      # with dev_batch_size = 1: every text is encoded alone, in both runs
    one = str(tmp_path / "one")
    mask_fields.main(dataset_name="amazon", lexical_index="unused", out=one, temp_dir=tmp, data=data, model_name="random-init:64x2",
                     field_names="title_dense,brand_dense,feature_dense", checkpoint_dir=out, dev_batch_size=1, additional_partition="test")
    two = str(tmp_path / "two")
    env = dict(os.environ, MFAR_DIST_BACKEND="gloo", MFAR_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "tests", "helpers", "two_rank_mask_fields.py"), data, str(tmp_path / "tmp2"), out, two]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    for fn in ("results_dicts-all-0.jsonl", "final-all-0.qres", "final-additional-all-0.qres"):
        assert open(f"{one}/{fn}").read() == open(f"{two}/{fn}").read(), fn


def test_sparse_fields_over_two_row_shards(tmp_path, monkeypatch):
    """Sparse (BM25) fields with two ranks (gloo, sharing cuda:0): the dense rows are sharded, the BM25 indices replicated (the
    reference evaluates sparse fields on every rank, index.py:97-124); the hybrid step merges the shards' dense lists, forms the
    same candidate union everywhere and gathers the dense score columns from their owners -- the files of the whole mask sweep
    equal a single process's, byte for byte."""
    sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_gpu_cli as T
    from mfar.commands import create_bm25s_index, mask_fields, train
    data, lex = str(tmp_path / "data"), str(tmp_path / "lex")
    T._write_dataset(data, relations=True)
    create_bm25s_index.main(data_path=data, dataset_name="amazon", output_path=lex, fields_str="single_sparse")
    fields = "title_dense,brand_dense,title_sparse,feature_sparse"
    out, tmp = str(tmp_path / "out"), str(tmp_path / "tmp")
    train.main(dataset_name="amazon", lexical_index=lex, out=out, temp_dir=tmp, data=data, model_name="random-init:64x2", field_names=fields,
               weights_lr=1e-2, encoder_lr=1e-4, train_batch_size=8, dev_batch_size=16, max_epochs=1, precision="32",
               negative_sampling_params=(20, 5, 1), additional_partition="test")
    monkeypatch.setenv("MFAR_ENCODE_TOKEN_BUDGET", "0")      # with dev_batch_size = 1: every text is encoded alone, in both runs
    one = str(tmp_path / "one")
    mask_fields.main(dataset_name="amazon", lexical_index=lex, out=one, temp_dir=tmp, data=data, model_name="random-init:64x2",
                     field_names=fields, checkpoint_dir=out, dev_batch_size=1, additional_partition="test")
    two = str(tmp_path / "two")
    env = dict(os.environ, MFAR_DIST_BACKEND="gloo", MFAR_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "tests", "helpers", "two_rank_mask_fields.py"), data, str(tmp_path / "tmp2"), out, two,
           fields, lex]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    for fn in ("results_dicts-all-0.jsonl", "final-all-0.qres", "final-additional-all-0.qres"):
        assert open(f"{one}/{fn}").read() == open(f"{two}/{fn}").read(), fn
