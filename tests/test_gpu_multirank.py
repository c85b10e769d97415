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


def _bench(n, extra_env, dtype="f32"):
    env = dict(os.environ, **extra_env)
    args = ["--docs", "70000", "--fields", "4", "--dim", "128", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs",
            "--dtype", dtype]
    if n == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
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
