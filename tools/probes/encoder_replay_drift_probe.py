"""Probe (round 6): do REPLAYS of a captured encoder forward stay equal to the eager forward?  (64, 512) under fp16 autocast differed by the
size of the largest element after some replays (tools/probes/encoder_graph_probe.py).  Per shape / precision / SDPA backend: eager
reference, capture, 30 replays, difference after each; where the differing elements sit."""
import contextlib
import os
import sys

import torch
from torch.nn.attention import SDPBackend, sdpa_kernel

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multifield-adaptive-retrieval_amd"))
from mfar.modeling.util import prepare_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tok, enc, _ = prepare_model("random-init:768x12", normalize=False, with_decoder=False)
    enc = enc.to(dev).eval()
    pool = torch.cuda.graph_pool_handle()
    for backend in ("default", "math"):
        ctx = (lambda: sdpa_kernel(SDPBackend.MATH)) if backend == "math" else contextlib.nullcontext
        for n, L in ((64, 512), (128, 256), (512, 64)):
            for name, ac in (("fp16", torch.float16), ("bf16", torch.bfloat16), ("fp32", None)):
                torch.manual_seed(1)
                ids = torch.randint(5, 60, (n, L), device=dev)
                lens = torch.randint(L // 2, L + 1, (n,), device=dev)
                mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
                f = {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}
                with torch.no_grad(), ctx():
                    with torch.autocast("cuda", dtype=ac, enabled=ac is not None):
                        ref = enc(f)["sentence_embedding"].float().clone()
                        ref2 = enc(f)["sentence_embedding"].float().clone()
                    noise = float((ref - ref2).abs().max())
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        with torch.autocast("cuda", dtype=ac, enabled=ac is not None, cache_enabled=False):
                            enc(f)["sentence_embedding"].float()
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, stream=s):
                        with torch.autocast("cuda", dtype=ac, enabled=ac is not None, cache_enabled=False):
                            out = enc(f)["sentence_embedding"].float()
                    diffs = []
                    worst = None
                    for r in range(30):
                        g.replay()
                        torch.cuda.synchronize()
                        d = (out - ref).abs()
                        diffs.append(float(d.max()))
                        if diffs[-1] > 0.1 and worst is None:
                            rows = (d.max(1).values > 0.1).nonzero().flatten().tolist()
                            worst = (r, len(rows), rows[:8], bool(torch.isnan(out).any()))
                    print(f"sdpa={backend} n={n} L={L} {name}: eager run-to-run {noise:.3g}; replay - eager after replay 1 / 2 / 10 / 30: "
                          f"{diffs[0]:.3g} / {diffs[1]:.3g} / {diffs[9]:.3g} / {diffs[29]:.3g}; first large difference: {worst}", flush=True)
                    del g, out


if __name__ == "__main__":
    main()
