"""Probe (round 6): how much of a corpus-encode forward is host time, and does a captured (hipGraph) forward remove it?
Eager HF forward under fp16 autocast against a torch.cuda.CUDAGraph replay of the same forward, BERT-base-shaped random-init encoder,
three batch shapes of the token budget 64 x 512.  Prints wall time per forward with the queue kept full, and GPU time (events)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multifield-adaptive-retrieval_amd"))
from mfar.modeling.util import prepare_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tok, enc, _ = prepare_model("random-init:768x12", normalize=False, with_decoder=False)
    enc = enc.to(dev).eval()
    for n, L in ((64, 512), (512, 64), (4096, 8)):
        ids = torch.randint(5, 60, (n, L), device=dev)
        mask = torch.ones(n, L, dtype=torch.long, device=dev)
        mask[:, L - L // 4:] = 0
        f = {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for _ in range(3):
                ref = enc(f)["sentence_embedding"].float()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = enc(f)["sentence_embedding"].float()
            e1.record()
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 20 * 1e3
            eager_gpu = e0.elapsed_time(e1) / 20
            # host time alone: launch without waiting
            t0 = time.perf_counter()
            for _ in range(5):
                out = enc(f)["sentence_embedding"].float()
            host = (time.perf_counter() - t0) / 5 * 1e3
            torch.cuda.synchronize()
            try:
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for _ in range(2):
                        enc(f)["sentence_embedding"].float()
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(g):
                    gout = enc(f)["sentence_embedding"].float()
                g.replay()
                torch.cuda.synchronize()
                same = bool(torch.equal(gout, ref))
                t0 = time.perf_counter()
                e0.record()
                for _ in range(20):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                graph = (time.perf_counter() - t0) / 20 * 1e3
                graph_gpu = e0.elapsed_time(e1) / 20
                print(f"n={n} L={L}: eager {eager:.2f} ms/forward (events {eager_gpu:.2f}, host alone {host:.2f}); graph replay {graph:.2f} ms "
                      f"(events {graph_gpu:.2f}); bits equal: {same}; max diff {float((gout - ref).abs().max()):.3g}", flush=True)
            except Exception as e:      # noqa: BLE001
                print(f"n={n} L={L}: eager {eager:.2f} ms/forward (host alone {host:.2f}); capture FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    main()
