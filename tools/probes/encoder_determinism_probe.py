"""Probe (round 6): is the eager encoder forward deterministic run to run?  (tools/probes/encoder_graph_probe.py saw a graph replay of the
(64, 512) forward differ from the eager result by 3.2 in one element.)  Same input twice, eager, per shape / precision / SDPA backend."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multifield-adaptive-retrieval_amd"))
from mfar.modeling.util import prepare_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tok, enc, _ = prepare_model("random-init:768x12", normalize=False, with_decoder=False)
    enc = enc.to(dev).eval()
    print("attention implementation:", getattr(enc.auto_model.config, "_attn_implementation", None), flush=True)
    for n, L in ((64, 512), (128, 256), (256, 128), (512, 64)):
        torch.manual_seed(1)
        ids = torch.randint(5, 60, (n, L), device=dev)
        lens = torch.randint(L // 2, L + 1, (n,), device=dev)
        mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
        f = {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}
        for name, ac in (("fp32", None), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
            with torch.no_grad(), torch.autocast("cuda", dtype=ac, enabled=ac is not None):
                outs = [enc(f)["sentence_embedding"].float().clone() for _ in range(4)]
                tokens = [enc(f)["token_embeddings"].float().clone() for _ in range(2)]
            torch.cuda.synchronize()
            d = max(float((o - outs[0]).abs().max()) for o in outs[1:])
            td = (tokens[1] - tokens[0]).abs()
            valid = mask.bool()
            d_valid = float(td[valid].max())
            d_pad = float(td[~valid].max()) if (~valid).any() else 0.0
            nan = bool(torch.isnan(outs[0]).any())
            print(f"n={n} L={L} {name}: sentence max diff over 4 runs {d:.3g}; token embeddings run to run: valid positions {d_valid:.3g}, padded positions {d_pad:.3g}; "
                  f"nan {nan}; |emb| max {float(outs[0].abs().max()):.3g}", flush=True)


if __name__ == "__main__":
    main()
