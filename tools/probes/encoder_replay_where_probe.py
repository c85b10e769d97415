"""Probe (round 6): WHERE a replayed (64, 512) forward first departs from the eager one (tools/probes/encoder_replay_drift_probe.py: right on
replay 1, wrong in every row from replay 2 on, L = 512 only).  Forward hooks keep every sub-module's output (static inside the graph)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "multifield-adaptive-retrieval_amd"))
from mfar.modeling.util import prepare_model  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    tok, enc, _ = prepare_model("random-init:768x2", normalize=False, with_decoder=False)
    enc = enc.to(dev).eval()
    for n, L in ((64, 512), (16, 512), (64, 504), (128, 256)):
        torch.manual_seed(1)
        ids = torch.randint(5, 60, (n, L), device=dev)
        lens = torch.randint(L // 2, L + 1, (n,), device=dev)
        mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
        f = {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids)}
        keep = {}
        hooks = []
        for name, mod in enc.named_modules():
            if name and len(list(mod.children())) == 0 or name.endswith(("attention.self", "embeddings")):
                def hook(m, a, out, name=name):
                    o = out[0] if isinstance(out, tuple) else out
                    if torch.is_tensor(o):
                        keep[name] = o
                hooks.append(mod.register_forward_hook(hook))
        with torch.no_grad():
            ref_out = enc(f)["sentence_embedding"].float().clone()
            ref = {k: v.float().clone() for k, v in keep.items()}
            keep.clear()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                enc(f)
            torch.cuda.current_stream().wait_stream(s)
            keep.clear()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                out = enc(f)["sentence_embedding"].float()
            static = dict(keep)
            for r in range(3):
                g.replay()
                torch.cuda.synchronize()
                junk = [torch.randn(1 << 20, device=dev) for _ in range(8)]          # allocations between replays, like a real loop
                bad = [(k, float((static[k].float() - ref[k]).abs().max())) for k in ref if k in static and float((static[k].float() - ref[k]).abs().max()) > 1e-3]
                ids_ok = bool(torch.equal(f["input_ids"], ids)) and bool(torch.equal(f["attention_mask"], mask))
                print(f"n={n} L={L} replay {r + 1}: output diff {float((out - ref_out).abs().max()):.3g}; inputs intact {ids_ok}; first departing modules: {bad[:4]}", flush=True)
                del junk
        for h in hooks:
            h.remove()
        del g


if __name__ == "__main__":
    main()
