"""Probe (round 6): the corpus encode's two halves, each ALONE, on the 50 000 x 22 prime-shaped corpus of tools/encode_bench.py.
  producer alone  -- format + distinct texts + Rust tokenisation + padding into pinned buffers, batches thrown away;
  consumer alone  -- the same batches (kept in memory) through copy -> forward -> scatter, eager and graph replays.
Whichever is slower bounds `on_eval_start` (they run side by side: contrastive.py `_encode_fields_prefetched`)."""
import json
import os
import sys
import threading
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "multifield-adaptive-retrieval_amd"))


def halves(module, st):
    import torch
    from mfar.data.format import format_documents
    from mfar.data.typedef import FieldType
    dense = [(k, f) for k, f in module.field_info.items() if f.field_type == FieldType.DENSE]
    bs = module.dev_batch_size
    max_len = int(module.encoder.get_max_seq_length())
    budget = bs * max_len
    out = {}
    for graphs in (False, True):
        t_prep = t_tok = 0.0
        kept = []
        stop = threading.Event()
        t0 = time.perf_counter()
        for key, field in dense:
            a = time.perf_counter()
            docs = format_documents(module.corpus, field.name, field.dataset)
            texts = [t for _, t in docs]
            uniq = list(dict.fromkeys(texts))
            slot = {t: i for i, t in enumerate(uniq)}
            order = sorted(range(len(uniq)), key=lambda i: len(uniq[i]))
            t_prep += time.perf_counter() - a
            a = time.perf_counter()
            for feats, rows in module._token_batches(uniq, order, bs, budget, max_len, stop, graphs):
                kept.append((feats, rows, len(uniq)))
            t_tok += time.perf_counter() - a
        producer = time.perf_counter() - t0
        tokens = sum(int(f["attention_mask"].sum()) for f, _, _ in kept)
        padded = sum(f["input_ids"].numel() for f, _, _ in kept)
        name = "graph_shapes" if graphs else "eager_shapes"
        out[name] = {"producer_alone_s": producer, "of_which_format_and_distinct_s": t_prep, "of_which_tokenise_and_pad_s": t_tok,
                     "batches": len(kept), "real_tokens": tokens, "padded_tokens": padded, "fill": tokens / padded}
        # consumer alone
        emb = torch.empty(max(n for _, _, n in kept), module.slab.dim, device=module.device)
        module.encoder.eval()
        module._graphed = None
        if graphs:
            module._use_graphs()
        for rep in range(3 if graphs else 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                for feats, rows, _ in kept:
                    module._forward_rows(feats, rows, emb, torch.float16)
            torch.cuda.synchronize()
            out[name][f"consumer_alone_s_pass{rep}"] = time.perf_counter() - t0
        module._graphed = None
    return out


if __name__ == "__main__":
    import encode_bench
    res = encode_bench.run(50000, 64, sweep=False, quiet=True, modes=("fp16",), probe=False, hooks=(halves,))
    print(json.dumps({"encode_s": res["autocast_fp16"]["seconds"], "tokens_per_encode": res["tokens_per_encode"], **res["autocast_fp16"]["halves"]}, indent=1))
