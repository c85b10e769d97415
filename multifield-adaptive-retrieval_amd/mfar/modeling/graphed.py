"""Corpus-encode forwards replayed from captured graphs (hipGraphs through `torch.cuda.CUDAGraph`).

Why (round 6; `profiles/r06_j_encode_halves.txt`, `r06_j_encode_chunks_ab.txt`, `r06_j_encoder_graph_probe.txt`; 50 000 x 22 prime-shaped
texts, fp16 autocast): the two halves of `on_eval_start`, each ALONE, take 3.4 s (producer thread: format, distinct texts, Rust tokenizer,
padding into pinned buffers) and 9.4 s (this thread: copy -> forward -> scatter; GPU-bound, 2.2 M tokens/s), side by side 12.3 s.  A forward
is 12-15 ms of GPU time and ~10 ms of Python on the launching thread (300 kernels behind HF's module tree and autocast's dispatch): alone that
is hidden -- the queue stays full, a replay of the same forward is 1 % faster -- but the launching thread shares the interpreter lock with
the producer, and whenever the producer holds it the launches stop.  A replay is one call: the launching thread hardly needs the lock.
Measured (`r06_k2_encode_graphs_ab.txt`): every RE-encode of a process 10.0-10.4 s (eager 11.4-12.0 s); a process's FIRST encode gains
nothing (12.5 s: 72 captures, each checked against the eager forward) -- hence OPT-IN, `MFAR_ENCODE_GRAPHS=1`, for runs that encode the
corpus many times (training with an evaluation per epoch).

What is captured: `encoder(features)["sentence_embedding"].float()` for ONE static shape `[n, L]` per graph, in eval mode, under the run's
autocast setting with autocast's weight-cast cache OFF -- the casts are part of the graph, so a replay reads the CURRENT parameter values
(in-place optimizer steps between two encodes are seen; parameters that MOVED are detected by `signature()` and the graphs dropped).
The producer cuts batches to a small family of shapes (`shape_ladder`, lengths rounded up to 8 tokens); a shape is captured the second
time it shows up (a shape seen once is not worth the capture and its check), at most `max_graphs` of them; anything else runs eagerly.  All graphs share
one memory pool: they are replayed one at a time on one stream and every output is consumed (scattered into the field's rows) on that
stream before the next replay.

Not the product path: `csrc/` never sees this file; it feeds rows to `mfar_index_write_rows` faster.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch


def shape_ladder(bs: int, cap: int = 4096) -> List[int]:
    """Batch sizes (texts per forward) the producer may cut, descending: `cap`, then 3/4 steps down to `bs` (the floor: the largest
    forward the caller sized memory for is `bs` texts of the encoder's full length)."""
    out, v = [], max(cap, bs)
    while v > bs:
        out.append(v)
        v = max(bs, (v * 3) // 4)
    out.append(bs)
    return out


def round_len(L: int, max_len: int, step: int = 8) -> int:
    return max(1, min(max_len, (max(1, L) + step - 1) // step * step))


class GraphedForward:
    def __init__(self, encoder, device: torch.device, max_graphs: int = 160):
        self.encoder = encoder
        self.device = device
        self.max_graphs = int(os.environ.get("MFAR_ENCODE_MAX_GRAPHS", max_graphs))
        self.pool = None
        self.side = None
        self.warmed = set()
        self.graphs: Dict[Tuple, Tuple] = {}       # (n, L, autocast dtype) -> (graph, static inputs, static output)
        self.seen: Dict[Tuple, int] = {}
        self.rejected = set()                      # shapes whose graph did not reproduce the eager forward
        self.sig = None
        self.failed = False                        # a capture raised: the encoder does not capture (data-dependent control flow); eager from then on
        self.n_replays = self.n_eager = 0
        names = getattr(getattr(encoder, "tokenizer", None), "model_input_names", None) or ()
        self.with_types = "token_type_ids" in names

    @staticmethod
    def enabled(device: torch.device) -> bool:
        return device.type == "cuda" and os.environ.get("MFAR_ENCODE_GRAPHS", "0") == "1"

    def signature(self):
        """Where the parameters live.  A graph holds addresses: after `.to()`, `load_state_dict(assign=True)` or a dtype change the old ones
        are stale (in-place updates are fine)."""
        return tuple((p.data_ptr(), p.dtype) for p in self.encoder.parameters())

    def begin(self) -> None:
        """Start of an encode: drop graphs whose parameters moved."""
        sig = self.signature()
        if sig != self.sig:
            self.reset()
            self.sig = sig

    def reset(self) -> None:
        self.graphs.clear()
        self.seen.clear()
        self.rejected.clear()
        self.pool = None

    def _eager(self, f, ac):
        with torch.autocast(device_type="cuda", dtype=ac, enabled=ac is not None):
            return self.encoder(f)["sentence_embedding"].float()

    def _capture(self, key, f, ac):
        static_in = {k: torch.empty_like(v) for k, v in f.items()}
        for k, v in f.items():
            static_in[k].copy_(v)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        cur = torch.cuda.current_stream(self.device)
        if self.side is None:
            self.side = torch.cuda.Stream(self.device)          # ONE capture stream: its GEMM workspaces are set up once
        side = self.side
        side.wait_stream(cur)
        if ac not in self.warmed:
            # one warm-up forward on the capture stream per precision (lazy initialisation, per-stream workspaces).  Later shapes need
            # none: a shape is captured the second time it shows up, so its kernels were already chosen by the eager forward before
            with torch.cuda.stream(side):
                with torch.autocast(device_type="cuda", dtype=ac, enabled=ac is not None, cache_enabled=False):
                    self.encoder(static_in)["sentence_embedding"].float()
            self.warmed.add(ac)
        cur.wait_stream(side)
        g = torch.cuda.CUDAGraph()
        # thread_local: the producer thread keeps allocating pinned buffers while this thread captures
        with torch.cuda.graph(g, pool=self.pool, stream=side, capture_error_mode="thread_local"):
            with torch.autocast(device_type="cuda", dtype=ac, enabled=ac is not None, cache_enabled=False):
                out = self.encoder(static_in)["sentence_embedding"].float()
        self.graphs[key] = (g, static_in, out)

    @torch.no_grad()
    def __call__(self, feats: Dict[str, torch.Tensor], ac: Optional[torch.dtype]) -> torch.Tensor:
        """feats: host (pinned) or device `input_ids`, `attention_mask` of one batch -> [n, E] fp32 on the device.  The result of a replay
        is the graph's static output: consume it on the current stream before the next call."""
        n, L = feats["input_ids"].shape
        key = (int(n), int(L), ac)
        hit = self.graphs.get(key)
        if hit is not None:
            g, static_in, out = hit
            for k in ("input_ids", "attention_mask"):
                static_in[k].copy_(feats[k], non_blocking=True)
            g.replay()
            self.n_replays += 1
            return out
        f = {k: feats[k].to(self.device, non_blocking=True) for k in ("input_ids", "attention_mask")}
        if self.with_types:
            f["token_type_ids"] = torch.zeros_like(f["input_ids"])
        count = self.seen.get(key, 0) + 1
        self.seen[key] = count
        if count >= 2 and not self.failed and key not in self.rejected and len(self.graphs) < self.max_graphs and not self.encoder.training:
            try:
                self._capture(key, f, ac)
                g, _, out = self.graphs[key]
                # every graph is CHECKED before it is trusted: two replays, the second compared with the eager forward of the same input
                # (what was found wrong on this stack -- SentenceEncoder.forward on `(tok * m).sum(1)` at L >= 512 -- was right on the first
                # replay and wrong from the second on).  Costs two forwards per shape, once.
                want = self._eager(f, ac)
                g.replay()
                g.replay()
                self.n_replays += 1
                tol = 0.0 if ac is None else 4e-3 * float(want.abs().max())     # (fp16 GEMMs of some shapes are not run-to-run identical: 1e-3)
                if bool(((out - want).abs() <= tol).all()):
                    return out
                import warnings
                self.rejected.add(key)
                self.graphs.pop(key, None)
                warnings.warn(f"corpus-encode graph of shape {key[:2]} does not replay the eager forward's result; that shape runs eagerly")
                return want
            except Exception as e:      # noqa: BLE001 -- an encoder that does not capture: eager from now on, loudly
                import warnings
                self.failed = True
                self.graphs.pop(key, None)
                torch.cuda.synchronize(self.device)
                warnings.warn(f"corpus-encode graph capture failed ({type(e).__name__}: {str(e)[:200]}); forwards run eagerly")
        self.n_eager += 1
        return self._eager(f, ac)
