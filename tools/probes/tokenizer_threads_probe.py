"""Probe (round 6): what the corpus encode's PRODUCER can deliver on the GPU box -- tokens/s of `backend_tokenizer.encode_batch` (Rust, rayon
pool) over prime-shaped texts for several RAYON_NUM_THREADS (one child process each: the pool is sized once per process)."""
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r"""
import sys, time, os
sys.path.insert(0, os.path.join(%r, "..", "..", "multifield-adaptive-retrieval_amd"))
import numpy as np
from mfar.modeling.util import prepare_model
tok, _, _ = prepare_model("random-init:32x1")
be = tok.backend_tokenizer
rng = np.random.default_rng(0)
words = ["alpha", "beta", "gamma", "retrieval", "protein", "kinase", "interacts", "with", "disease", "x1", "transporter", "phenotype"]
short = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(5, 20, size=32768)]
long_ = [" ".join(rng.choice(words, size=int(n))) for n in rng.integers(60, 120, size=4096)]
be.enable_truncation(max_length=512)
be.no_padding()
for name, texts in (("short", short), ("long", long_)):
    best = 0.0
    for _ in range(3):
        n = 0
        t0 = time.perf_counter()
        for c in range(0, len(texts), 8192):
            n += sum(len(e.ids) for e in be.encode_batch(texts[c:c + 8192], add_special_tokens=True))
        best = max(best, n / (time.perf_counter() - t0))
    print(f"RAYON_NUM_THREADS={os.environ.get('RAYON_NUM_THREADS', 'default')} cpus={len(os.sched_getaffinity(0))} {name}: {best / 1e6:.2f} M tokens/s", flush=True)
""" % HERE

for n in ("", "4", "8", "16", "32", "64"):
    env = dict(os.environ)
    env.pop("RAYON_NUM_THREADS", None)
    if n:
        env["RAYON_NUM_THREADS"] = n
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
