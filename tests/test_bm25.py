"""BM25 sparse index + negative sampler.  The reference's arithmetic lives in the un-vendored `bm25s==0.1.10`
(poetry.lock:768-769), which is not installed: PARITY UNPINNED.  What is pinned here: the published lucene-variant formula
(hand-computed), the retrieval / scoring surface of reference mfar/data/index.py:39-157 and the sampling rule of
mfar/data/negative_sampler.py:40-60."""
import math
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multifield-adaptive-retrieval_amd"))

from mfar.data import bm25
from mfar.data.index import BM25sSparseIndex
from mfar.data.negative_sampler import IndexNegativeSampler
from mfar.data.typedef import Corpus, Query

DOCS = {
    "d0": "The quick brown fox jumps over the lazy dog",
    "d1": "A quick quick fox",
    "d2": "Lazy dogs sleep all day and the fox is not in it",
    "d3": "",
    "d4": "Graph neural networks for protein protein interaction",
    "d5": "fox",
}


def test_tokenizer_rules():
    assert bm25.tokenize("The quick brown fox, it's a FOX-trot: x y2 ab")[0] == ["quick", "brown", "fox", "fox", "trot", "y2", "ab"]
    assert bm25.tokenize(["a an the", "Über-cool naïve"]) == [[], ["über", "cool", "naïve"]]      # unicode word characters
    assert bm25.tokenize("running dogs", stemmer=lambda w: w.rstrip("s"))[0] == ["running", "dog"]


def test_lucene_formula_by_hand():
    toks = bm25.tokenize(list(DOCS.values()))
    ix = bm25.BM25().index(toks)
    N = len(DOCS)
    lens = [len(t) for t in toks]
    avg = sum(lens) / N
    def expect(term, d):
        tf = toks[d].count(term)
        df = sum(1 for t in toks if term in t)
        idf = math.log(1 + (N - df + 0.5) / (df + 0.5))
        return idf * tf / (tf + 1.2 * (1 - 0.75 + 0.75 * lens[d] / avg)) if tf else 0.0
    s = ix.get_scores(["fox", "quick"])
    for d in range(N):
        assert s[d] == pytest.approx(expect("fox", d) + expect("quick", d), rel=1e-6, abs=1e-7), d
    assert s.dtype == np.float32 and s[3] == 0 and s[4] == 0
    # a repeated query token counts twice; unknown tokens are ignored
    assert np.allclose(ix.get_scores(["fox", "fox", "zzz"]), 2 * ix.get_scores(["fox"]))
    # the shortest document with the term wins among equal tf
    assert ix.get_scores(["fox"]).argmax() == 5


def test_retrieve_is_canonical_topk_and_k_bound():
    rng = random.Random(3)
    vocab = [f"w{i}" for i in range(40)]
    docs = [" ".join(rng.choice(vocab) for _ in range(rng.randrange(0, 30))) for _ in range(300)]
    docs[17] = docs[203] = docs[5]                         # identical documents -> exact score ties
    ix = bm25.BM25().index(bm25.tokenize(docs))
    qs = bm25.tokenize(["w1 w2 w3", "w7", "nothing here", "w5 w5 w9"])
    ids, sc = ix.retrieve(qs, k=25)
    for i, q in enumerate(qs):
        s = ix.get_scores(q)
        order = np.lexsort((np.arange(300), -s.astype(np.float64)))[:25]
        assert np.array_equal(ids[i], order) and np.array_equal(sc[i], s[order])
    with pytest.raises(ValueError):
        ix.retrieve(qs, k=301)


def test_sparse_index_surface_and_roundtrip(tmp_path):
    corpus = Corpus.from_docs_dict(DOCS, "amazon")
    si = BM25sSparseIndex.create(corpus, dataset_name="amazon")
    assert si.index_limit == 5000 and BM25sSparseIndex.create(corpus, dataset_name="prime").index_limit == 12000
    top = si.retrieve("quick fox", top_k=3)
    assert [k for k, _ in top] == ["d1", "d0", "d5"] and top[0][1] > top[1][1] > top[2][1] > 0      # d5 holds "fox" only
    batch = si.retrieve_batch(["quick fox", "protein"], top_k=2)
    assert batch[0] == top[:2] and batch[1][0][0] == "d4" and batch[1][1][1] == 0.0
    s = si.score("quick fox", ["d5", "d1", "d3"])
    assert s[1] == top[0][1] and s[0] == top[2][1] and s[2] == 0
    with pytest.raises(KeyError):
        si.score("fox", ["nope"])
    sb = si.score_batch(["quick fox", "lazy"], ["d1", "nope", "d2"])
    assert tuple(sb.shape) == (2, 3) and float(sb[0, 1]) == 0 and float(sb[0, 0]) == pytest.approx(float(top[0][1])) and float(sb[1, 2]) > 0
    cached = si.score_batch_with_cache([7, 8], ["d0", "d2"], {7: {0: 1.5}, 9: {2: 3.0}})
    assert cached.tolist() == [[1.5, 0], [0, 0]]
    si.set_safe_docs({1, 5})
    assert set(si.get_scores_sparse("quick fox")) == {1, 5}
    si.save(str(tmp_path))
    assert sorted(os.listdir(tmp_path / "index")) == ["data.csc.index.npy", "indices.csc.index.npy", "indptr.csc.index.npy",
                                                      "params.index.json", "vocab.index.json"]
    again = BM25sSparseIndex.load(str(tmp_path))
    assert again.keys == si.keys and again.retrieve("quick fox", 3) == top


def test_negative_sampler_rule():
    corpus = Corpus.from_docs_dict(DOCS)
    si = BM25sSparseIndex.create(corpus)
    ns = IndexNegativeSampler(si, DOCS, n_retrieve=4, n_bottom=2, n_sample=1, rng=random.Random(0))
    q = Query("q1", "quick fox")
    ranked = [k for k, _ in si.retrieve("quick fox", 4)]                      # d1 d0 d5 d2
    for _ in range(10):                                                       # the two LOWEST-scored non-positives
        got = ns.sample(q, {"q1": {"d1"}})
        assert len(got) == 1 and got[0]._id in ranked[-2:] and got[0].text == DOCS[got[0]._id]
    # every retrieved document is a positive -> retrieve len(positives) + n_bottom and try again
    ns2 = IndexNegativeSampler(si, DOCS, n_retrieve=2, n_bottom=2, n_sample=2, rng=random.Random(1))
    got = ns2.sample(q, {"q1": set(ranked[:2])})
    assert {d._id for d in got} == set(ranked[2:4])
    assert [len(x) for x in ns2.sample_batch([q, q], {"q1": set()})] == [2, 2]


def test_training_negatives_are_mined_per_rank_and_do_not_depend_on_the_world_size():
    """commands/train.py `_Instances.batches`: a rank mines negatives only for ITS slice of a global batch, the draw
    of an instance depends on (seed, pass, position) and not on the number of ranks, the BM25 candidate list of a query is computed
    once, and every instance is covered (padding to a multiple of the world size like DistributedSampler, drop_last=False)."""
    import random
    from mfar.commands.train import _Instances
    from mfar.data.index import Index
    from mfar.data.negative_sampler import IndexNegativeSampler

    class CountingIndex(Index):
        def __init__(self):
            self.calls = 0

        def retrieve(self, query, top_k):
            self.calls += 1
            r = random.Random(query)
            docs = [f"d{i}" for i in range(30)]
            r.shuffle(docs)
            return [(d, float(top_k - j)) for j, d in enumerate(docs[:top_k])]

    def make():
        inst = _Instances.__new__(_Instances)
        inst.queries = {f"q{i}": f"text of query {i}" for i in range(5)}
        inst.qrels = [type("R", (), {"doc_id": f"d{2 * i + 1}", "query_id": f"q{i % 5}"})() for i in range(11)]
        inst.corpus = [(f"d{i}", {}) for i in range(30)]
        inst.key_to_row = {k: i for i, (k, _) in enumerate(inst.corpus)}
        inst.pos_for_each_qid = {}
        for r in inst.qrels:
            inst.pos_for_each_qid.setdefault(r.query_id, set()).add(r.doc_id)
        inst.seed, inst.rng = 5, random.Random(5)
        ix = CountingIndex()
        inst.sampler = IndexNegativeSampler(ix, {}, n_retrieve=12, n_bottom=4, n_sample=1)
        return inst, ix

    def collect(world, passes=2):
        per_pass = []
        insts = [make() for _ in range(world)]
        for _ in range(passes):
            seen = {}
            steps = []
            for rank, (inst, _) in enumerate(insts):
                n = 0
                for rows, negs in inst.batches(4, rank, world, shuffle=True):
                    assert len(rows) == len(negs) <= 4
                    for r, ng in zip(rows, negs):
                        assert inst.corpus[ng][0] not in inst.pos_for_each_qid[r.query_id]
                        seen.setdefault((r.query_id, r.doc_id), []).append(ng)
                    n += 1
                steps.append(n)
            assert len(set(steps)) == 1                              # every rank runs the same number of steps
            per_pass.append(seen)
        return per_pass, [ix.calls for _, ix in insts]

    one, calls1 = collect(1)
    three, calls3 = collect(3)
    assert len(one[0]) == 11 and set(three[0]) == set(one[0])        # every instance covered, with 1 rank and with 3
    for p in range(2):                                               # same negative for the same instance whatever the world size
        for key, negs in one[p].items():
            assert three[p][key][0] == negs[0], (p, key)
    assert one[0] != one[1]                                          # a new pass draws anew
    assert calls1 == [5] and all(c <= 5 for c in calls3) and sum(calls3) < 11 * 2      # one retrieval per distinct query per rank, not per instance
