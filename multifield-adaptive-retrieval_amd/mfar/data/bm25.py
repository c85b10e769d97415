"""BM25 inverted index on the host CPU: the arithmetic behind the reference's sparse fields and hard-negative mining.

The reference builds these with the third-party package `bm25s==0.1.10` (poetry.lock:768-769; call sites
mfar/data/index.py:39-157: `bm25s.tokenize(..., stopwords="en", stemmer=None)`, `bm25s.BM25(method="lucene", k1=1.2,
b=0.75)`, `.index()`, `.get_scores()`, `.retrieve(..., backend_selection="numpy")`, `.save()` / `.load()`).  That package
is not vendored in the reference and not installed here, so this module RESTATES its published algorithm (PARITY UNPINNED:
no reference fixture holds BM25 numbers; the properties are pinned by tests/test_bm25.py instead):

  tokenize   lower-case, tokens = regex `(?u)\\b\\w\\w+\\b` (two or more word characters), minus the 33 English stop words of
             Lucene's default set (bm25s `STOPWORDS_EN`); no stemming (the reference passes stemmer=None: index.py:139-141)
  index      "eager" scoring: every (token, doc) posting stores its final contribution
                 idf(t) * tf / (tf + k1 * (1 - b + b * len(doc) / avg_len)),   idf(t) = ln(1 + (N - df + 0.5) / (df + 0.5))
             (the `lucene` variant: no (k1 + 1) factor), float32, in CSC order (one contiguous run of postings per token)
  score      sum of the posting values of the query's tokens, a token that occurs twice counts twice; unknown tokens are ignored
  retrieve   top-k documents by that score.  bm25s' tie order is whatever argpartition leaves; here ties follow the repo-wide
             canonical order (score desc, doc number asc)

This is host-side integer / sparse work (a few postings per query token); it feeds candidate lists and score columns into
the same C-ABI mixer (`mfar_mix_topk`) as the dense fields -- the HBM-resident hot path is the dense scan.
"""
import json
import os
import re
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

STOPWORDS_EN = frozenset((
    "a", "an", "and", "are", "as", "at", "be", "but", "by", "for", "if", "in", "into", "is", "it", "no", "not", "of", "on",
    "or", "such", "that", "the", "their", "then", "there", "these", "they", "this", "to", "was", "will", "with"))
_TOKEN = re.compile(r"(?u)\b\w\w+\b")


def tokenize_text(text: str, stopwords=STOPWORDS_EN, stemmer=None) -> List[str]:
    toks = [t for t in _TOKEN.findall(text.lower()) if t not in stopwords]
    if stemmer is not None:            # anything with stemWords(list) (PyStemmer) or a plain callable per word
        toks = list(stemmer.stemWords(toks)) if hasattr(stemmer, "stemWords") else [stemmer(t) for t in toks]
    return toks


class BM25:
    """Lucene-variant BM25 with precomputed posting scores (what `bm25s.BM25(method="lucene")` holds after `.index()`)."""

    def __init__(self, k1: float = 1.2, b: float = 0.75, method: str = "lucene"):
        if method != "lucene":
            raise ValueError("only the lucene variant is restated (the one the reference uses, index.py:140)")
        self.k1, self.b, self.method = float(k1), float(b), method
        self.vocab: Dict[str, int] = {}
        self.data = np.zeros(0, np.float32)        # posting scores, grouped by token
        self.indices = np.zeros(0, np.int32)       # posting doc numbers
        self.indptr = np.zeros(1, np.int64)        # [n_tokens + 1]
        self.num_docs = 0

    # ---- build
    def index(self, corpus_tokens: Sequence[Sequence[str]]) -> "BM25":
        n = len(corpus_tokens)
        vocab: Dict[str, int] = {}
        tok_ids, doc_ids = [], []
        lens = np.zeros(n, np.float64)
        for d, toks in enumerate(corpus_tokens):
            lens[d] = len(toks)
            for t in toks:
                i = vocab.get(t)
                if i is None:
                    i = vocab[t] = len(vocab)
                tok_ids.append(i)
                doc_ids.append(d)
        V = len(vocab)
        tok = np.asarray(tok_ids, np.int64)
        doc = np.asarray(doc_ids, np.int64)
        # term frequency per (token, doc): sort the occurrences, count runs
        order = np.lexsort((doc, tok))
        tok, doc = tok[order], doc[order]
        if tok.size:
            new = np.ones(tok.size, bool)
            new[1:] = (tok[1:] != tok[:-1]) | (doc[1:] != doc[:-1])
            starts = np.nonzero(new)[0]
            p_tok, p_doc = tok[starts], doc[starts]
            tf = np.diff(np.append(starts, tok.size)).astype(np.float64)
        else:
            p_tok = p_doc = np.zeros(0, np.int64)
            tf = np.zeros(0, np.float64)
        df = np.bincount(p_tok, minlength=V).astype(np.float64)
        idf = np.log(1.0 + (n - df + 0.5) / (df + 0.5))
        avg = float(lens.mean()) if n else 0.0
        norm = self.k1 * (1.0 - self.b + self.b * (lens[p_doc] / avg if avg > 0 else 0.0))
        self.data = (idf[p_tok] * (tf / (tf + norm))).astype(np.float32)
        self.indices = p_doc.astype(np.int32)
        self.indptr = np.concatenate([[0], np.cumsum(np.bincount(p_tok, minlength=V))]).astype(np.int64)
        self.vocab, self.num_docs = vocab, n
        return self

    # ---- score
    def get_tokens_ids(self, tokens: Iterable[str]) -> List[int]:
        return [self.vocab[t] for t in tokens if t in self.vocab]

    def get_scores(self, query_tokens: Sequence[str]) -> np.ndarray:
        """[num_docs] float32; postings are added token by token in query order (float32 accumulation like bm25s)."""
        scores = np.zeros(self.num_docs, np.float32)
        for t in self.get_tokens_ids(query_tokens):
            a, b = self.indptr[t], self.indptr[t + 1]
            scores[self.indices[a:b]] += self.data[a:b]        # a doc occurs at most once in a token's run
        return scores

    def retrieve(self, queries_tokens: Sequence[Sequence[str]], k: int) -> Tuple[np.ndarray, np.ndarray]:
        """-> (doc numbers [Q, k] int64, scores [Q, k] float32), canonical order (score desc, doc asc)."""
        if k > self.num_docs:
            raise ValueError(f"k of {k} is larger than the number of available scores, which is {self.num_docs}")   # as bm25s does
        ids = np.zeros((len(queries_tokens), k), np.int64)
        sc = np.zeros((len(queries_tokens), k), np.float32)
        for i, toks in enumerate(queries_tokens):
            s = self.get_scores(toks)
            if k < self.num_docs:
                # candidates: everything at least as good as the k-th best score (ties included), then the canonical order
                kth = np.partition(s, self.num_docs - k)[self.num_docs - k]
                cand = np.nonzero(s >= kth)[0]
            else:
                cand = np.arange(self.num_docs)
            o = cand[np.lexsort((cand, -s[cand].astype(np.float64)))][:k]
            ids[i], sc[i] = o, s[o]
        return ids, sc

    # ---- persistence: the file set bm25s writes (`<dir>/data.csc.index.npy`, `indices...`, `indptr...`, `vocab.index.json`,
    # `params.index.json`), so that an index directory is interchangeable as far as the layout goes
    def save(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "data.csc.index.npy"), self.data)
        np.save(os.path.join(path, "indices.csc.index.npy"), self.indices)
        np.save(os.path.join(path, "indptr.csc.index.npy"), self.indptr)
        with open(os.path.join(path, "vocab.index.json"), "w") as f:
            json.dump(self.vocab, f)
        with open(os.path.join(path, "params.index.json"), "w") as f:
            json.dump(dict(k1=self.k1, b=self.b, delta=0.5, method=self.method, idf_method=self.method, dtype="float32",
                           int_dtype="int32", num_docs=self.num_docs, version="0.1.10"), f)

    @classmethod
    def load(cls, path: str, mmap: bool = False) -> "BM25":
        with open(os.path.join(path, "params.index.json")) as f:
            p = json.load(f)
        self = cls(k1=p["k1"], b=p["b"], method=p.get("method", "lucene"))
        mode = "r" if mmap else None
        self.data = np.load(os.path.join(path, "data.csc.index.npy"), mmap_mode=mode)
        self.indices = np.load(os.path.join(path, "indices.csc.index.npy"), mmap_mode=mode)
        self.indptr = np.load(os.path.join(path, "indptr.csc.index.npy"), mmap_mode=mode)
        with open(os.path.join(path, "vocab.index.json")) as f:
            self.vocab = {k: int(v) for k, v in json.load(f).items()}
        self.num_docs = int(p["num_docs"])
        return self


def tokenize(texts, stopwords="en", stemmer=None) -> List[List[str]]:
    """`bm25s.tokenize(texts, stopwords="en", stemmer=..., return_ids=False)`: a string -> [tokens], a list -> [[tokens]]."""
    sw = STOPWORDS_EN if stopwords in ("en", "english", True) else (frozenset(stopwords) if stopwords else frozenset())
    if isinstance(texts, str):
        return [tokenize_text(texts, sw, stemmer)]
    return [tokenize_text(t, sw, stemmer) for t in texts]
