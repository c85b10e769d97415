"""Hard-negative mining for training (reference mfar/data/negative_sampler.py:10-60): retrieve `n_retrieve` documents for
the query text from an `Index` (the reference uses the BM25 index over the whole-document text,
`{lexical_index}/single_sparse_sparse_index`, contrastive.py:71-77), drop the positives, keep the `n_bottom` LOWEST-scored of
what is left and sample `n_sample` of those."""
import random
from abc import ABC
from typing import AbstractSet, List, Mapping, Tuple

from mfar.data.index import Index
from mfar.data.typedef import Document, Query


class NegativeSampler(ABC):
    @property
    def n_sample(self) -> int:
        raise NotImplementedError

    def sample(self, query: Query, pos_for_each_qid: Mapping[str, AbstractSet[str]]) -> List[Document]:
        raise NotImplementedError

    def sample_batch(self, queries: List[Query], pos_for_each_qid: Mapping[str, AbstractSet[str]]) -> List[List[Document]]:
        raise NotImplementedError


class IndexNegativeSampler(NegativeSampler):
    def __init__(self, index: Index, documents: Mapping[str, str], n_retrieve: int = 50, n_bottom: int = 5, n_sample: int = 1,
                 rng: random.Random = None):
        self.index, self.documents = index, documents
        self.n_retrieve, self.n_bottom, self._n_sample = n_retrieve, n_bottom, n_sample
        self.rng = rng if rng is not None else random        # the reference draws from the global `random` module

    @property
    def n_sample(self) -> int:
        return self._n_sample

    def _negatives(self, query: Query, positives: AbstractSet[str], top_k: int) -> List[Tuple[str, float]]:
        return [(doc_id, score) for doc_id, score in self.index.retrieve(query.text, top_k=top_k) if doc_id not in positives]

    def bottom_candidates(self, query: Query, pos_for_each_qid: Mapping[str, AbstractSet[str]]) -> List[str]:
        """The deterministic half of `sample` (negative_sampler.py:40-56): retrieve, drop the positives, keep the `n_bottom`
        lowest-scored ids.  A pure function of the query: callers that sample the same query every epoch cache it."""
        positives = pos_for_each_qid[query._id]
        cand = self._negatives(query, positives, self.n_retrieve)
        if not cand:                                        # every retrieved document was a positive: look deeper (:47-53)
            cand = self._negatives(query, positives, len(positives) + self.n_bottom)
        cand.sort(key=lambda x: x[1], reverse=True)         # stable: equal scores keep the index's order
        return [doc_id for doc_id, _ in cand[-self.n_bottom:]]

    def pick(self, bottom: List[str], rng=None) -> List[str]:
        """The random half (:57): `n_sample` of the bottom candidates."""
        rng = rng if rng is not None else self.rng
        return [bottom[i] for i in rng.sample(range(len(bottom)), self.n_sample)]

    def sample(self, query: Query, pos_for_each_qid: Mapping[str, AbstractSet[str]]) -> List[Document]:
        picked = self.pick(self.bottom_candidates(query, pos_for_each_qid))
        return [Document(i, self.documents.get(i, "")) for i in picked]

    def sample_batch(self, queries: List[Query], pos_for_each_qid: Mapping[str, AbstractSet[str]]) -> List[List[Document]]:
        return [self.sample(q, pos_for_each_qid) for q in queries]
