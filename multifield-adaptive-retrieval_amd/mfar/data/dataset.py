"""Query-side dataset of the evaluation path (reference mfar/data/dataset.py:138-179).  The training datasets
(ContrastiveTrainingDataset, DecomposedInstance, ...) are training-data plumbing and out of scope."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Dict, List, Optional, Set

from mfar.data.typedef import FieldType, Query


class Kind(Enum):
    QUERY = "query"
    HYBRID = "hybrid"


@dataclass
class InstanceBatch:
    mode: Kind
    query: Optional[dict] = None
    pos_cand: Optional[dict] = None
    neg_cands: Optional[dict] = None
    instances: List[Query] = field(default_factory=list)


class QueryDataset:
    """{query_id: text} -> Query items; `collate` tokenises a batch with padding='longest' and NO truncation flag
    (dataset.py:168-173).  Queries shorter than 5 characters are replaced by "what" (dataset.py:159-160)."""

    def __init__(self, tokenizer, queries: Dict[str, str], max_length: int = 512, field_types: Set[FieldType] = None):
        self.queries, self.tokenizer, self.max_length = queries, tokenizer, max_length
        self.ids = list(queries.keys())
        self.field_types = field_types

    def __len__(self):
        return len(self.queries)

    def __getitem__(self, idx: int) -> Query:
        qid = self.ids[idx]
        text = self.queries[qid]
        if len(text.strip()) < 5:
            text = "what"
        return Query(qid, text)

    def collate(self, instances: List[Query]) -> InstanceBatch:
        toks = self.tokenizer([q.text for q in instances], max_length=self.max_length, padding="longest", return_tensors="pt")
        return InstanceBatch(mode=Kind.QUERY, query={FieldType.DENSE: toks}, instances=instances)
