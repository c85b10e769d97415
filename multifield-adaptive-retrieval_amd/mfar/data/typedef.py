"""Small data types of the path (reference mfar/data/typedef.py): `FieldType`, `Field`, `Query`, `Document`, `Corpus`.
Only the shapes the scoring path needs are kept; the gzip/JSON-mixin loaders of the reference are training-data
plumbing and out of scope."""
import json
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, Iterator, List, Optional


class FieldType(Enum):
    SPARSE = 1   # typedef.py:69-71
    DENSE = 2


class Field:
    """One searchable field of a dataset (typedef.py:73-122): `key` is '<name>_dense' / '<name>_sparse', `name` the
    JSON key inside a document, `max_seq_length` the training-time token budget (schema.py:11-69)."""

    def __init__(self, key: str, name: str, field_type: FieldType, max_seq_length: int = 512, dataset=None):
        self.key, self.name, self.field_type = key, name, field_type
        self.max_seq_length, self.dataset = max_seq_length, dataset

    def serialize(self) -> dict:            # checkpoint hyper-parameter form (contrastive.py:634-640)
        return dict(key=self.key, name=self.name, field_type=self.field_type.name, max_seq_length=self.max_seq_length,
                    dataset=self.dataset)

    @classmethod
    def deserialize(cls, d: dict) -> "Field":
        return cls(d["key"], d["name"], FieldType[d["field_type"]], d["max_seq_length"], d["dataset"])

    def __dict__(self) -> dict:             # callable like the reference's (train.py:71 calls v.__dict__())
        return dict(name=self.name, field_type=self.field_type.name, max_seq_length=self.max_seq_length)

    def __str__(self) -> str:
        return json.dumps(self.__dict__())

    def __repr__(self) -> str:
        return f"Field({self.key!r}, {self.name!r}, {self.field_type.name}, {self.max_seq_length}, {self.dataset!r})"

    def __copy__(self):
        return Field(self.key, self.name, self.field_type, self.max_seq_length, self.dataset)

    def __deepcopy__(self, memo):
        return self.__copy__()


@dataclass
class Query:
    _id: str
    text: str
    metadata: Any = None


@dataclass
class Document:
    _id: str
    text: Any
    title: Optional[str] = None
    metadata: Any = None


@dataclass
class Corpus:
    """Ordered documents with key -> position lookup (typedef.py:125-171); what `BM25sSparseIndex.create` consumes."""
    docs: List[Document]
    dataset_name: Optional[str] = None

    def __post_init__(self):
        self.key_to_id = {doc._id: i for i, doc in enumerate(self.docs)}

    def keys(self) -> Iterator[str]:
        return (doc._id for doc in self.docs)

    def __len__(self):
        return len(self.docs)

    def get_text_by_id(self, doc_id: int) -> str:
        return self.docs[doc_id].text

    def get_text_by_key(self, key: str) -> str:
        return self.docs[self.key_to_id[key]].text

    def get_doc_by_id(self, doc_id: int) -> Document:
        return self.docs[doc_id]

    def get_doc_by_key(self, key: str) -> Document:
        try:
            return self.docs[self.key_to_id[key]]
        except KeyError:
            raise KeyError(f"Key {key} not found in corpus.")

    def pairs(self):
        return ((doc._id, doc.text) for doc in self.docs)

    @classmethod
    def from_docs_dict(cls, docs_dict: Dict[Any, str], dataset_name: str = None) -> "Corpus":
        return cls([Document(key, text) for key, text in docs_dict.items()], dataset_name)
