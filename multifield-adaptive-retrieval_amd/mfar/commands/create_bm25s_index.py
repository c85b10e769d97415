"""`python -m mfar.commands.create_bm25s_index --data_path D --dataset_name N --output_path O` -- build and save the BM25
indices of the sparse fields (reference mfar/commands/create_bm25s_index.py): one directory `{output_path}/{field_key}_sparse_index`
per field; `single_sparse_sparse_index` is the one hard-negative mining reads (contrastive.py:72)."""
from mfar.commands._cli import run
from mfar.data import trec
from mfar.data.format import format_documents
from mfar.data.index import BM25sSparseIndex
from mfar.data.schema import resolve_fields
from mfar.data.typedef import Corpus


def main(data_path: str, dataset_name: str, output_path: str, fields_str: str = "all_sparse,single_sparse"):
    fields = resolve_fields(fields_str, dataset_name)
    corpus = list(trec.read_corpus(f"{data_path}/corpus"))
    for field_name, field in fields.items():
        formatted = format_documents(corpus, field.name, field.dataset)
        docs = Corpus.from_docs_dict({item[0]: item[1] for item in formatted}, dataset_name=dataset_name)
        BM25sSparseIndex.create(docs, dataset_name=dataset_name).save(f"{output_path}/{field_name}_sparse_index")


if __name__ == "__main__":
    run(main)
