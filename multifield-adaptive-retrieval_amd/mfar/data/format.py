"""Document JSON -> per-field text (reference mfar/data/format.py:7-110).  This is the text the corpus encoder sees, so
its exact shape matters for parity: pinned by tests/golden/format_documents.json (captured from the reference).

Rules of `format_documents` (format.py:26-61): a missing field gives "" (and is still encoded -- every document
lacking a field yields the same vector, which is where exact score ties come from); str as is; numbers via str();
list of str joined by ", "; list of dicts -> per item "key: value" lines (minus a fixed set of bookkeeping keys),
items joined by newlines; None -> ""; dict -> `format_dict`.
The whole-document 'single' field of the single_dense baseline (format.py:20-22, 113-415) is rendered by
`mfar.data.format_single` (pinned by tests/golden/format_single.json)."""
from typing import Any, List, Tuple

_DROP_KEYS = {"reviewerID", "style", "verified", "overall", "reviewTime", "vote", "questionType", "answerType", "answerTime"}
_NL = chr(10)


def _scalar(v) -> bool:
    return isinstance(v, (str, int, float))


def format_dict(d: dict) -> str:
    """Nested prime-style dict -> text (format.py:64-110)."""
    parts: List[str] = []
    for key, value in d.items():
        if _scalar(value):
            parts.append(f"{key}: {value}")
        elif isinstance(value, list):
            if not value:
                parts.append(f"{key}: ")
            elif isinstance(value[0], dict):
                grouped = {}
                for item in value:
                    for k, v in item.items():
                        bucket = grouped.setdefault(k, [])
                        if isinstance(v, dict):
                            bucket.extend(v.values())
                        else:
                            bucket.append(v)
                parts.append("".join(f"{k}: {', '.join(str(x) for x in xs)}; " for k, xs in grouped.items()))
            elif isinstance(value[0], list):
                raise NotImplementedError("Nested list not supported!")
            else:
                parts.append(f"{key}: " + ", ".join(value))
        elif isinstance(value, dict):
            parts.append(", ".join(f"{k}: {value[k]}" for k in value))
        else:
            parts.append(", ".join(value))
    return "; ".join(parts)


def _field_text(value: Any) -> str:
    if isinstance(value, str):
        return value
    if isinstance(value, (int, float)):
        return str(value)
    if isinstance(value, list):
        if not value:
            return ""
        if isinstance(value[0], dict):
            return _NL.join(_NL.join(f"{k}: {v}" for k, v in item.items() if k not in _DROP_KEYS) for item in value)
        if isinstance(value[0], list):
            raise NotImplementedError("Nested list not supported!")
        return ", ".join(value)
    if value is None:
        return ""
    return format_dict(value)


def format_documents(documents, field_name: str, dataset_name: str) -> List[Tuple[str, str]]:
    """[(doc_id, doc_json)] -> [(doc_id, text of `field_name`)]."""
    if field_name == "single":
        from mfar.data.format_single import format_single_documents
        return format_single_documents(documents, dataset_name)
    out = []
    for doc_id, body in documents:
        out.append((doc_id, _field_text(body[field_name]) if field_name in body else ""))
    return out
