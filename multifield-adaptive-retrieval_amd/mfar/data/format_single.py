"""Whole-document text of the `single_dense` / `single_sparse` field (reference mfar/data/format.py:113-415,
`format_stark` and its per-dataset helpers): the STaRK-style rendering of one record that the single-field baseline
encodes instead of one text per field.  `format_documents(docs, "single", dataset)` (format.py:20-22) lands here.

The output text is what the encoder sees, so it has to match the reference character for character; it is pinned by
tests/golden/format_single.json (captured from the reference's own functions by tools/gen_golden.py), including the
documents the reference cannot format: an amazon record without `also_buy` / `also_view` and a mag record that is not a
paper make it raise UnboundLocalError (format.py:196-207, 233-238), and so do these functions.

Each renderer is a list of section builders; a section contributes "" when its source keys are absent.
"""
from typing import Any, Callable, Dict, List, Tuple


def _numbered(items, render: Callable[[int, Any], str]) -> str:
    return "".join(render(i + 1, x) for i, x in enumerate(items))


# ------------------------------------------------------------------------------------------------ amazon
def _amazon_relations(rec: dict) -> str:
    missing = [k for k in ("also_buy", "also_view") if k not in rec]
    if missing:     # the reference reads both lists unconditionally after binding them conditionally (format.py:196-207)
        raise UnboundLocalError(f"amazon record without {missing[0]!r}: the reference cannot format its relations")
    bought = _numbered(rec["also_buy"], lambda n, x: f"#{n}: {x}\n")
    viewed = _numbered(rec["also_view"], lambda n, x: f"#{n}: {x}\n")
    body = ""
    if bought:
        body += "  products also purchased: \n" + bought
    if viewed:
        body += "  products also viewed: \n" + viewed
    if "brand" in rec:
        body += f"  brand: {rec['brand']}\n"
    return " - relations:\n" + body if body else ""


def _amazon(rec: dict) -> str:
    out = [f"- product: {rec['title']}\n"]
    if "brand" in rec:
        out.append(f"- brand: {rec['brand']}\n")
    if "description" in rec:
        text = " ".join(rec["description"]).strip(" ")
        if text:
            out.append(f"- description: {text}\n")
    if "feature" in rec:        # numbering follows the position in the record, skipped entries keep their number
        out.append("- features: \n" + "".join(f"#{n}: {x}\n" for n, x in enumerate(rec["feature"], 1)
                                              if x and "asin" not in x.lower()))
    if "review" in rec:
        out.append("- reviews: \n" + _numbered(rec["review"], lambda n, r: f"#{n}:\nsummary: {r['summary']}\ntext: \"{r['reviewText']}\"\n"))
    if "qa" in rec:
        out.append("- QA: \n" + _numbered(rec["qa"], lambda n, r: f"#{n}:\nquestion: {r['question']}\nanswer: {r['answer']}\n"))
    out.append(_amazon_relations(rec))
    return "".join(out)


# ------------------------------------------------------------------------------------------------ mag
def _mag(rec: dict) -> str:
    if rec["type"] != "paper":  # only papers have a body in the reference; anything else leaves `doc` unbound (format.py:233-238)
        raise UnboundLocalError("mag record that is not a paper: the reference cannot format it")
    abstract = rec["abstract"].replace("\r", "").rstrip("\n")
    head = f" - paper title: {rec['title']}\n - abstract: {abstract}\n"
    rel = []
    if "paper___cites___paper" in rec:
        rel.append("paper cites paper: (" + ", ".join(f'"{t}"' for t in rec["paper___cites___paper"]) + ")")
    if "paper___has_topic___field_of_study" in rec:
        rel.append("paper has_topic field_of_study: (" + ", ".join(rec["paper___has_topic___field_of_study"]) + ")")
    if "author___affiliated_with___institution" in rec:
        who = rec["author___affiliated_with___institution"]
        rel.append("(" + ", ".join(f"{a} ({', '.join(who[a])})" for a in who) + ")")
    rel = [r for r in rel if r]
    return head + (" - relations:\n\n" + ",\n".join(rel) if rel else "")


# ------------------------------------------------------------------------------------------------ prime
# detail keys of gene/protein records that carry an explanation in the rendered text (facts about STaRK-prime)
_GENE_KEY_NOTES = dict(item.split("=") for item in (
    "name=gene name|type_of_gene=gene types|alias=other gene names|other_names=extended other gene names|"
    "genomic_pos=genomic position|generif=PubMed text|interpro=protein family and classification information|"
    "summary=protein summary text").split("|"))
# relation blocks appear in this fixed order
_PRIME_RELATIONS = ("ppi|carrier|enzyme|target|transporter|contraindication|indication|off-label use|synergistic interaction|"
                    "associated with|parent-child|phenotype absent|phenotype present|side effect|interacts with|linked to|"
                    "expression present|expression absent").split("|")


def _prime_detail_value(key: str, value: Any) -> Any:
    if key == "interpro" and isinstance(value, dict):
        return [value["desc"]]
    if key == "generif":
        joined = "; ".join(v["text"] for v in value)
        return " ".join(joined.split(" ")[:50000])
    if key == "genomic_pos" and isinstance(value, list):
        return value[0]
    return value


def _prime(rec: dict) -> str:
    if "name" not in rec:
        print(f"format_prime Error: \"name\" not found in {rec}. This should be required.")
        return ""
    out = [f"- name: {rec['name']}\n- type: {rec['type']}\n- source: {rec['source']}\n"]
    lines = []
    for key, value in rec.get("details", {}).items():
        if str(value) in ("", "nan") or key.startswith("_") or "_id" in key:
            continue
        if rec["type"] == "gene/protein" and key in _GENE_KEY_NOTES:
            lines.append(f"  - {key} ({_GENE_KEY_NOTES[key]}): {_prime_detail_value(key, value)}\n")
        else:
            lines.append(f"  - {key}: {value}\n")
    if lines:
        out.append("- details: \n" + "".join(lines))
    blocks = []
    for rel in _PRIME_RELATIONS:
        if rel in rec:
            inner = ", ".join(f"{k.replace(' ', '_')}: ({', '.join(rec[rel][k])})" for k in rec[rel])
            blocks.append(f"  {rel.replace(' ', '_')}: {{{inner}}}")
    if blocks:
        out.append(" - relations:\n" + "\n".join(blocks))
    return "".join(out)


# ------------------------------------------------------------------------------------------------ whatsthatbook / tomt
_BOOK_LINES_BEFORE_DATES = (("title", "title"), ("author", "author"), ("author_url", "author url"), ("description", "description"),
                            ("isbn", "isbn"))
_BOOK_LINES_AFTER_DATES = (("image_link", "image link"), ("num_ratings", "number of ratings"), ("num_reviews", "number of reviews"))


def _books(rec: dict) -> str:
    out = [f"- {label}: {rec[key]}\n" for key, label in _BOOK_LINES_BEFORE_DATES if key in rec]
    dates = [d for d in (rec.get("parsed_dates") or []) if d is not None]
    if dates:
        out.append(f"- parsed dates: {', '.join(dates)}\n")
    out += [f"- {label}: {rec[key]}\n" for key, label in _BOOK_LINES_AFTER_DATES if key in rec]
    if rec.get("genres"):
        out.append(f"- genres: {', '.join(rec['genres'])}\n")
    if "id" in rec:
        out.append(f"- id: {rec['id']}")        # last line: no newline
    return "".join(out)


_RENDERERS: Dict[str, Callable[[dict], str]] = {"amazon": _amazon, "mag": _mag, "prime": _prime, "whatsthatbook": _books, "tomt": _books}


def format_stark(data: Tuple[str, Any], dataset_name: str) -> Tuple[str, str]:
    """(doc_id, record) -> (doc_id, whole-document text)   (format.py:113-137)."""
    doc_id, rec = data
    if dataset_name not in _RENDERERS:
        raise ValueError("Select a valid STaRK dataset!")
    return doc_id, _RENDERERS[dataset_name](rec)


def format_single_documents(documents, dataset_name: str) -> List[Tuple[str, str]]:
    return [format_stark(d, dataset_name) for d in documents]
