"""Field presets per dataset and `resolve_fields` (reference mfar/data/schema.py).

`resolve_fields` fixes the number of fields F and -- more importantly -- their ORDER: sorted dense keys first, then
sorted sparse keys (schema.py:131-134).  That order is the column order of the field-weight matrix W [E, F], the row
order of the mask, and the field order inside the HBM slab.
"""
from typing import Dict, Iterable, Union

from mfar.data.typedef import Field, FieldType

SPARSE_MAX = 1048576

# Per dataset: "field name=training-time max token length" items (facts about the STaRK datasets; reference
# schema.py:11-69).  Kept as one compact text block and parsed once at import.
_PRESET_TEXT = {
    "mag": "abstract=512; author___affiliated_with___institution=512; paper___cites___paper=512; "
           "paper___has_topic___field_of_study=64; title=64",
    "prime": "associated with=256; carrier=8; contraindication=128; details=512; enzyme=64; expression absent=64; "
             "expression present=512; indication=32; interacts with=512; linked to=8; name=64; off-label use=8; "
             "parent-child=256; phenotype absent=8; phenotype present=512; ppi=512; side effect=128; source=8; "
             "synergistic interaction=512; target=64; transporter=8; type=8",
    "amazon": "also_buy=512; also_view=512; brand=16; description=512; feature=512; qa=512; review=512; title=128",
    "whatsthatbook": "author=16; author_url=64; date=64; description=512; genres=64; id=16; image_link=64; isbn_13=16; "
                     "parsed_dates=16; ratings=16; reviews=16; title=64",
}


def _parse_presets(text: str) -> Dict[str, int]:
    out = {}
    for item in text.split(";"):
        name, _, length = item.strip().rpartition("=")
        out[name] = int(length)
    return out


FIELD_PRESETS = {ds: _parse_presets(txt) for ds, txt in _PRESET_TEXT.items()}
DATASET_NAMES = list(FIELD_PRESETS)


def _schema(dataset: str) -> Dict[str, Field]:
    out = {}
    for name, max_len in FIELD_PRESETS[dataset].items():
        out[f"{name}_sparse"] = Field(f"{name}_sparse", name, FieldType.SPARSE, SPARSE_MAX, dataset=dataset)
        out[f"{name}_dense"] = Field(f"{name}_dense", name, FieldType.DENSE, max_len, dataset=dataset)
    return out


SCHEMAS = {d: _schema(d) for d in DATASET_NAMES}
STARK_SCHEMAS = {d: {"single_sparse": Field("single_sparse", "single", FieldType.SPARSE, SPARSE_MAX, d),
                     "single_dense": Field("single_dense", "single", FieldType.DENSE, 512, d)} for d in DATASET_NAMES}


def resolve_fields(field_names: Union[str, Iterable[str]], dataset: str) -> Dict[str, Field]:
    """'all_dense' / 'all_sparse' / 'single_dense' / 'single_sparse' / explicit '<name>_dense' keys (comma separated
    string or list; '.' stands for a blank inside a name, schema.py:108-110) -> ordered {key: Field}."""
    base = dataset.split("/")[-1]
    ds = next((d for d in DATASET_NAMES if d in base), None)
    if ds is None:
        raise NotImplementedError(f"Dataset {dataset} is not supported!")
    table = SCHEMAS[ds]
    if isinstance(field_names, str):
        field_names = [n.replace(".", " ") for n in field_names.split(",")]
    picked: Dict[str, Field] = {}
    for n in field_names:
        if n in ("all_sparse", "all_dense"):
            want = FieldType.SPARSE if n == "all_sparse" else FieldType.DENSE
            picked.update({k: f for k, f in table.items() if f.field_type == want})
        elif n in ("single_sparse", "single_dense"):
            picked[n] = STARK_SCHEMAS[ds][n]
        elif n in table:
            picked[n] = table[n]
        else:
            raise ValueError(f"Field {n} not found in dataset {dataset}")
    keys = sorted(picked)
    ordered = [k for k in keys if picked[k].field_type == FieldType.DENSE] + \
              [k for k in keys if picked[k].field_type == FieldType.SPARSE]
    return {k: picked[k] for k in ordered}
