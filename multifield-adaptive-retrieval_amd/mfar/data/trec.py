"""TREC file formats and retrieval metrics (reference mfar/data/trec.py).

`QRels` / `QRes` keep the reference's line formats (trec.py:22-23, 49-50).  The reference shells out to the `trec_eval`
binary (trec.py:84-93); it is used here too when it is on PATH, otherwise the same metrics are computed in Python from
their trec_eval definitions (`compute_metrics`) -- the binary exists on neither box, so those values are unpinned."""
import csv
import json
import math
import shutil
import subprocess
import sys
from collections import defaultdict
from dataclasses import dataclass
from typing import Dict, Iterable, List, TextIO, Tuple

csv.field_size_limit(sys.maxsize)
_TAB = chr(9)
_NL = chr(10)


@dataclass
class QRels:
    query_id: str
    doc_id: str
    relevance: float
    _iter: str = "0"

    def __str__(self):
        return _TAB.join([self.query_id, self._iter, self.doc_id, str(self.relevance)])

    @classmethod
    def from_str(cls, s: str) -> "QRels":
        q, it, d, rel = s.split(_TAB)
        return cls(q, d, float(rel), it)

    @classmethod
    def from_text_io(cls, f: TextIO) -> List["QRels"]:
        return [cls.from_str(line.strip()) for line in f]


@dataclass
class QRes:
    query_id: str
    doc_id: str
    sim: float
    run_id: str = "0"
    _iter: str = "0"
    _rank: int = 0

    def __str__(self):
        return _TAB.join([self.query_id, self._iter, self.doc_id, str(self._rank), str(self.sim), self.run_id])

    @classmethod
    def from_str(cls, s: str) -> "QRes":
        q, it, d, rank, sim, run = s.split()
        return cls(q, d, float(sim), run, it, int(rank))

    @classmethod
    def from_text_io(cls, f: TextIO) -> List["QRes"]:
        return [cls.from_str(line.strip()) for line in f]


_NOT_METRICS = {"runid", "num_q", "num_ret", "num_rel", "num_rel_ret"}


def parse_trec_eval_output(output: str) -> Dict[str, float]:
    metrics = {}
    for line in output.split(_NL):
        if not line:
            continue
        name, _, value = line.strip().split(_TAB)
        name = name.strip()
        if name not in _NOT_METRICS:
            metrics[name] = float(value.strip())
    return metrics


def _dcg(gains) -> float:
    return sum(g / math.log2(i + 2) for i, g in enumerate(gains))


_CUTS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)
_DISCOUNT = [math.log2(i + 2) for i in range(1024)]


def _dcg_prefix(gains) -> List[float]:
    """out[k] = _dcg(gains[:k]) with the additions in the same left-to-right order."""
    out, tot = [0.0], 0
    disc = _DISCOUNT
    for i, g in enumerate(gains):
        tot = tot + g / (disc[i] if i < 1024 else math.log2(i + 2))
        out.append(tot)
    return out


def read_qres_tuples(f: TextIO) -> List[Tuple[str, str, float]]:
    """(query_id, doc_id, sim) of every line of a .qres file: what `QRes.from_text_io` parses, without the dataclasses (an
    evaluation reads 100 lines per query; a mask sweep reads them 2 F + 2 times)."""
    out = []
    for line in f:
        q, _, d, _, sim, _ = line.split()
        out.append((q, d, float(sim)))
    return out


def compute_metrics(qrels: Iterable[QRels], qres) -> Dict[str, float]:
    """map, recip_rank, Rprec, ndcg, ndcg_cut_k, recall_k, success_k, P_k averaged over the queries that appear in the
    run and have at least one relevant document (trec_eval's default averaging).  Ranking: sim descending, ties by
    doc id descending (trec_eval's tie rule); the rank column is ignored like trec_eval does.
    `qres`: QRes records or (query_id, doc_id, sim) tuples."""
    rel = defaultdict(dict)
    for r in qrels:
        rel[r.query_id][r.doc_id] = r.relevance
    runs = defaultdict(list)
    for r in qres:
        if isinstance(r, tuple):
            runs[r[0]].append((r[2], r[1]))
        else:
            runs[r.query_id].append((r.sim, r.doc_id))
    sums = defaultdict(float)
    nq = 0
    for qid, docs in runs.items():
        pos = {d: g for d, g in rel.get(qid, {}).items() if g > 0}
        if not pos:
            continue
        nq += 1
        docs = sorted(docs, key=lambda t: t[1], reverse=True)
        docs.sort(key=lambda t: t[0], reverse=True)
        ranked = [d for _, d in docs]
        n = len(ranked)
        R = len(pos)
        cum, ap, rr = 0, 0.0, 0.0
        hits_prefix = [0]                                  # hits_prefix[k] = relevant documents among the first k
        gains = []
        for i, d in enumerate(ranked):
            g = pos.get(d)
            if g is not None:
                cum += 1
                ap += cum / (i + 1)
                if rr == 0.0:
                    rr = 1.0 / (i + 1)
                gains.append(g)
            else:
                gains.append(0.0)
            hits_prefix.append(cum)
        sums["map"] += ap / R
        sums["recip_rank"] += rr
        sums["Rprec"] += hits_prefix[min(R, n)] / R
        ideal = sorted(pos.values(), reverse=True)
        dcg, idcg = _dcg_prefix(gains), _dcg_prefix(ideal)
        sums["ndcg"] += dcg[n] / idcg[R]
        for k in _CUTS:
            top = hits_prefix[min(k, n)]
            sums[f"recall_{k}"] += top / R
            sums[f"P_{k}"] += top / k
            sums[f"ndcg_cut_{k}"] += dcg[min(k, n)] / idcg[min(k, R)]
        for k in (1, 5, 10):
            sums[f"success_{k}"] += 1.0 if hits_prefix[min(k, n)] > 0 else 0.0
    return {k: (v / nq if nq else 0.0) for k, v in sums.items()}


def call_trec_eval_and_get_metrics(qrels: str, qres: str) -> Dict[str, float]:
    """trec.py:84-93: `trec_eval -m all_trec qrels qres` when the binary exists, else the Python metrics."""
    if shutil.which("trec_eval"):
        proc = subprocess.run(["trec_eval", "-m", "all_trec", qrels, qres], stdout=subprocess.PIPE, check=True)
        return parse_trec_eval_output(proc.stdout.decode("utf-8"))
    with open(qrels) as f1, open(qres) as f2:
        return compute_metrics(QRels.from_text_io(f1), read_qres_tuples(f2))


def read_corpus(path: str) -> Iterable[Tuple[str, object]]:
    """Tab separated 'id, json' rows (trec.py:96-106): missing body -> "", unparsable body -> the raw remainder."""
    with open(path, "r") as f:
        for row in csv.reader(f, delimiter=_TAB):
            if len(row) < 2:
                yield row[0], ""
                continue
            try:
                yield row[0], json.loads(row[1])
            except Exception:
                yield row[0], _TAB.join(row[1:])
