"""Training-time scorer + loss (reference mfar/modeling/losses.py:149-360), plain PyTorch-ROCm.

This is the training twin of the evaluation scorer: the same per-field query.doc dot products, DIVIDED BY THE
TEMPERATURE (losses.py:184,187; evaluation does not divide, contrastive.py:685-694), optionally batch-normalised over
the field axis (losses.py:346), mixed by `LinearWeights` conditioned on the query (losses.py:347), and trained with a
bidirectional in-batch softmax NLL (query -> documents plus document -> queries, losses.py:288-301).  Tensors are tiny
(B ~ 12-24) and need autograd, so this stays stock torch ops; with several ranks the
embeddings are all-gathered with the autograd-aware collective (RCCL on ROCm) exactly where the reference does
(losses.py:255-257).  Pinned by tests/golden/hybrid_loss.npz (loss value and gradients captured from the reference).
Sparse (BM25) fields enter as extra score columns behind the dense ones (losses.py:303-345): the caller computes them on
the host from the `BM25sSparseIndex` objects (`score_batch`, or `score_batch_with_cache` over precomputed scores) and hands
them to `forward`; they are NOT divided by the temperature (losses.py:339-343), exactly like the reference.
"""
from typing import Optional

import torch


def _all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    import torch.distributed.nn.functional as dist_f
    return torch.cat(dist_f.all_gather(t), dim=0)


class HybridContrastiveLoss(torch.nn.Module):
    def __init__(self, temperature: float = 0.01, in_batch_negative: bool = True, reverse: bool = True,
                 all_gather_multi_gpu: bool = True, mixture_of_fields_layer: torch.nn.Module = None,
                 sparse_indices_dict=None, num_fields: int = 0, use_batchnorm: bool = False):
        super().__init__()
        self.sparse_indices_dict = sparse_indices_dict or {}
        self.temperature, self.in_batch_negative, self.reverse = temperature, in_batch_negative, reverse
        self.all_gather_multi_gpu = all_gather_multi_gpu
        self.mixture_of_fields_layer = mixture_of_fields_layer
        self.bn = torch.nn.BatchNorm1d(num_fields, track_running_stats=True) if use_batchnorm else torch.nn.Identity()

    # per-field components: q [Bq,E], d_pos [Bd,F,E], d_neg [Bd,F,N,E] -> [Bq,Bd,F], [Bq,Bd*N,F]   (losses.py:176-188)
    def field_components(self, q, d_pos, d_neg: Optional[torch.Tensor]):
        pos = torch.einsum("qe,dfe->qdf", q, d_pos) / self.temperature
        if d_neg is None or d_neg.numel() == 0:
            return pos, pos.new_zeros(pos.size(0), 0, pos.size(2))
        Bd, F, N, E = d_neg.shape
        neg = torch.einsum("qe,dfne->qdnf", q, d_neg).reshape(q.size(0), Bd * N, F) / self.temperature
        return pos, neg

    def _mix(self, x, q):       # x [Bq, S, F]: BatchNorm1d over the field axis, then query-conditioned field weights
        x = self.bn(x.permute(0, 2, 1)).permute(0, 2, 1)
        return self.mixture_of_fields_layer(x, q)

    @staticmethod
    def _sliced_nll(scores, batch, rank):           # losses.py:59-65: the positives sit on this rank's diagonal block
        lp = torch.log_softmax(scores, dim=1)[:, batch * rank: batch * (rank + 1)]
        return -torch.mean(torch.diag(lp))

    def forward(self, q, d_pos, d_neg: Optional[torch.Tensor] = None, sparse_pos: Optional[torch.Tensor] = None,
                sparse_neg: Optional[torch.Tensor] = None, sparse_rev: Optional[torch.Tensor] = None) -> torch.Tensor:
        """q [B,E], d_pos [B,Fd,E], d_neg [B,Fd,N,E] or None -> scalar loss (mean over ranks).
        Sparse score columns (raw BM25, no temperature): sparse_pos [B, ws*B, Fs] (this rank's queries x all positives),
        sparse_neg [B, ws*B*N, Fs], sparse_rev [ws*B, B, Fs] (all queries x this rank's positives); None without sparse fields."""
        import torch.distributed as dist
        multi = self.all_gather_multi_gpu and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        rank = dist.get_rank() if multi else 0
        all_q, all_pos = (_all_gather_cat(q), _all_gather_cat(d_pos)) if multi else (q, d_pos)
        all_neg = _all_gather_cat(d_neg) if (multi and d_neg is not None) else d_neg
        B = q.size(0)
        pos, neg = self.field_components(q, all_pos, all_neg)
        if sparse_pos is not None:                                                     # losses.py:339-343
            pos = torch.cat([pos, sparse_pos.to(pos)], dim=-1)
            neg = torch.cat([neg, (sparse_neg if sparse_neg is not None else pos.new_zeros(pos.size(0), 0, sparse_pos.size(2))).to(neg)], dim=-1)
        mixed = self._mix(torch.cat([pos, neg], dim=1), q)                           # [B, ws*B + ws*B*N]
        nll = self._sliced_nll(mixed, B, rank)
        if self.reverse:                                                               # document -> queries (losses.py:352-360)
            rev = torch.einsum("dfe,qe->qdf", d_pos, all_q) / self.temperature        # [ws*B, B, F]
            if sparse_rev is not None:                                                 # losses.py:356-358
                rev = torch.cat([rev, sparse_rev.to(rev)], dim=-1)
            nll = nll + self._sliced_nll(self._mix(rev, all_q).t(), B, rank)
        if multi:
            import torch.distributed.nn.functional as dist_f
            nll = dist_f.all_reduce(nll) / dist.get_world_size()
        return nll
