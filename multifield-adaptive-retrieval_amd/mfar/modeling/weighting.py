"""Field-weight mixer (reference mfar/modeling/weighting.py:3-29).

`LinearWeights` keeps the reference's constructor, parameter name/shape/initialisation and `forward` (plain torch ops,
autograd-capable -- the training loss needs gradients through it, reference mfar/modeling/losses.py:347,360).
The evaluation hot path does not call `forward`: `mix_topk` hands mask * scores -> softmax head -> weighted sum ->
top-k to the HIP kernel (mfar_mix_topk in include/mfar_hip.h), and `MultiFieldIndex.search` fuses it with the
retrieval stages.
"""
import torch


class LinearWeights(torch.nn.Module):
    """Linear field weights: (1) a plain mixture of fields, or (2) query-conditioned weights (weighting.py:4-8)."""

    def __init__(self, emb_size, num_fields, query_cond=False):
        super().__init__()
        self.query_cond = query_cond
        # emb_size = num_fields and num_fields = 1 when not query-conditioned (contrastive.py:286-287)
        self.weight = torch.nn.Parameter(torch.ones(emb_size, num_fields), requires_grad=True)

    def forward(self, x, q) -> torch.Tensor:
        """x [Batch, Samples, Field] (or [Samples, Field]), q [Batch, Emb] -> [Batch, Samples] (weighting.py:17-29)."""
        if self.query_cond:
            weights = q @ self.weight
        else:
            weights = self.weight.transpose(1, 0)
        weights_dist = torch.softmax(weights, dim=1)
        return torch.sum(weights_dist.unsqueeze(1) * x, dim=-1)

    @torch.no_grad()
    def mix_topk(self, cand_scores, cand_ids, q, mask=None, n_cand=None, k: int = 100):
        """Evaluation path on the GPU: cand_scores [Q, C, F], cand_ids [Q, C] -> top-k (ids, scores, n_valid).
        Arrays are numpy or CUDA tensors (see mfar.data.index.mix_topk)."""
        from mfar.data.index import mix_topk
        W = self.weight.detach()
        dev = W.device.index if W.is_cuda else 0
        if not self.query_cond:
            W = W.reshape(-1)
        on_dev = hasattr(cand_scores, "is_cuda") and cand_scores.is_cuda
        Wx = W.contiguous().float() if on_dev else W.cpu().float().numpy()
        return mix_topk(cand_scores, cand_ids, q, Wx, mask, n_cand, k=k, query_cond=self.query_cond, device=dev)
