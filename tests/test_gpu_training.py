"""`-m gpu` twin of the training-scorer checks: the training-time scorer + loss of the reference
(mfar/modeling/losses.py:176-188, 275-360) on the DEVICE against the loss value and gradients captured from the reference, and
the same per-field dot products against the HIP stage 2 (the evaluation-time scorer, mfar/data/index.py:227-232)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_hybrid_contrastive_loss_golden_on_device(golden_dir):
    """Loss + grad(W, q, d_pos) on cuda:0 == the reference's (plain / BatchNorm over fields / sparse score columns)."""
    from helpers.loss_check import check_hybrid_loss_golden
    assert check_hybrid_loss_golden(golden_dir, "cuda:0") == ["plain", "bn", "sparse"]


def test_training_scorer_equals_hip_stage2_over_temperature():
    """losses.py:176-188 computes q . d_f / temperature with torch ops on the device; the evaluation path computes q . d_f with
    `mfar_score_candidates` (the arithmetic contract's fma chain): the two scorers agree to fp32 summation-order noise, on the
    same rows, for every field -- training and evaluation rank documents by the same numbers."""
    import torch
    from mfar.data.index import MultiFieldIndex
    from mfar.modeling.losses import HybridContrastiveLoss
    from mfar.modeling.weighting import LinearWeights
    rng = np.random.default_rng(77)
    F, D, E, B, N, T = 5, 4000, 768, 12, 2, 0.05
    slab = (rng.standard_normal((F, D, E)) * 0.05 + 0.02).astype(np.float32)
    q = (rng.standard_normal((B, E)) * 0.05 + 0.02).astype(np.float32)
    ix = MultiFieldIndex(D, F, E, device=0)
    for f in range(F):
        ix.write_rows(f, 0, slab[f])
    pos_ids = rng.choice(D, B, replace=False)
    neg_ids = rng.choice(D, (B, N), replace=False)
    dev = torch.device("cuda:0")
    d_pos = torch.from_numpy(slab[:, pos_ids].transpose(1, 0, 2).copy()).to(dev)                 # [B, F, E]
    d_neg = torch.from_numpy(slab[:, neg_ids].transpose(1, 0, 2, 3).copy()).to(dev)              # [B, F, N, E]
    fn = HybridContrastiveLoss(temperature=T, mixture_of_fields_layer=LinearWeights(E, F, query_cond=True), num_fields=F).to(dev)
    pos, neg = fn.field_components(torch.from_numpy(q).to(dev), d_pos, d_neg)                  # [B, B, F], [B, B*N, F]
    cand = np.concatenate([np.tile(pos_ids, (B, 1)), np.tile(neg_ids.reshape(-1), (B, 1))], axis=1).astype(np.int64)
    x = ix.score_candidates(q, cand) / np.float32(T)                                           # [B, B + B*N, F]
    got = torch.cat([pos, neg], dim=1).cpu().numpy()
    np.testing.assert_allclose(got, x, rtol=2e-5, atol=2e-4)
    ix.close()
