"""HybridContrastiveLoss (the training-time scorer + loss, reference mfar/modeling/losses.py:176-188, 275-360) against the loss
value and gradients captured from the reference (tests/golden/hybrid_loss.npz, tools/gen_golden.py): dense columns divided by
the temperature, optional BatchNorm over the field axis, raw sparse (BM25) columns behind the dense ones.  Runs on any device:
the CPU suite calls it with "cpu", the `-m gpu` suite with "cuda:0"."""
import os

import numpy as np


def check_hybrid_loss_golden(golden_dir: str, device: str, rtol: float = 1e-4, atol: float = 1e-5):
    import torch
    from mfar.modeling.losses import HybridContrastiveLoss
    from mfar.modeling.weighting import LinearWeights
    z = np.load(os.path.join(golden_dir, "hybrid_loss.npz"))
    dev = torch.device(device)
    seen = []
    for name, use_bn in (("plain", False), ("bn", True), ("sparse", True)):
        t = lambda k: torch.tensor(z[f"{name}__{k}"], device=dev)
        q, d_pos, d_neg = (t(k).requires_grad_() for k in ("q", "d_pos", "d_neg"))
        E, F = z[f"{name}__W"].shape
        lw = LinearWeights(E, F, query_cond=True)
        lw.weight.data = torch.from_numpy(z[f"{name}__W"].copy())
        fn = HybridContrastiveLoss(temperature=float(z["temperature"]), mixture_of_fields_layer=lw, sparse_indices_dict={},
                                   num_fields=F, use_batchnorm=use_bn).to(dev)
        fn.train()
        sparse = dict(sparse_pos=t("sparse_pos"), sparse_neg=t("sparse_neg"), sparse_rev=t("sparse_rev")) if name == "sparse" else {}
        loss = fn(q, d_pos, d_neg, **sparse)
        loss.backward()
        assert loss.device.type == dev.type and lw.weight.grad.device.type == dev.type
        assert abs(float(loss) - float(z[f"{name}__loss"])) <= 1e-5 + 1e-5 * abs(float(z[f"{name}__loss"])), (name, float(loss))
        for got, key in ((lw.weight.grad, "grad_W"), (q.grad, "grad_q"), (d_pos.grad, "grad_d_pos")):
            np.testing.assert_allclose(got.cpu().numpy(), z[f"{name}__{key}"], rtol=rtol, atol=atol, err_msg=f"{name} {key} on {device}")
        # the training-time per-field scorer itself (losses.py:176-188): q . d_f / temperature
        pos, neg = fn.field_components(q.detach(), d_pos.detach(), d_neg.detach())
        want = np.einsum("qe,dfe->qdf", z[f"{name}__q"].astype(np.float64), z[f"{name}__d_pos"].astype(np.float64)) / float(z["temperature"])
        np.testing.assert_allclose(pos.cpu().numpy(), want, rtol=1e-5, atol=1e-4)
        seen.append(name)
    return seen
