"""The error bound behind the fp16 screen's certificate (csrc/mfar_screen.h), checked on the CPU against a numpy emulation
of the screen arithmetic: centring on the field mean, power-of-two scaling, fp16 rounding of the rows, the fp16 query (two-term split for
the 64-column pass, ONE term for the wide 128-column pass, whose rounding enters the bound at first order), products
summed without further error (float64) -- i.e. every error source except the MFMA's own fp32 accumulation, which the bound
budgets separately with (4K + 64) u32.  The exact side is the oracle's fma chain (the arithmetic contract)."""
import numpy as np
import pytest

from oracle import mfar_oracle as O

U16, U32, SLACK = 2.0 ** -11, 2.0 ** -24, 1.25


def _pow2_scale(amax):
    if amax == 0 or not np.isfinite(amax):
        return 1.0
    e = 13 - int(np.floor(np.log2(amax)))
    return float(2.0 ** max(-100, min(100, e)))


def _eps(qn, dn, mn, sq, sf, K, terms=2):
    c_rel = (1.02 if terms == 2 else 2.04) * U16 + (4.0 * K + 66.0) * U32
    c_abs = U32 * np.sqrt(K) * 1.0001
    return SLACK * (c_rel * qn * dn + K * U32 * qn * (dn + 2.0 * mn) + c_abs * (qn / sf + dn / sq))


@pytest.mark.parametrize("E,scale,seed,shift", [(32, 1.0, 0, 0.0), (768, 1.0, 1, 0.0), (768, 1e-3, 2, 5.0), (256, 3e4, 3, 0.3),
                                                 (96, 1e-20, 4, 0.0), (768, 1.0, 5, 40.0)])
def test_screen_error_bound_holds(E, scale, seed, shift):
    rng = np.random.default_rng(seed)
    D, Q = 4000, 16
    common = rng.standard_normal(E) * shift                 # a large shared component, like real sentence embeddings
    docs = ((rng.standard_normal((D, E)) * rng.lognormal(0, 1.5, (D, 1)) + common) * scale).astype(np.float32)
    docs[::97] *= np.float32(1e-4)                      # rows deep in the fp16 subnormal range after scaling
    docs[5] = 0
    q = (rng.standard_normal((Q, E)) * rng.lognormal(0, 1, (Q, 1))).astype(np.float32)
    q[3, ::2] *= np.float32(1e-6)
    exact = O.c_scores(docs, q)
    m = docs.mean(0, dtype=np.float32)                      # any vector works; the kernels use the field's mean
    c = (docs - m).astype(np.float32)                       # fl(d - m)
    sf = _pow2_scale(float(np.abs(c).max()))
    dn = float(np.sqrt((c.astype(np.float64) ** 2).sum(1).max()))
    mn = float(np.sqrt((m.astype(np.float64) ** 2).sum()))
    d16 = (c * np.float32(sf)).astype(np.float16).astype(np.float64)
    worst = 0.0
    for i in range(Q):
        sq = _pow2_scale(float(np.abs(q[i]).max()))
        qs = q[i] * np.float32(sq)
        a = qs.astype(np.float16)
        b = (qs - a.astype(np.float32)).astype(np.float16)
        qm = np.float32(0)
        for e in range(E):                                  # q . m in fp32, some summation order
            qm = np.float32(qm + np.float32(q[i, e] * m[e]))
        qn = float(np.sqrt((q[i].astype(np.float64) ** 2).sum()))
        for terms, qq in ((2, a.astype(np.float64) + b.astype(np.float64)), (1, a.astype(np.float64))):
            approx = (d16 @ qq) / (sq * sf) + float(qm)
            eps = _eps(qn, dn, mn, sq, sf, E, terms)
            err = np.abs(approx - exact[i].astype(np.float64)).max()
            assert err <= eps, (i, terms, err, eps)
            worst = max(worst, err / eps)
    assert worst < 0.5            # the rigorous bound is comfortably loose on real numbers


def test_stage2_approximate_scores_stay_inside_the_rigorous_bound():
    """The two-level stage 2's bound (csrc/mfar_select.h: mfar_s2_prep_kernel), measured: |approximate - exact| of every (query,
    row) pair against eps, the approximate score re-derived in numpy from the same centring / scale rule (fp16 of
    (d - mean) * 2^e, fp32 query) -- an independent restatement of the SRC_F16G arithmetic."""
    rng = np.random.default_rng(304)
    F, D, E, Q = 3, 5000, 768, 8
    mu = rng.standard_normal(E).astype(np.float32)
    mu /= np.linalg.norm(mu)
    slab = (rng.standard_normal((F, D, E)) * 0.04 + mu).astype(np.float32)
    q = (rng.standard_normal((Q, E)) * 0.04 + mu).astype(np.float32)
    worst = 0.0
    for f in range(F):
        mean = slab[f].mean(0, dtype=np.float64).astype(np.float32)
        c = slab[f] - mean
        amax = np.abs(c).max()
        sf = np.float32(2.0 ** (13 - int(np.floor(np.log2(amax)))))
        h = (c * sf).astype(np.float16).astype(np.float64)
        approx = (q.astype(np.float64) @ h.T) / float(sf) + (q.astype(np.float64) @ mean.astype(np.float64))[:, None]
        exact = q.astype(np.float64) @ slab[f].astype(np.float64).T
        qn = np.linalg.norm(q, axis=1)
        dmax, mn = np.linalg.norm(c, axis=1).max(), np.linalg.norm(mean)
        K, u16, u32 = float(E), 2.0 ** -11, 2.0 ** -24
        eps = 1.25 * ((u16 + 1.01 * (K + 2) * u32) * qn * dmax + 1.01 * (K + 1) * u32 * qn * (dmax + mn) + 1.01 * K * u32 * qn * mn +
                      u32 * np.sqrt(K) * qn / float(sf))
        ratio = np.abs(approx - exact).max(1) / eps
        worst = max(worst, float(ratio.max()))
    assert worst < 0.5, worst       # typical errors sit far inside the worst-case bound


def _bf16_round_bits(x):
    """fp32 -> bf16 bits, round to nearest even (csrc/mfar_device.h f2bf)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _bf16_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def _cv_bf16_to_f16(bits, scale):
    """The in-register conversion of the converted-docs pass (csrc/mfar_stage1.h s1_bf16x2_to_f16x2), element-wise: magnitude clamped
    up to the smallest bf16 pattern that is a normal fp16 number under the field's scale, exponent rebiased, mantissa shifted."""
    rebias = 112 - int(np.log2(scale))
    mag = (bits & 0x7FFF).astype(np.uint32)
    mag = np.maximum(mag, (rebias + 1) << 7)
    mag = ((mag - (rebias << 7)) << 3) & 0xFFFF
    out = (mag & 0x7FFF) | (bits & 0x8000)
    return out.astype(np.uint16).view(np.float16)


@pytest.mark.parametrize("E,scale,seed", [(64, 1.0, 0), (768, 1.0, 1), (768, 1e-3, 2), (256, 3e4, 3), (96, 1e-20, 4)])
def test_bf16_certified_pass_bounds_hold(E, scale, seed):
    """The two certified passes over a bf16 slab (csrc/mfar_screen.h "bf16 indexes"): (a) two bf16 query terms against the raw rows
    (mfar_direct_queries_kernel), (b) the rows converted to fp16 in registers against one fp16 query term
    (mfar_screen_queries_kernel, direct = 1).  Every error source except the MFMA's own fp32 accumulation (budgeted with
    (4 K + 66) u32) is emulated in numpy; the exact side is the bf16 contract's natural-order fp32 chain."""
    rng = np.random.default_rng(seed)
    D, Q = 3000, 12
    docs = ((rng.standard_normal((D, E)) * rng.lognormal(0, 1.5, (D, 1)) + rng.standard_normal(E) * 0.5) * scale).astype(np.float32)
    docs[::89] *= np.float32(1e-6)                       # values far below the fp16 normal range after scaling: clamped by (b)
    docs[7] = 0
    bits = _bf16_round_bits(docs)
    rows = _bf16_to_f32(bits)                           # what the index holds
    q = (rng.standard_normal((Q, E)) * rng.lognormal(0, 1, (Q, 1))).astype(np.float32)
    q[2, ::2] *= np.float32(1e-6)
    with O.chain("natural"):
        exact = O.c_scores(rows, q).astype(np.float64)
    K = float(E)
    dn = float(np.sqrt((rows.astype(np.float64) ** 2).sum(1).max())) * 1.0001
    sf = _pow2_scale(float(np.abs(rows).max()))
    d16 = _cv_bf16_to_f16(bits, sf).astype(np.float64)
    # the conversion is exact wherever the scaled value is a normal fp16 number, and off by at most 2^-14 (scaled) elsewhere
    scaled = rows.astype(np.float64) * sf
    assert np.abs(d16 - scaled).max() <= 2.0 ** -14
    assert np.array_equal(d16[np.abs(scaled) >= 2.0 ** -14], scaled[np.abs(scaled) >= 2.0 ** -14])
    worst = 0.0
    for i in range(Q):
        qn = float(np.sqrt((q[i].astype(np.float64) ** 2).sum())) * 1.0001
        # (a) two bf16 terms
        hi = _bf16_to_f32(_bf16_round_bits(q[i]))
        mid = _bf16_to_f32(_bf16_round_bits(q[i] - hi))
        approx = rows.astype(np.float64) @ (hi.astype(np.float64) + mid.astype(np.float64))
        eps_a = SLACK * ((1.01 * 2.0 ** -16 + (5.0 * K + 66.0) * U32) * qn * dn + 3.0e-36 * (np.sqrt(K) * (qn + dn) + K))
        err = np.abs(approx - exact[i]).max()
        assert err <= eps_a, ("two bf16 terms", i, err, eps_a)
        worst = max(worst, err / eps_a)
        # (b) converted docs x one fp16 term
        sq = _pow2_scale(float(np.abs(q[i]).max()))
        a16 = (q[i] * np.float32(sq)).astype(np.float16).astype(np.float64)
        approx = (d16 @ a16) / (sq * sf)
        c_rel = 1.02 * U16 + (4.0 * K + 66.0) * U32
        eps_b = SLACK * (c_rel * qn * dn + K * U32 * qn * dn + 2.0 ** -14 * 1.001 * np.sqrt(K) * 1.0001 * qn / sf +
                         U32 * np.sqrt(K) * 1.0001 * dn / sq)
        err = np.abs(approx - exact[i]).max()
        assert err <= eps_b, ("converted docs", i, err, eps_b)
        worst = max(worst, err / eps_b)
    assert worst < 0.6
