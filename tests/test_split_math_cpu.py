"""The arithmetic of csrc/conv_split.hip restated in numpy (no GPU): fp32 operands as pairs of fp16 values,
a 2^k = h1 + 2^-11 h2, w 2^e = w1 + w2, a w = 2^-(e+k) (h1 w1 + h1 w2 + h2 (w1 2^-11)) -- the 3x3 kernels in the form
h1 w1 + 2^-11 (h1 (w2 2^11) + h2 w1), the folded up-conv with w1 2^-11 formed per weight set.  Checks the representation
error of the scheme against fp64 and the exponent rule of the host mirror (ops.act_exponent_for)."""
import math

import numpy as np
import pytest

import kbnet_amd as kb


def split_dot(a, w, k, folded=False):
    """sum_c a[c] w[c] the way the kernel forms it (products exact, accumulation here in fp64: the test isolates the
    representation error; the GPU tests cover the fp32 accumulation)."""
    wmax = np.abs(w).max()
    e = 13 - math.frexp(float(wmax))[1] if wmax > 0 else 0
    ap = (a * np.float32(2.0 ** k)).astype(np.float32)
    h1 = ap.astype(np.float16)
    h1 = np.where(np.abs(h1.astype(np.float32)) < 2.0 ** -14, np.float16(0), h1)      # subnormal results are flushed
    h2 = ((ap - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    ws = (w * np.float32(2.0 ** e)).astype(np.float32)
    w1 = ws.astype(np.float16)
    flush = lambda x: np.where(np.abs(x.astype(np.float32)) < 2.0 ** -14, np.float16(0), x)   # the matrix core drops fp16 subnormals
    if folded:   # the folded up-conv: one accumulator, w2 unscaled, w1 2^-11 formed in registers
        w2 = flush((ws - w1.astype(np.float32)).astype(np.float16))
        w1s = flush((w1.astype(np.float32) * np.float32(2.0 ** -11)).astype(np.float16))
        h1, h2, w1, w2, w1s = (x.astype(np.float64) for x in (h1, h2, w1, w2, w1s))
        return float((h1 * w1 + h1 * w2 + h2 * w1s).sum() * 2.0 ** -(e + k))
    # the 3x3 kernels: the weight residual scaled by 2^11 like the activations', both small terms in one accumulator
    w2 = flush(((ws - w1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16))
    h1, h2, w1, w2 = (x.astype(np.float64) for x in (h1, h2, w1, w2))
    return float(((h1 * w1).sum() + (h1 * w2 + h2 * w1).sum() * 2.0 ** -11) * 2.0 ** -(e + k))


@pytest.mark.parametrize("folded", [False, True])
@pytest.mark.parametrize("amag", [1.0, 1e-4, 3e5])
def test_split_product_representation_error(amag, folded):
    rng = np.random.default_rng(3)
    K, trials = 2304, 64
    errs, scale = [], []
    for _ in range(trials):
        a = rng.standard_normal(K).astype(np.float32)
        a = (np.where(a < 0, 0.2 * a, a) * amag).astype(np.float32)
        w = (rng.standard_normal(K) / math.sqrt(K)).astype(np.float32)
        k = kb.ops.act_exponent_for(float(np.abs(a).max()))
        ref = float((a.astype(np.float64) * w.astype(np.float64)).sum())
        errs.append(split_dot(a, w, k, folded) - ref)
        scale.append(ref)
    rms = math.sqrt(np.mean(np.square(errs))) / math.sqrt(np.mean(np.square(scale)))
    assert rms < 4e-7, rms        # fp32 rounding of the operands alone would leave ~4e-8; an fmaf chain of this length ~8e-7


def test_act_exponent_rule():
    """ops.act_exponent_for restates the device rule (sp_act_scale, csrc/conv_split.hip): the frame's maximum lands in
    [2^14, 2^15) of the fp16 window -- below the overflow at 65504 even after h1's rounding -- and degenerate maxima
    give finite scales."""
    for amax in (1e-30, 1e-6, 0.0039, 1.0, 255.9, 256.0, 300.0, 65504.0, 4.0e6, 3.0e30):
        k = kb.ops.act_exponent_for(amax)
        assert 2.0 ** 14 <= amax * 2.0 ** k < 2.0 ** 15 or k in (-100, 100)
        assert float(np.float16(np.float32(amax) * np.float32(2.0 ** k))) <= 32768.0
    assert kb.ops.act_exponent_for(0.0) == 100 and kb.ops.act_exponent_for(1e-45) == 100
    assert kb.ops.act_exponent_for(float("inf")) == -100 and kb.ops.act_exponent_for(float("nan")) == -100


def test_window_slack_costs_nothing():
    """Why the window may sit anywhere within ~2^16 of the data: an activation below the full-precision range (h1 flushed)
    is carried by the scaled residual alone, with an ABSOLUTE error below 2^-40 of the window's top -- a frame whose
    maximum is 2^-16 of the top still has every value within 2^-24 of its maximum."""
    rng = np.random.default_rng(5)
    a = (rng.standard_normal(4096) * 2.0 ** -7).astype(np.float32)      # max |a| 2^k ~ 2^-5: 20 binades below the top
    k = 0
    ap = a * np.float32(2.0 ** k)
    h1 = ap.astype(np.float16)
    h1 = np.where(np.abs(h1.astype(np.float32)) < 2.0 ** -14, np.float16(0), h1)     # the kernels flush subnormal halves
    h2 = ((ap - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    h2 = np.where(np.abs(h2.astype(np.float32)) < 2.0 ** -14, np.float16(0), h2)
    back = h1.astype(np.float64) + h2.astype(np.float64) * 2.0 ** -11
    err = np.abs(back - ap.astype(np.float64)).max()
    assert err <= 2.0 ** -25 and err / np.abs(ap).max() < 2.0 ** -19


def _bound_scale(bound):
    """sp_scale_of_bound (csrc/conv_split.hip): 2^k with bound 2^k in [2^14, 2^15), finite for 0 / Inf / NaN."""
    bits = int(np.float32(bound).view(np.uint32))
    k = 14 + 127 - ((bits >> 23) & 255)
    return 2.0 ** max(-100, min(100, k))


def test_pair_tensor_window_from_a_bound():
    """The producer-written split format (include/kbnet_hip.h, "PAIR tensors") fixes its per-frame 2^k from a BOUND of the
    output -- input maxima x weight norms -- not from the output: the window never overflows whatever the overshoot, and
    an overshoot of up to 2^13 (KITTI layers: 2^5-2^7) leaves every value within 2^-22 of itself plus 2^-25 of a window
    unit (the flushed fp16 subnormals): what the consumers multiply is the fp32 value to fp32 accuracy."""
    rng = np.random.default_rng(11)
    w = rng.standard_normal((64, 288)).astype(np.float32) / 17.0
    x = np.abs(rng.standard_normal((288, 500))).astype(np.float32) * 3.0
    out = w.astype(np.float64) @ x.astype(np.float64)
    bound = float(np.abs(x).max()) * float(np.abs(w).sum(axis=1).max())
    assert np.abs(out).max() <= bound
    for slack in (1.0, 2.0 ** 7, 2.0 ** 13):
        scale = _bound_scale(bound * slack)
        t = (out * scale).astype(np.float32)
        assert np.abs(t).max() < 2.0 ** 15 / slack * 1.0001
        h1 = t.astype(np.float16)
        h1 = np.where(np.abs(h1.astype(np.float32)) < 2.0 ** -14, np.float16(0), h1)
        h2 = ((t - h1.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        h2 = np.where(np.abs(h2.astype(np.float32)) < 2.0 ** -14, np.float16(0), h2)
        back = h1.astype(np.float64) + h2.astype(np.float64) / 2048.0
        err = np.abs(back - t.astype(np.float64))
        assert (err <= np.abs(t) * 2.0 ** -22 + 2.0 ** -25).all()
        # relative to the frame's maximum: 2^-25 window units / (2^14 / slack): 2^-26 at the largest overshoot tested
        assert err.max() / np.abs(t).max() <= 2.0 ** -22
    assert _bound_scale(0.0) == 2.0 ** 100 and _bound_scale(float("inf")) == 2.0 ** -100 and _bound_scale(float("nan")) == 2.0 ** -100


def test_default_window_limits():
    """What the ABI default exponent (-6) covers: full precision from 2^-8 to fp16's maximum x 64."""
    for a in (2.0 ** -8, 1.0, 4.0e6):
        ap = np.float32(a) * np.float32(2.0 ** -6)
        assert 2.0 ** -14 <= float(ap) <= 65504.0
