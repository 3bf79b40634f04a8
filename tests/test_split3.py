"""The arithmetic claim behind the split-operand product (nmfgpu_amd/csrc/split3.h, kernels_x3.hip), checked on the CPU:
an fp32 value is the EXACT sum of three bf16 values obtained by round-to-nearest cuts of the successive residuals, and the
six cross products the kernel keeps reproduce a product to ~2^-23.  (The GPU tests check the kernel itself.)"""
import numpy as np


def round_bf16(a):
    """fp32 -> nearest bf16 (ties to even), returned as fp32."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(a))


def split3(v):
    v = np.asarray(v, dtype=np.float32)
    hi = round_bf16(v)
    r1 = (v - hi).astype(np.float32)
    mid = round_bf16(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = round_bf16(r2)
    return hi, mid, lo, r2


def test_three_bf16_terms_reproduce_every_fp32_bit():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(200000) * np.exp(rng.uniform(-60, 60, 200000))).astype(np.float32)
    v = np.concatenate([v, np.float32([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 2.0 ** -100])])
    hi, mid, lo, r2 = split3(v)
    # the residual subtractions are exact (each cut removes the leading eight significand bits) ...
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + r2.astype(np.float64)), v.astype(np.float64))
    # ... and the last residual already is a bf16 value
    assert np.array_equal(lo, r2)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), v.astype(np.float64))
    # every term is a bf16 value
    for t in (hi, mid, lo):
        assert np.array_equal(round_bf16(t), t)
    # magnitudes fall by 2^-8 per term (round to nearest: the residual is at most half an ulp of the cut)
    nz = v != 0
    assert (np.abs(mid[nz]) <= np.abs(v[nz]) * 2.0 ** -8).all() and (np.abs(lo[nz]) <= np.abs(v[nz]) * 2.0 ** -16).all()


def test_six_cross_terms_give_the_product_to_fp32_accuracy():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    a1, a2, a3, _ = (x.astype(np.float64) for x in split3(a))
    b1, b2, b3, _ = (x.astype(np.float64) for x in split3(b))
    kept = a1 * b1 + (a1 * b2 + a2 * b1) + (a2 * b2 + a1 * b3 + a3 * b1)
    exact = a.astype(np.float64) * b.astype(np.float64)
    dropped = a2 * b3 + a3 * b2 + a3 * b3
    assert np.allclose(kept + dropped, exact, rtol=0, atol=0)             # nothing else is missing
    assert (np.abs(kept - exact) <= 2.0 ** -22 * np.abs(exact) + 1e-300).all()
    # every kept term is exact in fp32: a product of two 8-bit significands has at most 16 bits
    for x, y in ((a1, b1), (a1, b2), (a2, b1), (a2, b2), (a1, b3), (a3, b1)):
        p = x * y
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)
