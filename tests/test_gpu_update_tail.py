"""The rank-64 update as the TAIL of the product launch that feeds it (UpdateTail, csrc/kernels.h; Engine::tail_for) against the same update as a launch of its own
(NMFAMD_NO_FUSED_TAIL=1): the same instructions on the same values, so factors and error values must be the same BITS.  Shapes: two resident images and one (config 2's),
a narrow column shard's (many K slices, W^T W in K slices), a rank below the padded 64, lengths that are no multiple of 32."""
import gc

import numpy as np
import pytest

import nmfgpu_amd as na

pytestmark = pytest.mark.gpu
F = np.asfortranarray


def _problem(m, n, r, seed):
    rng = np.random.default_rng(seed)
    V = F(rng.random((m, n), dtype=np.float32))
    W = F((1.0 - rng.random((m, r))).astype(np.float32))
    H = F((1.0 - rng.random((r, n))).astype(np.float32))
    return V, W, H


def _run(V, W, H, r, iters, monkeypatch, tail):
    if tail:
        monkeypatch.delenv("NMFAMD_NO_FUSED_TAIL", raising=False)
    else:
        monkeypatch.setenv("NMFAMD_NO_FUSED_TAIL", "1")
    gc.collect()                       # (a second engine alive on the device switches the tail off: Engine::tail_for)
    m, n = V.shape
    eng = na.Engine(m, n, r, "mu")
    eng.upload(V); eng.set_factors(W, H)
    errs = []
    for k in range(1, iters + 1):
        eng.iterate(1, first_iteration=k, error_every=3, last_iteration=iters)
        if k % 3 == 0 or k == iters:
            errs.append(eng.frobenius)
    Wg, Hg = eng.get_factors()
    geo = eng.geometry()
    del eng
    gc.collect()
    return Wg, Hg, errs, geo


@pytest.mark.parametrize("m,n,r", [
    (1000, 700, 64),           # two images, a handful of x-tiles
    (10000, 5000, 64),         # config 2: one resident image, y-tiled W^T V
    (10000, 625, 64),          # the shard of an 8-GPU run: 26 K slices of W^T V, W^T W in four K slices
    (10000, 1250, 64),
    (2999, 1001, 40),          # rank below the padded rank, ragged lengths
])
def test_update_tail_leaves_the_same_bits(m, n, r, monkeypatch):
    V, W, H = _problem(m, n, r, 7)
    iters = 7
    Wa, Ha, ea, geo_a = _run(V, W, H, r, iters, monkeypatch, tail=True)
    Wb, Hb, eb, geo_b = _run(V, W, H, r, iters, monkeypatch, tail=False)
    assert geo_a["fused_tail"] != 0, geo_a          # (the form under test really ran)
    assert geo_b["fused_tail"] == 0, geo_b
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    assert ea == eb and all(np.isfinite(ea))


def test_update_tail_is_off_while_two_engines_share_the_device(monkeypatch):
    monkeypatch.delenv("NMFAMD_NO_FUSED_TAIL", raising=False)
    V, W, H = _problem(1000, 700, 64, 3)
    gc.collect()
    a = na.Engine(1000, 700, 64, "mu"); a.upload(V); a.set_factors(W, H)
    b = na.Engine(1000, 700, 64, "mu"); b.upload(V); b.set_factors(W, H)
    a.iterate(3, first_iteration=1, error_every=0); b.iterate(3, first_iteration=1, error_every=0)
    assert a.geometry()["fused_tail"] == 0 and b.geometry()["fused_tail"] == 0
    Wa, Ha = a.get_factors(); Wb, Hb = b.get_factors()
    assert np.array_equal(Wa, Wb) and np.array_equal(Ha, Hb)
    del b
    gc.collect()
    a.iterate(3, first_iteration=4, error_every=0)
    assert a.geometry()["fused_tail"] != 0           # alone again
