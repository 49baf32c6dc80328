"""CPU tier: the oracle against its known answers, its float64 shadow and the committed fixtures."""
import os

import numpy as np
import pytest

from oracle import make_golden, philox
from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary, synthetic_params, synthetic_real


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (also SURVEY.md 8(c))."""
    kats = [((0, 0, 0, 0), (0, 0), "6627e8d5 e169c58d bc57ac4c 9b00dbd8"),
            ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), "408f276d 41c83b0e a20bc7c6 6d5451fd"),
            ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
             "d16cfe09 94fdcceb 5001e420 24126ea1")]
    for ctr, key, want in kats:
        w = philox.philox4x32_10(tuple(np.array([c], dtype=np.uint64) for c in ctr), key)
        assert " ".join("%08x" % int(x[0]) for x in w) == want


def test_uniform_contract():
    u = philox.uniform(64, 33, seed=7, stream_id=2, step=5)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    # global-row addressing: a shard with row0 reproduces the rows of the full matrix
    assert np.array_equal(philox.uniform(16, 33, 7, 2, 5, row0=32), u[32:48])
    # the four words of one block are four consecutive rows of one column
    w = philox.philox4x32_10((np.array([5]), np.array([3]), np.array([2]), np.array([5])), (7, 0))
    assert [int(x[0]) for x in w] == [int(x) for x in philox.block_words(16, 33, 7, 2, 5)[12:16, 5]]
    # streams, steps and seeds are independent planes
    assert not np.array_equal(u, philox.uniform(64, 33, 7, 3, 5))
    assert not np.array_equal(u, philox.uniform(64, 33, 7, 2, 6))
    assert not np.array_equal(u, philox.uniform(64, 33, 8, 2, 5))
    assert abs(float(u.mean()) - 0.5) < 0.03


def test_normal_moments():
    z = philox.normal(512, 128, seed=1, stream_id=5, step=0)
    assert np.all(np.isfinite(z))
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02


def test_fixtures_are_current(golden_dir, tmp_path, monkeypatch):
    """Regenerating the fixtures from the oracle reproduces the committed files exactly."""
    monkeypatch.setattr(make_golden, "GOLDEN_DIR", str(tmp_path))
    make_golden.main()
    # (ref_*.npz come from RUNNING reference code, oracle/make_ref_fixtures.py: tests/test_ref_fixtures.py)
    names = sorted(f for f in os.listdir(golden_dir) if f.endswith(".npz") and not f.startswith("ref_"))
    assert names == sorted(os.listdir(tmp_path))
    for name in names:
        a, b = np.load(os.path.join(golden_dir, name)), np.load(os.path.join(str(tmp_path), name))
        assert sorted(a.files) == sorted(b.files)
        for k in a.files:
            if a[k].dtype.kind == "f":
                assert np.allclose(a[k], b[k], rtol=1e-5, atol=1e-6, equal_nan=True), (name, k)   # BLAS order
            else:
                assert np.array_equal(a[k], b[k]), (name, k)


def test_float64_shadow():
    """float32 oracle vs the same code in float64: within the 1e-4 bar the kernels are held to."""
    nv, nh, B = 200, 120, 64
    W, b_h, b_v = synthetic_params(nv, nh, 1)
    v = synthetic_binary(B, nv, 2, p=0.3)
    W64, bh64, bv64, v64 = (x.astype(np.float64) for x in (W, b_h, b_v, v))
    _, _, _, c32, s32 = O.cd_step_fused(W, b_h, b_v, v, 0.01, 3, 0)
    _, _, _, c64, s64 = O.cd_step_fused(W64, bh64, bv64, v64, 0.01, 3, 0)
    assert np.array_equal(c32["u_h0"], c64["u_h0"])
    assert np.max(np.abs(c32["p_h0"] - c64["p_h0"])) < 1e-6
    if np.array_equal(c32["h_pos"], c64["h_pos"]) and np.array_equal(c32["v_neg"], c64["v_neg"]):
        for a, b in zip(s32, s64):
            assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < 1e-4


def test_cd_semantics():
    """The reading of rbm.py:119-134 the oracle encodes."""
    nv, nh, B = 30, 20, 16
    W, b_h, b_v = synthetic_params(nv, nh, 4)
    v = synthetic_real(B, nv, 5)
    Wn, bhn, bvn, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, 0.1, 9, 2)
    assert set(np.unique(ch["h_pos"])) <= {0.0, 1.0}               # h_pos enters as a binary sample
    assert set(np.unique(ch["v_neg"])) <= {0.0, 1.0}               # v_neg is a binary sample
    assert 0 < ch["h_neg"].min() and ch["h_neg"].max() < 1 and len(np.unique(ch["h_neg"])) > 2   # probabilities
    assert np.allclose(dW, v.T @ ch["h_pos"] - ch["v_neg"].T @ ch["h_neg"], atol=1e-5)
    assert np.allclose(Wn - W, 0.1 * dW, atol=1e-6)                # sums, scaled by lr only
    assert np.allclose(bvn - b_v, 0.1 * (v.sum(0) - ch["v_neg"].sum(0)), atol=1e-6)
    # reference_sequential: the W update equals the fused one (same chain 0), the biases use new chains
    Ws, bhs, bvs = O.cd_step_reference_sequential(W, b_h, b_v, v, 0.1, 9, 2)
    assert np.array_equal(Ws, Wn)
    assert not np.array_equal(bhs, bhn)
    # CD-k and persistent chains reduce to CD-1 at k = 1 / no chain
    _, _, _, ch2, _ = O.cd_step_fused(W, b_h, b_v, v, 0.1, 9, 2, k=1, v_chain=None)
    assert np.array_equal(ch2["v_neg"], ch["v_neg"])
    _, _, _, ch3, _ = O.cd_step_fused(W, b_h, b_v, v, 0.1, 9, 2, k=3)
    assert not np.array_equal(ch3["v_neg"], ch["v_neg"])


def test_batches_and_fit_loop():
    assert O.batch_slices(150, 64) == [(0, 64), (64, 128), (128, 150)]       # remainder last (rbm.py:110-111)
    assert O.batch_slices(128, 64) == [(0, 64), (64, 128)]
    nv, nh = 20, 12
    W, b_h, b_v = synthetic_params(nv, nh, 6)
    V = synthetic_binary(50, nv, 7, p=0.4)
    hps = {"batch_size": 16, "epochs": 2, "lr": 0.05}
    W1, bh1, bv1, scores, step = O.fit(W, b_h, b_v, V, hps, seed=3, with_score=True)
    assert step == 8 and len(scores) == 8 and all(np.isfinite(scores))
    # replaying by hand gives the same trajectory
    Wm, bhm, bvm = W, b_h, b_v
    s = 0
    for _ in range(2):
        for lo, hi in O.batch_slices(50, 16):
            Wm, bhm, bvm, _, _ = O.cd_step_fused(Wm, bhm, bvm, V[lo:hi], 0.05, 3, s)
            s += 1
    assert np.array_equal(W1, Wm) and np.array_equal(bh1, bhm) and np.array_equal(bv1, bvm)


def test_free_energy_forms():
    nv, nh = 24, 16
    W, b_h, b_v = synthetic_params(nv, nh, 8)
    v = synthetic_real(9, nv, 9)
    assert np.allclose(O.free_energy(v, W, b_h, b_v, True), O.free_energy(v, W, b_h, b_v, False), rtol=1e-6)
    big = v * 1e5
    with np.errstate(all="ignore"):
        assert np.all(np.isfinite(O.free_energy(big, W, b_h, b_v, True)))


def test_dbn_control_flow():
    hps = {"batch_size": 8, "epochs": 1, "lr": 0.05}
    with pytest.raises(ValueError):
        O.dbn_fit([], np.zeros((4, 4), np.float32))
    V = synthetic_binary(20, 16, 10, p=0.5)
    layers = [O.OracleLayer(*synthetic_params(16, 12, 11), hps, 1), O.OracleLayer(*synthetic_params(12, 6, 12), hps, 2)]
    top = O.dbn_fit(layers, V)
    assert top.shape == (20, 6) and layers[0].step == 3 and layers[1].step == 3
    feat = O.dbn_transform(layers, V)
    assert feat.shape == (20, 6) and set(np.unique(feat)) <= {0.0, 1.0}
    back = O.dbn_inv_transform(layers, feat)
    assert back.shape == (20, 16)


def test_second_opinion_torch():
    """The oracle's arithmetic against an independent implementation (torch CPU, float64): half steps,
    free energy and the CD statistics -- a typo in the restatement would have to exist twice."""
    import torch
    nv, nh, B = 50, 34, 20
    W, b_h, b_v = synthetic_params(nv, nh, 13)
    v = synthetic_real(B, nv, 14)
    tW, tbh, tbv, tv = (torch.from_numpy(x.astype(np.float64)) for x in (W, b_h, b_v, v))
    assert np.allclose(O.hidden_prob(v, W, b_h), torch.sigmoid(tv @ tW + tbh).numpy(), atol=2e-6)
    h = synthetic_binary(B, nh, 15, p=0.5)
    th = torch.from_numpy(h.astype(np.float64))
    assert np.allclose(O.visible_prob(h, W, b_v), torch.sigmoid(th @ tW.T + tbv).numpy(), atol=2e-6)
    F = -(tv @ tbv + torch.nn.functional.softplus(tv @ tW + tbh).sum(-1))
    assert np.allclose(O.free_energy(v, W, b_h, b_v), F.numpy(), rtol=1e-5)
    _, _, _, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, 0.1, 1, 0)
    tp = lambda k: torch.from_numpy(ch[k].astype(np.float64))
    assert np.allclose(dW, (tv.T @ tp("h_pos") - tp("v_neg").T @ tp("h_neg")).numpy(), atol=1e-4)
    assert np.allclose(dbv, (tv.sum(0) - tp("v_neg").sum(0)).numpy(), atol=1e-4)
    # bf16 rounding helper against torch's own conversion
    x = np.linspace(-3, 3, 1001, dtype=np.float32) * np.float32(1.2345)
    assert np.array_equal(O.bf16_round(x), torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy())


def test_bf16_split3_is_exact():
    """x == hi + mid + lo in float32 for every finite float32, each piece a bf16 value (x3 path)."""
    g = np.random.default_rng(7)
    x = np.concatenate([
        g.standard_normal(200000).astype(np.float32) * np.float32(0.05),
        g.random(200000).astype(np.float32),                              # probabilities
        (g.standard_normal(1000) * 1e-30).astype(np.float32), (g.standard_normal(1000) * 1e30).astype(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, 0.1, 1.0 - 2.0 ** -24, 2.0 ** -126, 3.0 * 2.0 ** -126, np.float32(1) / 3], np.float32),
        np.frombuffer(g.bytes(4 * 100000), dtype=np.uint32).astype(np.uint32).view(np.float32),
    ])
    x = x[np.isfinite(x) & (np.abs(x) < 3e38)]
    x = x[(np.abs(x) >= 2.0 ** -100) | (x == 0)]          # pieces of values this small underflow bf16's own range
    hi, mid, lo = O.bf16_split3(x)
    for p in (hi, mid, lo):
        assert np.array_equal(p, O.bf16_round(p)), "a piece must be exactly a bf16 value"
    assert np.array_equal((hi + mid) + lo, x) and np.array_equal(hi.astype(np.float64) + mid + lo, x.astype(np.float64))
    # 0/1 samples are a single piece
    s = (g.random(1000) < 0.3).astype(np.float32)
    hi, mid, lo = O.bf16_split3(s)
    assert np.array_equal(hi, s) and not mid.any() and not lo.any()
