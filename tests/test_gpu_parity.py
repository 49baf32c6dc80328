"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bars (north_star): uniforms bit-exact; probabilities / weight updates within 1e-4 (relative to
max(1, |ref|) for batch sums); Bernoulli samples equal to (u < p_gpu) exactly, and equal to the
oracle's samples except where |u - p| < 1e-5 (the GEMM rounding band).  Downstream stages are
checked teacher-forced: the oracle is fed the GPU's own samples of the previous stage.
"""
import os

import numpy as np
import pytest
import torch

from oracle import philox
from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary, synthetic_params, synthetic_real

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star tolerance, fp32
FLIP_BAND = 1e-5    # |u - p| below which a sample may legitimately differ


def _engine(W, b_h, b_v, device):
    from keras_unsupervised_amd.ebm.engine import DeviceRBM
    return DeviceRBM(W, b_h, b_v, device)


@pytest.fixture
def ctx_option(gpu_device):
    """Set experiment knobs on the device's context for one test (kurbm_ctx_set_option); restored afterwards."""
    from keras_unsupervised_amd._lib import Context
    ctx = Context.get(gpu_device.index)
    touched = {}

    def set_option(name, value, default):
        touched[name] = default
        ctx.set_option(name, value)

    yield set_option
    for name, default in touched.items():
        ctx.set_option(name, default)


def _dm(x, device):
    from keras_unsupervised_amd.ebm.engine import DeviceMatrix
    return DeviceMatrix.from_host(x, device)


def rel_err(a, ref):
    return np.max(np.abs(a.astype(np.float64) - ref.astype(np.float64)) / np.maximum(1.0, np.abs(ref.astype(np.float64))))


def check_half_step(out, p_ref, u_ref, s_ref):
    """Shared assertions for one Bernoulli half step."""
    p, u, s = out["prob"].to_numpy(), out["u"].to_numpy(), out["sample"].to_numpy()
    assert np.array_equal(u.view(np.uint32), u_ref.view(np.uint32)), "uniforms must be bit-exact"
    assert np.max(np.abs(p - p_ref)) <= TOL
    assert np.array_equal(s, (u < p).astype(np.float32)), "sample must be exactly (u < p) of the GPU's own p"
    diff = s != s_ref
    if diff.any():
        assert np.all(np.abs(u_ref[diff] - p_ref[diff]) < FLIP_BAND), "sample differs outside the rounding band"
    return int(diff.sum())


# ---------------------------------------------------------------------------------------
def test_philox_uniform_bit_exact(gpu_device, golden_dir):
    W, b_h, b_v = synthetic_params(8, 8, 1)
    e = _engine(W, b_h, b_v, gpu_device)
    g = np.load(os.path.join(golden_dir, "philox.npz"))
    for s in range(4):
        u = e.philox_uniform(8, 8, 0, s, 0).to_numpy()
        assert np.array_equal(u.view(np.uint32), g["uniforms"][s].view(np.uint32))
    u = e.philox_uniform(8, 8, 42, 3, 7, row0=8).to_numpy()
    assert np.array_equal(u.view(np.uint32), g["uniforms_seed42_row0_8"].view(np.uint32))
    # ragged shape, 64-bit seed, large row offset: against the live oracle
    seed = (0xDEADBEEF << 32) | 0x12345678
    u = e.philox_uniform(37, 101, seed, 0x80000005, 0xFFFFFFF0, row0=(1 << 33) + 4).to_numpy()
    ref = philox.uniform(37, 101, seed, 0x80000005, 0xFFFFFFF0, row0=(1 << 33) + 4)
    assert np.array_equal(u.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("B", [8, 64])
def test_half_steps_config1_golden(gpu_device, golden_dir, B):
    """784 x 256 (config 1 shape): both half steps against the golden fixture and the live oracle."""
    W, b_h, b_v = synthetic_params(784, 256, seed=11)
    v = synthetic_binary(B, 784, seed=12)
    e = _engine(W, b_h, b_v, gpu_device)
    g = np.load(os.path.join(golden_dir, "half_step_B%d.npz" % B))
    sub = int(g["row_stride"])
    vd = _dm(v, gpu_device)
    out = e.half_step("vh", vd, B, 0, 0, 1, 42, 0, 5, want_prob=True, want_u=True)
    rng = O.Rng(42, 5)
    p_ref, u_ref, h_ref = O.sample_hidden(v, W, b_h, rng, 0)
    check_half_step(out, p_ref, u_ref, h_ref)
    assert np.array_equal(out["u"].to_numpy()[::sub].view(np.uint32), g["u_h"].view(np.uint32))
    assert np.max(np.abs(out["prob"].to_numpy()[::sub] - g["p_h"])) <= TOL
    h_gold = np.unpackbits(g["h"], axis=1)[:, :256].astype(np.float32)
    assert np.array_equal(out["sample"].to_numpy(), h_gold)
    # h -> v, teacher-forced with the GPU's h
    h = out["sample"].to_numpy()
    out2 = e.half_step("hv", out["sample"], B, 0, 0, 1, 42, 1, 5, want_prob=True, want_u=True)
    p_ref, u_ref, v_ref = O.sample_visible(h, W, b_v, rng, 1)
    check_half_step(out2, p_ref, u_ref, v_ref)
    v_gold = np.unpackbits(g["v1"], axis=1)[:, :784].astype(np.float32)
    assert np.array_equal(out2["sample"].to_numpy(), v_gold)


@pytest.mark.parametrize("shape", [(1, 5, 3), (3, 17, 9), (130, 100, 70), (257, 113, 129), (100, 784, 128),
                                   (260, 224, 240)])
def test_half_steps_ragged_shapes(gpu_device, shape):
    """Edge shapes: single row, sizes that are not multiples of the tiles or of 4, real-valued input."""
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=100 + B)
    v = synthetic_real(B, nv, seed=200 + B)
    e = _engine(W, b_h, b_v, gpu_device)
    rng = O.Rng(9, 2, row0=8)
    out = e.half_step("vh", _dm(v, gpu_device), B, 0, 0, 1, 9, 4, 2, row0=8, want_prob=True, want_u=True)
    p_ref, u_ref, h_ref = O.sample_hidden(v, W, b_h, rng, 4)
    check_half_step(out, p_ref, u_ref, h_ref)
    h = synthetic_real(B, nh, seed=300 + B)
    out = e.half_step("hv", _dm(h, gpu_device), B, 0, 0, 1, 9, 5, 2, row0=8, want_prob=True, want_u=True)
    p_ref, u_ref, v_ref = O.sample_visible(h, W, b_v, rng, 5)
    check_half_step(out, p_ref, u_ref, v_ref)
    # padding columns of the outputs stay untouched (zero)
    t = out["sample"].t
    if t.shape[1] > nv:
        assert float(t[:, nv:].abs().max().item()) == 0.0


def test_half_step_row_window_and_prob_only(gpu_device):
    """A batch is a row window of the resident data matrix; NOISE_NONE writes probabilities only."""
    W, b_h, b_v = synthetic_params(96, 80, seed=5)
    V = synthetic_real(50, 96, seed=6)
    e = _engine(W, b_h, b_v, gpu_device)
    out = e.half_step("vh", _dm(V, gpu_device), 20, 12, 0, 0, 0, 0, 0, want_sample=False, want_prob=True)
    ref = O.hidden_prob(V[12:32], W, b_h)
    assert np.max(np.abs(out["prob"].to_numpy() - ref)) <= TOL


def test_gaussian_mode_half_steps(gpu_device):
    """relu-threshold hidden units and N(loc, 1) visible units (rbm.py:57-67)."""
    W, b_h, b_v = synthetic_params(72, 40, seed=8)
    W = W * 20.0  # spread the pre-activations so relu(x) covers (0, 1) and beyond
    v = synthetic_real(33, 72, seed=9)
    e = _engine(W, b_h, b_v, gpu_device)
    rng = O.Rng(3, 1)
    out = e.half_step("vh", _dm(v, gpu_device), 33, 0, 1, 1, 3, 0, 1, want_prob=True, want_u=True)
    p_ref, u_ref, h_ref = O.sample_hidden(v, W, b_h, rng, 0, O.MODE_VISIBLE_GAUSSIAN)
    check_half_step(out, p_ref, u_ref, h_ref)
    h = out["sample"].to_numpy()
    out = e.half_step("hv", out["sample"], 33, 0, 2, 2, 3, 1, 1, want_prob=True)
    loc, z, v1 = O.sample_visible(h, W, b_v, rng, 1, O.MODE_VISIBLE_GAUSSIAN)
    assert np.max(np.abs(out["prob"].to_numpy() - loc)) <= TOL
    assert np.max(np.abs(out["sample"].to_numpy() - v1)) <= TOL    # loc + Box-Muller(u_a, u_b), both sides in fp32
    assert abs(float(z.mean())) < 0.1


def test_free_energy(gpu_device, golden_dir):
    nv, nh = 64, 48
    W, b_h, b_v = synthetic_params(nv, nh, seed=51)
    v = synthetic_real(6, nv, seed=52)
    v[5] *= 4000.0
    W2 = W.copy()
    W2[:, 0] = 0.05
    e = _engine(W2, b_h, b_v, gpu_device)
    F = e.free_energy(_dm(v, gpu_device), 6).cpu().numpy()
    g = np.load(os.path.join(golden_dir, "free_energy.npz"))
    assert np.all(np.isfinite(F))
    assert rel_err(F, g["F_stable"]) <= TOL
    assert np.isinf(g["F_naive"][5])            # the reference's literal softplus overflows here
    # a bigger ragged case against the float64 oracle
    W, b_h, b_v = synthetic_params(300, 200, seed=53)
    v = synthetic_real(150, 300, seed=54)
    F = _engine(W, b_h, b_v, gpu_device).free_energy(_dm(v, gpu_device), 150).cpu().numpy()
    ref = O.free_energy(v.astype(np.float64), W.astype(np.float64), b_h.astype(np.float64), b_v.astype(np.float64))
    assert rel_err(F, ref) <= TOL


# ---------------------------------------------------------------------------------------
def _gpu_cd_delta(e, vd, rows, lr, seed, step, **kw):
    e.cd_step(vd, rows, 0, lr, seed, step, apply=False, emit_delta=True, **kw)
    torch.cuda.synchronize()
    return e.delta_buffer().cpu().numpy().copy()


def _split(delta, nv, nh):
    return delta[: nv * nh].reshape(nv, nh), delta[nv * nh: nv * nh + nh], delta[nv * nh + nh:]


def test_cd_step_small_golden(gpu_device, golden_dir):
    """One fused CD-1 update, 64 x 48, B = 24: delta and updated parameters vs golden + oracle."""
    nv, nh, B = 64, 48, 24
    W, b_h, b_v = synthetic_params(nv, nh, seed=21)
    v = synthetic_binary(B, nv, seed=22, p=0.3)
    g = np.load(os.path.join(golden_dir, "cd_step_small.npz"))
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    dW, dbh, dbv = _split(_gpu_cd_delta(e, vd, B, 0.01, 7, 3), nv, nh)
    assert rel_err(dW, g["dW"]) <= TOL
    assert rel_err(dbh, g["dbh"]) <= TOL
    assert np.array_equal(dbv, g["dbv"])          # integer-valued: exact
    # in-place apply
    e.cd_step(vd, B, 0, 0.01, 7, 3)
    Wn, bhn, bvn = e.get_weights()
    assert np.max(np.abs(Wn - g["W_fused"])) <= TOL
    assert np.max(np.abs(bhn - g["bh_fused"])) <= TOL
    assert np.max(np.abs(bvn - g["bv_fused"])) <= TOL
    # the reference's three sequential chains (rbm.py:214-216)
    e2 = _engine(W, b_h, b_v, gpu_device)
    for chain, which in ((0, 1), (1, 2), (2, 4)):
        e2.cd_step(vd, B, 0, 0.01, 7, 3, chain=chain, which=which)
    Wn, bhn, bvn = e2.get_weights()
    assert np.max(np.abs(Wn - g["W_seq"])) <= TOL
    assert np.max(np.abs(bhn - g["bh_seq"])) <= TOL
    assert np.max(np.abs(bvn - g["bv_seq"])) <= TOL


def test_cd_step_config1_golden(gpu_device, golden_dir):
    W, b_h, b_v = synthetic_params(784, 256, seed=11)
    v = synthetic_binary(64, 784, seed=12)
    g = np.load(os.path.join(golden_dir, "cd_step_config1.npz"))
    e = _engine(W, b_h, b_v, gpu_device)
    dW, dbh, dbv = _split(_gpu_cd_delta(e, _dm(v, gpu_device), 64, 1e-3, 42, 0), 784, 256)
    assert rel_err(dW[::49, ::16], g["dW_sub"]) <= TOL
    assert abs(dW.astype(np.float64).sum() - float(g["dW_sum"])) <= TOL * float(g["dW_abs_sum"])
    assert rel_err(dbh, g["dbh"]) <= TOL
    assert np.array_equal(dbv, g["dbv"])


def test_small_step_config1_golden(gpu_device, golden_dir):
    """kurbm_cd_step_small -- the whole CD-1 update in ONE launch (csrc/kurbm_small.hip) -- at BASELINE config 1's shape against the
    golden statistics: the applied update (W_new - W) / lr is the golden dW (fp32 rounding of W_new: ~4e-6 of an lr-sized step)."""
    W, b_h, b_v = synthetic_params(784, 256, seed=11)
    v = synthetic_binary(64, 784, seed=12)
    g = np.load(os.path.join(golden_dir, "cd_step_config1.npz"))
    e = _engine(W, b_h, b_v, gpu_device)
    e.cd_step(_dm(v, gpu_device), 64, 0, 1e-3, 42, 0, compute="small")
    Wn, bhn, bvn = e.get_weights()
    assert rel_err((Wn - W)[::49, ::16] / 1e-3, g["dW_sub"]) <= TOL
    assert rel_err((bhn - b_h) / 1e-3, g["dbh"]) <= TOL and rel_err((bvn - b_v) / 1e-3, g["dbv"]) <= TOL


@pytest.mark.parametrize("cfg", [dict(B=64, nv=784, nh=256), dict(B=128, nv=784, nh=128), dict(B=16, nv=784, nh=128), dict(B=50, nv=70, nh=90),
                                 dict(B=37, nv=100, nh=33, gauss=True), dict(B=128, nv=784, nh=128, gauss=True), dict(B=200, nv=300, nh=520), dict(B=5, nv=3, nh=2)])
@pytest.mark.parametrize("local", [1, 0])
def test_small_step_vs_oracle(gpu_device, ctx_option, cfg, local):
    """The one-launch step against the oracle's fused CD-1 step (same Philox counters: the draws are the oracle's), the parameters
    after TWO steps at the 1e-4 bar (the second step runs on the first one's weights and on the barrier state it left); ragged
    shapes, Gaussian visibles; `which` restricts the update as kurbm_cd_step's does; the epoch call is its step loop, bit for bit;
    and the five-launch fp32 path gives the same parameters to fp32 rounding.  Both schedules: phases 1-3 inside one XCD each
    (KURBM_SMALL_LOCAL=1, the default) and over the whole grid."""
    ctx_option("KURBM_SMALL_LOCAL", local, 1)
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg.get("gauss") else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=2600 + B)
    v = synthetic_real(2 * B, nv, seed=2601 + B) if cfg.get("gauss") else synthetic_binary(2 * B, nv, seed=2601 + B, p=0.3)
    lr = 1e-3
    ref = (W, b_h, b_v)
    for step in range(2):
        Wr, bhr, bvr, _, _ = O.cd_step_fused(*ref, v[step * B:(step + 1) * B], lr, 9, step, mode=mode)
        ref = (Wr, bhr, bvr)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    for step in range(2):
        e.cd_step(vd, B, step * B, lr, 9, step, mode=mode, compute="small")
    got = e.get_weights()
    for a, b in zip(got, ref):
        assert np.max(np.abs(a - b)) <= TOL
    e5 = _engine(W, b_h, b_v, gpu_device)
    for step in range(2):
        e5.cd_step(vd, B, step * B, lr, 9, step, mode=mode, compute="fp32")
    for a, b in zip(got, e5.get_weights()):
        assert np.max(np.abs(a - b)) <= 1e-5
    # one call for the epoch = the step loop
    e2 = _engine(W, b_h, b_v, gpu_device)
    assert e2.cd_epoch(vd, 2 * B, B, lr, 9, 0, mode=mode, compute="small") == 2
    for a, b in zip(got, e2.get_weights()):
        assert np.array_equal(a, b)
    # `which`: only the named parameters move, and they move as in the full step's first update
    from keras_unsupervised_amd import _lib
    full = _engine(W, b_h, b_v, gpu_device)
    full.cd_step(vd, B, 0, lr, 9, 0, mode=mode, compute="small")
    for which, idx in ((_lib.WHICH_W, 0), (_lib.WHICH_BH, 1), (_lib.WHICH_BV, 2)):
        ew = _engine(W, b_h, b_v, gpu_device)
        ew.cd_step(vd, B, 0, lr, 9, 0, mode=mode, which=which, compute="small")
        for j, (a, b0, f) in enumerate(zip(ew.get_weights(), (W, b_h, b_v), full.get_weights())):
            assert np.array_equal(a, f if j == idx else b0), (which, j)
    e.check_status()


@pytest.mark.parametrize("shape", [(128, 784, 128), (64, 784, 256), (40, 100, 72)])
def test_small_step_local_schedule_is_deterministic(gpu_device, shape):
    """600 one-launch steps (the XCD-local schedule: planes and barrier words handed over through an XCD's L2) twice from the same
    weights, other kernels of odd grid sizes in between (a launch starts on the XCD where the last one stopped): bit-identical
    parameters, finite, no status bit -- a stale read of a plane or of a barrier word would show here."""
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=77)
    v = synthetic_binary(8 * B, nv, seed=78, p=0.3)
    vd = _dm(v, gpu_device)
    runs = []
    for _ in range(2):
        e = _engine(W, b_h, b_v, gpu_device)
        for step in range(600):
            if step % 37 == 0:      # other grids in between: the next launch starts on whichever XCD the dispatcher stands at
                junk = torch.empty(1000 * (1 + step % 7) + 13, device=gpu_device).fill_(1.0)
            e.cd_step(vd, B, (step % 8) * B, 1e-2 / B, 5, step, compute="small")
        e.check_status()
        runs.append(e.get_weights())
    for a, b in zip(*runs):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    assert np.max(np.abs(runs[0][0] - W)) > 1e-3      # (it did train)


@pytest.mark.parametrize("cfg", [dict(B=50, nv=70, nh=90, k=1), dict(B=133, nv=200, nh=120, k=3),
                                 dict(B=64, nv=784, nh=256, k=1), dict(B=40, nv=48, nh=64, k=2, pcd=True),
                                 dict(B=30, nv=52, nh=44, k=1, gauss=True)])
def test_cd_step_vs_oracle(gpu_device, cfg):
    """CD-k, persistent chains and Gaussian mode against the live oracle; the run is deterministic."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    gauss = cfg.get("gauss", False)
    mode = O.MODE_VISIBLE_GAUSSIAN if gauss else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=60 + B)
    v = synthetic_real(B, nv, seed=61 + B) if gauss else synthetic_binary(B, nv, seed=61 + B, p=0.3)
    chain0 = synthetic_binary(B, nv, seed=62 + B, p=0.5) if cfg.get("pcd") else None
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    cd = _dm(chain0, gpu_device) if chain0 is not None else None
    d1 = _gpu_cd_delta(e, vd, B, 0.05, 77, 9, k=k, mode=mode, v_chain=cd)
    _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.05, 77, 9, k=k, mode=mode, v_chain=chain0)
    dW, dbh, dbv = _split(d1, nv, nh)
    if gauss:   # real-valued sums: reference in float64 (numpy's own fp32 matmul / sum error is of the order of the bar)
        dW_ref, dbh_ref, dbv_ref = O.cd_statistics({k_: ch[k_].astype(np.float64) for k_ in ("v_pos", "h_pos", "v_neg", "h_neg")})
    assert rel_err(dW, dW_ref) <= TOL
    assert rel_err(dbh, dbh_ref) <= TOL
    assert rel_err(dbv, dbv_ref) <= TOL
    if cd is not None:   # the chain buffer now holds v_neg
        assert np.array_equal(cd.to_numpy(), ch["v_neg"])
        cd = _dm(chain0, gpu_device)
    d2 = _gpu_cd_delta(e, vd, B, 0.05, 77, 9, k=k, mode=mode, v_chain=cd)
    assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32)), "same inputs, same counters -> same bits"


def test_full_size_step_properties(gpu_device):
    """Config 2 (784 x 1024, B = 4096): stage-by-stage, teacher-forced, plus size-independent properties."""
    B, nv, nh = 4096, 784, 1024
    W, b_h, b_v = synthetic_params(nv, nh, seed=1)
    v = synthetic_binary(B, nv, seed=1234)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    seed, step = 42, 17
    rng = O.Rng(seed, step)
    # stage 1: h_pos
    o1 = e.half_step("vh", vd, B, 0, 0, 1, seed, 0, step, want_prob=True, want_u=True)
    n_flip = check_half_step(o1, *O.sample_hidden(v, W, b_h, rng, 0))
    h_pos = o1["sample"].to_numpy()
    # stage 2: v_neg from the GPU's h_pos
    o2 = e.half_step("hv", o1["sample"], B, 0, 0, 1, seed, 1, step, want_prob=True, want_u=True)
    n_flip += check_half_step(o2, *O.sample_visible(h_pos, W, b_v, rng, 1))
    v_neg = o2["sample"].to_numpy()
    # stage 3: h_neg probabilities from the GPU's v_neg
    o3 = e.half_step("vh", o2["sample"], B, 0, 0, 0, seed, 0, step, want_sample=False, want_prob=True)
    h_neg = o3["prob"].to_numpy()
    assert np.max(np.abs(h_neg - O.hidden_prob(v_neg, W, b_h))) <= TOL
    assert n_flip < 64, "far more borderline samples than fp32 rounding explains"
    # stage 4: statistics from the GPU's own states, float64 reference
    dW = e.outer_delta(vd, o1["sample"], o2["sample"], o3["prob"], B).cpu().numpy()
    pos = v.astype(np.float64).T @ h_pos.astype(np.float64)
    neg = v_neg.astype(np.float64).T @ h_neg.astype(np.float64)
    ref = pos - neg
    # entries are differences of two sums of ~1e3 each: measure the error against the un-cancelled
    # magnitude (an fp32 accumulation bound), and the applied update lr * dW against the 1e-4 bar
    assert np.max(np.abs(dW - ref) / (pos + neg + 1.0)) <= 4e-6     # fp32 fma chains of 2048 terms, 4 slabs
    assert np.max(np.abs(1e-3 * dW - 1e-3 * ref)) <= TOL
    # the fused launch sequence reproduces exactly these stages
    d = _gpu_cd_delta(e, vd, B, 1e-3, seed, step)
    dW2, dbh, dbv = _split(d, nv, nh)
    assert np.array_equal(dW2.view(np.uint32), dW.view(np.uint32))
    assert np.array_equal(dbv, v.sum(0) - v_neg.sum(0))                       # integers: exact
    assert rel_err(dbh, h_pos.astype(np.float64).sum(0) - h_neg.astype(np.float64).sum(0)) <= TOL
    # linearity of the statistics in the batch: two half batches sum to the whole
    e.cd_step(vd, B // 2, 0, 1e-3, seed, step, apply=False, emit_delta=True)
    da = e.delta_buffer().cpu().numpy().copy()
    e.cd_step(vd, B // 2, B // 2, 1e-3, seed, step, apply=False, emit_delta=True, row0=B // 2)
    db = e.delta_buffer().cpu().numpy().copy()
    assert np.max(np.abs(da + db - d)[: nv * nh] / (pos + neg + 1.0).ravel()) <= 4e-6
    # apply: W_new - W_old == lr * dW
    e.apply_delta(1e-3, delta=torch.from_numpy(d).to(gpu_device))
    Wn = e.get_weights()[0]
    assert np.max(np.abs((Wn - W) - 1e-3 * dW)) <= 1e-6


# ---------------------------------------------------------------------------------------
def test_rbm_fit_trajectory_golden(gpu_device, golden_dir, capsys):
    """RBM.fit through the reference's class surface: N = 150, bs = 64 (remainder batch), both modes."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    nv, nh = 64, 48
    W, b_h, b_v = synthetic_params(nv, nh, seed=31)
    V = synthetic_binary(150, nv, seed=32, p=0.3)
    g = np.load(os.path.join(golden_dir, "fit_trajectory.npz"))
    hps = {"batch_size": 64, "epochs": 1, "lr": 0.01}
    for mode_name in ("fused", "reference_sequential"):
        rbm = RBM(hps, nh, name="rbm_1", mode=MODE_VISIBLE_BERNOULLI, seed=5, update_mode=mode_name,
                  weights=(W, b_h, b_v))
        assert rbm.fit(V) is None
        assert np.max(np.abs(rbm.rbm_weight - g["W_" + mode_name])) <= TOL
        assert np.max(np.abs(rbm.hidden_bias - g["bh_" + mode_name])) <= TOL
        assert np.max(np.abs(rbm.visible_bias - g["bv_" + mode_name])) <= TOL
        assert np.allclose(rbm.last_scores, g["scores_" + mode_name], rtol=1e-3, atol=1e-3)
        out = capsys.readouterr().out
        assert "1 / 1  epochs" in out and "3/3, score:" in out          # rbm.py:115, :234


def test_rbm_transform_surface(gpu_device):
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    nv, nh = 40, 24
    W, b_h, b_v = synthetic_params(nv, nh, seed=3)
    V = synthetic_binary(10, nv, seed=4)
    rbm = RBM({"batch_size": 4, "epochs": 1, "lr": 0.1}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=11, weights=(W, b_h, b_v))
    rbm.build((None, nv))
    H = rbm.transform(V)
    assert isinstance(H, list) and len(H) == 1 and H[0].shape == (10, nh)      # K.function list return
    assert np.array_equal(H[0], O.transform(W, b_h, V, 11, 0))
    Vr = rbm.inv_transform(H)                                                  # accepts the list form
    assert np.array_equal(Vr[0], O.inv_transform(W, b_v, H[0], 11, 1))
    fe = rbm.cal_free_energy([V])
    assert rel_err(fe[0], O.free_energy(V, W, b_h, b_v)) <= TOL
    assert rbm(V).shape == (10, nh)
    assert rbm.compute_output_shape((7, nv)) == (7, nh)
    cfg = rbm.get_config()
    assert cfg["output_dim"] == nh and cfg["mode"] == MODE_VISIBLE_BERNOULLI


def test_rbm_call_input_kinds(gpu_device):
    """call / transform / cal_free_energy take numpy arrays, torch tensors (host or device) and DeviceMatrix objects, and
    answer in kind: ndarray -> ndarray, tensor -> tensor ON THE INPUT'S DEVICE, DeviceMatrix -> DeviceMatrix."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    from keras_unsupervised_amd.ebm.engine import DeviceMatrix
    nv, nh = 40, 24
    W, b_h, b_v = synthetic_params(nv, nh, seed=3)
    V = synthetic_binary(10, nv, seed=4)

    def fresh():
        return RBM({"batch_size": 4, "epochs": 1, "lr": 0.1}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=11, weights=(W, b_h, b_v))

    ref = O.transform(W, b_h, V, 11, 0)
    out = fresh()(V)
    assert isinstance(out, np.ndarray) and np.array_equal(out, ref)
    out = fresh()(torch.from_numpy(V))                                   # host tensor in, host tensor out
    assert isinstance(out, torch.Tensor) and out.device.type == "cpu" and np.array_equal(out.numpy(), ref)
    out = fresh()(torch.from_numpy(V).to(gpu_device))
    assert out.device == gpu_device and np.array_equal(out.cpu().numpy(), ref)
    out = fresh()(DeviceMatrix.from_host(V, gpu_device))                 # a DeviceMatrix has no .shape: used to raise
    assert isinstance(out, DeviceMatrix) and np.array_equal(out.to_numpy(), ref)
    r = fresh()
    F = r.cal_free_energy(torch.from_numpy(V))
    assert isinstance(F, list) and F[0].device.type == "cpu"
    assert rel_err(F[0].numpy(), O.free_energy(V, W, b_h, b_v)) <= TOL
    H = r.transform(torch.from_numpy(V))
    assert isinstance(H, list) and H[0].device.type == "cpu"


def test_empty_inputs(gpu_device):
    """Empty feeds: transform / inv_transform / cal_free_energy of a [0, n] array give [0, m] / [0] results, fit of an
    empty matrix takes no step (rbm.py:110-111: num_step = 0), a DBN passes the empty matrix through its layers."""
    from keras_unsupervised_amd.ebm import DBN, MODE_VISIBLE_BERNOULLI, RBM
    nv, nh = 40, 24
    W0 = synthetic_params(nv, nh, seed=3)
    r = RBM({"batch_size": 4, "epochs": 2, "lr": 0.1}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=11, weights=W0)
    empty = np.zeros((0, nv), np.float32)
    H = r.transform(empty)
    assert H[0].shape == (0, nh)
    assert r.inv_transform(np.zeros((0, nh), np.float32))[0].shape == (0, nv)
    assert r.cal_free_energy(empty)[0].shape == (0,)
    assert r.fit(empty, verbose=1) is None and r._update_count == 0
    for x, y in zip(r.get_weights(), W0):
        assert np.array_equal(x, y)
    r3 = RBM({"batch_size": 1024, "epochs": 1, "lr": 0.1}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=11, weights=W0)     # x3 path
    assert r3.fit(empty, verbose=0) is None and r3._update_count == 0
    d = DBN()
    d.add_stack(r)
    d.add_stack(RBM({"batch_size": 4, "epochs": 1, "lr": 0.1}, 8, mode=MODE_VISIBLE_BERNOULLI, seed=2, weights=synthetic_params(nh, 8, seed=4)))
    assert d.transform(empty).shape == (0, 8)
    # a single row and a batch size larger than the data set
    one = synthetic_binary(1, nv, seed=5)
    r1 = RBM({"batch_size": 64, "epochs": 1, "lr": 0.01}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=11, weights=W0)
    r1.fit(one, verbose=0)
    Wr, bhr, bvr, _, _ = O.cd_step_fused(*W0, one, 0.01, 11, 0)
    assert np.max(np.abs(r1.rbm_weight - Wr)) <= TOL and np.max(np.abs(r1.visible_bias - bvr)) <= TOL


def test_dbn_golden(gpu_device, golden_dir, capsys):
    from keras_unsupervised_amd.ebm import DBN, MODE_VISIBLE_BERNOULLI, RBM
    g = np.load(os.path.join(golden_dir, "dbn_small.npz"))
    Vd = synthetic_binary(40, 32, seed=41, p=0.4)
    hps = {"batch_size": 16, "epochs": 2, "lr": 0.02}
    l0 = RBM(hps, 24, name="l0", mode=MODE_VISIBLE_BERNOULLI, seed=1, weights=synthetic_params(32, 24, seed=42))
    l1 = RBM(hps, 16, name="l1", mode=MODE_VISIBLE_BERNOULLI, seed=2, weights=synthetic_params(24, 16, seed=43))
    dbn = DBN()
    dbn.add_stack(l0)
    dbn.add_stack(l1)
    dbn.fit(Vd, verbose=0)
    assert "Train l0." in capsys.readouterr().out                               # dbn.py:53
    assert np.max(np.abs(l0.rbm_weight - g["W0"])) <= TOL
    assert np.max(np.abs(l1.rbm_weight - g["W1"])) <= TOL
    assert np.max(np.abs(l1.hidden_bias - g["bh1"])) <= TOL
    feat = dbn.transform(Vd)
    assert np.array_equal(feat, g["feat"].astype(np.float32))
    back = dbn.inv_transform(feat)
    assert np.array_equal(back, g["back"].astype(np.float32))
    bad = RBM(hps, 8, mode=MODE_VISIBLE_BERNOULLI)
    bad.build((None, 99))
    with pytest.raises(ValueError):
        dbn.add_stack(bad)                                                      # dbn.py:29-30
    with pytest.raises(ValueError):
        DBN().fit(Vd)                                                           # dbn.py:47-48


def test_c_abi_from_plain_c(gpu_device, tmp_path):
    """The boundary is a C ABI: examples/c/cd_step_demo.c (C99, gcc, HIP runtime API for memory, no Python or PyTorch in the
    process) runs one CD-1 update on the fp32-MFMA entry point, on the x3 entry point with resident planes and as ONE launch
    (kurbm_cd_step_small), compares them, and reads kurbm_score_small's score by polling pinned host memory."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "keras_unsupervised_amd", "csrc")
    exe = str(tmp_path / "cd_step_demo")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                    os.path.join(root, "examples", "c", "cd_step_demo.c"), "-L", csrc, "-lkurbm", "-L", "/opt/rocm/lib", "-lamdhip64",
                    "-lm", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi 5" in r.stdout and "polled" in r.stdout and "status 0" in r.stdout


def test_c_abi_error_behaviour(gpu_device):
    """Bad arguments come back as error codes with a message, never as a fault."""
    from keras_unsupervised_amd import _lib
    W, b_h, b_v = synthetic_params(16, 8, seed=1)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(synthetic_binary(8, 16, seed=2), gpu_device)
    with pytest.raises(_lib.KurbmError, match="multiple of 4"):
        e.half_step("vh", vd, 8, 0, 0, 1, 1, 0, 0, row0=2)
    with pytest.raises(_lib.KurbmError, match="k must be"):
        e.cd_step(vd, 8, 0, 0.1, 1, 0, k=0)
    with pytest.raises(ValueError):
        e.half_step("hv", vd, 8, 0, 0, 1, 1, 0, 0)      # 16 columns into an 8-wide hidden layer


def test_checkpoint_round_trip(gpu_device, tmp_path):
    """f-4: config + weights (+ RNG counters) survive save/load; the reloaded RBM continues bit-identically."""
    from keras_unsupervised_amd.ebm import DBN, MODE_VISIBLE_BERNOULLI, RBM, load_dbn, load_rbm, save_dbn, save_rbm
    nv, nh = 36, 20
    V = synthetic_binary(50, nv, seed=70, p=0.4)
    hps = {"batch_size": 16, "epochs": 1, "lr": 0.05}
    a = RBM(hps, nh, name="ck", mode=MODE_VISIBLE_BERNOULLI, seed=3, cd_k=2, weights=synthetic_params(nv, nh, 71))
    a.fit(V, verbose=0)
    save_rbm(a, str(tmp_path / "ck"))
    b = load_rbm(str(tmp_path / "ck"))
    assert b.mode == a.mode and b.cd_k == 2 and b.name == "ck" and b.hps == hps
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)                      # visible bias included (the reference drops it)
    a.fit(V, verbose=0)
    b.fit(V, verbose=0)
    assert np.array_equal(a.rbm_weight, b.rbm_weight)    # same counters -> same draws -> same bits
    assert np.array_equal(a.transform(V)[0], b.transform(V)[0])
    # persistent chains: the fantasy particles travel with the checkpoint, so the reloaded run continues the SAME chain
    for compute in ("fp32", "x3"):
        pa = RBM(hps, nh, name="pcd", mode=MODE_VISIBLE_BERNOULLI, seed=3, persistent=True, compute_dtype=compute,
                 weights=synthetic_params(nv, nh, 71))
        pa.fit(V, verbose=0)
        save_rbm(pa, str(tmp_path / "pcd"))
        pb = load_rbm(str(tmp_path / "pcd"))
        assert pb.persistent and np.array_equal(pb._v_chain.to_numpy(), pa._v_chain.to_numpy())
        pa.fit(V, verbose=0)
        pb.fit(V, verbose=0)
        assert np.array_equal(pa.rbm_weight, pb.rbm_weight)
        fresh = RBM(hps, nh, mode=MODE_VISIBLE_BERNOULLI, seed=3, persistent=True, compute_dtype=compute, weights=pb.get_weights())
        assert fresh._v_chain is None
        # a chain whose row count is not the batch size is refused -- by fit() (every step reads and rewrites batch_size rows of it
        # in place) and by load_rbm (a checkpoint edited, or saved under another batch size) -- instead of being read past its end
        pb.hps = dict(hps, batch_size=32)
        with pytest.raises(ValueError, match="persistent chain has shape"):
            pb.fit(V, verbose=0)
        pb.hps = hps
        import json as _json
        meta = _json.load(open(str(tmp_path / "pcd.json")))
        meta["config"]["hps"]["batch_size"] = 32
        _json.dump(meta, open(str(tmp_path / "pcd.json"), "w"))
        with pytest.raises(ValueError, match="checkpoint chain has shape"):
            load_rbm(str(tmp_path / "pcd"))
    d = DBN()
    d.add_stack(a)
    d.add_stack(RBM(hps, 8, name="top", mode=MODE_VISIBLE_BERNOULLI, seed=4, weights=synthetic_params(nh, 8, 72)))
    d.fit(V, verbose=0)
    save_dbn(d, str(tmp_path / "stack"))
    d2 = load_dbn(str(tmp_path / "stack"))
    assert np.array_equal(d.transform(V), d2.transform(V))


def test_example_pipeline_digits(gpu_device, tmp_path):
    """f-3: RBM features -> softmax head on the bundled 8x8 digits (upsampled to 784), end to end."""
    import importlib.util
    import json as _json
    spec = importlib.util.spec_from_file_location("rbm_softmax_digits", os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "rbm", "rbm_softmax_digits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    conf = _json.load(open(os.path.join(os.path.dirname(spec.origin), "rbm_softmax_conf.json")))
    conf["data"] = "digits"
    conf["hps"]["epochs"] = 15
    mc = mod.MNISTClassifier(conf, workdir=str(tmp_path))
    mc.train(verbose=0)
    acc = mc.test()
    assert acc is not None and acc > 0.6, "RBM features + softmax should beat chance (0.1) by far, got %r" % acc
    lines = open(tmp_path / "solution.csv").read().splitlines()
    assert lines[0] == "ImageId,Label" and len(lines) == 1 + 360
    conf["model_loading"] = True
    again = mod.MNISTClassifier(conf, workdir=str(tmp_path))
    assert np.array_equal(again.rbm.rbm_weight, mc.rbm.rbm_weight)


# ---------------------------------------------------------------------------------------
# bf16 operands / fp32 accumulate (extension; BASELINE.json config 5)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(64, 96, 80), (200, 300, 260), (130, 784, 256)])
def test_bf16_half_steps(gpu_device, shape):
    """bf16 half steps against the oracle fed bf16-rounded operands: uniforms bit-exact, p within 1e-4."""
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=400 + B)
    v = synthetic_real(B, nv, seed=401 + B)
    e = _engine(W, b_h, b_v, gpu_device)
    Wq = O.bf16_round(W)
    rng = O.Rng(5, 3)
    out = e.half_step_bf16("vh", _dm(v, gpu_device), B, 0, 1, 5, 2, 3)
    check_half_step(out, *O.sample_hidden(O.bf16_round(v), Wq, b_h, rng, 2))
    h = synthetic_binary(B, nh, seed=402 + B, p=0.5)
    out = e.half_step_bf16("hv", _dm(h, gpu_device), B, 0, 1, 5, 3, 3)
    check_half_step(out, *O.sample_visible(h, Wq, b_v, rng, 3))
    out = e.half_step_bf16("vh", _dm(v, gpu_device), B, 0, 0, 5, 0, 0)           # probabilities only
    assert np.max(np.abs(out["prob"].to_numpy() - O.hidden_prob(O.bf16_round(v), Wq, b_h))) <= TOL


@pytest.mark.parametrize("cfg", [dict(B=64, nv=96, nh=80, k=1), dict(B=150, nv=200, nh=136, k=3),
                                 dict(B=72, nv=128, nh=128, k=2, pcd=True), dict(B=256, nv=784, nh=256, k=1)])
def test_bf16_cd_step_vs_oracle(gpu_device, cfg):
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    W, b_h, b_v = synthetic_params(nv, nh, seed=500 + B)
    v = synthetic_binary(B, nv, seed=501 + B, p=0.3)
    chain0 = synthetic_binary(B, nv, seed=502 + B, p=0.5) if cfg.get("pcd") else None
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    cd = _dm(chain0, gpu_device) if chain0 is not None else None
    e.cd_step(vd, B, 0, 0.002, 77, 9, k=k, apply=False, emit_delta=True, v_chain=cd, bf16=True)
    torch.cuda.synchronize()
    d1 = e.delta_buffer().cpu().numpy().copy()
    _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused_bf16(W, b_h, b_v, v, 0.002, 77, 9, k=k, v_chain=chain0)
    dW, dbh, dbv = _split(d1, nv, nh)
    assert rel_err(dW, dW_ref) <= 2e-3          # h_neg enters the statistics rounded to bf16 (8 bits)
    assert rel_err(dbh, dbh_ref) <= TOL
    assert np.array_equal(dbv, dbv_ref)
    if cd is not None:
        assert np.array_equal(cd.to_numpy(), ch["v_neg"])
    # in-place apply keeps the fp32 master authoritative and the mirrors in sync
    e.cd_step(vd, B, 0, 0.002, 77, 9, k=k, v_chain=_dm(chain0, gpu_device) if chain0 is not None else None, bf16=True)
    Wn = e.get_weights()[0]
    assert np.max(np.abs(Wn - (W + np.float32(0.002) * dW))) <= 1e-5
    out = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0)
    assert np.max(np.abs(out["prob"].to_numpy() - O.hidden_prob(v, O.bf16_round(Wn), e.get_weights()[1]))) <= TOL


def test_bf16_rbm_fit(gpu_device):
    """RBM(compute_dtype='bf16').fit runs the bf16 path end to end and stays close to the fp32 run."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    nv, nh = 128, 96
    W0 = synthetic_params(nv, nh, seed=600)
    V = synthetic_binary(300, nv, seed=601, p=0.3)
    hps = {"batch_size": 128, "epochs": 2, "lr": 0.002}
    a = RBM(hps, nh, mode=MODE_VISIBLE_BERNOULLI, seed=1, weights=W0)
    b = RBM(hps, nh, mode=MODE_VISIBLE_BERNOULLI, seed=1, weights=W0, compute_dtype="bf16")
    a.fit(V, verbose=0)
    b.fit(V, verbose=0)
    assert b.get_config()["compute_dtype"] == "bf16"
    assert np.max(np.abs(a.rbm_weight - b.rbm_weight)) < 0.05 and np.isfinite(b.rbm_weight).all()
    assert not np.array_equal(a.rbm_weight, b.rbm_weight)


@pytest.mark.parametrize("resident", [False, True])
@pytest.mark.parametrize("shape", [(150, 64, 72, 40, 2, False), (1400, 256, 200, 136, 1, False), (512, 256, 784, 64, 1, False),
                                   (700, 256, 200, 136, 1, True)])
def test_x3_epoch_call_equals_step_loop(gpu_device, resident, shape):
    """kurbm_cd_epoch_x3 (one call per epoch) and the per-step host loop give bit-identical parameters -- with the
    per-step conversion of the batch and with RESIDENT planes (kurbm_x3_convert_rows once per window of rows, every step
    of every epoch reading them through opts->v_planes: the k_f32_to_bf16 launch is gone from the loop), for 0/1 data
    (one piece) and real-valued data (three), remainder batch included."""
    N, bs, nv, nh, k, real = shape
    W0 = synthetic_params(nv, nh, seed=700)
    V = synthetic_real(N, nv, seed=701) if real else synthetic_binary(N, nv, seed=701, p=0.3)
    a, b = _engine(*W0, gpu_device), _engine(*W0, gpu_device)
    vd = _dm(V, gpu_device)
    slices = O.batch_slices(N, bs)
    nsteps = len(slices)
    planes = a.make_planes(vd, [(lo, hi - lo) for lo, hi in slices]) if resident else None
    assert (planes is not None) == resident
    if resident:
        assert planes.uniform(0, N, bs) and planes.v_pieces == (3 if real else 1 | 0x10)   # (0/1 data: KURBM_V_BINARY)
    for epoch in range(2):
        assert a.cd_epoch(vd, N, bs, 0.01, 9, 5 + epoch * nsteps, k=k, compute="x3", planes=planes) == nsteps
        for i, (lo, hi) in enumerate(slices):
            b.cd_step(vd, hi - lo, lo, 0.01, 9, 5 + epoch * nsteps + i, k=k, compute="x3")
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)
    if resident:   # and step by step on the same planes
        c = _engine(*W0, gpu_device)
        for epoch in range(2):
            for i, (lo, hi) in enumerate(slices):
                c.cd_step(vd, hi - lo, lo, 0.01, 9, 5 + epoch * nsteps + i, k=k, compute="x3", planes=planes)
        for x, y in zip(c.get_weights(), b.get_weights()):
            assert np.array_equal(x, y)


def test_epoch_call_equals_step_loop(gpu_device):
    """kurbm_cd_epoch (one call per epoch) and the per-step host loop give bit-identical parameters."""
    nv, nh = 72, 40
    W0 = synthetic_params(nv, nh, seed=700)
    V = synthetic_binary(150, nv, seed=701, p=0.3)
    a, b = _engine(*W0, gpu_device), _engine(*W0, gpu_device)
    vd = _dm(V, gpu_device)
    assert a.cd_epoch(vd, 150, 64, 0.01, 9, 5, k=2) == 3
    for i, (lo, hi) in enumerate(O.batch_slices(150, 64)):
        b.cd_step(vd, hi - lo, lo, 0.01, 9, 5 + i, k=2)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.array_equal(x, y)


def test_random_shape_sweep(gpu_device):
    """Seeded sweep over awkward shapes (not multiples of 4 / 16 / 32 / the tiles; tiny and ragged
    batches; k tails of every residue class): half steps, CD-k delta and free energy vs the oracle."""
    rs = np.random.RandomState(2026)
    for case in range(14):
        B = int(rs.choice([1, 2, 5, 31, 33, 63, 65, 100, 129, 200, 257]))
        nv = int(rs.randint(3, 400))
        nh = int(rs.randint(3, 400))
        k = int(rs.choice([1, 1, 2]))
        W, b_h, b_v = synthetic_params(nv, nh, seed=900 + case)
        v = synthetic_binary(B, nv, seed=950 + case, p=0.4)
        e = _engine(W, b_h, b_v, gpu_device)
        vd = _dm(v, gpu_device)
        rng = O.Rng(case, 7)
        out = e.half_step("vh", vd, B, 0, 0, 1, case, 0, 7, want_prob=True, want_u=True)
        check_half_step(out, *O.sample_hidden(v, W, b_h, rng, 0))
        h = out["sample"].to_numpy()
        out2 = e.half_step("hv", out["sample"], B, 0, 0, 1, case, 1, 7, want_prob=True, want_u=True)
        check_half_step(out2, *O.sample_visible(h, W, b_v, rng, 1))
        F = e.free_energy(vd, B).cpu().numpy()
        assert rel_err(F, O.free_energy(v, W, b_h, b_v)) <= TOL, (case, B, nv, nh)
        d = _gpu_cd_delta(e, vd, B, 0.01, case, 3, k=k)
        _, _, _, _, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.01, case, 3, k=k)
        dW, dbh, dbv = _split(d, nv, nh)
        assert rel_err(dW, dW_ref) <= TOL and rel_err(dbh, dbh_ref) <= TOL and np.array_equal(dbv, dbv_ref), (case, B, nv, nh, k)


@pytest.mark.parametrize("compute", ["fp32", "x3"])
def test_data_parallel_emulation_two_shards(gpu_device, compute):
    """The N > 1 update sequence with the real kernels, two 'ranks' emulated on one GPU: each runs the
    chain on its shard (global-row Philox counters), the packed deltas are summed (the all-reduce) and
    applied on both replicas.  Replicas stay bit-identical and track the single-rank run within the fp32
    summation-order band -- draws are identical whatever the shard count."""
    from keras_unsupervised_amd.ebm import dp
    nv, nh, B = 120, 88, 150          # ragged: shards of 76 and 74 rows
    W0 = synthetic_params(nv, nh, seed=1000)
    V = synthetic_binary(2 * B, nv, seed=1001, p=0.3)
    vd = _dm(V, gpu_device)
    single = _engine(*W0, gpu_device)
    ranks = [_engine(*W0, gpu_device) for _ in range(2)]
    lr = 0.01
    for step in range(2):
        lo = step * B
        single.cd_step(vd, B, lo, lr, 21, step, k=2, compute=compute)
        deltas = []
        for r, e in enumerate(ranks):
            s_lo, s_hi = dp.shard_rows(B, 2, r)
            assert s_lo % 4 == 0
            e.cd_step(vd, s_hi - s_lo, lo + s_lo, lr, 21, step, k=2, apply=False, emit_delta=True, row0=s_lo,
                      compute=compute)
            deltas.append(e.delta_buffer().clone())
        total = deltas[0] + deltas[1]                      # what all_reduce(SUM) leaves on every rank
        for e in ranks:
            e.apply_delta(lr, delta=total, compute=compute)
    torch.cuda.synchronize()
    a, b, ref = ranks[0].get_weights(), ranks[1].get_weights(), single.get_weights()
    for x, y, z in zip(a, b, ref):
        assert np.array_equal(x, y)
        assert np.max(np.abs(x - z)) <= 1e-5


# ---------------------------------------------------------------------------------------
# x3: fp32 values as exact bf16 triples on the bf16 matrix cores -- held to the FP32 oracle and the
# fp32 tolerances (it is not a reduced-precision path)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(64, 96, 80), (200, 300, 260), (130, 784, 256), (5, 7, 3)])
def test_x3_half_steps(gpu_device, shape):
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=800 + B)
    W = (W * np.float32(3.7)).astype(np.float32)             # use all 24 significand bits
    e = _engine(W, b_h, b_v, gpu_device)
    rng = O.Rng(5, 3)
    for v in (synthetic_binary(B, nv, seed=801 + B, p=0.3), synthetic_real(B, nv, seed=802 + B)):
        vd = _dm(v, gpu_device)
        assert e.v_pieces(vd) == (1 if np.array_equal(v, O.bf16_round(v)) else 3)
        out = e.half_step_bf16("vh", vd, B, 0, 1, 5, 2, 3, pieces=3)
        check_half_step(out, *O.sample_hidden(v, W, b_h, rng, 2))
        out = e.half_step_bf16("vh", vd, B, 0, 0, 5, 0, 0, pieces=3)          # probabilities only
        assert np.max(np.abs(out["prob"].to_numpy() - O.hidden_prob(v, W, b_h))) <= TOL
    h = synthetic_binary(B, nh, seed=803 + B, p=0.5)
    out = e.half_step_bf16("hv", _dm(h, gpu_device), B, 0, 1, 5, 3, 3, pieces=3)
    check_half_step(out, *O.sample_visible(h, W, b_v, rng, 3))


def test_x3_is_not_reduced_precision(gpu_device):
    """Pre-activation error against float64: the x3 path is at least as accurate as the fp32 MFMA path
    (pieces are exact and their products are exact in fp32; only the order of fp32 additions differs),
    while the rounded-bf16 path is two orders of magnitude away."""
    B, nv, nh = 512, 784, 1024
    W, b_h, b_v = synthetic_params(nv, nh, seed=1)
    e = _engine(W, b_h, b_v, gpu_device)
    errs = {}
    for name, v in (("binary", synthetic_binary(B, nv, seed=5)), ("real", synthetic_real(B, nv, seed=6))):
        vd = _dm(v, gpu_device)
        x64 = v.astype(np.float64) @ W.astype(np.float64) + b_h.astype(np.float64)
        p64 = 1.0 / (1.0 + np.exp(-x64))
        p32 = e.half_step("vh", vd, B, 0, 0, 0, 0, 0, 0, want_sample=False, want_prob=True)["prob"].to_numpy()
        px3 = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=3)["prob"].to_numpy()
        pbf = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=1)["prob"].to_numpy()
        errs[name] = [float(np.max(np.abs(p - p64))) for p in (p32, px3, pbf)]
        assert errs[name][1] <= max(2.0 * errs[name][0], 5e-7), errs
        assert errs[name][1] <= 1e-6, errs
    assert errs["real"][2] > 20 * errs["real"][1], errs       # the rounded path IS reduced precision
    print("max |p - p_float64|  (fp32 MFMA, x3, rounded bf16):", errs)


@pytest.mark.parametrize("cfg", [dict(B=64, nv=96, nh=80, k=1), dict(B=150, nv=200, nh=136, k=3),
                                 dict(B=72, nv=128, nh=128, k=2, pcd=True), dict(B=256, nv=784, nh=256, k=1),
                                 dict(B=133, nv=100, nh=70, k=1, real=True), dict(B=9, nv=5, nh=3, k=1)])
def test_x3_cd_step_vs_oracle(gpu_device, cfg):
    """CD-k / PCD on the x3 path against the fp32 oracle, fp32 tolerances."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    W, b_h, b_v = synthetic_params(nv, nh, seed=900 + B)
    v = synthetic_real(B, nv, seed=901 + B) if cfg.get("real") else synthetic_binary(B, nv, seed=901 + B, p=0.3)
    chain0 = synthetic_binary(B, nv, seed=902 + B, p=0.5) if cfg.get("pcd") else None
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    cd = _dm(chain0, gpu_device) if chain0 is not None else None
    d1 = _gpu_cd_delta(e, vd, B, 0.05, 77, 9, k=k, v_chain=cd, compute="x3")
    _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.05, 77, 9, k=k, v_chain=chain0)
    dW, dbh, dbv = _split(d1, nv, nh)
    assert rel_err(dW, dW_ref) <= TOL
    assert rel_err(dbh, dbh_ref) <= TOL
    assert rel_err(dbv, dbv_ref) <= TOL
    if cd is not None:
        assert np.array_equal(cd.to_numpy(), ch["v_neg"])
        cd = _dm(chain0, gpu_device)
    d2 = _gpu_cd_delta(e, vd, B, 0.05, 77, 9, k=k, v_chain=cd, compute="x3")
    assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32)), "same inputs, same counters -> same bits"
    # in-place apply: fp32 master updated, the piece mirrors follow
    e.cd_step(vd, B, 0, 0.05, 77, 9, k=k, v_chain=_dm(chain0, gpu_device) if chain0 is not None else None, compute="x3")
    Wn, bhn, bvn = e.get_weights()
    assert np.max(np.abs(Wn - (W + np.float32(0.05) * dW))) <= 1e-5
    out = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=3)
    assert np.max(np.abs(out["prob"].to_numpy() - O.hidden_prob(v, Wn, bhn))) <= TOL


def test_x3_full_size_properties(gpu_device):
    """Config 2 on the x3 path: exact integer statistics, determinism, linearity in the batch, and
    agreement with the fp32 MFMA path (identical draws; samples differ only in the rounding band)."""
    B, nv, nh = 4096, 784, 1024
    W, b_h, b_v = synthetic_params(nv, nh, seed=1)
    v = synthetic_binary(B, nv, seed=1234)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    seed, step = 42, 17
    d = _gpu_cd_delta(e, vd, B, 1e-3, seed, step, compute="x3")
    d_again = _gpu_cd_delta(e, vd, B, 1e-3, seed, step, compute="x3")
    assert np.array_equal(d.view(np.uint32), d_again.view(np.uint32))
    dW, dbh, dbv = _split(d, nv, nh)
    assert np.array_equal(dbv, np.round(dbv)) and np.abs(dbv).max() <= B       # counts of 0/1 states
    # the fp32 MFMA path on the same counters
    dW32, dbh32, dbv32 = _split(_gpu_cd_delta(e, vd, B, 1e-3, seed, step), nv, nh)
    assert np.abs(dbv - dbv32).sum() <= 64                                     # a handful of borderline samples
    assert np.linalg.norm(dW - dW32) <= 2e-3 * np.linalg.norm(dW32)
    assert np.max(np.abs(1e-3 * dW - 1e-3 * dW32)) <= 1e-2                     # a flipped unit moves its row/column by <= lr
    # stage-wise, teacher-forced on the fp32 path's samples: the statistics GEMM of the x3 step equals a
    # float64 reference computed from ITS OWN chain states -- recover them through the half-step hook
    o1 = e.half_step_bf16("vh", vd, B, 0, 1, seed, 0, step, pieces=3)
    h_pos = o1["sample"].to_numpy()
    o2 = e.half_step_bf16("hv", o1["sample"], B, 0, 1, seed, 1, step, pieces=3)
    v_neg = o2["sample"].to_numpy()
    o3 = e.half_step_bf16("vh", o2["sample"], B, 0, 0, seed, 0, step, pieces=3)
    h_neg = o3["prob"].to_numpy()
    pos = v.astype(np.float64).T @ h_pos.astype(np.float64)
    neg = v_neg.astype(np.float64).T @ h_neg.astype(np.float64)
    assert np.max(np.abs(dW - (pos - neg)) / (pos + neg + 1.0)) <= 4e-6
    assert np.array_equal(dbv, v.sum(0) - v_neg.sum(0))
    assert rel_err(dbh, h_pos.astype(np.float64).sum(0) - h_neg.astype(np.float64).sum(0)) <= TOL
    # linearity: two half batches sum to the whole
    da = _gpu_cd_delta(e, vd, B // 2, 1e-3, seed, step, compute="x3")
    e.cd_step(vd, B // 2, B // 2, 1e-3, seed, step, apply=False, emit_delta=True, row0=B // 2, compute="x3")
    torch.cuda.synchronize()
    db = e.delta_buffer().cpu().numpy().copy()
    assert np.max(np.abs(da + db - d)[: nv * nh] / (pos + neg + 1.0).ravel()) <= 4e-6


def test_x3_rbm_fit_trajectory_golden(gpu_device, golden_dir):
    """RBM(compute_dtype='x3').fit reproduces the fp32 oracle's golden trajectory."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    nv, nh = 64, 48
    W, b_h, b_v = synthetic_params(nv, nh, seed=31)
    V = synthetic_binary(150, nv, seed=32, p=0.3)
    g = np.load(os.path.join(golden_dir, "fit_trajectory.npz"))
    hps = {"batch_size": 64, "epochs": 1, "lr": 0.01}
    for um in ("fused", "reference_sequential"):
        r = RBM(hps, nh, mode=MODE_VISIBLE_BERNOULLI, seed=5, update_mode=um, weights=(W, b_h, b_v), compute_dtype="x3")
        r.fit(V, verbose=0)
        assert np.max(np.abs(r.rbm_weight - g["W_" + um])) <= TOL
        assert np.max(np.abs(r.hidden_bias - g["bh_" + um])) <= TOL
        assert np.max(np.abs(r.visible_bias - g["bv_" + um])) <= TOL


@pytest.fixture(scope="module")
def one_rank_comm(gpu_device):
    """A 1-rank RCCL communicator through the C ABI (kurbm_comm_unique_id / kurbm_comm_init_rank): what a one-GPU box can
    exercise of the exchange step -- RCCL bound and called from inside libkurbm.so, the comm stream, the events."""
    from keras_unsupervised_amd.ebm import dp
    comm = dp.Comm(gpu_device, 0, 1, dp.Comm.new_unique_id())
    yield comm
    comm.destroy()


def test_comm_through_c_abi(gpu_device, one_rank_comm):
    comm = one_rank_comm
    assert comm.count() == 1                                   # ncclCommCount
    assert comm.lib.kurbm_comm_rank(comm.handle) == 0          # ncclCommUserRank
    x = torch.randn(804624, device=gpu_device)                 # the packed [dW | db_h | db_v] of 784 x 1024
    y = x.clone()
    comm.allreduce_sum_(y)
    comm.barrier()
    assert torch.equal(x, y)                                   # a sum over one rank
    from keras_unsupervised_amd import _lib
    with pytest.raises(_lib.KurbmError):                       # a null communicator is an error code, not a fault
        _lib.check(comm.lib.kurbm_allreduce_sum_f32(None, y.data_ptr(), 4, None))
    # one process driving its GPUs (ncclCommInitAll): the array form, here with the one device of this box
    import ctypes as C
    devs = (C.c_int * 1)(gpu_device.index)
    out = (C.c_void_p * 1)()
    _lib.check(comm.lib.kurbm_comm_init_all(1, devs, out))
    assert comm.lib.kurbm_comm_count(out[0]) == 1 and comm.lib.kurbm_comm_rank(out[0]) == 0
    st = C.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)
    _lib.check(comm.lib.kurbm_allreduce_sum_f32(out[0], y.data_ptr(), y.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    comm.lib.kurbm_comm_destroy(out[0])
    with pytest.raises(_lib.KurbmError, match="rank 3 of 2"):   # bad rank: refused before RCCL is asked
        _lib.check(comm.lib.kurbm_comm_init_rank(gpu_device.index, 2, 3, C.create_string_buffer(128), 128, C.byref(C.c_void_p())))


@pytest.mark.parametrize("cfg", [dict(B=300, nv=784, nh=256, k=1), dict(B=260, nv=1100, nh=200, k=2, pcd=True),
                                 dict(B=64, nv=100, nh=80, k=1), dict(B=256, nv=640, nh=136, k=1, gauss=True),
                                 dict(B=4096, nv=784, nh=1024, k=1)])
def test_x3_dp_step_one_call(gpu_device, one_rank_comm, cfg):
    """kurbm_cd_step_x3_dp -- chain, statistics in row ranges of dW with each range all-reduced on the library's comm
    stream while the next is computed, fused apply + weight-piece mirror -- against the plain sequence
    emit -> all-reduce -> apply: bit-identical with one range, within fp32 summation order with several (the split-K
    slicing of a row range differs); Gaussian mode (the reference's default) included; a rank without rows joins with
    zeros and leaves the parameters where they were."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg.get("gauss") else O.MODE_VISIBLE_BERNOULLI
    W0 = synthetic_params(nv, nh, seed=1100 + B)
    V = synthetic_real(B, nv, seed=1101 + B) if cfg.get("gauss") else synthetic_binary(B, nv, seed=1101 + B, p=0.3)
    chain0 = synthetic_binary(B, nv, seed=1102 + B, p=0.5) if cfg.get("pcd") else None
    vd = _dm(V, gpu_device)
    ref = _engine(*W0, gpu_device)
    cr = _dm(chain0, gpu_device) if chain0 is not None else None
    ref.cd_step(vd, B, 0, 0.01, 5, 3, k=k, mode=mode, apply=False, emit_delta=True, row0=8, v_chain=cr, compute="x3")
    d_ref = ref.delta_buffer().clone()
    ref.apply_delta(0.01, compute="x3")
    for n_chunks in (1, 2, 3, 0):
        e = _engine(*W0, gpu_device)
        cd = _dm(chain0, gpu_device) if chain0 is not None else None
        e.cd_step_dp(one_rank_comm, vd, B, 0, 0.01, 5, 3, k=k, mode=mode, row0=8, v_chain=cd, compute="x3", n_chunks=n_chunks)
        torch.cuda.synchronize()
        if cd is not None:
            assert np.array_equal(cd.to_numpy(), cr.to_numpy())
        got, want = e.delta_buffer().cpu().numpy(), d_ref.cpu().numpy()
        if n_chunks == 1:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
            for x, y in zip(e.get_weights(), ref.get_weights()):
                assert np.array_equal(x, y)
        else:
            dW, dbh, dbv = _split(got, nv, nh)
            rW, rbh, rbv = _split(want, nv, nh)
            assert np.max(np.abs(dW - rW)) <= 1e-4 * max(1.0, float(np.abs(rW).max()))
            assert np.array_equal(dbv.view(np.uint32), rbv.view(np.uint32)) and np.array_equal(dbh.view(np.uint32), rbh.view(np.uint32))
            for x, y in zip(e.get_weights(), ref.get_weights()):
                assert np.max(np.abs(x - y)) <= 1e-6 * max(1.0, float(np.abs(y).max()))    # a few ulp: other split-K slicing
        # the rewritten weight pieces are the new weights
        p1 = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=3)["prob"].to_numpy()
        assert np.max(np.abs(p1 - O.hidden_prob(V, *e.get_weights()[:2]))) <= TOL
    # a rank that owns no rows of the batch
    e = _engine(*W0, gpu_device)
    e.cd_step_dp(one_rank_comm, vd, 0, 0, 0.01, 5, 3, k=k, mode=mode, compute="x3")
    torch.cuda.synchronize()
    assert float(e.delta_buffer().abs().max().item()) == 0.0
    for x, y in zip(e.get_weights(), W0):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_dp_step_other_paths(gpu_device, one_rank_comm, compute):
    """The data-parallel step of the fp32-MFMA and rounded-bf16 paths (emit -> kurbm_allreduce_sum_f32 -> apply)."""
    B, nv, nh = 200, 300, 140
    W0 = synthetic_params(nv, nh, seed=1200)
    vd = _dm(synthetic_binary(B, nv, seed=1201, p=0.3), gpu_device)
    a, b = _engine(*W0, gpu_device), _engine(*W0, gpu_device)
    a.cd_step_dp(one_rank_comm, vd, B, 0, 0.01, 5, 3, k=2, row0=4, compute=compute)
    b.cd_step(vd, B, 0, 0.01, 5, 3, k=2, row0=4, compute=compute)
    for x, y in zip(a.get_weights(), b.get_weights()):
        assert np.max(np.abs(x - y)) <= 1e-6


def test_x3_random_shape_sweep(gpu_device):
    """The seeded sweep of awkward shapes (tiny and ragged batches, unit counts that are not multiples of 4 / 16 /
    64 / the tiles, k of every residue class, 0/1 and real-valued data, CD-1 / CD-2, with and without persistent
    chains) on the x3 path: half steps and CD-k sums against the fp32 oracle, fp32 bars; in-place apply and the
    data-parallel apply (fused weight-piece rewrite) agree."""
    rs = np.random.RandomState(4052)
    for case in range(16):
        B = int(rs.choice([1, 2, 5, 31, 33, 63, 65, 100, 129, 200, 257, 300]))
        nv = int(rs.randint(3, 500))
        nh = int(rs.randint(3, 500))
        k = int(rs.choice([1, 1, 2]))
        real = bool(case % 3 == 2)
        pcd = bool(case % 4 == 3)
        W, b_h, b_v = synthetic_params(nv, nh, seed=1900 + case)
        v = synthetic_real(B, nv, seed=1950 + case) if real else synthetic_binary(B, nv, seed=1950 + case, p=0.4)
        chain0 = synthetic_binary(B, nv, seed=1975 + case, p=0.5) if pcd else None
        e = _engine(W, b_h, b_v, gpu_device)
        vd = _dm(v, gpu_device)
        rng = O.Rng(case, 7)
        out = e.half_step_bf16("vh", vd, B, 0, 1, case, 0, 7, pieces=3)
        check_half_step(out, *O.sample_hidden(v, W, b_h, rng, 0))
        h = out["sample"].to_numpy()
        out2 = e.half_step_bf16("hv", out["sample"], B, 0, 1, case, 1, 7, pieces=3)
        check_half_step(out2, *O.sample_visible(h, W, b_v, rng, 1))
        cd = _dm(chain0, gpu_device) if pcd else None
        d = _gpu_cd_delta(e, vd, B, 0.01, case, 3, k=k, v_chain=cd, compute="x3")
        _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.01, case, 3, k=k, v_chain=chain0)
        dW, dbh, dbv = _split(d, nv, nh)
        tag = (case, B, nv, nh, k, real, pcd)
        # the bias sums against float64 sums of the oracle's states (numpy's float32 row-by-row sum of 300
        # probabilities is itself ~1e-4 off; the kernels add the 64-row partials in double)
        dbh64 = ch["h_pos"].astype(np.float64).sum(0) - ch["h_neg"].astype(np.float64).sum(0)
        dbv64 = v.astype(np.float64).sum(0) - ch["v_neg"].astype(np.float64).sum(0)
        assert rel_err(dW, dW_ref) <= TOL and rel_err(dbh, dbh64) <= TOL and rel_err(dbv, dbv64) <= TOL, tag
        if pcd:
            assert np.array_equal(cd.to_numpy(), ch["v_neg"]), tag
        # in-place apply (slab reduce writes W and its pieces) == emit + data-parallel apply (fused rewrite)
        e2 = _engine(W, b_h, b_v, gpu_device)
        e2.cd_step(vd, B, 0, 0.01, case, 3, k=k, v_chain=_dm(chain0, gpu_device) if pcd else None, compute="x3")
        e.apply_delta(0.01, compute="x3")
        for x, y in zip(e.get_weights(), e2.get_weights()):
            assert np.max(np.abs(x - y)) <= 1e-6, tag
        p1 = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=3)["prob"].to_numpy()      # reads the rewritten pieces
        p2 = e2.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=3)["prob"].to_numpy()
        assert np.max(np.abs(p1 - p2)) <= 1e-6, tag


def test_x3_transform_surface(gpu_device):
    """transform / inv_transform on a large Bernoulli-mode input run on the x3 kernels: the samples equal those of the
    fp32 MFMA kernels (same counters) except where |u - p| is inside the rounding band."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    nv, nh, N = 96, 80, 2048
    W0 = synthetic_params(nv, nh, seed=1300)
    V = synthetic_binary(N, nv, seed=1301, p=0.3)
    a = RBM({"batch_size": 64, "epochs": 1, "lr": 0.01}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=3, weights=W0, compute_dtype="fp32")
    b = RBM({"batch_size": 64, "epochs": 1, "lr": 0.01}, nh, mode=MODE_VISIBLE_BERNOULLI, seed=3, weights=W0)
    Ha, Hb = a.transform(V)[0], b.transform(V)[0]
    assert Ha.shape == Hb.shape == (N, nh) and set(np.unique(Hb)) <= {0.0, 1.0}
    assert (Ha != Hb).sum() <= 8          # borderline draws only
    rng = O.Rng(3, 0)
    _, _, h_ref = O.sample_hidden(V, *W0[:2], rng, 0x100)
    assert (Hb != h_ref).sum() <= 8
    Va, Vb = a.inv_transform(Ha)[0], b.inv_transform(Ha)[0]
    assert (Va != Vb).sum() <= 8
    # Gaussian visibles: relu-threshold hidden draws of real-valued data, N(loc, 1) visibles
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_GAUSSIAN
    Vr = synthetic_real(N, nv, seed=1302)
    a = RBM({"batch_size": 64, "epochs": 1, "lr": 0.01}, nh, mode=MODE_VISIBLE_GAUSSIAN, seed=3, weights=W0, compute_dtype="fp32")
    b = RBM({"batch_size": 64, "epochs": 1, "lr": 0.01}, nh, mode=MODE_VISIBLE_GAUSSIAN, seed=3, weights=W0)
    Ha, Hb = a.transform(Vr)[0], b.transform(Vr)[0]
    assert (Ha != Hb).sum() <= 8
    Va, Vb = a.inv_transform(Ha)[0], b.inv_transform(Ha)[0]
    assert np.max(np.abs(Va - Vb)) <= TOL


@pytest.mark.parametrize("cfg", [dict(B=4096, nv=784, nh=1024, gauss=True), dict(B=4096, nv=784, nh=1024, gauss=False),
                                 dict(B=300, nv=260, nh=136, gauss=True), dict(B=200, nv=300, nh=140, gauss=False),
                                 dict(B=1024, nv=100, nh=1030, gauss=True)])
def test_x3_real_valued_data_walks_agree(gpu_device, ctx_option, cfg):
    """Real-valued data (grey levels: three bf16 pieces per value) on the x3 path, both modes.  Round 4 gave it two new walks --
    the half steps on a three-piece batch as TWO tiles per k position on 128 x 128 tiles (KURBM_X3_PAIR), and the statistics as two
    launches, the positive half as the transposed problem on a byte plane of h_pos^T (KURBM_X3_SPLIT_STATS).  Every combination of
    the two knobs gives the same chain bit for bit (samples are (u < p) of probabilities that differ in the last ulps at most: the
    test holds them to the oracle's tolerance) and statistics that agree with float64 statistics of the oracle's chain to the
    1e-4 bar; the planes the step kept (h_pos^T as bytes, v_neg^T as bytes or pieces) decode to the chain's own states."""
    from keras_unsupervised_amd import _lib
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg["gauss"] else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=2500 + B)
    v = (np.floor(synthetic_real(B, nv, seed=2501 + B) * 256.0) / 255.0).astype(np.float32)      # grey levels k / 255
    _, _, _, ch, _ = O.cd_step_fused(W, b_h, b_v, v, 0.01, 31, 2, k=1, mode=mode)
    got = {}
    for pair in (1, 0):
        for split in (1, 0):
            ctx_option("KURBM_X3_PAIR", pair, 1)
            ctx_option("KURBM_X3_SPLIT_STATS", split, 1)
            e = _engine(W, b_h, b_v, gpu_device)
            vd = _dm(v, gpu_device)
            d = _gpu_cd_delta(e, vd, B, 0.01, 31, 2, mode=mode, compute="x3")
            dW, dbh, dbv = _split(d, nv, nh)
            hp = e.dump_plane(_lib.PLANE_H_POS, vd, B, mode).to_numpy()
            hpT = e.dump_plane(_lib.PLANE_H_POS_T, vd, B, mode).to_numpy()
            vn = e.dump_plane(_lib.PLANE_V_NEG, vd, B, mode).to_numpy()
            vnT = e.dump_plane(_lib.PLANE_V_NEG_T, vd, B, mode).to_numpy()
            hn = e.dump_plane(_lib.PLANE_H_NEG_T, vd, B, mode).to_numpy()
            assert np.array_equal(hp, hpT) and np.array_equal(vn, vnT), (pair, split)
            assert set(np.unique(hp)) <= {0.0, 1.0}
            # the chain against the oracle's (a borderline draw -- |u - p| inside the rounding band -- may flip a unit)
            assert np.mean(hp != ch["h_pos"]) < 1e-4
            if np.array_equal(hp, ch["h_pos"]):
                assert np.max(np.abs(vn - ch["v_neg"])) <= TOL if cfg["gauss"] else np.mean(vn != ch["v_neg"]) < 1e-4
            # the statistics against float64 statistics of the chain this step ran (its own planes, decoded)
            v64, hp64, vn64, hn64 = (x.astype(np.float64) for x in (v, hp, vn, hn))
            dW64, dbh64, dbv64 = O.cd_statistics(dict(v_pos=v64, h_pos=hp64, v_neg=vn64, h_neg=hn64))
            scale = np.abs(v64).T @ hp64 + np.abs(vn64).T @ hn64 + 1.0          # the un-cancelled magnitude (fp32 accumulation bound)
            assert np.max(np.abs(dW - dW64) / scale) <= 4e-6, (pair, split)
            assert np.max(np.abs(0.01 * (dW - dW64))) <= TOL, (pair, split)      # the applied update lr * dW at the 1e-4 bar
            assert rel_err(dbh, dbh64) <= TOL and rel_err(dbv, dbv64) <= TOL, (pair, split)
            got[(pair, split)] = (d, hp, vn, scale)
            continue
            got[(pair, split)] = (d, hp, vn)
    # the chain does not depend on how the statistics are cut; the statistics agree to the order of their fp32 additions
    for pair in (1, 0):
        assert np.array_equal(got[(pair, 1)][1], got[(pair, 0)][1]) and np.array_equal(got[(pair, 1)][2], got[(pair, 0)][2])
        dd = np.abs(got[(pair, 1)][0] - got[(pair, 0)][0])[: nv * nh] / got[(pair, 1)][3].ravel()
        assert np.max(dd) <= 4e-6
    assert np.mean(got[(1, 1)][1] != got[(0, 1)][1]) < 1e-4      # (paired / unpaired walks add the six piece products in another order)


@pytest.mark.parametrize("cfg", [dict(B=30, nv=52, nh=44, k=1), dict(B=200, nv=300, nh=140, k=2),
                                 dict(B=130, nv=96, nh=260, k=1, pcd=True), dict(B=512, nv=784, nh=256, k=1)])
def test_x3_gaussian_visibles_vs_oracle(gpu_device, cfg):
    """MODE_VISIBLE_GAUSSIAN on the x3 path: the negative visibles are real-valued and travel as three exact pieces
    (row-major for the next half step, transposed for the statistics).  Same oracle, same 1e-4 bar as everything else;
    the real-valued sums are compared with float64 statistics of the oracle's chain states."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    mode = O.MODE_VISIBLE_GAUSSIAN
    W, b_h, b_v = synthetic_params(nv, nh, seed=1400 + B)
    v = synthetic_real(B, nv, seed=1401 + B)
    chain0 = synthetic_real(B, nv, seed=1402 + B) if cfg.get("pcd") else None
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(v, gpu_device)
    cd = _dm(chain0, gpu_device) if chain0 is not None else None
    d1 = _gpu_cd_delta(e, vd, B, 0.01, 77, 9, k=k, mode=mode, v_chain=cd, compute="x3")
    _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.01, 77, 9, k=k, mode=mode, v_chain=chain0)
    dW, dbh, dbv = _split(d1, nv, nh)
    dW_ref, dbh_ref, dbv_ref = O.cd_statistics({k_: ch[k_].astype(np.float64) for k_ in ("v_pos", "h_pos", "v_neg", "h_neg")})
    assert rel_err(dW, dW_ref) <= TOL and rel_err(dbh, dbh_ref) <= TOL and rel_err(dbv, dbv_ref) <= TOL
    if cd is not None:
        assert np.max(np.abs(cd.to_numpy() - ch["v_neg"])) <= TOL
        cd = _dm(chain0, gpu_device)
    d2 = _gpu_cd_delta(e, vd, B, 0.01, 77, 9, k=k, mode=mode, v_chain=cd, compute="x3")
    assert np.array_equal(d1.view(np.uint32), d2.view(np.uint32))
    # against the fp32 MFMA kernels on the same counters
    d32 = _gpu_cd_delta(e, vd, B, 0.01, 77, 9, k=k, mode=mode, v_chain=_dm(chain0, gpu_device) if chain0 is not None else None)
    assert rel_err(d1[: nv * nh], d32[: nv * nh]) <= TOL
    # half step hook: h -> v with N(loc, 1) noise
    h = synthetic_binary(B, nh, seed=1403 + B, p=0.5)
    out = e.half_step_bf16("hv", _dm(h, gpu_device), B, 2, 2, 3, 1, 1, pieces=3)
    loc, z, v1 = O.sample_visible(h, W, b_v, O.Rng(3, 1), 1, mode)
    assert np.max(np.abs(out["prob"].to_numpy() - loc)) <= TOL and np.max(np.abs(out["sample"].to_numpy() - v1)) <= TOL


def test_x3_gaussian_persistent_chain_is_not_rounded(gpu_device):
    """Gaussian mode with persistent chains on 0/1 DATA: the data is one bf16 piece, but the chain holds real-valued
    N(loc, 1) draws after the first step and must travel as three pieces.  (A cached "this buffer is bf16-exact" from
    the chain's initial contents would round the fantasy particles to 8 bits: the second step's sums would then be off
    by ~1e-2, not 1e-5.)"""
    B, nv, nh = 256, 200, 136
    mode = O.MODE_VISIBLE_GAUSSIAN
    W, b_h, b_v = synthetic_params(nv, nh, seed=2100)
    v = synthetic_binary(B, nv, seed=2101, p=0.3)
    e = _engine(W, b_h, b_v, gpu_device)
    vd, cd = _dm(v, gpu_device), _dm(v, gpu_device)              # chain starts as a copy of the (bf16-exact) data
    assert e.v_pieces(vd) == 1
    e.cd_step(vd, B, 0, 1e-3, 7, 0, mode=mode, v_chain=cd, apply=False, emit_delta=True, compute="x3")
    chain1 = cd.to_numpy().copy()
    assert not np.array_equal(chain1, O.bf16_round(chain1))      # real-valued now
    d2 = _gpu_cd_delta(e, vd, B, 1e-3, 7, 1, mode=mode, v_chain=cd, compute="x3")
    _, _, _, ch, _ = O.cd_step_fused(W, b_h, b_v, v, 1e-3, 7, 1, mode=mode, v_chain=chain1)
    dW_ref, dbh_ref, dbv_ref = O.cd_statistics({k_: ch[k_].astype(np.float64) for k_ in ("v_pos", "h_pos", "v_neg", "h_neg")})
    dW, dbh, dbv = _split(d2, nv, nh)
    assert rel_err(dW, dW_ref) <= TOL and rel_err(dbh, dbh_ref) <= TOL and rel_err(dbv, dbv_ref) <= TOL
    assert np.max(np.abs(cd.to_numpy() - ch["v_neg"])) <= TOL


@pytest.mark.parametrize("shape", [(6, 64, 48), (150, 300, 200), (1024, 784, 1024), (1100, 130, 257)])
def test_x3_free_energy_vs_oracle(gpu_device, shape):
    """F(v) with the v.W product on the bf16 pieces (kurbm_free_energy_x3): same bar as the fp32 MFMA kernel, for 0/1
    rows (one piece) and real-valued rows (three pieces), incl. a row large enough to overflow a literal softplus."""
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=1500 + B)
    e = _engine(W, b_h, b_v, gpu_device)
    args64 = [a.astype(np.float64) for a in (W, b_h, b_v)]
    for v in (synthetic_binary(B, nv, seed=1501 + B, p=0.4), synthetic_real(B, nv, seed=1502 + B)):
        if v.max() > 1.0:
            v[B - 1] *= 4000.0
        vd = _dm(v, gpu_device)
        F = e.free_energy(vd, B, compute="x3").cpu().numpy()
        F32 = e.free_energy(vd, B).cpu().numpy()
        ref = O.free_energy(v.astype(np.float64), *args64)
        assert np.all(np.isfinite(F))
        assert rel_err(F, ref) <= TOL
        assert rel_err(F, F32) <= TOL
    # a row window of a larger matrix
    v = synthetic_real(B + 70, nv, seed=1503 + B)
    F = e.free_energy(_dm(v, gpu_device), B, 64, compute="x3").cpu().numpy()
    assert rel_err(F, O.free_energy(v[64:64 + B].astype(np.float64), *args64)) <= TOL


@pytest.mark.parametrize("mode_name", ["bernoulli", "gaussian"])
def test_x3_score_matches_fp32_score(gpu_device, mode_name, capsys):
    """fit(verbose=1) at batch >= 1024 scores on the x3 kernels: same draws as the fp32 MFMA scoring, so the printed
    score agrees to the rounding of a mean of |F - F'| (borderline draws may flip a handful of units)."""
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, RBM
    mode = MODE_VISIBLE_BERNOULLI if mode_name == "bernoulli" else MODE_VISIBLE_GAUSSIAN
    nv, nh, N = 96, 80, 2048
    W0 = synthetic_params(nv, nh, seed=1600)
    V = synthetic_binary(N, nv, seed=1601, p=0.3) if mode_name == "bernoulli" else synthetic_real(N, nv, seed=1602)
    hps = {"batch_size": 1024, "epochs": 1, "lr": 0.01}
    a = RBM(hps, nh, mode=mode, seed=5, weights=W0, compute_dtype="fp32")
    b = RBM(hps, nh, mode=mode, seed=5, weights=W0)
    a.fit(V, verbose=1)
    b.fit(V, verbose=1)
    capsys.readouterr()
    assert len(a.last_scores) == len(b.last_scores) == 2
    for sa, sb in zip(a.last_scores, b.last_scores):
        assert abs(sa - sb) <= 1e-2 * max(1.0, abs(sa)), (sa, sb)
    Fa, Fb = a.cal_free_energy(V)[0], b.cal_free_energy(V)[0]
    assert rel_err(Fb, Fa) <= 5 * TOL      # the two fits differ by fp32 rounding of the updates


@pytest.mark.parametrize("cfg", [dict(B=4096, nv=784, nh=1024, gauss=False), dict(B=512, nv=784, nh=200, gauss=False),
                                 dict(B=200, nv=300, nh=140, gauss=True), dict(B=2048, nv=200, nh=1000, gauss=True)])
def test_x3_step_does_not_read_uninitialised_workspace(gpu_device, cfg):
    """The C ABI promises nothing about the workspace's contents: a step on a workspace (and weight mirror) full of 0xFF
    bytes (bf16 NaNs in every k padding the kernels do not write themselves) gives the bits a zeroed workspace gives."""
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg["gauss"] else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=1700 + B)
    v = synthetic_real(B, nv, seed=1701 + B) if cfg["gauss"] else synthetic_binary(B, nv, seed=1701 + B, p=0.3)
    got = []
    for fill in (0, 0xFF):
        e = _engine(W, b_h, b_v, gpu_device)
        vd = _dm(v, gpu_device)
        e.workspace_bf16(B, 1, 3, e.v_pieces(vd)).fill_(fill)
        e.mirror(3).fill_(fill)
        e._mirrors[3][1] = True        # stale: the library rewrites it from W (pads included) before the step
        e.cd_step(vd, B, 0, 1e-3, 9, 0, mode=mode, compute="x3")
        e.cd_step(vd, B, 0, 1e-3, 9, 1, mode=mode, compute="x3")
        got.append([x.copy() for x in e.get_weights()])
    for a, b in zip(*got):
        assert np.all(np.isfinite(b))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(4096, 784, 1024), (200, 300, 140)])
def test_step_does_not_read_uninitialised_workspace(gpu_device, compute, shape):
    """Same promise on the fp32 MFMA path and the rounded-bf16 path: 0xFF-filled scratch gives the bits zeroed scratch gives."""
    B, nv, nh = shape
    W, b_h, b_v = synthetic_params(nv, nh, seed=1800 + B)
    v = synthetic_binary(B, nv, seed=1801 + B, p=0.3)
    got = []
    for fill in (0, 0xFF):
        e = _engine(W, b_h, b_v, gpu_device)
        vd = _dm(v, gpu_device)
        if compute == "fp32":
            e.workspace(B).fill_(fill)
        else:
            e.workspace_bf16(B, 1, 1, 1).fill_(fill)
            e.mirror(1).fill_(fill)
            e._mirrors[1][1] = True
        for step in range(2):
            e.cd_step(vd, B, 0, 1e-3, 9, step, compute=compute)
        got.append([x.copy() for x in e.get_weights()])
    for a, b in zip(*got):
        assert np.all(np.isfinite(b))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("tall", ["0", "1"])
@pytest.mark.parametrize("shape", [(256, 64, 64), (512, 100, 500), (1024, 784, 200), (768, 200, 136), (2048, 130, 1024),
                                   (1300, 784, 257), (4096, 1024, 784), (4096, 784, 1024)])
def test_x3_half_steps_tile_configs(gpu_device, ctx_option, tall, shape):
    """The x3 half steps on forced 128 x 128 and forced 256 x 64 tiles (KURBM_X3_TALL; the tall kernels need an even number
    of 128-row tiles and are skipped otherwise), with whatever XCD block factorisation the grid admits: probabilities and
    draws against the oracle, row-major and transposed planes through a second half step on the output."""
    B, nv, nh = shape
    ctx_option("KURBM_X3_TALL", int(tall), -1)
    W, b_h, b_v = synthetic_params(nv, nh, seed=2000 + B)
    e = _engine(W, b_h, b_v, gpu_device)
    rng = O.Rng(5, 3)
    for v in (synthetic_binary(B, nv, seed=2001 + B, p=0.3), synthetic_real(B, nv, seed=2002 + B)):
        out = e.half_step_bf16("vh", _dm(v, gpu_device), B, 0, 1, 5, 2, 3, pieces=3)
        check_half_step(out, *O.sample_hidden(v, W, b_h, rng, 2))
    h = synthetic_binary(B, nh, seed=2003 + B, p=0.5)
    out = e.half_step_bf16("hv", _dm(h, gpu_device), B, 0, 1, 5, 3, 3, pieces=3)
    check_half_step(out, *O.sample_visible(h, W, b_v, rng, 3))
    # a whole step on the same tiles: updates within the fp32 bar of the fp32 MFMA path's (same draws; a borderline
    # draw may flip a unit, which moves single entries of dW by lr * 1)
    v = synthetic_binary(B, nv, seed=2004 + B, p=0.3)
    a, b = _engine(W, b_h, b_v, gpu_device), _engine(W, b_h, b_v, gpu_device)
    a.cd_step(_dm(v, gpu_device), B, 0, 1e-3, 11, 0, compute="x3")
    b.cd_step(_dm(v, gpu_device), B, 0, 1e-3, 11, 0, compute="fp32")
    dW = np.abs(a.get_weights()[0] - b.get_weights()[0])
    assert np.mean(dW > 1e-6) < 1e-3 and np.all(np.isfinite(a.get_weights()[0]))


def test_bf16_exact_flag_bits(gpu_device):
    """kurbm_bf16_exact: bit 0 = some value is not a bf16 value, bit 1 = some value is neither 0 nor 1 (include/kurbm.h)."""
    from keras_unsupervised_amd import _lib
    e = _engine(*synthetic_params(24, 16, seed=1), gpu_device)
    g = np.random.default_rng(3)
    cases = [(synthetic_binary(130, 24, seed=2, p=0.4), 1 | _lib.V_BINARY, True),
             ((g.integers(0, 256, (130, 24)) / 256.0).astype(np.float32), 1, False),     # grey levels: bf16-exact, not 0/1
             (g.random((130, 24)).astype(np.float32), 3, False)]
    for x, want_pieces, want_binary in cases:
        d = _dm(x, gpu_device)
        assert e._x3_pieces(d, None, O.MODE_VISIBLE_BERNOULLI) == want_pieces
        assert bool(d.binary) == want_binary


@pytest.mark.parametrize("cfg", [dict(B=200, nv=100, nh=80, k=1), dict(B=1024, nv=784, nh=256, k=2),
                                 dict(B=384, nv=260, nh=520, k=1, planes=True), dict(B=256, nv=300, nh=128, k=1, gauss=True)])
def test_x3_fp8_positive_statistics(gpu_device, ctx_option, cfg):
    """0/1 data: v_pos^T h_pos on the fp8 matrix cores (fp8 transposed planes from the conversion kernel / the resident
    planes and from the h_pos epilogue, 128-deep k-tiles) against the same step on bf16 planes (KURBM_X3_F8POS=0) and
    against the oracle: the positive products are counts, exact either way, so the two differ only in the order in which
    the negative phase's fp32 terms meet them."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg.get("gauss") else O.MODE_VISIBLE_BERNOULLI
    W0 = synthetic_params(nv, nh, seed=1300 + B)
    V = synthetic_binary(B, nv, seed=1301 + B, p=0.3)
    got = {}
    for f8 in (1, 0):
        ctx_option("KURBM_X3_F8POS", f8, 1)
        e = _engine(*W0, gpu_device)
        vd = _dm(V, gpu_device)
        planes = e.make_planes(vd, [(0, B)], mode) if cfg.get("planes") else None
        e.cd_step(vd, B, 0, 0.01, 9, 2, k=k, mode=mode, apply=False, emit_delta=True, compute="x3", planes=planes)
        torch.cuda.synchronize()
        got[f8] = _split(e.delta_buffer().cpu().numpy(), nv, nh)
    for a, b in zip(got[1], got[0]):
        assert np.max(np.abs(a - b)) <= 2e-6 * max(1.0, float(np.abs(b).max()))
    assert np.array_equal(got[1][2], got[0][2]) and np.array_equal(got[1][1], got[0][1])   # the bias sums never see the planes
    if not cfg.get("gauss") and k == 1:   # (k > 1 / Gaussian: a flip in the |u - p| < 1e-5 band changes the whole chain)
        dW = O.cd_step_fused(*W0, V, 1.0, 9, 2)[4][0]
        assert np.max(np.abs(got[1][0] - dW)) <= 1e-4 * max(1.0, float(np.abs(dW).max())) or \
            np.mean(np.abs(got[1][0] - dW) > 1e-3) < 0.05   # (a band flip moves one row / column of dW by one count)


@pytest.mark.parametrize("shape", [(4096, 784, 1024), (1024, 1000, 300), (512, 260, 1024), (300, 784, 100)])
def test_paired_walk_half_step_equals_segment_by_segment(gpu_device, ctx_option, shape):
    """A real-valued A operand on the x3 path: the paired walk (KURBM_X3_PAIR=1, the default: two tiles per k position on 128 x 128
    tiles, the six piece pairs of a position together) against the three segments one after the other on the generic walk
    (=0).  The same piece pairs, multiplied exactly, summed in another order: the probabilities agree to fp32 rounding, the
    uniforms are the same numbers, and both sit within the fp32 tolerance of the oracle.  rbm.py:214."""
    B, nv, nh = shape
    W0 = synthetic_params(nv, nh, seed=2500 + B)
    V = synthetic_real(B, nv, seed=2501 + B)
    got = {}
    for share in (1, 0):
        ctx_option("KURBM_X3_PAIR", share, 1)
        e = _engine(*W0, gpu_device)
        out = e.half_step_bf16("vh", _dm(V, gpu_device), B, 0, 1, 5, 2, 3, pieces=3)
        torch.cuda.synchronize()
        got[share] = {k: out[k].to_numpy() for k in ("prob", "u", "sample")}
    assert np.array_equal(got[1]["u"], got[0]["u"])
    assert np.max(np.abs(got[1]["prob"] - got[0]["prob"])) <= 2e-6
    flips = got[1]["sample"] != got[0]["sample"]
    assert np.all(np.abs(got[1]["u"] - got[1]["prob"])[flips] < 1e-5)
    ref = O.sigmoid(V.astype(np.float64) @ W0[0].astype(np.float64) + W0[1].astype(np.float64))
    assert np.max(np.abs(got[1]["prob"] - ref)) <= 1e-5


@pytest.mark.parametrize("cfg", [dict(B=4096, nv=784, nh=1024), dict(B=1024, nv=784, nh=256, xcd2d=0), dict(B=512, nv=300, nh=200),
                                 dict(B=1024, nv=1024, nh=1024, compute="bf16"), dict(B=2048, nv=1024, nh=784, real=True, xcd2d=0)])
def test_block_mapping_by_division_is_the_same_mapping(gpu_device, ctx_option, cfg):
    """k_gemm_pb finds its tile from its block index with multiply-high constants of the launcher (exact while dividend x
    divisor < 2^32); a grid beyond that takes the same mapping by integer division (GemmArgsB::map_slow).  KURBM_MAP_SLOW=1 forces
    that path on ordinary grids, in the XCD-block order and in the linear one: the step must come out bit for bit the same."""
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    compute = cfg.get("compute", "x3")
    if "xcd2d" in cfg:
        ctx_option("KURBM_X3_XCD2D", cfg["xcd2d"], 1)
    W0 = synthetic_params(nv, nh, seed=2400 + B)
    V = synthetic_real(B, nv, seed=2401 + B) if cfg.get("real") else synthetic_binary(B, nv, seed=2401 + B, p=0.3)
    got = {}
    for slow in (1, 0):
        ctx_option("KURBM_MAP_SLOW", slow, 0)
        e = _engine(*W0, gpu_device)
        vd = _dm(V, gpu_device)
        for step in range(2):
            e.cd_step(vd, B, 0, 1e-3, 9, step, compute=compute)
        torch.cuda.synchronize()
        got[slow] = [x.copy() for x in e.get_weights()]
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(got[1][0])) and not np.array_equal(got[1][0], W0[0])


@pytest.mark.parametrize("cfg", [dict(B=4096, nv=784, nh=1024), dict(B=4096, nv=784, nh=1024, planes=True, steps=3),
                                 dict(B=1024, nv=784, nh=256), dict(B=512, nv=300, nh=200), dict(B=384, nv=260, nh=70),
                                 dict(B=640, nv=784, nh=1024, k=2, persistent=True), dict(B=4096, nv=784, nh=1024, split=3),
                                 dict(B=4096, nv=784, nh=1024, split=1), dict(B=2048, nv=1500, nh=333, split=2)])
def test_stats_byte_plane_is_bit_identical(gpu_device, ctx_option, cfg):
    """The statistics GEMM with v_neg^T as a byte plane (KURBM_X3_STATS_BYTES=1, the default wherever the positive half runs on
    fp8 planes: k_gemm_pb<..., EPI_SLAB, ..., AB>, its own tile schedule) against the same launch on the bf16 plane: bytes are
    2.0 behind a zero low byte, the fp8 tile is scaled by 2 through its exponent and the slab epilogue halves -- powers of two,
    the same products summed in the same order -- so W, b_h, b_v and the weight-piece mirror are bit for bit the same.
    rbm.py:125-134."""
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    steps = cfg.get("steps", 2)
    mode = O.MODE_VISIBLE_BERNOULLI
    if "split" in cfg:
        ctx_option("KURBM_BF16_SPLIT", cfg["split"], -1)
    W0 = synthetic_params(nv, nh, seed=2300 + B)
    V = synthetic_binary(B, nv, seed=2301 + B, p=0.3)
    got = {}
    for byt in (1, 0):
        ctx_option("KURBM_X3_STATS_BYTES", byt, 1)
        e = _engine(*W0, gpu_device)
        vd = _dm(V, gpu_device)
        e.workspace_bf16(B, cfg.get("k", 1), 3, e._x3_pieces(vd, None, mode)).fill_(0xFF if byt else 0)
        chain = None
        if cfg.get("persistent"):
            from keras_unsupervised_amd.ebm.engine import DeviceMatrix
            chain = DeviceMatrix.from_host(synthetic_binary(B, nv, seed=2302 + B, p=0.5), gpu_device)
        planes = e.make_planes(vd, [(0, B)], mode, chain) if cfg.get("planes") else None
        for step in range(steps):
            e.cd_step(vd, B, 0, 1e-3, 9, step, k=cfg.get("k", 1), mode=mode, compute="x3", planes=planes, v_chain=chain)
        torch.cuda.synchronize()
        got[byt] = [x.copy() for x in e.get_weights()] + [e._mirrors[3][0].cpu().numpy().copy()]
    for a, b in zip(got[1], got[0]):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(got[1][0])) and not np.array_equal(got[1][0], W0[0])


@pytest.mark.parametrize("cfg", [dict(B=300, nv=784, nh=256, k=1, chunks=(1, 2, 3)), dict(B=260, nv=1100, nh=200, k=2, pcd=True, chunks=(1, 3)),
                                 dict(B=256, nv=2048, nh=2048, k=1, chunks=(0, 4), auto=2), dict(B=128, nv=300, nh=150, k=1, chunks=(2,))])
def test_bf16_dp_step_one_call(gpu_device, one_rank_comm, cfg):
    """kurbm_cd_step_bf16_dp (the data-parallel step of BASELINE.json config 5's path): chain, statistics in row ranges of dW,
    range i all-reduced AND applied (its rows of W, its part of the weight mirror) on the library's comm stream while range
    i + 1 is multiplied -- against the plain sequence emit -> all-reduce -> apply.  One range: bit-identical; several: the
    split-K slicing of a row range differs, so dW agrees to the order of the fp32 additions.  n_chunks = 0 is the
    library's automatic choice: ONE range whatever the size (ranges are an opt-in until they have run on a real node).  n_hid % 4 != 0 takes the unfused apply.  The rewritten mirror is the new weights."""
    B, nv, nh, k = cfg["B"], cfg["nv"], cfg["nh"], cfg["k"]
    W0 = synthetic_params(nv, nh, seed=2300 + B)
    V = synthetic_binary(B, nv, seed=2301 + B, p=0.3)
    chain0 = synthetic_binary(B, nv, seed=2302 + B, p=0.5) if cfg.get("pcd") else None
    vd = _dm(V, gpu_device)
    ref = _engine(*W0, gpu_device)
    cr = _dm(chain0, gpu_device) if chain0 is not None else None
    ref.cd_step(vd, B, 0, 0.01, 5, 3, k=k, apply=False, emit_delta=True, row0=8, v_chain=cr, compute="bf16")
    d_ref = ref.delta_buffer().clone()
    ref.apply_delta(0.01)
    for n_chunks in cfg["chunks"]:
        e = _engine(*W0, gpu_device)
        cd = _dm(chain0, gpu_device) if chain0 is not None else None
        e.cd_step_dp(one_rank_comm, vd, B, 0, 0.01, 5, 3, k=k, row0=8, v_chain=cd, compute="bf16", n_chunks=n_chunks)
        torch.cuda.synchronize()
        if cd is not None:
            assert np.array_equal(cd.to_numpy(), cr.to_numpy())
        got, want = e.delta_buffer().cpu().numpy(), d_ref.cpu().numpy()
        dW, dbh, dbv = _split(got, nv, nh)
        rW, rbh, rbv = _split(want, nv, nh)
        assert np.array_equal(dbv.view(np.uint32), rbv.view(np.uint32)) and np.array_equal(dbh.view(np.uint32), rbh.view(np.uint32))
        if n_chunks in (0, 1):
            assert np.array_equal(dW.view(np.uint32), rW.view(np.uint32))
        else:
            assert np.max(np.abs(dW - rW)) <= 1e-4 * max(1.0, float(np.abs(rW).max()))
        for x, y in zip(e.get_weights(), ref.get_weights()):
            assert np.max(np.abs(x - y)) <= 1e-6 * max(1.0, float(np.abs(y).max()))
        # the mirror the library rewrote (per range, on its comm stream) is the rounded image of the new weights
        stale = _engine(*e.get_weights(), gpu_device)
        p_fresh = stale.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=1)["prob"].to_numpy()
        p_kept = e.half_step_bf16("vh", vd, B, 0, 0, 0, 0, 0, pieces=1)["prob"].to_numpy()
        assert np.array_equal(p_fresh, p_kept)
        h = synthetic_binary(B, nh, seed=2303 + B, p=0.5)
        q_fresh = stale.half_step_bf16("hv", _dm(h, gpu_device), B, 0, 0, 0, 0, 0, pieces=1)["prob"].to_numpy()
        q_kept = e.half_step_bf16("hv", _dm(h, gpu_device), B, 0, 0, 0, 0, 0, pieces=1)["prob"].to_numpy()
        assert np.array_equal(q_fresh, q_kept)
    # a rank that owns no rows of the batch joins with zeros
    e = _engine(*W0, gpu_device)
    e.cd_step_dp(one_rank_comm, vd, 0, 0, 0.01, 5, 3, k=k, compute="bf16", n_chunks=cfg["chunks"][-1])
    torch.cuda.synchronize()
    assert float(e.delta_buffer().abs().max().item()) == 0.0
    for x, y in zip(e.get_weights(), W0):
        assert np.array_equal(x, y)
    # a refused call returns before any collective was enqueued (every check precedes the first all-reduce)
    from keras_unsupervised_amd import _lib
    with pytest.raises(_lib.KurbmError, match="k must be"):
        e.cd_step_dp(one_rank_comm, vd, B, 0, 0.01, 5, 3, k=99, compute="bf16")
    with pytest.raises(_lib.KurbmError, match="k must be"):
        e.cd_step_dp(one_rank_comm, vd, B, 0, 0.01, 5, 3, k=99, compute="x3")


@pytest.mark.parametrize("cfg", [dict(B=4096, nv=784, nh=1024), dict(B=4096, nv=784, nh=1024, real=True), dict(B=1024, nv=300, nh=200, gauss=True),
                                 dict(B=1100, nv=260, nh=136), dict(B=2048, nv=784, nh=256, planes=True), dict(B=4096, nv=20, nh=16)])
def test_score_one_call_vs_oracle(gpu_device, cfg):
    """kurbm_score_x3 -- F(v), a fresh one-step reconstruction v', F(v'), mean |F - F'| into a device float, one library call,
    no host synchronisation (rbm.py:225-233; what fit(verbose=1) prints every step) -- against the oracle's step_score with
    the same counters; resident planes change nothing; the free energies it can also return against O.free_energy."""
    import ctypes as C
    from keras_unsupervised_amd._lib import CdOpts, check
    from keras_unsupervised_amd.ebm.engine import CHAIN_SCORE
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg.get("gauss") else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=2400 + B)
    V = synthetic_real(B, nv, seed=2401 + B) if (cfg.get("real") or cfg.get("gauss")) else synthetic_binary(B, nv, seed=2401 + B, p=0.3)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(V, gpu_device)
    planes = e.make_planes(vd, [(0, B)], mode) if cfg.get("planes") else None
    s = e.score_x3(vd, B, 0, 7, 3, mode, CHAIN_SCORE, planes=planes)
    s2 = e.score_x3(vd, B, 0, 7, 3, mode, CHAIN_SCORE)
    torch.cuda.synchronize()
    got = float(s[0].item())
    assert got == float(s2[0].item())                                  # same counters -> same draws -> same bits
    want = O.step_score(W, b_h, b_v, V, 7, 3, mode)
    assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (got, want)
    # the free energies themselves (F(v) is a function of v alone; F(v') follows the reconstruction)
    vp = e._x3_pieces(vd, None, mode)
    mir, ws = e.mirror(3), e.workspace_bf16(B, 1, 3, vp)
    opts = CdOpts(1, int(mode), 0.0, 0, None, None, 7, 0, 3, CHAIN_SCORE)
    F = torch.empty(2 * B, dtype=torch.float32, device=gpu_device)
    out = torch.empty(4, dtype=torch.float32, device=gpu_device)
    check(e.lib.kurbm_score_x3(e.ctx.handle, C.byref(e.params), mir.data_ptr(), mir.numel(), vd.ptr(), vp, B, vd.ld, C.byref(opts),
                               out.data_ptr(), F.data_ptr(), ws.data_ptr(), ws.numel(), e._stream()))
    torch.cuda.synchronize()
    Fh = F.cpu().numpy()
    assert rel_err(Fh[:B], O.free_energy(V, W, b_h, b_v)) <= TOL
    assert abs(float(np.mean(np.abs(Fh[:B].astype(np.float64) - Fh[B:].astype(np.float64)))) - got) <= 1e-5 * max(1.0, abs(got))
    assert float(out[0].item()) == got


@pytest.mark.parametrize("local", [1, 0])
@pytest.mark.parametrize("cfg", [dict(B=128, nv=784, nh=128), dict(B=64, nv=784, nh=256, gauss=True), dict(B=128, nv=784, nh=128, real=True, gauss=True),
                                 dict(B=37, nv=100, nh=33), dict(B=200, nv=300, nh=520, gauss=True), dict(B=16, nv=784, nh=1024), dict(B=5, nv=3, nh=2)])
def test_small_score_vs_oracle(gpu_device, ctx_option, cfg, local):
    """kurbm_score_small -- the score of fit(verbose=1) for a small RBM in ONE launch (csrc/kurbm_small.hip: k_score_small), both
    schedules -- against the oracle's step_score with the same counters, the free energies against O.free_energy, twice the same
    bits, and against the five-call form the host class used before (free energy, two half steps, free energy)."""
    from keras_unsupervised_amd.ebm.engine import CHAIN_SCORE
    ctx_option("KURBM_SMALL_LOCAL", local, 1)
    B, nv, nh = cfg["B"], cfg["nv"], cfg["nh"]
    mode = O.MODE_VISIBLE_GAUSSIAN if cfg.get("gauss") else O.MODE_VISIBLE_BERNOULLI
    W, b_h, b_v = synthetic_params(nv, nh, seed=2700 + B)
    V = synthetic_real(B, nv, seed=2701 + B) if cfg.get("real") else synthetic_binary(B, nv, seed=2701 + B, p=0.3)
    e = _engine(W, b_h, b_v, gpu_device)
    vd = _dm(V, gpu_device)
    s, F = e.score_small(vd, B, 0, 7, 3, mode, CHAIN_SCORE, want_F=True)
    s2 = e.score_small(vd, B, 0, 7, 3, mode, CHAIN_SCORE)
    torch.cuda.synchronize()
    e.check_status()
    got = float(s[0].item())
    assert np.isfinite(got) and got == float(s2[0].item())
    want = O.step_score(W, b_h, b_v, V, 7, 3, mode)
    assert abs(got - want) <= 1e-3 * max(1.0, abs(want)), (got, want)
    Fh = F.cpu().numpy()
    assert rel_err(Fh[0], O.free_energy(V, W, b_h, b_v)) <= TOL
    assert abs(float(np.mean(np.abs(Fh[0].astype(np.float64) - Fh[1].astype(np.float64)))) - got) <= 1e-5 * max(1.0, abs(got))
    # a training step in between leaves the barrier state the next score starts from
    e.cd_step(vd, B, 0, 1e-3, 9, 0, mode=mode, compute="small")
    s3 = e.score_small(vd, B, 0, 7, 4, mode, CHAIN_SCORE)
    torch.cuda.synchronize()
    e.check_status()
    Wn, bhn, bvn = e.get_weights()
    want3 = O.step_score(Wn, bhn, bvn, V, 7, 4, mode)
    assert abs(float(s3[0].item()) - want3) <= 1e-3 * max(1.0, abs(want3))
