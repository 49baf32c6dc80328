"""Random shapes, data kinds and modes: one CD step on the x3 path against the same step on the fp32 MFMA kernels -- an
independent implementation on the GPU (other kernels, other tilings, the same Philox uniforms: rbm.py:113-134).  The
probabilities of the two agree to fp32 rounding, so a sample flips only inside the |u - p| band and the updates agree except
for the rare entries such a flip moves by one count.  (tools/fuzz_shapes.py is the long form.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_x3_step_equals_fp32_step_on_random_shapes(gpu_device, seed):
    from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM
    g = np.random.default_rng(seed)
    for _ in range(6):
        B = int(g.choice([64, 100, 200, 256, 300, 384, 640, 777, 1024, 1500]))
        nv, nh = int(g.integers(40, 1400)), int(g.integers(40, 1400))
        real = bool(g.integers(0, 2))
        mode = int(g.integers(0, 2)) if real else 0
        k = int(g.choice([1, 1, 2]))
        persistent = bool(g.integers(0, 2)) and mode == 0
        W0 = g.uniform(-0.1, 0.1, (nv, nh)).astype(np.float32)
        bh = g.uniform(-0.1, 0.1, nh).astype(np.float32)
        bv = g.uniform(-0.1, 0.1, nv).astype(np.float32)
        Vh = ((np.floor(g.random((B, nv)) * 256.0) / 255.0) if real else (g.random((B, nv)) < 0.3)).astype(np.float32)
        Ch = (np.random.default_rng(7).random((B, nv)) < 0.5).astype(np.float32)
        out = {}
        for compute in ("x3", "fp32"):
            e = DeviceRBM(W0, bh, bv, gpu_device)
            chain = DeviceMatrix.from_host(Ch, gpu_device) if persistent else None
            e.cd_step(DeviceMatrix.from_host(Vh, gpu_device), B, 0, 1.0, 11, 3, k=k, mode=mode, compute=compute, v_chain=chain)
            torch.cuda.synchronize()
            out[compute] = [x.copy() for x in e.get_weights()]
        what = "B %d nv %d nh %d real %d mode %d k %d pcd %d" % (B, nv, nh, real, mode, k, persistent)
        scale = max(1.0, float(np.abs(out["fp32"][0] - W0).max()))
        assert np.all(np.isfinite(out["x3"][0])), what
        assert float(np.mean(np.abs(out["x3"][0] - out["fp32"][0]) > 1e-3 * scale)) < 0.02, what
        for a, b in zip(out["x3"][1:], out["fp32"][1:]):
            assert float(np.mean(np.abs(a - b) > 1e-3 * scale)) < 0.05, what
