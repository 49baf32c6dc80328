#!/usr/bin/env python3
"""Soak / determinism check: two independent runs of N CD-1 steps at the headline shape must end
with bit-identical parameters (no atomics, fixed reduction orders), finite, and the training signal
must move the weights.   python tools/soak.py [--steps 3000]"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--compute", default="x3", choices=("x3", "fp32", "bf16"))
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--planes", action="store_true", help="x3: run 1 reads resident data planes, run 0 converts per step -- same bits")
ap.add_argument("--data", default="binary", choices=("binary", "real"), help="real: grey levels k / 255 (the batch travels as three pieces)")
ap.add_argument("--gauss", action="store_true", help="Gaussian visibles (the reference's default mode)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, NV, NH = 4096, 784, 1024
g = np.random.default_rng(1)
W0 = g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32)
torch.manual_seed(0)
V = DeviceMatrix.from_host((torch.rand(8 * B, NV, device=dev) < 0.19).float() if a.data == "binary"
                           else torch.floor(torch.rand(8 * B, NV, device=dev) * 256.0) / 255.0, dev)
mode = 1 if a.gauss else 0
digests = []
for run in range(2):
    eng = DeviceRBM(W0, np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
    planes = eng.make_planes(V, [(j * B, B) for j in range(8)], mode) if (a.planes and run == 1 and a.compute == "x3") else None
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.cd_step(V, B, (i % 8) * B, 1e-3 / B, 42, i, mode=mode, compute=a.compute, planes=planes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    W, bh, bv = eng.get_weights()
    assert np.isfinite(W).all() and np.isfinite(bh).all() and np.isfinite(bv).all()
    digests.append(hashlib.sha256(W.tobytes() + bh.tobytes() + bv.tobytes()).hexdigest())
    print("run %d: %d steps in %.2f s (%.0f steps/s), |W - W0| max %.4f, digest %s"
          % (run, a.steps, dt, a.steps / dt, float(np.abs(W - W0).max()), digests[-1][:16]))
assert digests[0] == digests[1], "runs differ: the path is not deterministic"
print("deterministic: OK")
