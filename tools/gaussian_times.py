#!/usr/bin/env python3
"""Config-2 CD-1 step with Gaussian visibles (the reference's default mode, real-valued data): fp32 MFMA vs x3."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

dev = torch.device("cuda", 0)
B, NV, NH = 4096, 784, 1024
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host(g.random((B, NV)).astype(np.float32), dev)


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("Gaussian visibles, real-valued data: fp32 MFMA %.1f us/step | x3 %.1f us/step"
      % (t(lambda: eng.cd_step(V, B, 0, 1e-7, 42, 0, mode=1)), t(lambda: eng.cd_step(V, B, 0, 1e-7, 42, 0, mode=1, compute="x3"))))
