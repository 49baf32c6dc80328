#!/usr/bin/env python3
"""v -> h half step over a whole data set (DBN inter-layer transform, rbm.py:88-89): fp32 MFMA kernel vs the x3 hook."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

N, NV, NH = 65536, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((torch.rand(N, NV, device=dev) < 0.19).float(), dev)


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("transform of %d rows: fp32 MFMA %.3f ms | x3 hook %.3f ms"
      % (N, t(lambda: eng.half_step("vh", V, N, 0, 0, 1, 1, 0x100, 0)), t(lambda: eng.half_step_bf16("vh", V, N, 0, 1, 1, 0x100, 0, pieces=3))))
