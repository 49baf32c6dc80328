#!/usr/bin/env python3
"""Host time to ENQUEUE one x3 CD-1 step (ctypes call + 8 launches) against its GPU time: is the step loop host-bound?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
for B, NV, NH in ((4096, 784, 1024), (64, 784, 256)):
    eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
    V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
    for comp in ("x3", "fp32"):
        for _ in range(20):
            eng.cd_step(V, B, 0, 1e-7, 42, 0, compute=comp)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            eng.cd_step(V, B, 0, 1e-7, 42, i, compute=comp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("B=%d %dx%d %-4s: host enqueue %.1f us/step, until the GPU is done %.1f us/step" % (B, NV, NH, comp, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
