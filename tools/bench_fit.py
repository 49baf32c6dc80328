#!/usr/bin/env python3
"""RBM.fit(verbose=0) wall time per step, 784 x 1024, Bernoulli mode, over batch sizes and compute paths:
where should compute_dtype='auto' switch from the fp32 MFMA kernels to x3?"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ku.ebm import RBM  # noqa: E402

V = (np.random.default_rng(0).random((32768, 784)) < 0.19).astype(np.float32)
Vd = torch.from_numpy(V).cuda()
for bs in (256, 512, 1024, 2048, 4096):
    line = "batch %5d:" % bs
    for c in ("fp32", "x3"):
        rbm = RBM({"batch_size": bs, "epochs": 1, "lr": 1e-3 / bs}, 1024, mode=0, compute_dtype=c)
        rbm.fit(Vd[: 4 * bs], verbose=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rbm.fit(Vd, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = -(-len(V) // bs)
        line += "  %s %7.1f us/step (%6.2f Mrows/s)" % (c, dt / steps * 1e6, len(V) / dt / 1e6)
    print(line)
