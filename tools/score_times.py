#!/usr/bin/env python3
"""The per-step score of fit(verbose=1) (rbm.py:225-233: F(v), a one-step reconstruction v', F(v')) and
cal_free_energy over a data set: fp32 MFMA kernels vs the x3 kernels, 784 x 1024."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix  # noqa: E402

N, NV, NH, B = 65536, 784, 1024, 4096
dev = torch.device("cuda", 0)
V = DeviceMatrix.from_host((torch.rand(N, NV, device=dev) < 0.19).float(), dev)


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for compute in ("fp32", "auto"):
    r = RBM({"batch_size": B, "epochs": 1, "lr": 0.01}, NH, mode=MODE_VISIBLE_BERNOULLI, seed=1, compute_dtype=compute)
    r.build((None, NV))
    d = r._dev
    x3 = "x3" if compute == "auto" else None
    print("%-5s score (batch %d): %.3f ms | F(v) of one batch: %.3f ms | F(v) of %d rows: %.3f ms"
          % (compute, B, t(lambda: r._score(V, 0, B, 0)), t(lambda: d.free_energy(V, B, 0, compute=x3)),
             N, t(lambda: d.free_energy(V, N, 0, compute=x3), 5)))
