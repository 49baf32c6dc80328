#!/usr/bin/env python3
"""The reference's own example configuration (examples/rbm/rbm_softmax_mnist_conf.json: 784 -> 128,
batch 128, lr 1e-3) on 42 000 synthetic MNIST-sized rows: wall time of RBM.fit for one epoch.
Small-batch, launch-latency-bound regime (BASELINE.json configs[0] family)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ku.ebm import RBM  # noqa: E402

V = (np.random.default_rng(0).random((42000, 784)) < 0.19).astype(np.float32)
for bs in (128, 64, 1024):
    rbm = RBM({"batch_size": bs, "epochs": 1, "lr": 1e-3 / bs}, 128 if bs != 64 else 256, mode=0)
    rbm.fit(V[: 4 * bs], verbose=0)           # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rbm.fit(V, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = -(-len(V) // bs)
    print("784x%d batch %4d: %d steps in %.3f s = %.0f steps/s, %.1f us/step (incl. the %.0f MB upload)"
          % (rbm.output_dim, bs, steps, dt, steps / dt, dt / steps * 1e6, V.nbytes / 1e6))
