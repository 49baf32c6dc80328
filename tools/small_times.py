#!/usr/bin/env python3
"""The one-launch CD-1 step of small RBMs (kurbm_cd_step_small) against the five-launch fp32 path: microseconds per step by HIP
events (steps queued back to back through the library's epoch call, so the host is not in the way), and RBM.fit wall time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import RBM  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

dev = torch.device("cuda", 0)
g = np.random.default_rng(0)
for bs, nh in ((128, 128), (64, 128), (64, 256), (128, 256), (16, 128), (200, 512)):
    nv, steps = 784, 200
    V = DeviceMatrix.from_host((g.random((bs * steps, nv)) < 0.19).astype(np.float32), dev)
    out = []
    for compute in ("small", "fp32"):
        eng = DeviceRBM(g.uniform(-0.05, 0.05, (nv, nh)).astype(np.float32), np.zeros(nh, np.float32), np.zeros(nv, np.float32), dev)
        eng.cd_epoch(V, bs * 20, bs, 1e-3 / bs, 1, 0, compute=compute)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        eng.cd_epoch(V, bs * steps, bs, 1e-3 / bs, 1, 0, compute=compute)
        b.record()
        host = (time.perf_counter() - t0) / steps * 1e6
        torch.cuda.synchronize()
        out.append("%s %6.1f us/step (host enqueue %5.1f)" % (compute, a.elapsed_time(b) / steps * 1e3, host))
        eng.check_status()
    print("784 x %3d, batch %3d: " % (nh, bs) + "   ".join(out), flush=True)
V = (g.random((42000, 784)) < 0.19).astype(np.float32)
for bs, nh in ((128, 128), (64, 256)):
    for compute in ("auto", "fp32"):
        rbm = RBM({"batch_size": bs, "epochs": 1, "lr": 1e-3 / bs}, nh, mode=0, compute_dtype=compute)
        rbm.fit(V[: 4 * bs], verbose=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rbm.fit(V, verbose=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = -(-len(V) // bs)
        print("RBM.fit 784 x %d batch %d compute_dtype=%s: %.1f us/step wall (incl. the 132 MB upload)" % (nh, bs, compute, dt / steps * 1e6), flush=True)
