#!/usr/bin/env python3
"""GPU time (HIP events) of the data-parallel x3 step variants on one GPU, no process group (all-reduces are no-ops)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import dp  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
lr = 1e-3 / B


def t(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def one_shot():
    eng.cd_step(V, B, 0, lr, 42, 0, apply=False, emit_delta=True, compute="x3")
    eng.apply_delta(lr, compute="x3")


def overlapped():
    dp.x3_sums_overlapped(eng, V, B, 0, lr, 42, 0)
    eng.apply_delta(lr, compute="x3")


print("in-place step %.1f us | emit + apply %.1f us | chain + two row ranges + apply %.1f us | chain alone %.1f us"
      % (t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3")), t(one_shot), t(overlapped),
         t(lambda: eng.cd_chain_x3(V, B, 0, lr, 42, 0))))
print("stats rows [0,384) %.1f us | [384,784) %.1f us | [0,784) %.1f us"
      % (t(lambda: eng.x3_stats_rows(V, B, 0, 0, 384, lr, 42, 0)), t(lambda: eng.x3_stats_rows(V, B, 0, 384, 784, lr, 42, 0)),
         t(lambda: eng.x3_stats_rows(V, B, 0, 0, 784, lr, 42, 0))))
