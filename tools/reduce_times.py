#!/usr/bin/env python3
"""Stage 5 of the x3 CD-1 step alone (slab reduce + W update + weight-piece mirror + bias sums, one launch),
config 2.  KURBM_REDUCE_TR=16|32|64 forces the tile height."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
lr = 1e-3 / B
for _ in range(3):
    eng.cd_step(V, B, 0, lr, 42, 0, compute="x3")


def t(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("KURBM_REDUCE_TR=%s: conversion %.2f us, reduce + mirror %.2f us, whole step %.1f us"
      % (os.environ.get("KURBM_REDUCE_TR", "auto"), t(lambda: eng.cd_step_x3_stage(V, B, 0, lr, 42, 0, 0)),
         t(lambda: eng.cd_step_x3_stage(V, B, 0, lr, 42, 0, 5)), t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3"))))
