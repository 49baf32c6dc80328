#!/usr/bin/env python3
"""Gaussian-visible half step (rbm.py:64-66): max |v_gpu - v_oracle| on the fp32-MFMA and x3 kernels, and the share of it
that is Box-Muller (GPU logf / sqrtf / cospif vs numpy float32 vs a float64 evaluation of the same two uniforms)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402
from oracle import philox, rbm_oracle as O  # noqa: E402
from oracle.make_golden import synthetic_binary, synthetic_params  # noqa: E402

dev = torch.device("cuda", 0)
B, nv, nh = 2048, 784, 1024
W, b_h, b_v = synthetic_params(nv, nh, seed=8)
h = synthetic_binary(B, nh, seed=9, p=0.5)
e = DeviceRBM(W, b_h, b_v, dev)
hd = DeviceMatrix.from_host(h, dev)
loc, z, v1 = O.sample_visible(h, W, b_v, O.Rng(3, 1), 1, O.MODE_VISIBLE_GAUSSIAN)
ua = philox.uniform(B, nv, 3, 1, 1).astype(np.float64)
ub = philox.uniform(B, nv, 3, 1 | 0x80000000, 1).astype(np.float64)
z64 = np.sqrt(-2.0 * np.log(1.0 - ua)) * np.cos(2.0 * np.pi * ub)
loc64 = h.astype(np.float64) @ W.astype(np.float64).T + b_v
print("numpy fp32 z vs float64 z: %.3g" % np.max(np.abs(z - z64)))
for name, out in (("fp32", e.half_step("hv", hd, B, 0, 2, 2, 3, 1, 1, want_prob=True)),
                  ("x3", e.half_step_bf16("hv", hd, B, 2, 2, 3, 1, 1, pieces=3))):
    s, p = out["sample"].to_numpy(), out["prob"].to_numpy()
    print("%-4s  |loc - oracle| %.3g  |loc - f64| %.3g  |v - oracle| %.3g  |v - f64| %.3g  |(v - loc) - z64| %.3g"
          % (name, np.max(np.abs(p - loc)), np.max(np.abs(p - loc64)), np.max(np.abs(s - v1)), np.max(np.abs(s - (loc64 + z64))),
             np.max(np.abs((s.astype(np.float64) - p) - z64))))
