#!/usr/bin/env python3
"""Upper bound of what overlapping the dependent launches of a CD-1 step could gain (config 2, x3): the same step with its
half steps (KURBM_ANYORDER bit 0) and its statistics GEMM (bit 1) dispatched WITHOUT the AQL barrier bit, so that a launch's
workgroups start as CUs come free under the tail of the launch in front.  No dependency is tracked -- the results race; this is
a TIMING of the overlap a flag-based (row-block dependency) version could at most reach.  A/B/A/B in one process."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd._lib import Context  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
planes = eng.make_planes(V, [(0, B)])
ctx = Context.get(0)


def t(iters=300):
    for i in range(20):
        eng.cd_step(V, B, 0, 1e-3 / B, 42, i, compute="x3", planes=planes)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        eng.cd_step(V, B, 0, 1e-3 / B, 42, i, compute="x3", planes=planes)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for rep in range(2):
    for knob in (0, 1, 3, 0):
        ctx.set_option("KURBM_ANYORDER", knob)
        print("KURBM_ANYORDER=%d  %.1f us per step" % (knob, t()))
ctx.set_option("KURBM_ANYORDER", 0)
