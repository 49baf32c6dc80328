#!/usr/bin/env python3
"""HIP-event time of each kernel of the config-2 CD-1 step, for tile-configuration sweeps:

    KURBM_CFG_VH=3 KURBM_CFG_HV=6 KURBM_CFG_OUTER=7 KURBM_SPLIT=4 python tools/kernel_times.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)


def t(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


h_pos = eng.half_step("vh", V, B, 0, 0, 1, 42, 0, 0)["sample"]
v_neg = eng.half_step("hv", h_pos, B, 0, 0, 1, 42, 1, 0)["sample"]
h_neg = eng.half_step("vh", v_neg, B, 0, 0, 0, 42, 0, 0, want_sample=False, want_prob=True)["prob"]
res = {
    "vh_s": t(lambda: eng.half_step("vh", V, B, 0, 0, 1, 42, 0, 0)),
    "hv_s": t(lambda: eng.half_step("hv", h_pos, B, 0, 0, 1, 42, 1, 0)),
    "vh_p": t(lambda: eng.half_step("vh", v_neg, B, 0, 0, 0, 42, 0, 0, want_sample=False, want_prob=True)),
    "outer": t(lambda: eng.outer_delta(V, h_pos, v_neg, h_neg, B)),
    "step": t(lambda: eng.cd_step(V, B, 0, 1e-7, 42, 0)),
}
env = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("KURBM_") and k != "KURBM_LIB")
print("%-40s " % env + "  ".join("%s %6.1f" % kv for kv in res.items()) + "  us")
