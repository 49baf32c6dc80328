#!/usr/bin/env python3
"""Config-2 CD-1 step time on the three compute paths (HIP events, torch current stream)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 784, 1024)))
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
Vr = DeviceMatrix.from_host(g.random((B, NV)).astype(np.float32), dev)


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


lr = 1e-3 / B
env = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("KURBM_") and k != "KURBM_LIB")
res = {c: t(lambda c=c: eng.cd_step(V, B, 0, lr, 42, 0, compute=c)) for c in ("fp32", "x3", "bf16")}
res["x3_real"] = t(lambda: eng.cd_step(Vr, B, 0, lr, 42, 0, compute="x3"))
print("%-30s " % env + "  ".join("%s %6.1f us" % kv for kv in res.items()))
