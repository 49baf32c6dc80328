#!/usr/bin/env python3
"""Each launch of the x3 CD-1 step replayed alone (config 2): conversion, three half steps, statistics, reduce + mirror."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 784, 1024)))
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


env = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("KURBM_") and k != "KURBM_LIB")
names = ["conv", "vh sample", "hv sample", "vh prob", "stats", "reduce+mirror"]
times = [t(lambda s=s: eng.cd_step_x3_stage(V, B, 0, lr, 42, 0, s)) for s in range(6)]
print("%-24s " % env + "  ".join("%s %.1f" % kv for kv in zip(names, times)) + "  | sum %.1f  step %.1f us"
      % (sum(times), t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3"))))
