#!/usr/bin/env python3
"""Each launch of the x3 CD-1 step replayed alone, for any data kind and mode:

    python tools/stage_times.py [binary|grey] [bern|gauss] [B NV NH]

binary = Bernoulli(0.19) 0/1 data (BASELINE config 2), grey = grey levels k/255 (three bf16 pieces per value); bern / gauss =
the RBM mode (gauss: the reference's constructor default, rbm.py:22).  Prints microseconds per launch (HIP events, 200
replays each on the planes of a complete step) and per whole step; KURBM_* knobs of the environment are echoed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "binary"
mode = 1 if (len(sys.argv) > 2 and sys.argv[2] == "gauss") else 0
B, NV, NH = (int(x) for x in (sys.argv[3:6] if len(sys.argv) > 5 else (4096, 784, 1024)))
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
if kind == "binary":
    V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
else:
    V = DeviceMatrix.from_host((np.floor(g.random((B, NV)) * 256.0) / 255.0).astype(np.float32), dev)
lr = 1e-3 / B
planes = eng.make_planes(V, [(0, B)], mode)
for _ in range(3):
    eng.cd_step(V, B, 0, lr, 42, 0, mode=mode, compute="x3", planes=planes)


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
names = ["vh sample", "hv sample", "vh prob", "stats", "reduce+mirror"]
times = [t(lambda s=s: eng.cd_step_x3_stage(V, B, 0, lr, 42, 0, s, mode=mode, planes=planes)) for s in range(1, 6)]
print("%-6s %-5s %-22s " % (kind, "gauss" if mode else "bern", env) + "  ".join("%s %.1f" % kv for kv in zip(names, times))
      + "  | sum %.1f  step %.1f us" % (sum(times), t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, mode=mode, compute="x3", planes=planes))))
