#!/usr/bin/env python3
"""Is a launch power-limited?  Replays ONE stage of the x3 step (tools/stage_times.py numbering) for a few seconds and samples the
GPU's clock and power beside it (rocm-smi), then prints the stage's time and the samples' medians.

    python tools/power_probe.py [binary|grey] [bern|gauss] STAGE [seconds]
"""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

kind, mode, stage = sys.argv[1], 1 if sys.argv[2] == "gauss" else 0, int(sys.argv[3])
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
B, NV, NH = 4096, 784, 1024
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
torch.cuda.synchronize()
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:  # noqa: BLE001
            out = str(e)
        samples.append(out)
        time.sleep(0.05)


th = threading.Thread(target=sampler)
th.start()
fn = (lambda: eng.cd_step(V, B, 0, lr, 42, 0, mode=mode, compute="x3", planes=planes)) if stage < 0 else \
     (lambda: eng.cd_step_x3_stage(V, B, 0, lr, 42, 0, stage, mode=mode, planes=planes))
t_end = time.time() + secs
n = 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
while time.time() < t_end:
    for _ in range(200):
        fn()
    n += 200
    torch.cuda.synchronize()
b.record()
torch.cuda.synchronize()
stop[0] = True
th.join()
us = a.elapsed_time(b) / n * 1e3
pw = [float(x) for s in samples for x in re.findall(r"Power \(W\):\s*([0-9.]+)", s)]
ck = [float(x) for s in samples for x in re.findall(r"sclk clock level:.*?\((\d+)Mhz\)", s)]
env = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("KURBM_") and k != "KURBM_LIB")
print("%-6s %-5s stage %2d %-16s %.1f us/launch | %d samples: power median %s W max %s | sclk median %s MHz min %s"
      % (kind, "gauss" if mode else "bern", stage, env, us, len(samples), np.median(pw) if pw else None, max(pw) if pw else None,
         np.median(ck) if ck else None, min(ck) if ck else None))
if not pw:
    print(samples[-1][:1500] if samples else "no samples")
