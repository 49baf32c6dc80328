#!/usr/bin/env python3
"""Random shapes: one CD step on the x3 path against the same step on the fp32 MFMA kernels (an independent implementation:
other kernels, other tilings, the same Philox uniforms).  Probabilities agree to ~1e-6, so a sample flips only inside the
|u - p| rounding band and the updates agree except for the rare row / column such a flip moves by one count.

    python tools/fuzz_shapes.py [--n 40] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = np.random.default_rng(a.seed)
worst = 0.0
for it in range(a.n):
    B = int(g.choice([64, 100, 128, 200, 256, 300, 384, 512, 640, 777, 1024, 1500, 2048]))
    nv = int(g.integers(40, 1400))
    nh = int(g.integers(40, 1400))
    kind = str(g.choice(["binary", "real"]))
    mode = int(g.integers(0, 2)) if kind == "real" else 0
    k = int(g.choice([1, 1, 2]))
    persistent = bool(g.integers(0, 2)) and mode == 0
    W0 = g.uniform(-0.1, 0.1, (nv, nh)).astype(np.float32)
    bh = g.uniform(-0.1, 0.1, nh).astype(np.float32)
    bv = g.uniform(-0.1, 0.1, nv).astype(np.float32)
    Vh = (g.random((B, nv)) < 0.3).astype(np.float32) if kind == "binary" else (np.floor(g.random((B, nv)) * 256.0) / 255.0).astype(np.float32)
    out = {}
    for compute in ("x3", "fp32"):
        e = DeviceRBM(W0, bh, bv, dev)
        V = DeviceMatrix.from_host(Vh, dev)
        chain = DeviceMatrix.from_host((np.random.default_rng(7).random((B, nv)) < 0.5).astype(np.float32), dev) if persistent else None
        e.cd_step(V, B, 0, 1.0, 11, 3, k=k, mode=mode, compute=compute, v_chain=chain)
        torch.cuda.synchronize()
        out[compute] = [x.copy() for x in e.get_weights()]
    dW = [out["x3"][i] - out["fp32"][i] for i in range(3)]
    scale = max(1.0, float(np.abs(out["fp32"][0] - W0).max()))
    frac = float(np.mean(np.abs(dW[0]) > 1e-3 * scale))
    mx = float(np.abs(dW[0]).max())
    ok = np.all(np.isfinite(out["x3"][0])) and frac < 0.02 and all(np.abs(d).max() <= 0.05 * scale * B for d in dW)
    worst = max(worst, frac)
    print("%2d  B %4d  nv %4d  nh %4d  %-6s mode %d  k %d  pcd %d | max |dW_x3 - dW_fp32| %.2e  entries off > 1e-3: %.4f  %s"
          % (it, B, nv, nh, kind, mode, k, int(persistent), mx, frac, "ok" if ok else "FAIL"), flush=True)
    assert ok
print("all %d shapes agree (worst share of entries a flipped sample moved: %.4f)" % (a.n, worst))
