#!/usr/bin/env python3
"""Config 5 of BASELINE.json on ONE GPU's share: 4096 x 4096 RBM, persistent CD-10, 1024 rows per GPU
(global batch 8192 over 8 GPUs): fp32 MFMA path, x3 path (exact bf16 triples) and rounded-bf16 path.  One JSON line.

    python tools/bench_config5.py [--rows 1024] [--k 10] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1024)
ap.add_argument("--nv", type=int, default=4096)
ap.add_argument("--nh", type=int, default=4096)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--only", default="", help="one compute path (fp32 / x3 / bf16): for rocprofv3 --kernel-trace --stats")
a = ap.parse_args()
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
W = g.uniform(-0.05, 0.05, (a.nv, a.nh)).astype(np.float32)
V = DeviceMatrix.from_host((torch.rand(a.rows, a.nv, device=dev) < 0.19).float(), dev)
flop = (4 * a.k + 8) * a.rows * a.nv * a.nh            # persistent CD-k
out = {"workload": "rbm_%dx%d_pcd%d_rows%d" % (a.nv, a.nh, a.k, a.rows), "flop_per_step": flop}
for name in ((a.only,) if a.only else ("fp32", "x3", "bf16")):
    eng = DeviceRBM(W, np.zeros(a.nh, np.float32), np.zeros(a.nv, np.float32), dev)
    chain = DeviceMatrix.from_host(V.view().clone(), dev)
    lr = 1e-3 / a.rows
    for i in range(3):
        eng.cd_step(V, a.rows, 0, lr, 42, i, k=a.k, v_chain=chain, compute=name)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        eng.cd_step(V, a.rows, 0, lr, 42, 3 + i, k=a.k, v_chain=chain, compute=name)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    assert bool(torch.isfinite(eng.W.t).all().item())
    out[name] = {"ms_per_step": dt * 1e3, "steps_per_s": 1 / dt, "tflops": flop / dt / 1e12,
                 "frac_of_fp32_mfma_peak": flop / dt / 157.3e12}
    del eng
print(json.dumps(out))
