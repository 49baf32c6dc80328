#!/usr/bin/env python3
"""Config 4 of BASELINE.json: 3-layer DBN 784 -> 1024 -> 1024 -> 1024, greedy layer-wise CD-1,
batch 4096, on one MI355X.  Prints per-layer CD-1 steps/s and MFMA-roofline fractions (one JSON line).

    python tools/bench_dbn.py [--rows 65536] [--epochs 2]
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
from keras_unsupervised_amd.ebm import DBN, MODE_VISIBLE_BERNOULLI, RBM  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--epochs", type=int, default=2)
args = ap.parse_args()
PEAK = 157.3e12
B = 4096
dev = torch.device("cuda", 0)
dims = [784, 1024, 1024, 1024]
hps = {"batch_size": B, "epochs": args.epochs, "lr": 1e-3 / B}
V = DeviceMatrix.from_host((torch.rand(args.rows, 784, device=dev) < 0.19).float(), dev)
layers = [RBM(hps, dims[i + 1], name="rbm_%d" % (i + 1), mode=MODE_VISIBLE_BERNOULLI, seed=i) for i in range(3)]
dbn = DBN()
out = {"workload": "dbn_784-1024-1024-1024_cd1_batch4096_fp32", "rows": args.rows, "epochs": args.epochs, "layers": []}
X = V
t_all = time.perf_counter()
for i, layer in enumerate(layers):
    layer.build((None, dims[i]))
    dbn.add_stack(layer)
    layer.fit(X.view()[:B].contiguous(), verbose=0)      # warm-up (workspace allocation, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    layer.fit(X, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = args.epochs * (args.rows // B)
    flop = 10.0 * B * dims[i] * dims[i + 1]
    t1 = time.perf_counter()
    X = layer.transform(X)                               # stays on the device between layers
    torch.cuda.synchronize()
    out["layers"].append({"shape": "%dx%d" % (dims[i], dims[i + 1]), "steps_per_s": steps / dt,
                          "ms_per_step": dt / steps * 1e3, "tflops": flop * steps / dt / 1e12,
                          "frac_fp32_mfma_peak": flop * steps / dt / PEAK,
                          "transform_ms": (time.perf_counter() - t1) * 1e3})
out["total_s"] = time.perf_counter() - t_all
print(json.dumps(out))
