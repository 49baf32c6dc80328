#!/usr/bin/env python3
"""60 config-2 CD-1 steps on one compute path, for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o x3 -- python3 tools/x3_profile_run.py x3
    ... x3_profile_run.py x3 4096 784 1024 real        grey-level data k / 255 (the batch travels as three pieces)
    ... x3_profile_run.py x3 4096 784 1024 real gauss  the reference's default Gaussian-visible mode on that data"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

compute = sys.argv[1] if len(sys.argv) > 1 else "x3"
B, NV, NH = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (4096, 784, 1024)
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
kind = sys.argv[5] if len(sys.argv) > 5 else "binary"
mode = 1 if (len(sys.argv) > 6 and sys.argv[6] == "gauss") else 0
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32) if kind == "binary"
                           else (np.floor(g.random((B, NV)) * 256.0) / 255.0).astype(np.float32), dev)
# as RBM.fit does on the x3 path: the bf16 planes of the batch are made once, the steps read them
planes = eng.make_planes(V, [(0, B)], mode) if compute == "x3" and os.environ.get("PROFILE_NO_PLANES", "0") != "1" else None
for i in range(60):
    eng.cd_step(V, B, 0, 1e-3 / B, 42, i, mode=mode, compute=compute, planes=planes)
torch.cuda.synchronize()
