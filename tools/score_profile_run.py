#!/usr/bin/env python3
"""60 score passes (kurbm_score_x3, config 2) for rocprofv3 --kernel-trace --stats:  ... -- python3 tools/score_profile_run.py [gauss]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, RBM  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
gauss = len(sys.argv) > 1 and sys.argv[1] == "gauss"
V = DeviceMatrix.from_host(torch.floor(torch.rand(B, NV, device=dev) * 256.0) / 255.0 if gauss else (torch.rand(B, NV, device=dev) < 0.19).float(), dev)
r = RBM({"batch_size": B, "epochs": 1, "lr": 1e-3 / B}, NH, mode=MODE_VISIBLE_GAUSSIAN if gauss else MODE_VISIBLE_BERNOULLI, seed=1)
r.build((None, NV))
r._planes = r._dev.make_planes(V, [(0, B)], r.mode)
for i in range(60):
    r._score(V, 0, B, i)
torch.cuda.synchronize()
