#!/usr/bin/env python3
"""Host enqueue time vs GPU time of the data-parallel x3 step on one rank (1-rank RCCL group), with and without
dp.X3Pipeline: is the host ahead of the GPU?"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import dp  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH, NB = 4096, 784, 1024, 16
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
if os.environ.get("HIPRI", "0") == "1":
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
else:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((NB * B, NV)) < 0.19).astype(np.float32), dev)
V.bf16_exact = True
lr = 1e-3 / B
pipe = dp.X3Pipeline(eng)


def plain(i):
    lo = (i % NB) * B
    eng.cd_step(V, B, lo, lr, 42, i, apply=False, emit_delta=True, compute="x3")
    dp.allreduce_sum_(eng.delta_buffer())
    eng.apply_delta(lr, compute="x3")


def piped(i):
    pipe.step(V, B, (i % NB) * B, lr, 42, i, nxt=(((i + 1) % NB) * B, B))


for name, fn in (("plain", plain), ("X3Pipeline", piped)):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        fn(20 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-10s host enqueue %.1f us/step, until the GPU is done %.1f us/step" % (name, (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
dist.destroy_process_group()
