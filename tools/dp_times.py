#!/usr/bin/env python3
"""Where the time of the data-parallel x3 step goes on ONE GPU (1-rank RCCL communicator): local step, the plain
sequence emit -> all-reduce -> apply, kurbm_cd_step_x3_dp with 1 / 2 / 3 row ranges, the all-reduce alone.
Wall clock per call over 200 calls (the host may be the bound) and HIP-event time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from keras_unsupervised_amd.ebm import dp  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

dev = torch.device("cuda", 0)
B, nv, nh = 4096, 784, 1024
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (nv, nh)).astype(np.float32), np.zeros(nh, np.float32), np.zeros(nv, np.float32), dev)
V = DeviceMatrix.from_host((torch.rand(B, nv, device=dev) < 0.19).float(), dev)
lr = 1e-3 / B
comm = dp.Comm(dev, 0, 1, dp.Comm.new_unique_id())
delta = eng.delta_buffer()


def t(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e6
    return "host enqueue %7.1f us  wall %7.1f us  events %7.1f us" % (host, wall, e0.elapsed_time(e1) / n * 1e3)


def plain():
    eng.cd_step(V, B, 0, lr, 42, 0, apply=False, emit_delta=True, compute="x3")
    comm.allreduce_sum_(delta)
    eng.apply_delta(lr, compute="x3")


print("local x3 step              ", t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3")))
print("emit / all-reduce / apply  ", t(plain))
for n in (1, 2, 3):
    print("kurbm_cd_step_x3_dp, %d range" % n, t(lambda n=n: eng.cd_step_dp(comm, V, B, 0, lr, 42, 0, compute="x3", n_chunks=n)))
print("all-reduce alone (3.2 MB)  ", t(lambda: comm.allreduce_sum_(delta)))
small = torch.zeros(1024, device=dev)
print("all-reduce alone (4 KB)    ", t(lambda: comm.allreduce_sum_(small)))
comm.destroy()

# cross-stream hand-off alone (torch streams / events): small kernel -> event -> other stream -> event -> back
for name, s2 in (("normal-priority side stream", torch.cuda.Stream(dev)), ("high-priority side stream", torch.cuda.Stream(dev, priority=-1))):
    x = torch.zeros(1 << 20, device=dev)
    main = torch.cuda.current_stream(dev)

    def hop():
        x.add_(1.0)
        e = torch.cuda.Event()
        e.record(main)
        s2.wait_event(e)
        with torch.cuda.stream(s2):
            x.add_(1.0)
        e2 = torch.cuda.Event()
        e2.record(s2)
        main.wait_event(e2)

    print("hop main -> %s -> main" % name, t(hop))

# the same with a NON-NULL main stream (torch's default stream is the legacy null stream)
comm = dp.Comm(dev, 0, 1, dp.Comm.new_unique_id())
own = torch.cuda.Stream(dev)
with torch.cuda.stream(own):
    print("--- launches on an own (non-null) stream")
    print("local x3 step              ", t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3")))
    for n in (1, 2, 3):
        print("kurbm_cd_step_x3_dp, %d range" % n, t(lambda n=n: eng.cd_step_dp(comm, V, B, 0, lr, 42, 0, compute="x3", n_chunks=n)))
    s2 = torch.cuda.Stream(dev)
    x = torch.zeros(1 << 20, device=dev)

    def hop2():
        x.add_(1.0)
        e = torch.cuda.Event()
        e.record(own)
        s2.wait_event(e)
        with torch.cuda.stream(s2):
            x.add_(1.0)
        e2 = torch.cuda.Event()
        e2.record(s2)
        own.wait_event(e2)

    print("hop own -> side -> own     ", t(hop2))
comm.destroy()
