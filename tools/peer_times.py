#!/usr/bin/env python3
"""What the peer exchange (kurbm_peer_*) adds to a step, measured where it can be on one GPU.

    python tools/peer_times.py            one process: local x3 step | RCCL 1-rank dp step | peer 1-rank dp step (HIP events)
    python -m torch.distributed.run --nproc-per-node 2 tools/peer_times.py     two PROCESSES on GPU 0: the peer step with a real
                                                                               partner (both share the GPU: the step time doubles,
                                                                               the protocol's own cost is the difference to 2 x local)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from keras_unsupervised_amd.ebm import dp  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
B, nv, nh = 4096, 784, 1024
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (nv, nh)).astype(np.float32), np.zeros(nh, np.float32), np.zeros(nv, np.float32), dev)
V = DeviceMatrix.from_host((torch.rand(B, nv, device=dev) < 0.19).float(), dev)
planes = eng.make_planes(V, [(0, B)])
lr = 1e-3 / B


def t(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return "host enqueue %6.1f us  events %6.1f us" % (host, e0.elapsed_time(e1) / n * 1e3)


def say(name, s):
    print("[rank %d of %d] %-44s %s" % (rank, world, name, s), flush=True)


say("local x3 step", t(lambda: eng.cd_step(V, B, 0, lr, 42, 0, compute="x3", planes=planes)))
peer = dp.PeerExchange(dev, rank, world, nv, nh)
say("peer step (kurbm_cd_step_x3_peer)", t(lambda: eng.cd_step_dp(peer, V, B, 0, lr, 42, 0, row0=rank * B, compute="x3", planes=planes)))
buf = torch.zeros(nv * nh + nh + nv, device=dev)
say("peer all-reduce alone (3.2 MB)", t(lambda: peer.allreduce_sum_(buf)))
small = torch.zeros(1024, device=dev)
say("peer all-reduce alone (4 KB)", t(lambda: peer.allreduce_sum_(small)))
if world == 1:
    comm = dp.Comm(dev, 0, 1, dp.Comm.new_unique_id())
    say("RCCL 1-rank step (kurbm_cd_step_x3_dp)", t(lambda: eng.cd_step_dp(comm, V, B, 0, lr, 42, 0, compute="x3", planes=planes)))
    say("RCCL 1-rank all-reduce alone (3.2 MB)", t(lambda: comm.allreduce_sum_(buf)))
    comm.destroy()
eng.check_status()
peer.destroy()
if world > 1:
    dist.destroy_process_group()
