"""CPU tier, world_size 2 over gloo: the data-parallel host path of RBM training.

Each rank runs the Gibbs chain on its shard with global-row Philox counters (here with the
oracle standing in for the kernels -- this test is about sharding, packing and the all-reduce),
the packed deltas are summed by keras_unsupervised_amd.ebm.dp.allreduce_sum_, and the result must
equal the single-process full-batch delta: identical draws, sums equal up to fp32 order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary, synthetic_params


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from keras_unsupervised_amd.ebm import dp
        nv, nh = 40, 24
        W, b_h, b_v = synthetic_params(nv, nh, 3)
        v = synthetic_binary(n_rows, nv, 4, p=0.3)
        assert dp.world() == (rank, world)
        lo, hi = dp.shard_rows(n_rows, world, rank)
        delta = torch.zeros(dp.packed_size(nv, nh), dtype=torch.float32)
        if hi > lo:
            _, _, _, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v[lo:hi], 0.01, 5, 2, row0=lo)
            dp.pack(torch.from_numpy(dW), torch.from_numpy(dbh), torch.from_numpy(dbv), out=delta)
        dp.allreduce_sum_(delta)
        q.put((rank, lo, hi, delta.numpy().copy()))
    finally:
        dist.destroy_process_group()


def _run(n_rows, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def _check(n_rows):
    nv, nh = 40, 24
    W, b_h, b_v = synthetic_params(nv, nh, 3)
    v = synthetic_binary(n_rows, nv, 4, p=0.3)
    _, _, _, _, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, 0.01, 5, 2)
    full = np.concatenate([dW.ravel(), dbh, dbv])
    res = _run(n_rows)
    assert np.array_equal(res[0][3], res[1][3])                      # every rank holds the same sum
    assert np.max(np.abs(res[0][3] - full) / np.maximum(1.0, np.abs(full))) <= 1e-5
    return res


def test_two_ranks_equal_one():
    res = _check(64)
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 32, 32, 64)


def test_ragged_batch_and_empty_shard():
    _check(22)                     # shards of 12 and 10 rows
    res = _check(3)                # rank 1 owns no rows and contributes zeros
    assert res[1][1] == res[1][2]
