"""CPU tier, world_size 2 over gloo: the data-parallel host path of RBM training.

No GPU here, so the two things that need one are replaced by test doubles that live in THIS file: the engine
(DeviceRBM -> OracleEngine: the oracle computes what the kernels would) and the communicator (dp.Comm, RCCL behind
the C ABI -> GlooComm).  Everything else is the product's own code: RBM.fit's batch loop, RBM._update_data_parallel,
dp.shard_rows / pack / unpack, the global-row counters.  Results must equal the single-process run: identical draws,
sums equal up to fp32 order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary, synthetic_params


class GlooComm:
    """Stand-in for dp.Comm on CPU tensors: the same interface, torch.distributed (gloo) underneath."""

    def allreduce_sum_(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def count(self):
        return dist.get_world_size()


class OracleEngine:
    """Stand-in for engine.DeviceRBM on the CPU: same constructor and the methods RBM.fit's data-parallel path calls,
    with oracle/rbm_oracle.py computing what the HIP kernels would."""

    def __init__(self, W, b_h, b_v, device=None):
        self.W, self.b_h, self.b_v = [np.array(a, dtype=np.float32) for a in (W, b_h, b_v)]
        self.n_vis, self.n_hid = self.W.shape
        self.device = torch.device("cpu")
        self.calls = []

    def get_weights(self):
        return self.W.copy(), self.b_h.copy(), self.b_v.copy()

    def check_status(self):
        """(DeviceRBM reads the context's sticky status word here; the double has no kernels to report)"""

    def cd_step_dp(self, comm, v, rows, row_start, lr, seed, step, k=1, mode=0, chain=0, row0=0, v_chain=None,
                   v_chain_row=0, compute="x3", n_chunks=0, planes=None):
        from keras_unsupervised_amd.ebm import dp
        self.calls.append((rows, row_start, row0, step))
        delta = torch.zeros(dp.packed_size(self.n_vis, self.n_hid), dtype=torch.float32)
        if rows > 0:
            vb = v.t[row_start:row_start + rows, :v.cols].numpy()
            vc = v_chain.t[v_chain_row:v_chain_row + rows, :v.cols].numpy() if v_chain is not None else None
            _, _, _, ch, (dW, dbh, dbv) = O.cd_step_fused(self.W, self.b_h, self.b_v, vb, lr, seed, step, k=k, row0=row0,
                                                          mode=mode, v_chain=vc)
            dp.pack(torch.from_numpy(dW), torch.from_numpy(dbh), torch.from_numpy(dbv), out=delta)
            if v_chain is not None:
                v_chain.t[v_chain_row:v_chain_row + rows, :v.cols] = torch.from_numpy(ch["v_neg"])
        comm.allreduce_sum_(delta)
        dW, dbh, dbv = [a.numpy() for a in dp.unpack(delta, self.n_vis, self.n_hid)]
        lr = np.float32(lr)
        self.W, self.b_h, self.b_v = self.W + lr * dW, self.b_h + lr * dbh, self.b_v + lr * dbv


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
        GlooComm().allreduce_sum_(delta)
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


# ---- RBM.fit itself at world size 2 ------------------------------------------------------------------------------
FIT_CASES = {"cd1": dict(N=150, bs=64, k=1, persistent=False, mode=O.MODE_VISIBLE_BERNOULLI),
             "pcd2": dict(N=100, bs=40, k=2, persistent=True, mode=O.MODE_VISIBLE_BERNOULLI),
             "gauss": dict(N=70, bs=32, k=1, persistent=False, mode=O.MODE_VISIBLE_GAUSSIAN),
             "tiny_tail": dict(N=67, bs=64, k=1, persistent=False, mode=O.MODE_VISIBLE_BERNOULLI)}   # 3-row remainder: rank 1 idle
NV, NH, LR, SEED = 40, 24, 0.01, 5


def _fit_data(case):
    c = FIT_CASES[case]
    W0 = synthetic_params(NV, NH, 3)
    V = synthetic_binary(c["N"], NV, 4, p=0.3)
    return c, W0, V


def _fit_worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from keras_unsupervised_amd.ebm import RBM, dp
        from keras_unsupervised_amd.ebm import rbm as rbm_mod
        rbm_mod.DeviceRBM = OracleEngine                                    # the two test doubles
        rbm_mod.resolve_device = lambda device=None: torch.device("cpu")
        dp.get_comm = lambda device: GlooComm()
        c, W0, V = _fit_data(case)
        r = RBM({"batch_size": c["bs"], "epochs": 2, "lr": LR}, NH, mode=c["mode"], seed=SEED, weights=W0,
                cd_k=c["k"], persistent=c["persistent"], compute_dtype="fp32")
        assert r.fit(V, verbose=0) is None
        # (collective: each rank owns a band of the chain's rows, RBM.full_chain assembles them -- what save_rbm stores)
        chain = r.full_chain()
        q.put((rank, [w.copy() for w in r.get_weights()], list(r._dev.calls), chain))
    finally:
        dist.destroy_process_group()


def _fit_reference(case):
    """The single-process run of the same schedule, straight from the oracle."""
    c, (W, b_h, b_v), V = _fit_data(case)
    chain = None
    if c["persistent"]:
        chain = np.zeros((c["bs"], NV), np.float32)
        chain[:min(c["bs"], c["N"])] = V[:c["bs"]]
    step = 0
    for _ in range(2):
        for lo, hi in O.batch_slices(c["N"], c["bs"]):
            vc = chain[:hi - lo] if chain is not None else None
            W, b_h, b_v, ch, _ = O.cd_step_fused(W, b_h, b_v, V[lo:hi], LR, SEED, step, k=c["k"], mode=c["mode"], v_chain=vc)
            if chain is not None:
                chain[:hi - lo] = ch["v_neg"]
            step += 1
    return W, b_h, b_v, chain


def _run_fit(case, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


def _check_fit(case):
    res = _run_fit(case)
    ref = _fit_reference(case)
    for a, b, r in zip(res[0][1], res[1][1], ref[:3]):
        assert np.array_equal(a, b)                                   # replicas stay bit-identical
        assert np.max(np.abs(a - r)) <= 1e-5                          # and track the single-process run
    if ref[3] is None:
        assert res[0][3] is None and res[1][3] is None
    else:
        # the assembled chain is the single-process chain on EVERY rank (a rank's own copy holds stale rows outside its band)
        assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][3], ref[3])
    return res


def test_rbm_fit_world2_cd1():
    res = _check_fit("cd1")
    # rank 1's first call: rows 32..63 of batch 0, counters offset by row0 = 32, parameter-update counter 0
    assert res[1][2][0] == (32, 32, 32, 0) and res[0][2][0] == (32, 0, 0, 0)
    assert res[1][2][2] == (10, 128 + 12, 12, 2)                      # remainder batch of 22 rows: shards of 12 and 10


def test_rbm_fit_world2_persistent_cd2():
    _check_fit("pcd2")


def test_rbm_fit_world2_gaussian_mode():
    _check_fit("gauss")


def test_rbm_fit_world2_idle_rank_on_tiny_remainder():
    res = _check_fit("tiny_tail")
    assert res[1][2][1][0] == 0                                       # rank 1 owns no rows of the 3-row batch, still joins
