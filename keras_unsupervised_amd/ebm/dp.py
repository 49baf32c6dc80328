"""Data-parallel sharding of a CD batch over the GPUs of one node.

The Gibbs chain of a row never looks at another row (reference ku/ebm/rbm.py:119-124), and
the updates are SUMS over the batch (rbm.py:125-134), so rank r runs the chain on its own
rows and one sum all-reduce of the packed [dW | db_h | db_v] buffer (RCCL over xGMI; `nccl`
backend of torch.distributed) reproduces the single-GPU update up to fp32 summation order.
The Philox counters use the global row index, so the draws do not depend on the GPU count.

Everything here is host logic on torch tensors; it runs unchanged on CPU tensors with the
`gloo` backend, which is how tests/test_dp_gloo.py covers the N > 1 path without GPUs.
"""
import torch
import torch.distributed as dist


import os

# Off by default.  Measured on one MI355X (tools/dp_step_times.py, config 2): the statistics of rows [0,384) and
# [384,784) take 25.6 + 33.9 us against 46.2 us in one piece, so hiding the first all-reduce (~30 us of an estimated
# ~56 us at 8 GPUs) under the second range nets ~3 us, and the extra launches make the step host-bound.
OVERLAP_ROW_RANGES = os.environ.get("KURBM_DP_OVERLAP", "0") == "1"
# x3 data-parallel step: convert the next batch while this step's all-reduce is in flight (X3Pipeline).  OFF by default:
# on one rank (1-rank RCCL group, nothing to hide under) the step got 10 us LONGER with the host well ahead of the GPU
# (tools/dp_host_time.py: 150.6 -> 161.6 us), and the N > 1 case cannot be measured on the one-GPU boxes of this build.
PRECONVERT = os.environ.get("KURBM_DP_PRECONVERT", "0") == "1"


def world():
    """(rank, world_size) of the default process group, (0, 1) when not distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rows(n_rows, world_size, rank):
    """Rows [lo, hi) of an n_rows batch owned by `rank`.

    Shard starts are multiples of 4: one Philox block covers 4 consecutive rows of a column
    (include/kurbm.h), so a shard must not split a block.  Later ranks may own no rows of a
    small remainder batch; they still take part in the all-reduce with a zero delta.
    """
    per = -(-n_rows // world_size)
    per = (per + 3) // 4 * 4
    lo = min(rank * per, n_rows)
    hi = min(lo + per, n_rows)
    return lo, hi


def allreduce_sum_(delta, async_op=False):
    """In-place sum of the packed delta over all ranks (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized():
        return dist.all_reduce(delta, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


class X3Pipeline:
    """The data-parallel x3 step with the NEXT batch's conversion (fp32 -> bf16 planes, both orientations, column sums;
    independent of the parameters) enqueued while this step's all-reduce is in flight:

        [convert t]  chain t, statistics t, packed sums   all-reduce t (RCCL stream)   apply t   chain t+1 ...
                                                          convert t+1 (compute stream)

    `nxt` = (row_start, rows) of the rows this rank takes next, or None.  Only batches of the same row count are
    pre-converted (the workspace layout follows the row count)."""

    def __init__(self, eng):
        self.eng = eng
        self.ready = None

    def step(self, V, rows, lo, lr, seed, step, nxt=None, **kw):
        eng = self.eng
        part = "rest" if self.ready == (id(V), lo, rows) else None
        eng.cd_step(V, rows, lo, lr, seed, step, apply=False, emit_delta=True, compute="x3", part=part, **kw)
        work = allreduce_sum_(eng.delta_buffer(), async_op=True)
        self.ready = None
        if nxt is not None and nxt[1] == rows:
            eng.cd_step(V, rows, nxt[0], lr, seed, step + 1, apply=False, emit_delta=True, compute="x3", part="convert", **kw)
            self.ready = (id(V), nxt[0], rows)
        if work is not None:
            work.wait()
        eng.apply_delta(lr, compute="x3")


def packed_size(n_vis, n_hid):
    return n_vis * n_hid + n_hid + n_vis


def unpack(delta, n_vis, n_hid):
    """Views (dW [n_vis, n_hid], db_h [n_hid], db_v [n_vis]) of a packed delta."""
    nw = n_vis * n_hid
    return delta[:nw].view(n_vis, n_hid), delta[nw:nw + n_hid], delta[nw + n_hid:nw + n_hid + n_vis]


def pack(dW, db_h, db_v, out=None):
    n_vis, n_hid = dW.shape
    if out is None:
        out = torch.empty(packed_size(n_vis, n_hid), dtype=dW.dtype, device=dW.device)
    a, b, c = unpack(out, n_vis, n_hid)
    a.copy_(dW)
    b.copy_(db_h)
    c.copy_(db_v)
    return out


def x3_sums_overlapped(eng, v, rows, row_start, lr, seed, step, k=1, row0=0, v_chain=None, v_chain_row=0):
    """Data-parallel x3 step up to the summed delta: the chain, then the statistics in two row ranges of dW,
    the all-reduce of the first range running while the second is still being computed.  Leaves the all-reduced
    packed sums in eng.delta_buffer().  (Without a process group the all-reduces are no-ops.)"""
    nv, nh = eng.n_vis, eng.n_hid
    delta = eng.delta_buffer()
    eng.cd_chain_x3(v, rows, row_start, lr, seed, step, k=k, row0=row0, v_chain=v_chain, v_chain_row=v_chain_row)
    m = (nv // 2) // 128 * 128
    if m < 128:                                   # too few visible rows to split on a tile boundary
        eng.x3_stats_rows(v, rows, row_start, 0, nv, lr, seed, step, k=k, row0=row0)
        allreduce_sum_(delta)
        return
    eng.x3_stats_rows(v, rows, row_start, 0, m, lr, seed, step, k=k, row0=row0)
    w1 = allreduce_sum_(delta[: m * nh], async_op=True)
    eng.x3_stats_rows(v, rows, row_start, m, nv, lr, seed, step, k=k, row0=row0)
    w2 = allreduce_sum_(delta[m * nh:], async_op=True)
    for w in (w1, w2):
        if w is not None:
            w.wait()
