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
