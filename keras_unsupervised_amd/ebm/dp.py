"""Data-parallel sharding of a CD batch over the GPUs of one node.

The Gibbs chain of a row never looks at another row (reference ku/ebm/rbm.py:119-124), and the updates are SUMS
over the batch (rbm.py:125-134), so rank r runs the chain on its own rows and one sum all-reduce of the packed
[dW | db_h | db_v] buffer reproduces the single-GPU update up to fp32 summation order.  The Philox counters use
the global row index, so the draws do not depend on the GPU count.

The collective is RCCL's ncclAllReduce over xGMI, called INSIDE libkurbm.so (include/kurbm.h: kurbm_comm_*,
kurbm_allreduce_sum_f32, kurbm_cd_step_x3_dp -- the last one overlaps the all-reduce of the first rows of dW with
the statistics GEMM of the rest, on a comm stream the library owns).  torch.distributed is used for two things
only: naming (rank, world size), and carrying the 128-byte RCCL unique id from rank 0 to the others when the
communicator is created.
"""
import ctypes as C

import torch
import torch.distributed as dist

from .. import _lib


def world():
    """(rank, world_size) of the default process group, (0, 1) when not distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rows(n_rows, world_size, rank, of=None):
    """Rows [lo, hi) of an n_rows batch owned by `rank`.

    Shard starts are multiples of 4: one Philox block covers 4 consecutive rows of a column
    (include/kurbm.h), so a shard must not split a block.  Later ranks may own no rows of a
    small remainder batch; they still take part in the all-reduce with a zero delta.
    `of` (>= n_rows): cut the shards as for a batch of `of` rows and clip them to n_rows.  Persistent chains need
    that -- row j of the fantasy particles must live on the same rank in the full batches and in the remainder
    batch, or a rank would continue a stale copy of it.
    """
    per = -(-(n_rows if of is None else max(of, n_rows)) // world_size)
    per = (per + 3) // 4 * 4
    lo = min(rank * per, n_rows)
    hi = min(lo + per, n_rows)
    return lo, hi


class Comm:
    """This process's RCCL communicator behind the C ABI (kurbm_comm_init_rank), plus the comm stream and events
    libkurbm.so owns for the chunked data-parallel step."""

    def __init__(self, device, rank, world_size, unique_id):
        self.lib = _lib.load()
        self.device = torch.device(device)
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
        _lib.check(self.lib.kurbm_comm_init_rank(int(self.device.index), int(world_size), int(rank), buf,
                                                 _lib.UNIQUE_ID_BYTES, C.byref(h)))
        self.handle = h
        self.rank, self.world_size = int(rank), int(world_size)

    @staticmethod
    def new_unique_id():
        lib = _lib.load()
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(lib.kurbm_comm_unique_id(buf, _lib.UNIQUE_ID_BYTES))
        return bytes(buf.raw)

    def count(self):
        """Ranks as RCCL itself reports them (ncclCommCount)."""
        n = self.lib.kurbm_comm_count(self.handle)
        if n < 0:
            _lib.check(n)
        return n

    def allreduce_sum_(self, t):
        """In-place sum over all ranks of a contiguous fp32 device tensor, ordered on torch's current stream."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.kurbm_allreduce_sum_f32(self.handle, t.data_ptr(), t.numel(), st))
        return t

    def barrier(self):
        """All ranks have reached this point and their devices are idle: a one-float all-reduce, then a device sync."""
        if getattr(self, "_token", None) is None:
            self._token = torch.zeros(4, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        self.allreduce_sum_(self._token)
        torch.cuda.synchronize(self.device)

    def destroy(self):
        if self.handle:
            self.lib.kurbm_comm_destroy(self.handle)
            self.handle = None


class PeerExchange:
    """The exchange through peer pointers (include/kurbm.h: kurbm_peer_*; keras_unsupervised_amd/csrc/kurbm_peer.hip): a two-shot
    all-reduce over hipIpc-mapped buffers, the second shot fused into the launch that applies the update.  No collective library:
    it also runs between processes that SHARE one GPU.  torch.distributed carries the 64-byte IPC handles, nothing else.
    Duck-types Comm (count / allreduce_sum_ / barrier / destroy), so RBM.fit's data-parallel loop does not care which it has."""

    def __init__(self, device, rank, world_size, n_vis, n_hid):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rank, self.world_size, self.n_vis, self.n_hid = int(rank), int(world_size), int(n_vis), int(n_hid)
        h = C.c_void_p()
        _lib.check(self.lib.kurbm_peer_create(int(self.device.index), self.world_size, self.rank, self.n_vis, self.n_hid, C.byref(h)))
        self.handle = h
        nb = int(self.lib.kurbm_peer_handle_bytes())
        buf = C.create_string_buffer(nb)
        _lib.check(self.lib.kurbm_peer_handle(self.handle, buf, nb))
        handles = [bytes(buf.raw)]
        if self.world_size > 1:
            handles = [None] * self.world_size
            dist.all_gather_object(handles, bytes(buf.raw))
            blob = b"".join(handles)
            err = None
            try:
                _lib.check(self.lib.kurbm_peer_connect(self.handle, C.create_string_buffer(blob, len(blob)), len(blob)))
            except _lib.KurbmError as e:      # (e.g. no peer access between two devices)
                err = e
            # every rank learns whether EVERY rank has mapped every buffer (this is also the barrier in front of the first step):
            # a rank that failed must not leave the others waiting for it inside a step
            oks = [None] * self.world_size
            dist.all_gather_object(oks, err is None)
            if not all(oks):
                self.destroy()
                raise _lib.KurbmError("peer exchange: ranks %s could not map the other ranks' buffers%s"
                                      % ([r for r, ok in enumerate(oks) if not ok], "" if err is None else " (this rank: %s)" % err))
        self.capacity = self.n_vis * self.n_hid + self.n_hid + self.n_vis

    def count(self):
        return int(self.lib.kurbm_peer_ranks(self.handle))

    def allreduce_sum_(self, t):
        """In-place sum over all ranks of a contiguous fp32 device tensor (in pieces of the exchange buffer's capacity)."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.device == self.device
        ctx = _lib.Context.get(self.device.index)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        flat = t.view(-1)
        step = self.capacity // 4 * 4
        for lo in range(0, flat.numel(), step):
            n = min(step, flat.numel() - lo)
            _lib.check(self.lib.kurbm_peer_allreduce_sum_f32(ctx.handle, self.handle, flat.data_ptr() + 4 * lo, n, st))
        return t

    def barrier(self):
        if getattr(self, "_token", None) is None:
            self._token = torch.zeros(4, dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        self.allreduce_sum_(self._token)
        torch.cuda.synchronize(self.device)

    def destroy(self):
        if self.handle:
            self.lib.kurbm_peer_destroy(self.handle)
            self.handle = None


def exchange_kind():
    """'rccl' (default: ncclAllReduce inside libkurbm.so) or 'peer' (KURBM_DP_EXCHANGE=peer: the two-shot exchange over hipIpc)."""
    import os
    kind = os.environ.get("KURBM_DP_EXCHANGE", "rccl").lower()
    if kind not in ("rccl", "peer"):
        raise ValueError("KURBM_DP_EXCHANGE must be 'rccl' or 'peer', got %r" % kind)
    return kind


_comms = {}
_peers = {}


def get_exchange(device, n_vis, n_hid):
    """The exchange of this process for an n_vis x n_hid RBM on `device`: the RCCL communicator, or (KURBM_DP_EXCHANGE=peer) the
    peer exchange of that shape, created on first use (collective)."""
    if exchange_kind() == "rccl":
        return get_comm(device)
    device = torch.device(device)
    key = (device, int(n_vis), int(n_hid))
    x = _peers.get(key)
    if x is None:
        rank, n = world()
        with torch.cuda.device(device):
            x = _peers[key] = PeerExchange(device, rank, n, n_vis, n_hid)
    return x


def get_comm(device):
    """The communicator of `device` in the default process group, created on first use (collective: every rank must
    reach its first data-parallel step).  Rank 0 draws the RCCL unique id; torch.distributed carries its bytes."""
    device = torch.device(device)
    comm = _comms.get(device)
    if comm is None:
        rank, n = world()
        box = [Comm.new_unique_id() if rank == 0 else None]
        if n > 1:
            with torch.cuda.device(device):
                dist.broadcast_object_list(box, src=0)
        comm = _comms[device] = Comm(device, rank, n, box[0])
    return comm


def destroy_comms():
    for comm in list(_comms.values()) + list(_peers.values()):
        comm.destroy()
    _comms.clear()
    _peers.clear()


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
