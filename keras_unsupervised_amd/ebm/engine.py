"""Device-side state of one RBM and the ctypes calls that drive the HIP kernels.

PyTorch is used for exactly three things here: owning device memory (``torch.zeros`` /
``data_ptr()``), naming the current HIP stream, and host<->device copies.  All arithmetic
of the contrastive-divergence path happens inside ``libkurbm.so``.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .._lib import (ACT_LINEAR, ACT_RELU, ACT_SIGMOID, NOISE_BERNOULLI, NOISE_GAUSSIAN, NOISE_NONE,
                    WHICH_ALL, CdOpts, Context, Params, Rng, check)

MODE_VISIBLE_BERNOULLI = 0
MODE_VISIBLE_GAUSSIAN = 1
MODE_COMPLEX = 2

# sampling-site ids of the RNG contract (include/kurbm.h; CPU statement: oracle/rbm_oracle.py)
CHAIN_STRIDE = 64
STREAM_TRANSFORM = 0x100
STREAM_INV_TRANSFORM = 0x101
CHAIN_W, CHAIN_BH, CHAIN_BV, CHAIN_SCORE = 0, 1, 2, 3


def round_up(x, m):
    return (x + m - 1) // m * m


def resolve_device(device=None):
    """The gfx950 device this process drives.  No GPU -> loud failure (no CPU path exists)."""
    if not torch.cuda.is_available():
        raise _lib.KurbmError(
            "no ROCm device visible: keras_unsupervised_amd runs its RBM/DBN path on MI355X (gfx950) "
            "HIP kernels only and has no CPU fallback")
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.KurbmError("device must be a ROCm ('cuda') device, got %s" % device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def device_guard(device):
    """`with` context that makes a ROCm device current (a no-op for any other device type)."""
    import contextlib
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


class DeviceMatrix:
    """Row-major fp32 [rows, ld] matrix on the device, ld % 4 == 0, padding zero."""

    __slots__ = ("t", "rows", "cols", "ld", "bf16_exact", "binary")

    def __init__(self, t, rows, cols, ld):
        self.t, self.rows, self.cols, self.ld = t, rows, cols, ld
        self.bf16_exact = None   # True / False once DeviceRBM.v_pieces() has looked (0/1 data is exact)
        self.binary = None       # ... and whether every element is 0.0 or 1.0

    @classmethod
    def zeros(cls, rows, cols, device):
        ld = round_up(cols, 4)
        return cls(torch.zeros((max(rows, 1), ld), dtype=torch.float32, device=device), rows, cols, ld)

    @classmethod
    def from_host(cls, x, device):
        """Upload a numpy / torch [rows, cols] array (any float dtype) as fp32."""
        if isinstance(x, torch.Tensor):
            src = x.detach().to(dtype=torch.float32)
        else:
            src = torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32))
        if src.dim() != 2:
            raise ValueError("expected a 2-d array, got shape %s" % (tuple(src.shape),))
        rows, cols = src.shape
        ld = round_up(cols, 4)
        if isinstance(x, torch.Tensor) and src.is_cuda and ld == cols and src.is_contiguous() \
                and src.data_ptr() % 16 == 0 and src.device == device:
            return cls(src, rows, cols, ld)  # already in the device layout: no copy
        m = cls.zeros(rows, cols, device)
        m.t[:rows, :cols].copy_(src, non_blocking=False)
        return m

    def ptr(self, row=0):
        return self.t.data_ptr() + row * self.ld * 4

    def to_numpy(self):
        return self.t[: self.rows, : self.cols].contiguous().cpu().numpy()      # (rows may be 0: an empty [0, cols] array)

    def view(self):
        """[rows, cols] torch view (device)."""
        return self.t[: self.rows, : self.cols]


class ResidentPlanes:
    """bf16 planes (kurbm_x3_convert_rows) of windows of rows of a data matrix, made once and read by every x3 step on
    those rows instead of a per-step conversion: fit() walks the same windows every epoch (rbm.py:113, :211)."""

    def __init__(self, eng, v, windows, v_pieces):
        self.v, self.v_pieces = v, int(v_pieces)
        self.windows = [(int(lo), int(rows)) for lo, rows in windows]
        max_rows = max([rows for _, rows in self.windows] + [1])
        self.stride = int(eng.lib.kurbm_x3_planes_bytes(eng.ctx.handle, max_rows, eng.n_vis, self.v_pieces))
        if self.stride == 0:
            raise _lib.KurbmError("kurbm_x3_planes_bytes failed")
        self.buf = torch.empty(max(len(self.windows), 1) * self.stride, dtype=torch.uint8, device=eng.device)
        self.slot = {}
        for t, (lo, rows) in enumerate(self.windows):
            if rows <= 0:
                continue
            check(eng.lib.kurbm_x3_convert_rows(eng.ctx.handle, v.ptr(lo), rows, v.ld, eng.n_vis, self.v_pieces,
                                                self.buf.data_ptr() + t * self.stride, self.stride, eng._stream()))
            self.slot[(lo, rows)] = t

    def ptr(self, v, row_start, rows, v_pieces):
        """Device address of the planes of rows [row_start, +rows) of v, or None if these are not planes of them."""
        t = self.slot.get((int(row_start), int(rows)))
        if t is None or v is not self.v or v_pieces != self.v_pieces:
            return None
        return self.buf.data_ptr() + t * self.stride

    def uniform(self, row_start, n_rows, batch_size):
        """True if the windows are exactly the contiguous batches of rows [row_start, +n_rows) (kurbm_cd_epoch_x3)."""
        want = [(row_start + lo, min(batch_size, n_rows - lo)) for lo in range(0, n_rows, batch_size)]
        return want == self.windows


class DeviceRBM:
    """W, b_h, b_v on one device + scratch, and one method per C-ABI entry point."""

    def __init__(self, W, b_h, b_v, device=None):
        self.device = resolve_device(device)
        self.ctx = Context.get(self.device.index)
        self.lib = self.ctx.lib
        W = np.asarray(W, dtype=np.float32)
        self.n_vis, self.n_hid = W.shape
        with torch.cuda.device(self.device):
            self.W = DeviceMatrix.from_host(W, self.device)
            if self.W.t.data_ptr() == 0:
                raise _lib.KurbmError("device allocation failed")
            self.b_h = torch.from_numpy(np.ascontiguousarray(b_h, dtype=np.float32)).to(self.device)
            self.b_v = torch.from_numpy(np.ascontiguousarray(b_v, dtype=np.float32)).to(self.device)
        assert self.b_h.numel() == self.n_hid and self.b_v.numel() == self.n_vis
        self.params = Params(self.n_vis, self.n_hid, self.W.ld, 0, self.W.t.data_ptr(),
                             self.b_h.data_ptr(), self.b_v.data_ptr())
        self._ws = None
        self._ws_rows = 0
        self.delta = None  # packed [V*H | H | V] sums, allocated on first use
        # bf16 images of W in both orientations, per number of pieces: 1 = rounded (compute 'bf16'),
        # 3 = exact hi/mid/lo split (compute 'x3');  [tensor, stale]
        self._mirrors = {}
        self._ws_b = {}

    # -- plumbing -------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, rows, k=1):
        if self._ws is None or rows > self._ws_rows:
            n = self.lib.kurbm_workspace_bytes(self.ctx.handle, rows, self.n_vis, self.n_hid, k)
            if n == 0:
                raise _lib.KurbmError("kurbm_workspace_bytes failed")
            self._ws = torch.zeros(n, dtype=torch.uint8, device=self.device)
            self._ws_rows = rows
        return self._ws

    def delta_buffer(self):
        if self.delta is None:
            n = self.n_vis * self.n_hid + self.n_hid + self.n_vis
            self.delta = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self.delta

    # -- bf16 mirrors (fp32 master weights stay authoritative) --------------------------------
    def _weights_written(self, kept=None):
        """The fp32 W changed; every mirror except the one the library refreshed itself is stale."""
        for pieces, m in self._mirrors.items():
            if pieces != kept:
                m[1] = True

    def mirror(self, pieces=1):
        """The bf16 images of W (1 piece: rounded; 3: exact split), refreshed if W was written elsewhere."""
        m = self._mirrors.get(pieces)
        fn_bytes = self.lib.kurbm_bf16_mirror_bytes if pieces == 1 else self.lib.kurbm_x3_mirror_bytes
        fn_refresh = self.lib.kurbm_bf16_mirror_refresh if pieces == 1 else self.lib.kurbm_x3_mirror_refresh
        if m is None:
            n = fn_bytes(self.ctx.handle, self.n_vis, self.n_hid)
            m = self._mirrors[pieces] = [torch.zeros(n, dtype=torch.uint8, device=self.device), True]
        if m[1]:
            check(fn_refresh(self.ctx.handle, C.byref(self.params), m[0].data_ptr(), m[0].numel(), self._stream()))
            m[1] = False
        return m[0]

    def workspace_bf16(self, rows, k=1, pieces=1, v_pieces=1):
        key = (pieces, v_pieces)
        have = self._ws_b.get(key)
        if have is None or rows > have[1]:
            if pieces == 1:
                n = self.lib.kurbm_bf16_workspace_bytes(self.ctx.handle, rows, self.n_vis, self.n_hid, k)
            else:
                n = self.lib.kurbm_x3_workspace_bytes(self.ctx.handle, rows, self.n_vis, self.n_hid, k, v_pieces)
            if n == 0:
                raise _lib.KurbmError("workspace size query failed")
            have = self._ws_b[key] = (torch.zeros(n, dtype=torch.uint8, device=self.device), rows)
        return have[0]

    def v_pieces(self, v):
        """1 if every element of DeviceMatrix v is exactly a bf16 value (0/1 data is), else 3.

        Looked at once per matrix (one pass over it on the device, one 4-byte read back)."""
        if v.bf16_exact is None:
            with torch.cuda.device(self.device):
                flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                check(self.lib.kurbm_bf16_exact(self.ctx.handle, v.ptr(), max(v.rows, 1), v.cols, v.ld, flag.data_ptr(),
                                                self._stream()))
                bits = int(flag.item())
                v.bf16_exact, v.binary = (bits & 1) == 0, bits == 0
        return 1 if v.bf16_exact else 3

    def check_status(self):
        """Raise if a kernel of this context reported a problem (kurbm_ctx_status; synchronises with the device).  Bits 1 and 2
        mean a device-side wait ran into its bound and an update was SKIPPED -- training must not go on from that state: bit 1 =
        the peer exchange (a rank never arrived), bit 2 = a barrier of the one-launch small step or score (its grid was not
        resident: a CU mask, a device shared with another process -- or the dispatcher did not deal its workgroups to the XCDs
        in turn: KURBM_SMALL_LOCAL=0 takes the grid-wide schedule)."""
        bits = self.ctx.status()
        if bits:
            what = [name for bit, name in ((2, "the peer exchange timed out waiting for a rank"),
                                           (4, "the one-launch small step found its grid not resident")) if bits & bit]
            raise _lib.KurbmError("kurbm status %#x: %s; an update was skipped and the replicas / weights are no longer what the "
                                  "step sequence defines -- reload the weights (set_weights)" % (bits, "; ".join(what) or "unknown bit"))

    def get_weights(self):
        out = self.W.to_numpy(), self.b_h.cpu().numpy(), self.b_v.cpu().numpy()     # (device -> host copies: a sync)
        self.check_status()
        return out

    def set_weights(self, W=None, b_h=None, b_v=None):
        if W is not None:
            W = np.asarray(W, dtype=np.float32)
            assert W.shape == (self.n_vis, self.n_hid)
            self.W.t[:, : self.n_hid].copy_(torch.from_numpy(np.ascontiguousarray(W)))
            self._weights_written()
        if b_h is not None:
            self.b_h.copy_(torch.from_numpy(np.ascontiguousarray(b_h, dtype=np.float32)))
        if b_v is not None:
            self.b_v.copy_(torch.from_numpy(np.ascontiguousarray(b_v, dtype=np.float32)))

    # -- kernels --------------------------------------------------------------------
    def half_step(self, direction, x, rows, row_start, act, noise, seed, stream_id, step, row0=0,
                  want_sample=True, want_prob=False, want_u=False):
        """One fused half step over rows [row_start, row_start+rows) of DeviceMatrix x.

        direction 'vh': x is [*, n_vis] -> outputs [rows, n_hid];  'hv': the converse.
        Returns dict of DeviceMatrix for the requested outputs.
        """
        n_out = self.n_hid if direction == "vh" else self.n_vis
        n_in = self.n_vis if direction == "vh" else self.n_hid
        if x.cols != n_in:
            raise ValueError("input has %d columns, expected %d" % (x.cols, n_in))
        out = {}
        with torch.cuda.device(self.device):
            for key, want in (("sample", want_sample), ("prob", want_prob), ("u", want_u)):
                out[key] = DeviceMatrix.zeros(rows, n_out, self.device) if want else None
            rng = Rng(int(seed), int(row0), int(stream_id) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF)
            fn = self.lib.kurbm_half_step_vh_dbg if direction == "vh" else self.lib.kurbm_half_step_hv_dbg
            ld_out = round_up(n_out, 4)
            check(fn(self.ctx.handle, C.byref(self.params), x.ptr(row_start), rows, x.ld, act, noise,
                     C.byref(rng), out["sample"].ptr() if want_sample else None,
                     out["prob"].ptr() if want_prob else None, out["u"].ptr() if want_u else None,
                     ld_out, self._stream()))
        return out

    def _x3_pieces(self, v, v_chain, mode):
        """Pieces per element of the batch (and of the persistent chain) on the x3 path: 1 when every value is exactly a
        bf16 value, else 3.  The data matrix is looked at once.  The chain is REWRITTEN by every step -- 0/1 states in
        Bernoulli mode, real-valued N(loc, 1) draws in Gaussian mode (rbm.py:64-66) -- so nothing is cached for it beyond
        what the library itself is known to have written there."""
        vp = self.v_pieces(v)
        if v_chain is not None and vp == 1:
            vp = 3 if mode == MODE_VISIBLE_GAUSSIAN else self.v_pieces(v_chain)
        if vp == 1 and v.binary and (v_chain is None or v_chain.binary):
            vp |= _lib.V_BINARY   # 0/1 data (and chain): byte / fp8 planes (include/kurbm.h)
        return vp

    def make_planes(self, v, windows, mode=MODE_VISIBLE_BERNOULLI, v_chain=None):
        """ResidentPlanes of `windows` = [(row_start, rows), ...] of DeviceMatrix v for the x3 steps of a fit(), or None
        when they would not fit comfortably in the free HBM (the steps then convert their rows themselves)."""
        with torch.cuda.device(self.device):
            vp = self._x3_pieces(v, v_chain, mode)
            max_rows = max([rows for _, rows in windows] + [1])
            need = len(windows) * int(self.lib.kurbm_x3_planes_bytes(self.ctx.handle, max_rows, self.n_vis, vp))
            free, _ = torch.cuda.mem_get_info(self.device)
            if need == 0 or need > free // 2:
                return None
            return ResidentPlanes(self, v, windows, vp)

    def _chain_written(self, v_chain, mode):
        if v_chain is not None:
            v_chain.bf16_exact = True if mode == MODE_VISIBLE_BERNOULLI else None
            v_chain.binary = v_chain.bf16_exact

    def cd_step(self, v, rows, row_start, lr, seed, step, k=1, mode=MODE_VISIBLE_BERNOULLI, chain=0,
                which=WHICH_ALL, apply=True, emit_delta=False, v_chain=None, row0=0, v_chain_row=0, bf16=False,
                compute=None, planes=None):
        """One CD-k update on rows [row_start, row_start+rows) of DeviceMatrix v.

        compute: 'fp32' (fp32 MFMA), 'x3' (fp32 values as exact bf16 triples on the bf16 MFMA),
        'bf16' (operands rounded to bf16).  planes: ResidentPlanes holding these rows (x3: no per-step conversion)."""
        compute = compute or ("bf16" if bf16 else "fp32")
        with torch.cuda.device(self.device):
            opts = CdOpts(int(k), int(mode), float(lr), 1 if apply else 0,
                          self.delta_buffer().data_ptr() if emit_delta else None,
                          v_chain.ptr(v_chain_row) if v_chain is not None else None,
                          int(seed), int(row0), int(step) & 0xFFFFFFFF, int(chain))
            if compute == "bf16":
                mir, ws = self.mirror(1), self.workspace_bf16(rows, k)
                check(self.lib.kurbm_cd_step_bf16(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                  v.ptr(row_start), rows, v.ld, C.byref(opts), int(which),
                                                  ws.data_ptr(), ws.numel(), self._stream()))
                if apply and (which & 1):
                    self._weights_written(kept=1)
            elif compute == "x3":
                vp = self._x3_pieces(v, v_chain, mode)
                mir, ws = self.mirror(3), self.workspace_bf16(rows, k, 3, vp)
                if planes is not None:
                    opts.v_planes = planes.ptr(v, row_start, rows, vp)
                check(self.lib.kurbm_cd_step_x3(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                v.ptr(row_start), vp, rows, v.ld, C.byref(opts), int(which),
                                                ws.data_ptr(), ws.numel(), self._stream()))
                if apply and (which & 1):
                    self._weights_written(kept=3)
            elif compute == "fp32":
                ws = self.workspace(rows, k)
                check(self.lib.kurbm_cd_step(self.ctx.handle, C.byref(self.params), v.ptr(row_start), rows, v.ld,
                                             C.byref(opts), int(which), ws.data_ptr(), ws.numel(), self._stream()))
                if apply and (which & 1):
                    self._weights_written()
            elif compute == "small":
                # the whole CD-1 step in one launch (kurbm_small.hip): small problems, where the step is launch latencies
                ws = self.workspace(rows, k)
                check(self.lib.kurbm_cd_step_small(self.ctx.handle, C.byref(self.params), v.ptr(row_start), rows, v.ld,
                                                   C.byref(opts), int(which), ws.data_ptr(), ws.numel(), self._stream()))
                if which & 1:
                    self._weights_written()
            else:
                raise ValueError("compute must be 'fp32', 'x3', 'bf16' or 'small', got %r" % (compute,))
        self._chain_written(v_chain, mode)

    def cd_step_dp(self, comm, v, rows, row_start, lr, seed, step, k=1, mode=MODE_VISIBLE_BERNOULLI, chain=0, row0=0,
                   v_chain=None, v_chain_row=0, compute="x3", n_chunks=0, planes=None):
        """One data-parallel CD-k update: this rank's chain on rows [row_start, +rows) of v (`row0` = their index in
        the global batch), the packed sums all-reduced over `comm` (dp.Comm), the summed update applied.  rows may be 0
        (a rank without rows of a remainder batch still joins the all-reduce).  On the x3 and rounded-bf16 paths all of it
        is ONE library call (kurbm_cd_step_x3_dp / _bf16_dp): one all-reduce of the packed sums on the launch stream (n_chunks > 1
        or KURBM_DP_CHUNKS > 1 opts into row ranges of dW, range i all-reduced + applied on the library's comm stream under the
        statistics GEMM of range i + 1); the fp32-MFMA path runs emit -> kurbm_allreduce_sum_f32 -> apply."""
        if hasattr(comm, "capacity") and compute == "x3":
            # the peer exchange (dp.PeerExchange): chain, statistics, two-shot exchange with the apply fused into its second shot
            with torch.cuda.device(self.device):
                vp = self._x3_pieces(v, v_chain, mode)
                mir, ws = self.mirror(3), self.workspace_bf16(max(rows, 1), k, 3, vp)
                opts = CdOpts(int(k), int(mode), float(lr), 1, None, v_chain.ptr(v_chain_row) if v_chain is not None else None,
                              int(seed), int(row0), int(step) & 0xFFFFFFFF, int(chain))
                if planes is not None and rows > 0:
                    opts.v_planes = planes.ptr(v, row_start, rows, vp)
                check(self.lib.kurbm_cd_step_x3_peer(self.ctx.handle, comm.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                     v.ptr(row_start), vp, int(rows), v.ld, C.byref(opts), ws.data_ptr(), ws.numel(),
                                                     self._stream()))
            self._weights_written(kept=3)
            self._mirrors[3][1] = False
            if rows > 0:
                self._chain_written(v_chain, mode)
            return
        delta = self.delta_buffer()
        # n_chunks 0: the library's choice (by the size of the exchange; ctx knob KURBM_DP_CHUNKS overrides)
        if compute in ("x3", "bf16") and not hasattr(comm, "capacity"):
            pieces = 3 if compute == "x3" else 1
            with torch.cuda.device(self.device):
                vp = self._x3_pieces(v, v_chain, mode) if pieces == 3 else 1
                mir, ws = self.mirror(pieces), self.workspace_bf16(max(rows, 1), k, pieces, vp)
                opts = CdOpts(int(k), int(mode), float(lr), 1, delta.data_ptr(),
                              v_chain.ptr(v_chain_row) if v_chain is not None else None,
                              int(seed), int(row0), int(step) & 0xFFFFFFFF, int(chain))
                if planes is not None and rows > 0 and pieces == 3:
                    opts.v_planes = planes.ptr(v, row_start, rows, vp)
                if pieces == 3:
                    check(self.lib.kurbm_cd_step_x3_dp(self.ctx.handle, comm.handle, C.byref(self.params), mir.data_ptr(),
                                                       mir.numel(), v.ptr(row_start), vp, int(rows), v.ld, C.byref(opts),
                                                       int(n_chunks), ws.data_ptr(), ws.numel(), self._stream()))
                else:
                    check(self.lib.kurbm_cd_step_bf16_dp(self.ctx.handle, comm.handle, C.byref(self.params), mir.data_ptr(),
                                                         mir.numel(), v.ptr(row_start), int(rows), v.ld, C.byref(opts),
                                                         int(n_chunks), ws.data_ptr(), ws.numel(), self._stream()))
            self._weights_written(kept=pieces)
            self._mirrors[pieces][1] = False
            if rows > 0:
                self._chain_written(v_chain, mode)
            return
        if rows > 0:
            self.cd_step(v, rows, row_start, lr, seed, step, k=k, mode=mode, chain=chain, apply=False, emit_delta=True,
                         v_chain=v_chain, row0=row0, v_chain_row=v_chain_row, compute=compute)
        else:
            delta.zero_()
        comm.allreduce_sum_(delta)
        self.apply_delta(lr, compute=compute)

    def cd_step_x3_stage(self, v, rows, row_start, lr, seed, step, stage, mode=MODE_VISIBLE_BERNOULLI, planes=None):
        """Measurement hook (bench.py, tools/stage_times.py): ONE launch of the x3 CD-1 sequence on the planes the previous
        complete x3 step (same rows, data and mode) left in the workspace; stage numbering as kurbm_cd_step_x3_stage."""
        with torch.cuda.device(self.device):
            vp = self._x3_pieces(v, None, mode)   # (as cd_step passes it: the planes must read the same)
            mir, ws = self.mirror(3), self.workspace_bf16(rows, 1, 3, vp)
            opts = CdOpts(1, int(mode), float(lr), 1, None, None, int(seed), 0, int(step) & 0xFFFFFFFF, 0)
            if planes is not None:
                opts.v_planes = planes.ptr(v, row_start, rows, vp)
            check(self.lib.kurbm_cd_step_x3_stage(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                  v.ptr(row_start), vp, rows, v.ld, C.byref(opts), WHICH_ALL, int(stage),
                                                  ws.data_ptr(), ws.numel(), self._stream()))
            if stage == 5:
                self._weights_written(kept=3)

    def cd_epoch(self, v, n_rows, batch_size, lr, seed, step0, k=1, mode=MODE_VISIBLE_BERNOULLI, v_chain=None,
                 compute="fp32", row_start=0, planes=None):
        """All batches of rows [row_start, row_start + n_rows) in ONE library call (fused updates, no score); returns #steps."""
        if compute == "x3":
            with torch.cuda.device(self.device):
                vp = self._x3_pieces(v, v_chain, mode)
                rows = min(batch_size, max(n_rows, 1))
                mir, ws = self.mirror(3), self.workspace_bf16(rows, k, 3, vp)
                opts = CdOpts(int(k), int(mode), float(lr), 1, None, v_chain.ptr() if v_chain is not None else None,
                              int(seed), 0, int(step0) & 0xFFFFFFFF, 0)
                if planes is not None and planes.v is v and planes.v_pieces == vp and planes.uniform(row_start, n_rows, batch_size):
                    opts.v_planes, opts.v_planes_stride = planes.buf.data_ptr(), planes.stride
                n = self.lib.kurbm_cd_epoch_x3(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(), v.ptr(row_start), vp,
                                               int(n_rows), v.ld, int(batch_size), C.byref(opts), ws.data_ptr(), ws.numel(),
                                               self._stream())
                if n < 0:
                    check(n)
            self._weights_written(kept=3)
            self._chain_written(v_chain, mode)
            return n
        with torch.cuda.device(self.device):
            ws = self.workspace(min(batch_size, max(n_rows, 1)), k)
            opts = CdOpts(int(k), int(mode), float(lr), 1, None, v_chain.ptr() if v_chain is not None else None,
                          int(seed), 0, int(step0) & 0xFFFFFFFF, 0)
            fn = self.lib.kurbm_cd_epoch_small if compute == "small" else self.lib.kurbm_cd_epoch
            n = fn(self.ctx.handle, C.byref(self.params), v.ptr(row_start), int(n_rows), v.ld,
                                        int(batch_size), C.byref(opts), ws.data_ptr(), ws.numel(), self._stream())
            if n < 0:
                check(n)
        self._weights_written()
        self._chain_written(v_chain, mode)
        return n

    def cd_epoch_small_scored(self, v, n_rows, batch_size, lr, seed, step0, mode, score_chain, row_start=0):
        """The batch loop of fit(verbose=1) for a small RBM as ONE library call (kurbm_cd_epoch_small_scored): every batch's
        one-launch update followed by its one-launch score.  Returns (#steps, scores): scores is a PINNED host tensor
        [steps, 2] that the device fills as it goes -- scores[i, 0] the score of step i, scores[i, 1] turning 1.0 when it has
        landed -- so the caller prints the lines while the device runs on, without synchronising."""
        steps = -(-int(n_rows) // int(batch_size)) if n_rows > 0 else 0
        scores = torch.zeros((max(steps, 1), 2), dtype=torch.float32, pin_memory=True)
        with torch.cuda.device(self.device):
            ws = self.workspace(min(batch_size, max(n_rows, 1)))
            opts = CdOpts(1, int(mode), float(lr), 1, None, None, int(seed), 0, int(step0) & 0xFFFFFFFF, 0)
            n = self.lib.kurbm_cd_epoch_small_scored(self.ctx.handle, C.byref(self.params), v.ptr(row_start), int(n_rows), v.ld,
                                                     int(batch_size), C.byref(opts), int(score_chain), scores.data_ptr(),
                                                     ws.data_ptr(), ws.numel(), self._stream())
            if n < 0:
                check(n)
        self._weights_written()
        return n, scores

    def half_step_bf16(self, direction, x, rows, act, noise, seed, stream_id, step, row0=0, pieces=1, row_start=0,
                       want_prob=True, want_u=True):
        """One half step with bf16 products (pieces=3: the exact split, used by transform() on the x3 path; pieces=1:
        rounded operands, a test hook); fp32 planes (sample, prob, u) as requested."""
        n_out = self.n_hid if direction == "vh" else self.n_vis
        with torch.cuda.device(self.device):
            xp = self.v_pieces(x) if pieces == 3 else 1
            mir, ws = self.mirror(pieces), self.workspace_bf16(rows, 1, pieces, xp if pieces == 3 else 1)
            want = {"sample": bool(noise), "prob": bool(want_prob) or not noise, "u": bool(want_u) and bool(noise)}
            out = {k: (DeviceMatrix.zeros(rows, n_out, self.device) if want[k] else None) for k in ("sample", "prob", "u")}
            rng = Rng(int(seed), int(row0), int(stream_id) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF)
            tail = (act, noise, C.byref(rng), out["sample"].ptr() if out["sample"] is not None else None,
                    out["prob"].ptr() if out["prob"] is not None else None, out["u"].ptr() if out["u"] is not None else None,
                    round_up(n_out, 4), ws.data_ptr(), ws.numel(), self._stream())
            d = 0 if direction == "vh" else 1
            if pieces == 3:
                check(self.lib.kurbm_half_step_x3(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(), d,
                                                  x.ptr(row_start), xp, rows, x.ld, *tail))
            else:
                check(self.lib.kurbm_half_step_bf16(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(), d,
                                                    x.ptr(row_start), rows, x.ld, *tail))
            if noise == NOISE_BERNOULLI and out["sample"] is not None:
                out["sample"].bf16_exact = out["sample"].binary = True        # 0/1: one bf16 piece, no need to look
            elif noise == NOISE_GAUSSIAN and out["sample"] is not None:
                out["sample"].bf16_exact = out["sample"].binary = False
        return out

    def apply_delta(self, lr, which=WHICH_ALL, delta=None, compute=None):
        """W, b_h, b_v += lr * delta (the all-reduced packed sums).  compute='x3' also rewrites the weight-piece
        mirror in the same launch."""
        delta = self.delta_buffer() if delta is None else delta
        with torch.cuda.device(self.device):
            if compute == "x3" and 3 in self._mirrors:
                mir = self._mirrors[3][0]
                check(self.lib.kurbm_x3_apply_delta(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                    delta.data_ptr(), float(lr), int(which), self._stream()))
                self._weights_written(kept=3)
                self._mirrors[3][1] = False
                return
            check(self.lib.kurbm_apply_delta(self.ctx.handle, C.byref(self.params), delta.data_ptr(), float(lr),
                                             int(which), self._stream()))
        self._weights_written()

    def free_energy(self, v, rows, row_start=0, compute=None):
        """F(v) per row (rbm.py:73-76).  compute='x3': the v.W product on the bf16 pieces (fp32-equivalent)."""
        with torch.cuda.device(self.device):
            F = torch.empty(rows, dtype=torch.float32, device=self.device)
            if compute == "x3":
                vp = self.v_pieces(v)
                mir, ws = self.mirror(3), self.workspace_bf16(rows, 1, 3, vp)
                check(self.lib.kurbm_free_energy_x3(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(),
                                                    v.ptr(row_start), vp, rows, v.ld, F.data_ptr(), ws.data_ptr(),
                                                    ws.numel(), self._stream()))
                return F
            ws = self.workspace(rows)
            check(self.lib.kurbm_free_energy(self.ctx.handle, C.byref(self.params), v.ptr(row_start), rows, v.ld,
                                             F.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()))
        return F

    def score_x3(self, v, rows, row_start, seed, step, mode, chain, planes=None):
        """mean |F(v) - F(v')| of rows [row_start, +rows) of v, v' a fresh one-step reconstruction (rbm.py:225-233), in ONE
        library call on the x3 kernels; returns a device tensor whose element 0 is the score -- nothing is read back here."""
        with torch.cuda.device(self.device):
            vp = self._x3_pieces(v, None, mode)
            mir, ws = self.mirror(3), self.workspace_bf16(rows, 1, 3, vp)
            opts = CdOpts(1, int(mode), 0.0, 0, None, None, int(seed), 0, int(step) & 0xFFFFFFFF, int(chain))
            if planes is not None:
                opts.v_planes = planes.ptr(v, row_start, rows, vp)
            out = torch.empty(4, dtype=torch.float32, device=self.device)
            check(self.lib.kurbm_score_x3(self.ctx.handle, C.byref(self.params), mir.data_ptr(), mir.numel(), v.ptr(row_start), vp,
                                          int(rows), v.ld, C.byref(opts), out.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                          self._stream()))
        return out

    def score_small(self, v, rows, row_start, seed, step, mode, chain, want_F=False):
        """The same score for a small RBM in ONE launch (kurbm_score_small: csrc/kurbm_small.hip; at most 512 rows); returns a
        device tensor whose element 0 is the score (want_F: and a [2, rows] tensor of F(v), F(v'))."""
        with torch.cuda.device(self.device):
            ws = self.workspace(rows)
            opts = CdOpts(1, int(mode), 0.0, 0, None, None, int(seed), 0, int(step) & 0xFFFFFFFF, int(chain))
            out = torch.empty(4, dtype=torch.float32, device=self.device)
            F = torch.empty((2, rows), dtype=torch.float32, device=self.device) if want_F else None
            check(self.lib.kurbm_score_small(self.ctx.handle, C.byref(self.params), v.ptr(row_start), int(rows), v.ld, C.byref(opts),
                                             out.data_ptr(), F.data_ptr() if want_F else None, ws.data_ptr(), ws.numel(),
                                             self._stream()))
        return (out, F) if want_F else out

    def dump_plane(self, which, v, rows, mode=MODE_VISIBLE_BERNOULLI):
        """Test hook: one of the planes the last complete x3 CD-1 step on `rows` rows of v left in the workspace, as fp32
        [rows, units] (which: _lib.PLANE_*)."""
        with torch.cuda.device(self.device):
            vp = self._x3_pieces(v, None, mode)
            ws = self.workspace_bf16(rows, 1, 3, vp)
            units = self.n_vis if which in (_lib.PLANE_V_NEG, _lib.PLANE_V_NEG_T) else self.n_hid
            out = DeviceMatrix.zeros(rows, units, self.device)
            check(self.lib.kurbm_x3_dump_plane(self.ctx.handle, int(which), int(rows), self.n_vis, self.n_hid, vp, int(mode),
                                               ws.data_ptr(), ws.numel(), out.ptr(), out.ld, self._stream()))
        return out

    def philox_uniform(self, rows, cols, seed, stream_id, step, row0=0):
        with torch.cuda.device(self.device):
            out = DeviceMatrix.zeros(rows, cols, self.device)
            rng = Rng(int(seed), int(row0), int(stream_id) & 0xFFFFFFFF, int(step) & 0xFFFFFFFF)
            check(self.lib.kurbm_philox_uniform(self.ctx.handle, out.ptr(), rows, cols, out.ld, C.byref(rng),
                                                self._stream()))
        return out

    def outer_delta(self, v_pos, h_pos, v_neg, h_neg, rows):
        """dW = v_pos^T.h_pos - v_neg^T.h_neg as a dense [n_vis, n_hid] tensor (test / bench hook)."""
        with torch.cuda.device(self.device):
            ws = self.workspace(rows)
            out = torch.empty((self.n_vis, self.n_hid), dtype=torch.float32, device=self.device)
            check(self.lib.kurbm_outer_delta(self.ctx.handle, v_pos.ptr(), h_pos.ptr(), v_neg.ptr(), h_neg.ptr(),
                                             rows, self.n_vis, self.n_hid, v_pos.ld, h_pos.ld, out.data_ptr(),
                                             ws.data_ptr(), ws.numel(), self._stream()))
        return out


def hidden_site(mode):
    """(activation, noise) of the v->h sampling site for an RBM mode (rbm.py:46-47 vs :58-59)."""
    return (ACT_RELU if mode == MODE_VISIBLE_GAUSSIAN else ACT_SIGMOID), NOISE_BERNOULLI


def visible_site(mode):
    """(activation, noise) of the h->v sampling site (rbm.py:52-53 vs :64-66)."""
    if mode == MODE_VISIBLE_GAUSSIAN:
        return ACT_LINEAR, NOISE_GAUSSIAN
    return ACT_SIGMOID, NOISE_BERNOULLI


__all__ = ["DeviceRBM", "DeviceMatrix", "ResidentPlanes", "resolve_device", "device_guard", "hidden_site", "visible_site", "round_up",
           "MODE_VISIBLE_BERNOULLI", "MODE_VISIBLE_GAUSSIAN", "MODE_COMPLEX", "NOISE_NONE"]
