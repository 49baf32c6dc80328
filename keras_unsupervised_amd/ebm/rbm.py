"""RBM with the class surface of the reference ``ku.ebm.RBM`` (reference ku/ebm/rbm.py:19-242),
driven by hand-written gfx950 HIP kernels through the C ABI of ``include/kurbm.h``.

What is kept verbatim from the reference: the constructor signature
``RBM(hps, output_dim, name=None, mode=MODE_VISIBLE_GAUSSIAN, **kwargs)`` (rbm.py:22), the hps
keys ``batch_size`` / ``epochs`` / ``lr`` (rbm.py:46,113,128), the methods ``build`` / ``call`` /
``transform`` / ``inv_transform`` / ``cal_free_energy`` / ``fit`` / ``compute_output_shape`` /
``get_config``, the attributes ``rbm_weight`` / ``hidden_bias`` / ``visible_bias``, the batch order
of ``fit`` (contiguous, unshuffled, remainder last: rbm.py:110-111, :211, :218), the update rules
(sums over the batch scaled by lr only: rbm.py:125-134) and the per-step score print
(rbm.py:225-234).  The TensorFlow graph machinery (placeholders, K.function) is gone.

Where the reference cannot run as written, the minimal repairs of SURVEY.md 8(a) apply (uniform
shape follows the probabilities, remainder-batch shape, list returns of K.function, ...).

Extensions, all off by default or reference-preserving: ``update_mode='fused'`` (one Gibbs chain
feeds all three updates; ``'reference_sequential'`` replays the reference's three chains),
``cd_k``, ``persistent`` chains, ``seed``, data-parallel training when torch.distributed is
initialised, ``verbose=0`` to skip the score passes.
"""
import collections
import time

import numpy as np
import torch

from .. import _lib
from . import dp
from .engine import (CHAIN_BH, CHAIN_BV, CHAIN_SCORE, CHAIN_STRIDE, CHAIN_W, MODE_COMPLEX,
                     MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, STREAM_INV_TRANSFORM, STREAM_TRANSFORM,
                     DeviceMatrix, DeviceRBM, device_guard, hidden_site, resolve_device, visible_site)

_UPDATE_MODES = ("fused", "reference_sequential")


def _unwrap(x):
    """K.function takes and returns 1-element lists (rbm.py:89, :211, :233): accept both forms."""
    if isinstance(x, (list, tuple)):
        if len(x) != 1:
            raise ValueError("expected an array or a 1-element list of arrays")
        return x[0]
    return x


class _ScoreRing:
    """The per-step score of fit(verbose = 1) on its way from the device to stdout without a host synchronisation per
    step: each score is copied (asynchronously, on the stream that computed it) into a slot of a small pinned host ring,
    an event marks the copy, and a step's line is handed to `sink` once the step AFTER it has been queued -- by then its
    event has normally fired, so the host stays one step ahead of the device instead of waiting for every step."""

    def __init__(self, device, sink, depth=8):
        self.cuda = torch.device(device).type == "cuda"
        self.buf = torch.empty((depth, 4), dtype=torch.float32, pin_memory=self.cuda)
        self.events = [torch.cuda.Event() if self.cuda else None for _ in range(depth)]
        self.pending, self.head, self.depth, self.sink = collections.deque(), 0, depth, sink

    def push(self, score_dev, label):
        slot, self.head = self.head, (self.head + 1) % self.depth
        self.buf[slot, :1].copy_(score_dev.reshape(-1)[:1], non_blocking=True)
        if self.cuda:
            self.events[slot].record()
        self.pending.append((slot, label))
        while len(self.pending) > 1:
            self._pop()

    def _pop(self):
        slot, label = self.pending.popleft()
        if self.cuda:
            self.events[slot].synchronize()
        self.sink(float(self.buf[slot, 0]), label)

    def flush(self):
        while self.pending:
            self._pop()


class RBM(object):
    """Restricted Boltzmann machine trained by contrastive divergence on MI355X."""

    def __init__(self, hps, output_dim, name=None, mode=MODE_VISIBLE_GAUSSIAN, **kwargs):
        self.hps = hps
        self.output_dim = int(output_dim)
        self.name = name
        self.mode = mode
        if mode not in (MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN):
            # MODE_COMPLEX is a TODO in the reference too (rbm.py:16, :68-70)
            raise ValueError("unsupported RBM mode %r" % (mode,))
        opt = lambda key, default: kwargs.pop(key, hps.get(key, default) if hasattr(hps, "get") else default)
        self.seed = int(opt("seed", 0))
        self.update_mode = opt("update_mode", "fused")
        if self.update_mode not in _UPDATE_MODES:
            raise ValueError("update_mode must be one of %s" % (_UPDATE_MODES,))
        self.cd_k = int(opt("cd_k", 1))
        self.persistent = bool(opt("persistent", False))
        # how the matrix products of fit() run (extension; storage, accumulation and results are fp32 in all):
        #   'fp32'  fp32 MFMA               'x3'  fp32 values as exact bf16 triples on the bf16 MFMA
        #   'bf16'  operands ROUNDED to bf16 (reduced precision, BASELINE.json config 5)
        #   'small' the whole CD-1 step in ONE launch (kurbm_cd_step_small; fp32 MFMA): the reference example's own sizes
        #   'auto'  (default) 'x3' at batch sizes >= 256 (tools/bench_fit.py, 784 x 1024, us per step fp32 / x3,
        #           end of round 3: batch 256 95 / 87, 512 102 / 87, 1024 122 / 95, 2048 194 / 105, 4096 341 / 129); below that
        #           'small' where it applies (CD-1 from the data, one GPU), else 'fp32' 
        self.compute_dtype = str(opt("compute_dtype", "auto"))
        if self.compute_dtype not in ("fp32", "x3", "bf16", "small", "auto"):
            raise ValueError("compute_dtype must be 'fp32', 'x3', 'bf16', 'small' or 'auto'")
        self.list_returns = bool(kwargs.pop("ku_compat_list_returns", True))
        self._device_arg = kwargs.pop("device", None)
        self._init_weights = kwargs.pop("weights", None)
        self._kwargs = kwargs          # remaining Keras Layer kwargs (dtype, trainable, ...) are kept for get_config
        self.built = False
        self.input_shape = None
        self.output_shape = None
        self._dev = None               # DeviceRBM
        self._update_count = 0         # `step` of the RNG contract for fit()
        self._call_count = 0           # `step` of the RNG contract for transform / inv_transform / call
        self._v_chain = None           # persistent fantasy particles
        self._planes = None            # engine.ResidentPlanes of the data matrix during an x3 fit()
        self.last_scores = []

    # ------------------------------------------------------------------ build ----------
    def build(self, input_shape):
        """Create W ~ U(-0.05, 0.05) [n_vis, n_hid], b_h, b_v on the device (rbm.py:29-40)."""
        n_vis = int(input_shape[1])
        if self._init_weights is not None:
            W, b_h, b_v = self._init_weights
        else:
            g = np.random.default_rng(self.seed)
            W = g.uniform(-0.05, 0.05, size=(n_vis, self.output_dim)).astype(np.float32)
            b_h = g.uniform(-0.05, 0.05, size=(self.output_dim,)).astype(np.float32)
            b_v = g.uniform(-0.05, 0.05, size=(n_vis,)).astype(np.float32)
        W = np.asarray(W, dtype=np.float32)
        if W.shape != (n_vis, self.output_dim):
            raise ValueError("weights have shape %s, expected %s" % (W.shape, (n_vis, self.output_dim)))
        self._dev = DeviceRBM(W, b_h, b_v, resolve_device(self._device_arg))
        self.input_shape = (None, n_vis)
        self.output_shape = (None, self.output_dim)
        self.built = True

    def _ensure_built(self, n_cols):
        if not self.built:
            self.build((None, n_cols))

    # ------------------------------------------------------------------ variables ------
    @property
    def rbm_weight(self):
        return self._dev.get_weights()[0]

    @rbm_weight.setter
    def rbm_weight(self, W):
        self._dev.set_weights(W=W)

    @property
    def hidden_bias(self):
        return self._dev.get_weights()[1]

    @hidden_bias.setter
    def hidden_bias(self, b):
        self._dev.set_weights(b_h=b)

    @property
    def visible_bias(self):
        return self._dev.get_weights()[2]

    @visible_bias.setter
    def visible_bias(self, b):
        self._dev.set_weights(b_v=b)

    def get_weights(self):
        """[rbm_weight, rbm_hidden_bias, rbm_visible_bias] -- unlike the reference, the visible bias
        is included (there it is a bare K.variable and is silently dropped: rbm.py:38-40)."""
        return list(self._dev.get_weights())

    def set_weights(self, weights):
        W, b_h = weights[0], weights[1]
        b_v = weights[2] if len(weights) > 2 else None
        self._dev.set_weights(W, b_h, b_v)

    # ------------------------------------------------------------------ forward passes --
    @staticmethod
    def _n_cols(x):
        x = _unwrap(x)
        return x.cols if isinstance(x, DeviceMatrix) else x.shape[1]

    def _as_device(self, x):
        """(DeviceMatrix, kind): kind = "device" for a DeviceMatrix, the input's torch.device for a tensor, "numpy"."""
        x = _unwrap(x)
        if isinstance(x, DeviceMatrix):
            return x, "device"
        kind = x.device if isinstance(x, torch.Tensor) else "numpy"
        dev = self._dev.device if self._dev is not None else resolve_device(self._device_arg)
        with device_guard(dev):
            return DeviceMatrix.from_host(x, dev), kind

    def _ret(self, m, kind, as_list):
        """Results go back where the input came from: a DeviceMatrix as is, a tensor on the input tensor's device,
        a numpy array on the host (the K.function convention of the reference)."""
        if kind == "device":
            out = m
        elif isinstance(kind, torch.device):
            out = m.view().to(kind)
        else:
            out = m.to_numpy()
        return [out] if as_list else out

    def _half(self, direction, x, act, noise, stream_id):
        if x.rows == 0:
            # an empty batch yields an empty result (K.function on a [0, n] feed does); the call still counts for the RNG
            self._call_count += 1
            n_out = self._dev.n_hid if direction == "vh" else self._dev.n_vis
            return DeviceMatrix.zeros(0, n_out, self._dev.device)
        # large inputs (a whole data set between DBN layers) go through the x3 kernels too
        if self._large(x.rows):
            out = self._dev.half_step_bf16(direction, x, x.rows, act, noise, self.seed, stream_id, self._call_count, pieces=3,
                                           want_prob=False, want_u=False)
        else:
            out = self._dev.half_step(direction, x, x.rows, 0, act, noise, self.seed, stream_id, self._call_count)
        self._call_count += 1
        return out["sample"]

    def _large(self, rows):
        """Inference-side calls of at least 1024 rows use the x3 kernels (as training does from that batch size on)."""
        return self.compute_dtype in ("auto", "x3") and rows >= 1024

    def _sample_hidden(self, x):
        act, noise = hidden_site(self.mode)
        return self._half("vh", x, act, noise, STREAM_TRANSFORM)

    def _sample_visible(self, h):
        act, noise = visible_site(self.mode)
        return self._half("hv", h, act, noise, STREAM_INV_TRANSFORM)

    def transform(self, v):
        """Sampled hidden states of v (transform_func, rbm.py:88-89 -> :46-48 / :58-60)."""
        self._ensure_built(self._n_cols(v))
        x, kind = self._as_device(v)
        return self._ret(self._sample_hidden(x), kind, self.list_returns and kind != "device")

    def inv_transform(self, h):
        """Sampled visible states of h (inv_transform_func, rbm.py:91-92 -> :52-54 / :64-67)."""
        if not self.built:
            raise ValueError("inv_transform needs a built RBM (call build / fit / transform first)")
        x, kind = self._as_device(h)
        return self._ret(self._sample_visible(x), kind, self.list_returns and kind != "device")

    def call(self, x):
        """The layer's forward pass: stochastic hidden features (rbm.py:80-86)."""
        self._ensure_built(self._n_cols(x))
        xd, kind = self._as_device(x)
        return self._ret(self._sample_hidden(xd), kind, False)

    __call__ = call

    def cal_free_energy(self, v):
        """F(v) = -(v.b_v + sum_j softplus((v.W + b_h)_j))   (free_energy_func, rbm.py:97-98 -> :73-76)."""
        self._ensure_built(self._n_cols(v))
        x, kind = self._as_device(v)
        if x.rows == 0:
            F = torch.empty(0, dtype=torch.float32, device=self._dev.device)
        else:
            F = self._dev.free_energy(x, x.rows, compute="x3" if self._large(x.rows) else None)
        if kind == "device":
            return F
        if isinstance(kind, torch.device):          # torch input: the result goes back to the input's device
            F = F.to(kind)
            return [F] if self.list_returns else F
        F = F.cpu().numpy()
        return [F] if self.list_returns else F

    def compute_output_shape(self, input_shape):
        return (input_shape[0], self.output_dim)          # rbm.py:94-95

    # ------------------------------------------------------------------ training --------
    def _score(self, Vd, lo, rows, step):
        """mean |F(v) - F(v')| with v' a fresh one-step reconstruction (rbm.py:225-233), as a DEVICE tensor (element 0):
        the launches are queued behind the update and nothing is read back here -- fit() prints a step's line when its
        score has landed in pinned host memory (_ScoreRing), so the reference's default verbose = 1 costs no host
        synchronisation per step."""
        d = self._dev
        base = CHAIN_SCORE * CHAIN_STRIDE
        if self._large(rows):
            # one library call on the x3 kernels: same draws (chain CHAIN_SCORE, sites 0 and 1), products on the bf16 pieces
            return d.score_x3(Vd, rows, lo, self.seed, step, self.mode, CHAIN_SCORE, planes=self._planes)
        if self._compute() == "small" and rows <= 512:
            # small RBMs: the whole score in one launch, as their step is (same draws)
            return d.score_small(Vd, rows, lo, self.seed, step, self.mode, CHAIN_SCORE)
        act_h, noise_h = hidden_site(self.mode)
        act_v, noise_v = visible_site(self.mode)
        fe = d.free_energy(Vd, rows, lo)
        h = d.half_step("vh", Vd, rows, lo, act_h, noise_h, self.seed, base + 0, step)["sample"]
        v1 = d.half_step("hv", h, rows, 0, act_v, noise_v, self.seed, base + 1, step)["sample"]
        fe_p = d.free_energy(v1, rows, 0)
        return (fe - fe_p).abs().mean().reshape(1)

    def fit(self, V, verbose=1):
        """Train the RBM on V [N, n_vis] by CD-k (rbm.py:100-234).

        V: numpy array, torch tensor (host or device) -- uploaded once and kept resident.
        verbose: 1 prints the epoch counter and the per-step score as the reference does
                 (rbm.py:114-115, :234); 0 prints nothing and skips the three score passes.
        Returns None.
        """
        V = _unwrap(V)
        n_cols = V.cols if isinstance(V, DeviceMatrix) else V.shape[1]
        self._ensure_built(n_cols)
        Vd, _ = self._as_device(V)
        if Vd.cols != self._dev.n_vis:
            raise ValueError("V has %d columns, the RBM has %d visible units" % (Vd.cols, self._dev.n_vis))
        bs = int(self.hps["batch_size"])
        lr = float(self.hps["lr"])
        n = Vd.rows
        num_step = n // bs if n % bs == 0 else n // bs + 1                     # rbm.py:110-111
        rank, world = dp.world()
        d = self._dev
        self.last_scores = []
        self._data_real = self.compute_dtype == "auto" and d.v_pieces(Vd) != 1   # ('auto' asks once what the data is)
        if self.persistent and self._v_chain is None:
            # fantasy particles start at the first batch of the data
            self._v_chain = DeviceMatrix.zeros(bs, d.n_vis, d.device)
            first = min(bs, n)
            self._v_chain.t[:first].copy_(Vd.t[:first])
        if self.persistent and (self._v_chain.rows != bs or self._v_chain.cols != d.n_vis):
            # every step reads and rewrites `rows` rows of the chain in place: a chain kept from a fit with another batch
            # size (or loaded from such a checkpoint) would be read past its end
            raise ValueError("the persistent chain has shape (%d, %d); hps['batch_size'] = %d and %d visible units need (%d, %d) -- "
                             "set rbm._v_chain = None to restart the chain from the data"
                             % (self._v_chain.rows, self._v_chain.cols, bs, d.n_vis, bs, d.n_vis))

        # x3: the bf16 planes of every window of rows this fit walks are made once, here (the data is the same every
        # epoch); the fallback -- not enough HBM -- is the per-step conversion inside the library
        planes = None
        if self._compute() == "x3" and n > 0:
            windows = []
            for i in range(num_step):
                lo, rows = i * bs, min((i + 1) * bs, n) - i * bs
                if world > 1:
                    s_lo, s_hi = self._shard(rows, rank, world)
                    lo, rows = lo + s_lo, s_hi - s_lo
                windows.append((lo, rows))
            planes = d.make_planes(Vd, windows, self.mode, self._v_chain if self.persistent else None)
        self._planes = planes
        # quiet single-GPU fused training (fp32 MFMA or x3): the whole batch loop of an epoch is one library call
        whole_epochs = (verbose != 1 and world == 1 and self.update_mode == "fused"
                        and self._compute() in ("fp32", "x3", "small"))
        # ... and with the per-step score (the reference's default), for the one-launch small path
        scored_epochs = (verbose == 1 and world == 1 and self.update_mode == "fused" and self._compute() == "small" and bs <= 512)
        def print_score(score, label):
            self.last_scores.append(score)
            print("\n{0:d}/{1:d}, score: {2:f}".format(label[0], label[1], score))   # rbm.py:234
        ring = _ScoreRing(d.device, print_score) if verbose == 1 else None
        for epoch in range(int(self.hps["epochs"])):                              # rbm.py:113
            if verbose == 1:
                print(epoch + 1, "/", self.hps["epochs"], " epochs", end="\r")   # rbm.py:115
            if whole_epochs:
                self._update_count += d.cd_epoch(Vd, n, bs, lr, self.seed, self._update_count, k=self.cd_k,
                                                 mode=self.mode, v_chain=self._v_chain if self.persistent else None,
                                                 compute=self._compute(), planes=planes)
                continue
            if scored_epochs:
                # small RBMs with the reference's default verbose = 1: the epoch's updates AND scores are queued by one library
                # call (two launches per step); the scores land in pinned host memory, and the lines are printed, in order, as they do
                steps, scores = d.cd_epoch_small_scored(Vd, n, bs, lr, self.seed, self._update_count, self.mode, CHAIN_SCORE)
                self._update_count += steps
                flags = scores.numpy()
                for i in range(steps):
                    spins = 0
                    while flags[i, 1] != 1.0:
                        spins += 1
                        if spins > 2000:
                            time.sleep(50e-6)
                        if spins > 400000:     # (~20 s: the device never wrote it -- a skipped launch; the status word says why)
                            d.check_status()
                            raise _lib.KurbmError("the score of step %d never arrived" % (i + 1))
                    print_score(float(flags[i, 0]), (i + 1, num_step))
                continue
            for i in range(num_step):                                            # rbm.py:163
                lo, hi = i * bs, min((i + 1) * bs, n)                            # rbm.py:211 / :218
                rows = hi - lo
                step = self._update_count
                if world == 1:
                    self._update_local(Vd, lo, rows, lr, step)
                else:
                    self._update_data_parallel(Vd, lo, rows, lr, step, rank, world)
                self._update_count += 1
                if verbose == 1:
                    # (the line of step i is printed when its score has landed: at most one step late, never out of order)
                    ring.push(self._score(Vd, lo, rows, step), (i + 1, num_step))
            if ring is not None:
                ring.flush()                                                     # an epoch's lines end with the epoch
        self._planes = None
        d.check_status()                  # fit() ends with one status read: no later call trains on from a skipped update
        return None

    def _compute(self):
        """The compute path of this fit.  'auto' reads tools/small_crossover.py's tables (784 visible units, one MI355X;
        profiles/r04_e_compute_path_crossover.txt):
        * the one-launch step (CD-1 from the data, one GPU) up to batch 128 with up to 1024 hidden units, batch 256 with up to 512 and
          batch 384 with up to 256, batch 512 with up to 128 -- 21-56 us per step where the multi-launch paths take 55-110;
        * 0/1 data in Bernoulli mode: x3 at every other size (it beats the fp32-MFMA launches from batch 64 on);
        * real-valued data or Gaussian visibles (three pieces per value, seven launches): x3 from rows x n_vis x n_hid >= 6e8 on
          (batch 1024 at 784 x 1024, 1536 at 784 x 512), the fp32-MFMA kernels below."""
        c = self.compute_dtype
        small_ok = self.cd_k == 1 and not self.persistent and dp.world()[1] == 1
        if c == "auto":
            b, h = int(self.hps["batch_size"]), int(self.output_dim)
            if small_ok and ((b <= 128 and h <= 1024) or (b <= 256 and h <= 512) or (b <= 384 and h <= 256) or (b <= 512 and h <= 128)):
                return "small"
            real = self.mode == MODE_VISIBLE_GAUSSIAN or getattr(self, "_data_real", False)
            n_vis = self._dev.n_vis if self._dev is not None else 784
            return "x3" if (not real or float(b) * n_vis * h >= 6e8) else "fp32"
        if c == "small" and not small_ok:
            return "fp32"     # the one-launch step is CD-1 from the data on one GPU; everything else runs the five-launch path
        return c

    def _update_local(self, Vd, lo, rows, lr, step):
        d = self._dev
        if self.update_mode == "fused":
            d.cd_step(Vd, rows, lo, lr, self.seed, step, k=self.cd_k, mode=self.mode, chain=CHAIN_W,
                      v_chain=self._v_chain if self.persistent else None, compute=self._compute(), planes=self._planes)
        else:
            # the reference's three K.function calls: each its own chain, each seeing the variables
            # the previous call already updated (rbm.py:214-216)
            for chain, which in ((CHAIN_W, _lib.WHICH_W), (CHAIN_BH, _lib.WHICH_BH), (CHAIN_BV, _lib.WHICH_BV)):
                d.cd_step(Vd, rows, lo, lr, self.seed, step, k=1, mode=self.mode, chain=chain, which=which,
                          compute=self._compute(), planes=self._planes)

    def _update_data_parallel(self, Vd, lo, rows, lr, step, rank, world):
        """Each rank: the chain on its rows of the batch -> packed sums -> sum all-reduce (RCCL, inside libkurbm.so) ->
        the identical update on every replica.  Philox counters use the row's index in the global batch (row0)."""
        if self.update_mode != "fused":
            raise ValueError("data-parallel training supports update_mode='fused' only")
        d = self._dev
        s_lo, s_hi = self._shard(rows, rank, world)
        d.cd_step_dp(dp.get_exchange(d.device, d.n_vis, d.n_hid), Vd, s_hi - s_lo, lo + s_lo, lr, self.seed, step, k=self.cd_k, mode=self.mode,
                     chain=CHAIN_W, row0=s_lo, v_chain=self._v_chain if self.persistent else None, v_chain_row=s_lo,
                     compute=self._compute(), planes=self._planes)

    def full_chain(self):
        """The persistent chain as ONE host array [batch_size, n_vis], or None.  Under data parallelism rank r advances
        only its own rows of the chain (`_shard`); every other row it holds is stale.  This call is then COLLECTIVE: each
        rank contributes its rows, zeros elsewhere, to a sum all-reduce on the library's communicator, and every rank
        returns the assembled chain."""
        if self._v_chain is None:
            return None
        rank, world = dp.world()
        if world == 1:
            return self._v_chain.to_numpy()
        lo, hi = self._shard(self._v_chain.rows, rank, world)
        with device_guard(self._dev.device):
            own = torch.zeros_like(self._v_chain.t)
            own[lo:hi].copy_(self._v_chain.t[lo:hi])
            dp.get_exchange(self._dev.device, self._dev.n_vis, self._dev.n_hid).allreduce_sum_(own.view(-1))
            return own[: self._v_chain.rows, : self._v_chain.cols].contiguous().cpu().numpy()

    def _shard(self, rows, rank, world):
        """This rank's rows [lo, hi) of a batch of `rows` rows (persistent chains: fixed ownership of the chain's rows,
        whatever the batch's row count)."""
        return dp.shard_rows(rows, world, rank, of=int(self.hps["batch_size"]) if self.persistent else None)

    # ------------------------------------------------------------------ config ----------
    def get_config(self):
        """hps, output_dim, name as in rbm.py:236-242, plus `mode` (dropped by the reference) and the
        extension knobs, so that RBM(**config) rebuilds the same layer."""
        config = {"hps": self.hps, "output_dim": self.output_dim, "name": self.name, "mode": self.mode,
                  "seed": self.seed, "update_mode": self.update_mode, "cd_k": self.cd_k,
                  "persistent": self.persistent, "compute_dtype": self.compute_dtype}
        return dict(list(self._kwargs.items()) + list(config.items()))


__all__ = ["RBM", "MODE_VISIBLE_BERNOULLI", "MODE_VISIBLE_GAUSSIAN", "MODE_COMPLEX"]
