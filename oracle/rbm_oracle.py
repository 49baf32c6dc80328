"""CPU restatement of the reference RBM / DBN contrastive-divergence path (numpy).

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; ``keras_unsupervised_amd`` never does and fails loudly when its HIP
library is missing.

PARITY UNPINNED.  The reference (``/root/reference/ku/ebm/rbm.py``, ``dbn.py``)
delegates all arithmetic to TensorFlow/Keras backend ops (TF ~2.3 implied by
``tensorflow-probability==0.11`` in the reference ``setup.py:70``), which is not
in ``/root/reference`` and not installed here; the reference has no tests,
golden vectors or fixtures for this path, and the path is not
runnable as written (SURVEY.md 8(a) "Defects").  This file therefore follows
``rbm.py`` / ``dbn.py`` line by line under the documented *minimal-repair*
reading, and the random stream is the build's own Philox contract
(``oracle/philox.py``).  Golden vectors under ``tests/golden/`` are produced by
this file (``oracle/make_golden.py``), cross-checked against its float64 shadow.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).
"""
import numpy as np

from . import philox

# ku/ebm/rbm.py:14-16
MODE_VISIBLE_BERNOULLI = 0
MODE_VISIBLE_GAUSSIAN = 1
MODE_COMPLEX = 2

# ---- sampling-site ("stream") ids of the build's RNG contract ----------------
# Gibbs chain of one parameter update: h_0 is the positive-phase hidden sample,
# v_t / h_t (t >= 1) the negative-phase states.  reference_sequential mode runs
# several independent chains per step; chain c offsets the ids by CHAIN_STRIDE*c.
CHAIN_STRIDE = 64
STREAM_TRANSFORM = 0x100      # RBM.transform / RBM.call draws      (rbm.py:46,82)
STREAM_INV_TRANSFORM = 0x101  # RBM.inv_transform draws              (rbm.py:52)
CHAIN_W, CHAIN_BH, CHAIN_BV, CHAIN_SCORE = 0, 1, 2, 3


def stream_h(t, chain=0):
    """Stream id of the hidden sample at Gibbs iteration t (t=0: positive phase)."""
    return chain * CHAIN_STRIDE + 2 * t


def stream_v(t, chain=0):
    """Stream id of the visible sample at Gibbs iteration t >= 1."""
    return chain * CHAIN_STRIDE + 2 * t - 1


def init_params(n_vis, n_hid, seed=0, dtype=np.float32):
    """Keras ``'uniform'`` initialiser = U(-0.05, 0.05) for W, b_h, b_v.

    ku/ebm/rbm.py:30-40.  (The reference's own stream is TF's; here a numpy
    Generator keyed by ``seed`` -- tests always pass explicit arrays to both sides.)
    """
    g = np.random.default_rng(seed)
    W = g.uniform(-0.05, 0.05, size=(n_vis, n_hid)).astype(dtype)
    b_h = g.uniform(-0.05, 0.05, size=(n_hid,)).astype(dtype)
    b_v = g.uniform(-0.05, 0.05, size=(n_vis,)).astype(dtype)
    return W, b_h, b_v


def bf16_round(x):
    """float32 -> nearest-even bf16 -> float32: what the bf16 kernels feed the matrix cores.

    Extension of the build (BASELINE.json config 5), absent from the reference.  Products of two bf16
    values are exact in float32 and the MFMA accumulates in float32, so a float32 matmul of rounded
    operands is the CPU statement of the bf16 path."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def bf16_split3(x):
    """Exact three-way split of float32 values into bf16 pieces, x == hi + mid + lo (the x3 compute path:
    include/kurbm.h, kurbm_device.h bf16_piece_bits).  Round to nearest even at each stage; both residuals
    are exact in float32 and the last one has at most 8 significant bits, so the sum is exact.  The x3
    kernels multiply the pieces on the bf16 matrix cores and accumulate in float32: their CPU statement
    is the plain float32 oracle (cd_step_fused), not a reduced-precision one."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_round(x)
    r1 = x - hi
    mid = bf16_round(r1)
    lo = bf16_round(r1 - mid)
    return hi, mid, lo


def cd_step_fused_bf16(W, b_h, b_v, v_batch, lr, seed, step, k=1, row0=0, mode=MODE_VISIBLE_BERNOULLI, v_chain=None):
    """cd_step_fused with every matrix-product operand rounded to bf16 (weights, data, the h_neg
    probabilities; 0/1 samples are exact); biases, activations, sums and the update stay float32 and
    the update is applied to the float32 master weights."""
    Wq = bf16_round(W)
    rng = Rng(seed, step, row0)
    ch = gibbs_chain(bf16_round(v_batch), Wq, b_h, b_v, rng, k=k, chain=CHAIN_W, mode=mode,
                     v_chain=None if v_chain is None else bf16_round(v_chain))
    vq, vn = bf16_round(v_batch), bf16_round(ch["v_neg"])
    dW = vq.T @ ch["h_pos"] - vn.T @ bf16_round(ch["h_neg"])
    db_h = ch["h_pos"].sum(axis=0) - ch["h_neg"].sum(axis=0)
    db_v = v_batch.sum(axis=0) - ch["v_neg"].sum(axis=0)
    lr = W.dtype.type(lr)
    return W + lr * dW, b_h + lr * db_h, b_v + lr * db_v, ch, (dW, db_h, db_v)


def sigmoid(x):
    one = x.dtype.type(1.0)
    return one / (one + np.exp(-x))


class Rng:
    """Names one (seed, step, row0) context; draws are pure functions of it."""

    def __init__(self, seed, step=0, row0=0):
        self.seed, self.step, self.row0 = int(seed), int(step), int(row0)

    def uniform(self, rows, cols, stream_id):
        return philox.uniform(rows, cols, self.seed, stream_id, self.step, self.row0)

    def normal(self, rows, cols, stream_id):
        return philox.normal(rows, cols, self.seed, stream_id, self.step, self.row0)


# ------------------------------------------------------------------------------
# half steps
# ------------------------------------------------------------------------------
def hidden_prob(v, W, b_h, mode=MODE_VISIBLE_BERNOULLI):
    """Hidden "probability" given visible.

    Bernoulli mode: sigmoid(v.W + b_h)     ku/ebm/rbm.py:47, :83, :124
    Gaussian  mode: relu(v.W + b_h) as the Bernoulli threshold   rbm.py:59, :86
    """
    a = v @ W + b_h
    if mode == MODE_VISIBLE_GAUSSIAN:
        return np.maximum(a, a.dtype.type(0))
    return sigmoid(a)


def sample_hidden(v, W, b_h, rng, stream_id, mode=MODE_VISIBLE_BERNOULLI):
    """v -> h half step: h = float(u < p_h), u ~ U[0,1).

    ku/ebm/rbm.py:46-47 (and :58-59, :82-86) under the minimal repair: the uniform has
    the shape of the probabilities (rows of the input x n_hid), cast to float32.
    Returns (p, u, h).
    """
    p = hidden_prob(v, W, b_h, mode)
    u = rng.uniform(p.shape[0], p.shape[1], stream_id)
    h = (u < p).astype(p.dtype)
    return p, u, h


def visible_prob(h, W, b_v):
    """sigmoid(h.W^T + b_v)   ku/ebm/rbm.py:53, :122."""
    return sigmoid(h @ W.T + b_v)


def sample_visible(h, W, b_v, rng, stream_id, mode=MODE_VISIBLE_BERNOULLI):
    """h -> v half step.

    Bernoulli mode: v = float(u < sigmoid(h.W^T + b_v))          rbm.py:52-53, :121-123
    Gaussian  mode: v ~ N(h.W^T + b_v, I)  (MultivariateNormalDiag with unit scale,
                    rbm.py:64-66, :143-144 -> backend_ext/tensorflow_backend.py:32-46)
    Returns (p_or_loc, noise, v).
    """
    if mode == MODE_VISIBLE_GAUSSIAN:
        loc = h @ W.T + b_v
        z = rng.normal(loc.shape[0], loc.shape[1], stream_id).astype(loc.dtype)
        return loc, z, loc + z
    p = visible_prob(h, W, b_v)
    u = rng.uniform(p.shape[0], p.shape[1], stream_id)
    v = (u < p).astype(p.dtype)
    return p, u, v


def free_energy(v, W, b_h, b_v, stable=True):
    """F(v) = -( v.b_v + sum_j log(1 + exp((v.W + b_h)_j)) )   ku/ebm/rbm.py:73-75.

    ``stable=False`` is the reference's literal ``log(1 + exp(x))`` (overflows to inf
    in float32 for x > ~88); ``stable=True`` is max(x,0) + log1p(exp(-|x|)), the form
    the kernels use -- identical wherever the literal form is finite.
    """
    a = v @ W + b_h
    if stable:
        sp = np.maximum(a, a.dtype.type(0)) + np.log1p(np.exp(-np.abs(a)))
    else:
        with np.errstate(over="ignore"):
            sp = np.log(a.dtype.type(1) + np.exp(a))
    return -(v @ b_v + sp.sum(axis=-1))


# ------------------------------------------------------------------------------
# one Gibbs chain and its sufficient statistics
# ------------------------------------------------------------------------------
def gibbs_chain(v_pos, W, b_h, b_v, rng, k=1, chain=0, mode=MODE_VISIBLE_BERNOULLI,
                v_chain=None):
    """CD-k chain as built in ku/ebm/rbm.py:119-124 (k = 1 there).

    v_pos -> h_pos (sample) -> v_1 (sample) -> h_1 ... -> v_k (sample) -> h_k (PROBABILITY,
    rbm.py:124 ``h_neg = K.sigmoid(...)`` -- sigmoid in both modes, rbm.py:145).
    ``v_chain`` (persistent CD, an extension absent from the reference): if given, the
    negative phase starts from h ~ p(h | v_chain) instead of h_pos.
    Returns dict with v_pos, h_pos, v_neg, h_neg and the intermediate probabilities.
    """
    out = {}
    p_h0, u_h0, h_pos = sample_hidden(v_pos, W, b_h, rng, stream_h(0, chain), mode)
    out.update(p_h0=p_h0, u_h0=u_h0, h_pos=h_pos)
    if v_chain is None:
        h = h_pos
    else:
        _, _, h = sample_hidden(v_chain, W, b_h, rng, stream_h(0, chain) + 32, mode)
    v = None
    for t in range(1, k + 1):
        p_v, n_v, v = sample_visible(h, W, b_v, rng, stream_v(t, chain), mode)
        if t == 1:
            out.update(p_v1=p_v, u_v1=n_v)
        if t < k:
            _, _, h = sample_hidden(v, W, b_h, rng, stream_h(t, chain), mode)
    h_neg = sigmoid(v @ W + b_h)  # rbm.py:124 / :145
    out.update(v_pos=v_pos, v_neg=v, h_neg=h_neg)
    return out


def cd_statistics(ch):
    """dW, db_h, db_v of one chain -- sums over the batch, not means.

    dW  = v_pos^T.h_pos - v_neg^T.h_neg                       ku/ebm/rbm.py:125-126
    db_h = sum_b h_pos - sum_b h_neg                           rbm.py:130-131
    db_v = sum_b v_pos - sum_b v_neg                           rbm.py:133-134
    """
    dW = ch["v_pos"].T @ ch["h_pos"] - ch["v_neg"].T @ ch["h_neg"]
    db_h = ch["h_pos"].sum(axis=0) - ch["h_neg"].sum(axis=0)
    db_v = ch["v_pos"].sum(axis=0) - ch["v_neg"].sum(axis=0)
    return dW, db_h, db_v


def cd_step_fused(W, b_h, b_v, v_batch, lr, seed, step, k=1, row0=0,
                  mode=MODE_VISIBLE_BERNOULLI, v_chain=None):
    """One parameter update from ONE chain (the algorithm the metric counts).

    W += lr*dW, b_h += lr*db_h, b_v += lr*db_v   (ku/ebm/rbm.py:127-134), all three
    from the same chain and the same pre-update parameters.
    Returns (W, b_h, b_v, chain_dict, (dW, db_h, db_v)).
    """
    lr = W.dtype.type(lr)
    rng = Rng(seed, step, row0)
    ch = gibbs_chain(v_batch, W, b_h, b_v, rng, k=k, chain=CHAIN_W, mode=mode, v_chain=v_chain)
    dW, db_h, db_v = cd_statistics(ch)
    return W + lr * dW, b_h + lr * db_h, b_v + lr * db_v, ch, (dW, db_h, db_v)


def cd_step_reference_sequential(W, b_h, b_v, v_batch, lr, seed, step,
                                 mode=MODE_VISIBLE_BERNOULLI):
    """The reference's three K.function calls, verbatim (ku/ebm/rbm.py:214-216 / :221-223).

    Each call re-executes its own graph with fresh random draws, and sees the variables
    the previous call already updated: chain 0 -> W; chain 1 (new W) -> b_h; chain 2
    (new W, new b_h) -> b_v.   [TF graph-mode semantics; SURVEY.md 8(a) item 3]
    """
    lr = W.dtype.type(lr)
    rng = Rng(seed, step, 0)
    ch = gibbs_chain(v_batch, W, b_h, b_v, rng, chain=CHAIN_W, mode=mode)
    W = W + lr * cd_statistics(ch)[0]
    ch = gibbs_chain(v_batch, W, b_h, b_v, rng, chain=CHAIN_BH, mode=mode)
    b_h = b_h + lr * cd_statistics(ch)[1]
    ch = gibbs_chain(v_batch, W, b_h, b_v, rng, chain=CHAIN_BV, mode=mode)
    b_v = b_v + lr * cd_statistics(ch)[2]
    return W, b_h, b_v


def step_score(W, b_h, b_v, v_batch, seed, step, mode=MODE_VISIBLE_BERNOULLI):
    """Per-step training score, ku/ebm/rbm.py:225-233.

    fe = F(v_batch); v' = sample_first_visible(v_batch) (a fresh chain, rbm.py:137-138,
    :230); fe' = F(v'); score = mean |fe - fe'|, all with the post-update parameters.
    """
    rng = Rng(seed, step, 0)
    fe = free_energy(v_batch, W, b_h, b_v)
    _, _, h = sample_hidden(v_batch, W, b_h, rng, stream_h(0, CHAIN_SCORE), mode)
    _, _, v1 = sample_visible(h, W, b_v, rng, stream_v(1, CHAIN_SCORE), mode)
    fe_p = free_energy(v1, W, b_h, b_v)
    return float(np.mean(np.abs(fe - fe_p)))


def batch_slices(n, batch_size):
    """Contiguous, in-order, unshuffled batches; last one is the remainder.

    ku/ebm/rbm.py:110-111 (num_step) and :211, :218 (slices); the remainder shape is the
    repaired ``int(a, b)`` of :169/:192.
    """
    num_step = n // batch_size if n % batch_size == 0 else n // batch_size + 1
    return [(i * batch_size, min((i + 1) * batch_size, n)) for i in range(num_step)]


def fit(W, b_h, b_v, V, hps, seed=0, update_mode="fused", k=1, mode=MODE_VISIBLE_BERNOULLI,
        step0=0, with_score=False):
    """RBM.fit, ku/ebm/rbm.py:100-234.

    for epoch in range(hps['epochs']):  for each batch in order:  three updates (or the
    fused single-chain update); score per step (rbm.py:227-233).
    Returns (W, b_h, b_v, scores, next_step).
    """
    scores = []
    step = step0
    for _ in range(hps["epochs"]):                      # rbm.py:113
        for lo, hi in batch_slices(V.shape[0], hps["batch_size"]):   # rbm.py:163
            vb = V[lo:hi]
            if update_mode == "fused":
                W, b_h, b_v, _, _ = cd_step_fused(W, b_h, b_v, vb, hps["lr"], seed, step, k=k, mode=mode)
            elif update_mode == "reference_sequential":
                W, b_h, b_v = cd_step_reference_sequential(W, b_h, b_v, vb, hps["lr"], seed, step, mode=mode)
            else:
                raise ValueError(update_mode)
            if with_score:
                scores.append(step_score(W, b_h, b_v, vb, seed, step, mode))
            step += 1
    return W, b_h, b_v, scores, step


def transform(W, b_h, v, seed, call_index, mode=MODE_VISIBLE_BERNOULLI, row0=0):
    """RBM.transform (ku/ebm/rbm.py:88-89 -> :46-48): sampled hidden states."""
    return sample_hidden(v, W, b_h, Rng(seed, call_index, row0), STREAM_TRANSFORM, mode)[2]


def inv_transform(W, b_v, h, seed, call_index, mode=MODE_VISIBLE_BERNOULLI, row0=0):
    """RBM.inv_transform (ku/ebm/rbm.py:91-92 -> :52-54 / :64-67): sampled visible states."""
    return sample_visible(h, W, b_v, Rng(seed, call_index, row0), STREAM_INV_TRANSFORM, mode)[2]


# ------------------------------------------------------------------------------
# DBN (ku/ebm/dbn.py) -- greedy layer-wise stack
# ------------------------------------------------------------------------------
class OracleLayer:
    """Parameters + counters of one RBM inside the oracle DBN."""

    def __init__(self, W, b_h, b_v, hps, seed, mode=MODE_VISIBLE_BERNOULLI):
        self.W, self.b_h, self.b_v = W, b_h, b_v
        self.hps, self.seed, self.mode = hps, seed, mode
        self.step = 0          # parameter-update counter
        self.calls = 0         # transform / inv_transform call counter

    def fit(self, V, update_mode="fused"):
        self.W, self.b_h, self.b_v, _, self.step = fit(
            self.W, self.b_h, self.b_v, V, self.hps, self.seed, update_mode, mode=self.mode,
            step0=self.step)

    def transform(self, V):
        out = transform(self.W, self.b_h, V, self.seed, self.calls, self.mode)
        self.calls += 1
        return out

    def inv_transform(self, H):
        out = inv_transform(self.W, self.b_v, H, self.seed, self.calls, self.mode)
        self.calls += 1
        return out


def dbn_fit(layers, V, update_mode="fused"):
    """DBN.fit, ku/ebm/dbn.py:34-55 (repaired ``self.rbm_layer`` -> loop variable):
    V_p = V.copy(); for each layer: layer.fit(V_p); V_p = layer.transform(V_p)."""
    if not layers:
        raise ValueError("Any rbm layer doesn't exist.")   # dbn.py:47-48
    V_p = V.copy()
    for layer in layers:
        layer.fit(V_p, update_mode)
        V_p = layer.transform(V_p)
    return V_p


def dbn_transform(layers, V):
    """DBN.transform, ku/ebm/dbn.py:57-75."""
    if not layers:
        raise ValueError("Any rbm layer doesn't exist.")   # dbn.py:68-69
    V_p = V.copy()
    for layer in layers:
        V_p = layer.transform(V_p)
    return V_p


def dbn_inv_transform(layers, H):
    """DBN.inv_transform, ku/ebm/dbn.py:77-95 with the repaired (reverse) loop of :92."""
    if not layers:
        raise ValueError("Any rbm layer doesn't exist.")   # dbn.py:88-89
    H_p = H.copy()
    for layer in reversed(layers):
        H_p = layer.inv_transform(H_p)
    return H_p
