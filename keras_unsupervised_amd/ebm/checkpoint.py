"""Checkpoint / config round trip of RBM and DBN objects.

The reference has no EBM-specific format: the example saves the whole Keras model to h5
(reference examples/rbm/rbm_softmax_mnist.py:94, :48) and `ku/utility.py:7-34` writes model JSON + h5
weights.  Both lose part of an RBM: `get_config` omits `mode` (reference ku/ebm/rbm.py:236-242) and
`visible_bias` is a bare variable outside `get_weights()` (rbm.py:38-40).  Here a checkpoint is

    <stem>.json          get_config() (hps, output_dim, name, mode, seed, update_mode, cd_k, persistent)
                         + input_dim + the RNG counters, so a reloaded RBM continues the same stream
    <stem>.safetensors   rbm_weight [n_vis, n_hid], rbm_hidden_bias [n_hid], rbm_visible_bias [n_vis], fp32;
                         with persistent=True also v_chain [batch_size, n_vis], the fantasy particles, so that a
                         reloaded RBM continues the chain it was saved with (data parallel: assembled over the ranks,
                         each of which owns a band of its rows; save_rbm is then a collective call, rank 0 writes)

and a DBN checkpoint is a JSON list of layer stems.
"""
import json
import os

import numpy as np
from safetensors.numpy import load_file, save_file

FORMAT_VERSION = 1


def save_rbm(rbm, stem):
    """Write <stem>.json and <stem>.safetensors for a built RBM.  Returns the two paths."""
    if not rbm.built:
        raise ValueError("cannot checkpoint an RBM that has not been built")
    W, b_h, b_v = rbm.get_weights()
    cfg = {k: v for k, v in rbm.get_config().items() if k in
           ("hps", "output_dim", "name", "mode", "seed", "update_mode", "cd_k", "persistent", "compute_dtype")}
    meta = {"format": "kurbm-rbm", "version": FORMAT_VERSION, "config": cfg, "input_dim": int(W.shape[0]),
            "update_count": int(rbm._update_count), "call_count": int(rbm._call_count)}
    os.makedirs(os.path.dirname(os.path.abspath(stem)), exist_ok=True)
    tensors = {"rbm_weight": np.ascontiguousarray(W, dtype=np.float32),
               "rbm_hidden_bias": np.ascontiguousarray(b_h, dtype=np.float32),
               "rbm_visible_bias": np.ascontiguousarray(b_v, dtype=np.float32)}
    if rbm._v_chain is not None:
        # data parallel: every rank advances only its own rows of the chain, so the chain is assembled over the ranks first
        # (RBM.full_chain: a collective -- every rank calls save_rbm; rank 0 writes the files)
        tensors["v_chain"] = np.ascontiguousarray(rbm.full_chain(), dtype=np.float32)
    from . import dp
    if dp.world()[0] != 0:
        return stem + ".json", stem + ".safetensors"
    with open(stem + ".json", "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    save_file(tensors, stem + ".safetensors")
    return stem + ".json", stem + ".safetensors"


def load_rbm(stem, device=None):
    """Rebuild the RBM saved by save_rbm (on `device`, default the current ROCm device)."""
    from .rbm import RBM
    with open(stem + ".json") as f:
        meta = json.load(f)
    if meta.get("format") != "kurbm-rbm" or meta.get("version") != FORMAT_VERSION:
        raise ValueError("%s.json is not a kurbm RBM checkpoint (format %r, version %r)"
                         % (stem, meta.get("format"), meta.get("version")))
    t = load_file(stem + ".safetensors")
    W, b_h, b_v = t["rbm_weight"], t["rbm_hidden_bias"], t["rbm_visible_bias"]
    cfg = dict(meta["config"])
    if W.shape != (meta["input_dim"], cfg["output_dim"]) or b_h.shape != (cfg["output_dim"],) \
            or b_v.shape != (meta["input_dim"],):
        raise ValueError("checkpoint tensors do not match its config")
    rbm = RBM(weights=(W, b_h, b_v), device=device, **cfg)
    rbm.build((None, meta["input_dim"]))
    rbm._update_count = int(meta.get("update_count", 0))
    rbm._call_count = int(meta.get("call_count", 0))
    if "v_chain" in t:
        from .engine import DeviceMatrix, device_guard
        # (fit() reads and rewrites batch_size rows of the chain in place: any other row count is refused here, not there)
        if t["v_chain"].shape != (int(cfg["hps"]["batch_size"]), meta["input_dim"]):
            raise ValueError("checkpoint chain has shape %s, its config needs (%d, %d)"
                             % (t["v_chain"].shape, int(cfg["hps"]["batch_size"]), meta["input_dim"]))
        with device_guard(rbm._dev.device):
            rbm._v_chain = DeviceMatrix.from_host(t["v_chain"], rbm._dev.device)
    return rbm


def save_dbn(dbn, stem):
    """One RBM checkpoint per layer + <stem>.json listing them."""
    layers = dbn._layers()
    names = []
    for i, layer in enumerate(layers):
        save_rbm(layer, "%s.layer%d" % (stem, i))
        names.append(os.path.basename("%s.layer%d" % (stem, i)))
    with open(stem + ".json", "w") as f:
        json.dump({"format": "kurbm-dbn", "version": FORMAT_VERSION, "layers": names}, f, indent=1)
    return stem + ".json"


def load_dbn(stem, device=None):
    from .dbn import DBN
    with open(stem + ".json") as f:
        meta = json.load(f)
    if meta.get("format") != "kurbm-dbn":
        raise ValueError("%s.json is not a kurbm DBN checkpoint" % stem)
    dbn = DBN()
    base = os.path.dirname(os.path.abspath(stem))
    for name in meta["layers"]:
        dbn.add_stack(load_rbm(os.path.join(base, name), device))
    return dbn
