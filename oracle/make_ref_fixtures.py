"""Fixtures made by RUNNING reference code: the only part of the hot path that executes in the build container.

    python -m oracle.make_ref_fixtures          (repo root, build container only: needs /root/reference)

TEST INFRASTRUCTURE.  `/root/reference/ku/ebm/dbn.py` has no imports (dbn.py:1-8), so it loads by file path without
TensorFlow; `ku/ebm/rbm.py` does not (rbm.py:7-11 import the Keras backend), which is why the ARITHMETIC of the path stays
"parity unpinned" (DESIGN.md section 2).  What runs here is the DBN's control flow, with stub layers whose `transform` /
`inv_transform` are the oracle's (oracle/rbm_oracle.py: OracleLayer), and it pins exactly that:

  * `DBN.add_stack` (dbn.py:14-32): the first call takes the `else` branch (:31-32) and creates `_rbm_layers`; the second
    call dies on `self.rbm_layer` (:25) -- recorded as the exception type and message, the defect SURVEY.md 8(a) lists;
  * `DBN.transform` (dbn.py:57-75) on a two-layer stack (the second layer appended to `_rbm_layers` by hand, since :25
    cannot): layers in order, one `transform` call each, the input array never written (`V.copy()`, :65);
  * `DBN.inv_transform` (dbn.py:77-95) AS WRITTEN: `range(len(layers), -1)` is empty (:92), so it returns `H.copy()`
    unchanged and calls no layer -- the repair (a reverse walk) is this build's, documented, and the fixture holds both;
  * `DBN.fit` (dbn.py:34-55): prints 'Train <name>.' (:53) and dies on `self.rbm_layer` (:54);
  * the three `ValueError`s of an empty DBN (dbn.py:47-48, :68-69, :88-89).

Nothing of the reference travels: the fixture `tests/golden/ref_dbn.npz` holds inputs' seeds, output arrays, call logs and
exception texts -- data, no source.  tests/test_ref_fixtures.py holds `O.dbn_transform` / `O.dbn_inv_transform` and the host
`DBN` class to it on the CPU tier.
"""
import contextlib
import importlib.util
import io
import json
import os

import numpy as np

from . import rbm_oracle as O
from .make_golden import GOLDEN_DIR, synthetic_binary, synthetic_params

REF_DBN = "/root/reference/ku/ebm/dbn.py"
DIMS = (12, 10, 6)          # visible -> hidden 1 -> hidden 2
ROWS = 8
SEEDS = (31, 32)            # parameter seeds of the two layers; sampler seeds are seed + 100
HPS = {"batch_size": 4, "epochs": 1, "lr": 1e-3}


# sha256 of the dbn.py this script was written against and has been read line by line (no imports, no module-level side
# effects): code from the public reference tree is executed only if it is still THAT file
REF_DBN_SHA256 = "87b38b3a7df6d080600191e1d21d17a2d7fee8db5e9653d885f5e93d658c673d"


def load_reference_dbn():
    import hashlib
    with open(REF_DBN, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    if digest != REF_DBN_SHA256:
        raise RuntimeError("%s is not the file this script was reviewed against (sha256 %s): not executing it" % (REF_DBN, digest))
    spec = importlib.util.spec_from_file_location("ref_ku_ebm_dbn", REF_DBN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def oracle_layers():
    layers = []
    for i in range(2):
        W, b_h, b_v = synthetic_params(DIMS[i], DIMS[i + 1], seed=SEEDS[i])
        layers.append(O.OracleLayer(W, b_h, b_v, dict(HPS), SEEDS[i] + 100))
    return layers


class StubLayer:
    """What dbn.py touches of an RBM: name, input_shape / output_shape, fit / transform / inv_transform."""

    def __init__(self, name, core, log):
        self.name, self.core, self.log = name, core, log
        self.input_shape = (None, core.W.shape[0])
        self.output_shape = (None, core.W.shape[1])

    def fit(self, V, verbose=1):
        self.log.append("%s.fit" % self.name)

    def transform(self, V):
        self.log.append("%s.transform" % self.name)
        return self.core.transform(V)

    def inv_transform(self, H):
        self.log.append("%s.inv_transform" % self.name)
        return self.core.inv_transform(H)


def outcome(fn):
    """('ok', result) or (exception type name, message)."""
    try:
        return "ok", fn()
    except Exception as e:  # noqa: BLE001 -- the exception IS the datum
        return type(e).__name__, str(e)


def main():
    ref = load_reference_dbn()
    V = synthetic_binary(ROWS, DIMS[0], seed=33)
    H = synthetic_binary(ROWS, DIMS[2], seed=34)
    facts = {"reference_file": "ku/ebm/dbn.py", "dims": DIMS, "rows": ROWS, "param_seeds": SEEDS, "data_seeds": [33, 34]}

    # ---- an empty DBN: three ValueErrors ------------------------------------------------------------
    empty = ref.DBN()
    facts["empty_fit"] = outcome(lambda: empty.fit(V))
    facts["empty_transform"] = outcome(lambda: empty.transform(V))
    facts["empty_inv_transform"] = outcome(lambda: empty.inv_transform(H))

    # ---- add_stack: first call works, second dies on self.rbm_layer --------------------------------
    log = []
    stubs = [StubLayer("rbm_%d" % (i + 1), core, log) for i, core in enumerate(oracle_layers())]
    dbn = ref.DBN()
    facts["add_stack_first"] = outcome(lambda: dbn.add_stack(stubs[0]))[0]
    facts["layers_after_first_add"] = len(dbn._rbm_layers)
    facts["add_stack_second"] = outcome(lambda: dbn.add_stack(stubs[1]))
    dbn._rbm_layers.append(stubs[1])           # what dbn.py:26 would have done

    # ---- transform: the chain, the call order, the input left alone -------------------------------
    V_in = V.copy()
    del log[:]
    status, top = outcome(lambda: dbn.transform(V_in))
    assert status == "ok", (status, top)
    facts["transform_calls"] = list(log)
    facts["transform_leaves_input"] = bool(np.array_equal(V_in, V))

    # ---- inv_transform as written: identity, no layer called ----------------------------------------
    del log[:]
    status, back = outcome(lambda: dbn.inv_transform(H.copy()))
    assert status == "ok", (status, back)
    facts["inv_transform_calls_as_written"] = list(log)
    facts["inv_transform_as_written_is_identity"] = bool(np.array_equal(back, H))

    # ---- fit: the print, then self.rbm_layer ----------------------------------------------------------
    del log[:]
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        facts["fit"] = outcome(lambda: dbn.fit(V))
    facts["fit_stdout"] = out.getvalue()
    facts["fit_calls"] = list(log)

    # the repaired reverse walk on FRESH oracle layers whose counters stand where the reference run left them after its
    # one transform call: what this build's DBN.inv_transform must return for the same H
    fresh = oracle_layers()
    O.dbn_transform(fresh, V)
    repaired = O.dbn_inv_transform(fresh, H)

    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "ref_dbn.npz")
    np.savez_compressed(path, facts=np.array(json.dumps(facts, sort_keys=True)), transform_out=top.astype(np.float32),
                        inv_transform_as_written=back.astype(np.float32), inv_transform_repaired=repaired.astype(np.float32))
    print("%s  %.1f KB" % (path, os.path.getsize(path) / 1024.0))
    print(json.dumps(facts, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
