"""The oracle's DBN control flow and the host DBN class against tests/golden/ref_dbn.npz -- the one fixture produced by
RUNNING reference code (oracle/make_ref_fixtures.py: /root/reference/ku/ebm/dbn.py loaded by path in the build container,
stub layers backed by the oracle).  It pins dbn.py's control flow only; the arithmetic of rbm.py stays unpinned."""
import json
import os

import numpy as np
import pytest

from keras_unsupervised_amd.ebm.dbn import DBN
from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary
from oracle.make_ref_fixtures import DIMS, ROWS, oracle_layers


@pytest.fixture(scope="module")
def ref(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_dbn.npz"))
    return json.loads(str(z["facts"])), z


def test_oracle_dbn_transform_equals_reference_run(ref):
    facts, z = ref
    assert tuple(facts["dims"]) == DIMS and facts["rows"] == ROWS
    V = synthetic_binary(ROWS, DIMS[0], seed=facts["data_seeds"][0])
    V_in = V.copy()
    layers = oracle_layers()
    top = O.dbn_transform(layers, V_in)
    assert np.array_equal(top, z["transform_out"])            # bit-exact: same layers, same order, one call each
    assert [l.calls for l in layers] == [1, 1]                # dbn.py:72-73: every layer's transform exactly once
    assert facts["transform_calls"] == ["rbm_1.transform", "rbm_2.transform"]
    assert np.array_equal(V_in, V) and facts["transform_leaves_input"]      # dbn.py:65: V.copy()


def test_inv_transform_repair_is_documented_against_the_reference_run(ref):
    facts, z = ref
    H = synthetic_binary(ROWS, DIMS[2], seed=facts["data_seeds"][1])
    # as written (dbn.py:92: range(len(layers), -1) is empty) the reference returns H unchanged and calls no layer
    assert facts["inv_transform_as_written_is_identity"] and facts["inv_transform_calls_as_written"] == []
    assert np.array_equal(z["inv_transform_as_written"], H)
    # the repair of this build: the reverse walk -- same counters as the fixture's generator
    layers = oracle_layers()
    O.dbn_transform(layers, synthetic_binary(ROWS, DIMS[0], seed=facts["data_seeds"][0]))
    back = O.dbn_inv_transform(layers, H)
    assert np.array_equal(back, z["inv_transform_repaired"])
    assert back.shape == (ROWS, DIMS[0]) and not np.array_equal(back.shape, H.shape)


def test_reference_defects_are_the_ones_the_survey_lists(ref):
    facts, _ = ref
    assert facts["add_stack_first"] == "ok" and facts["layers_after_first_add"] == 1         # dbn.py:31-32
    assert facts["add_stack_second"] == ["AttributeError", "'DBN' object has no attribute 'rbm_layer'"]   # dbn.py:25
    assert facts["fit"] == ["AttributeError", "'DBN' object has no attribute 'rbm_layer'"]   # dbn.py:54
    assert facts["fit_stdout"] == "Train rbm_1.\n" and facts["fit_calls"] == []              # dbn.py:53 ran, :54 did not


def test_host_dbn_raises_what_the_reference_raises(ref, capsys):
    """The host class (no GPU needed for these paths): same exception type and text for an empty stack (dbn.py:47-48,
    :68-69, :88-89), and the 'Train <name>.' line of dbn.py:53."""
    facts, _ = ref
    x = np.zeros((4, 4), np.float32)
    for key, call in (("empty_fit", lambda: DBN().fit(x)), ("empty_transform", lambda: DBN().transform(x)),
                      ("empty_inv_transform", lambda: DBN().inv_transform(x))):
        kind, msg = facts[key]
        assert kind == "ValueError"
        with pytest.raises(ValueError) as e:
            call()
        assert str(e.value) == msg

    class Stub:
        """Host-side stand-in for an RBM: what DBN touches, backed by the oracle (no device)."""
        built = True

        def __init__(self, name, core):
            self.name, self.core = name, core
            self.input_shape, self.output_shape = (None, core.W.shape[0]), (None, core.W.shape[1])
            self.output_dim = core.W.shape[1]
            self.fits = 0

        def _ensure_built(self, n):      # (a built RBM ignores the call, as RBM._ensure_built does)
            pass

        def _as_device(self, X):
            return np.asarray(X, np.float32), "numpy"

        def _ret(self, m, kind, as_list):
            return m

        def fit(self, V, verbose=1):
            self.fits += 1

        def transform(self, V):
            return self.core.transform(V)

        def inv_transform(self, H):
            return self.core.inv_transform(H)

    d = DBN()
    stubs = [Stub("rbm_%d" % (i + 1), core) for i, core in enumerate(oracle_layers())]
    for s in stubs:
        d.add_stack(s)                                        # (the repaired dbn.py:24-26 accepts the second layer)
    V = synthetic_binary(ROWS, DIMS[0], seed=facts["data_seeds"][0])
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_dbn.npz"))
    assert np.array_equal(d.transform(V), z["transform_out"])
    H = synthetic_binary(ROWS, DIMS[2], seed=facts["data_seeds"][1])
    assert np.array_equal(d.inv_transform(H), z["inv_transform_repaired"])
    capsys.readouterr()
    d.fit(V, verbose=0)
    out = capsys.readouterr().out
    assert out.startswith(facts["fit_stdout"]) and out == "Train rbm_1.\nTrain rbm_2.\n"
    assert [s.fits for s in stubs] == [1, 1]


@pytest.mark.skipif(os.environ.get("KURBM_RUN_REFERENCE") != "1" or not os.path.exists("/root/reference/ku/ebm/dbn.py"),
                    reason="opt-in (KURBM_RUN_REFERENCE=1, build container only): EXECUTES the reference's dbn.py")
def test_ref_fixture_is_current(golden_dir, tmp_path, monkeypatch):
    """Where the reference is present (the build container, never the GPU box) AND the run opts in (KURBM_RUN_REFERENCE=1):
    running the reference's dbn.py again reproduces the fixture.  Not part of the default CPU tier: it executes code from the
    public reference tree (oracle/make_ref_fixtures.py checks the file's sha256 against the reviewed one first)."""
    from oracle import make_ref_fixtures
    monkeypatch.setattr(make_ref_fixtures, "GOLDEN_DIR", str(tmp_path))
    make_ref_fixtures.main()
    a, b = np.load(os.path.join(golden_dir, "ref_dbn.npz")), np.load(os.path.join(str(tmp_path), "ref_dbn.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
