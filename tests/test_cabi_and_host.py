"""CPU tier: the C-ABI library loads and exports what include/kurbm.h declares; host logic that
needs no device (constructor surface, DBN stacking errors, sharding arithmetic, loud failure)."""
import os
import re

import numpy as np
import pytest
import torch

from keras_unsupervised_amd import _lib
from keras_unsupervised_amd.ebm import DBN, MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, RBM, dp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "kurbm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kurbm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), "libkurbm.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.kurbm_abi_version() == _lib.ABI_VERSION == 5


def test_header_is_plain_c_and_structs_match(tmp_path):
    """include/kurbm.h compiles as C99 with gcc (no C++, no HIP, no torch types in the boundary), and the struct layouts
    the ctypes table assumes are the ones the C compiler produces."""
    import ctypes as C
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "kurbm.h"\n'
        'int main(void) {\n'
        '  printf("%zu %zu %zu %zu %zu %zu %d %d %d\\n", sizeof(kurbm_params), offsetof(kurbm_params, W), sizeof(kurbm_rng),\n'
        '         sizeof(kurbm_cd_opts), offsetof(kurbm_cd_opts, v_planes), offsetof(kurbm_cd_opts, seed),\n'
        '         KURBM_ABI_VERSION, KURBM_UNIQUE_ID_BYTES, KURBM_V_BINARY);\n'
        '  return 0; }\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == [C.sizeof(_lib.Params), _lib.Params.W.offset, C.sizeof(_lib.Rng), C.sizeof(_lib.CdOpts),
                                     _lib.CdOpts.v_planes.offset, _lib.CdOpts.seed.offset, _lib.ABI_VERSION, _lib.UNIQUE_ID_BYTES,
                                     _lib.V_BINARY]


def test_struct_layouts_match_header():
    import ctypes as C
    assert C.sizeof(_lib.Params) == 40 and _lib.Params.W.offset == 16
    assert C.sizeof(_lib.Rng) == 24 and _lib.Rng.stream_id.offset == 16
    assert C.sizeof(_lib.CdOpts) == 72 and _lib.CdOpts.delta_out.offset == 16 and _lib.CdOpts.seed.offset == 32
    assert _lib.CdOpts.v_planes.offset == 56 and _lib.CdOpts.v_planes_stride.offset == 64


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-device failure mode")
def test_no_device_fails_loudly():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.kurbm_ctx_create(0, C.byref(h)) < 0          # error code, not a crash
    assert lib.kurbm_last_error()
    rbm = RBM({"batch_size": 4, "epochs": 1, "lr": 0.1}, 8, mode=MODE_VISIBLE_BERNOULLI)
    with pytest.raises(_lib.KurbmError, match="no CPU fallback"):
        rbm.build((None, 16))
    with pytest.raises(_lib.KurbmError):
        rbm.fit(np.zeros((8, 16), np.float32))


def test_rbm_constructor_surface():
    hps = {"batch_size": 128, "epochs": 1, "lr": 0.001}
    rbm = RBM(hps, 128, name="rbm_1")
    assert rbm.mode == MODE_VISIBLE_GAUSSIAN                 # the reference's default (rbm.py:22)
    assert rbm.hps is hps and rbm.output_dim == 128 and rbm.name == "rbm_1"
    assert rbm.compute_output_shape((32, 784)) == (32, 128)  # rbm.py:94-95
    cfg = rbm.get_config()
    assert cfg["hps"] == hps and cfg["output_dim"] == 128 and cfg["name"] == "rbm_1"
    clone = RBM(**cfg)
    assert clone.mode == rbm.mode and clone.update_mode == "fused" and clone.cd_k == 1
    with pytest.raises(ValueError):
        RBM(hps, 8, mode=2)                                  # MODE_COMPLEX is a TODO in the reference
    with pytest.raises(ValueError):
        RBM(hps, 8, update_mode="bogus")
    assert RBM(dict(hps, cd_k=10, persistent=True), 8).cd_k == 10


def test_dbn_stack_errors():
    with pytest.raises(ValueError, match="rbm layer"):
        DBN().fit(np.zeros((4, 4), np.float32))              # dbn.py:47-48
    with pytest.raises(ValueError):
        DBN().transform(np.zeros((4, 4), np.float32))        # dbn.py:68-69
    with pytest.raises(ValueError):
        DBN().inv_transform(np.zeros((4, 4), np.float32))    # dbn.py:88-89
    hps = {"batch_size": 4, "epochs": 1, "lr": 0.1}
    a, b = RBM(hps, 8), RBM(hps, 6)
    a.input_shape, a.output_shape = (None, 16), (None, 8)
    b.input_shape, b.output_shape = (None, 9), (None, 6)     # 9 != 8
    d = DBN()
    d.add_stack(a)
    with pytest.raises(ValueError, match="output dimension"):
        d.add_stack(b)                                       # dbn.py:29-30
    b.input_shape = (None, 8)
    d.add_stack(b)
    assert len(d._rbm_layers) == 2


def test_shard_rows_properties():
    for n in (1, 3, 4, 22, 64, 150, 4096, 4097, 32768):
        for world in (1, 2, 3, 4, 8):
            spans = [dp.shard_rows(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (lo, hi), (lo2, _) in zip(spans, spans[1:]):
                assert hi == lo2                              # contiguous, no overlap
            assert all(lo % 4 == 0 for lo, hi in spans if hi > lo)   # never splits a Philox block
    assert dp.shard_rows(32768, 8, 3) == (12288, 16384)


def test_pack_unpack_roundtrip():
    dW, dbh, dbv = torch.randn(5, 3), torch.randn(3), torch.randn(5)
    p = dp.pack(dW, dbh, dbv)
    assert p.numel() == dp.packed_size(5, 3)
    a, b, c = dp.unpack(p, 5, 3)
    assert torch.equal(a, dW) and torch.equal(b, dbh) and torch.equal(c, dbv)
    assert dp.world() == (0, 1)


def test_shard_rows_fixed_ownership_for_persistent_chains():
    """of=batch_size: row j of the fantasy particles stays on one rank in full and remainder batches alike."""
    for bs, world in ((40, 2), (4096, 8), (100, 3)):
        full = [dp.shard_rows(bs, world, r, of=bs) for r in range(world)]
        assert full == [dp.shard_rows(bs, world, r) for r in range(world)]
        for n in (1, bs // 3, bs // 2 + 1, bs - 1):
            part = [dp.shard_rows(n, world, r, of=bs) for r in range(world)]
            for (lo, hi), (flo, fhi) in zip(part, full):
                assert lo == min(flo, n) and hi == min(fhi, n)
            assert part[0][0] == 0 and max(hi for _, hi in part) == n


def test_graft_entry_build_check_matches_the_abi():
    """__graft_entry__.build() must not pin an ABI number of its own (it once asserted the previous one)."""
    import inspect
    import __graft_entry__ as g
    src = inspect.getsource(g.build)
    assert "_lib.ABI_VERSION" in src


def test_score_ring_prints_in_order_one_step_late():
    """fit(verbose=1) hands a step's score line to stdout when the NEXT step has been queued (never out of order, at most one
    step late, everything flushed at an epoch's end): the host-side ring on CPU tensors (no events, no pinned memory)."""
    from keras_unsupervised_amd.ebm.rbm import _ScoreRing
    seen = []
    ring = _ScoreRing(torch.device("cpu"), lambda score, label: seen.append((label, score)), depth=4)
    for i in range(10):
        ring.push(torch.tensor([float(i) * 0.5, -1.0, -1.0, -1.0]), (i + 1, 10))
        assert [l for l, _ in seen] == [(j + 1, 10) for j in range(i)]          # step i's line waits for step i + 1
    ring.flush()
    assert seen == [((j + 1, 10), j * 0.5) for j in range(10)]
    ring.flush()                                                                  # idempotent
    assert len(seen) == 10
