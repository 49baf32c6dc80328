"""The peer exchange (include/kurbm.h: kurbm_peer_*; csrc/kurbm_peer.hip) -- a two-shot all-reduce over hipIpc-mapped buffers whose
second shot is the launch that applies the update -- run as TWO (and three) PROCESSES ON ONE GPU: the protocol (flags, bands, bounded
waits, buffer reuse over many epochs) is the one N GPUs run, only the wire is missing.  What is held: the sums are the ranks' local
sums added in rank order, bit for bit; the fused step leaves exactly what the library's own apply makes of that sum; replicas stay
bit-identical through RBM.fit (CD-2, persistent chain, a remainder batch that leaves the last rank idle); the trajectory agrees with
the single-process fit to the order of the fp32 additions.  Reference: rbm.py:125-134 (updates are batch SUMS)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world, tmp_path):
    stem = str(tmp_path / ("peer_w%d" % world))
    env = dict(os.environ, KURBM_DP_EXCHANGE="peer", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "peer_worker.py"), stem]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    return [np.load(stem + ".rank%d.npz" % k) for k in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_peer_exchange_processes_on_one_gpu(gpu_device, tmp_path, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import peer_worker as w
    ranks = _run(world, tmp_path)
    # the plain all-reduce: the inputs added in rank order, identical on every rank
    for key in [k for k in ranks[0].files if k.startswith("ar_in_")]:
        want = ranks[0][key].copy()
        for r in ranks[1:]:
            want = want + r[key]
        for r in ranks:
            got = r[key.replace("ar_in_", "ar_out_")]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
    for name in w.CASES:
        S = ranks[0][name + "_D"].copy()
        for r in ranks[1:]:
            S = S + r[name + "_D"]
        for r in ranks:
            assert np.array_equal(r[name + "_S"].view(np.uint32), S.view(np.uint32)), name      # rank order, bit for bit
            for k in ("W", "bh", "bv", "mirror"):
                assert np.array_equal(r[name + "_peer_" + k], ranks[0][name + "_peer_" + k]), (name, k)   # replicas identical
                assert np.array_equal(r[name + "_peer_" + k], r[name + "_ref_" + k]), (name, k)           # = the library's apply of S
        assert not np.array_equal(ranks[0][name + "_peer_W"], np.asarray(w.synthetic_params(w.NV, w.NH, 3)[0]))
    for k in ("W", "bh", "bv", "chain"):
        for r in ranks[1:]:
            assert np.array_equal(r["fit_" + k], ranks[0]["fit_" + k]), k
    # against the same fit in ONE process (same draws: global row indices; sums in another order, borderline samples may flip)
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM
    N = 2 * w.B + 3
    one = RBM({"batch_size": w.B, "epochs": 2, "lr": w.LR}, w.NH, mode=MODE_VISIBLE_BERNOULLI, seed=w.SEED,
              weights=w.synthetic_params(w.NV, w.NH, 3), cd_k=2, persistent=True, compute_dtype="x3", device=str(gpu_device))
    one.fit(w.synthetic_binary(N, w.NV, 9, p=0.3), verbose=0)
    for k, a in zip(("W", "bh", "bv"), one.get_weights()):
        d = np.abs(ranks[0]["fit_" + k] - a)
        assert np.mean(d > 1e-5) < 0.02 and float(d.max()) <= 6 * 4 * w.LR + 1e-5, (k, float(d.max()))
    assert np.mean(ranks[0]["fit_chain"] != one.full_chain()) < 1e-3
