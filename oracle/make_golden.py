"""Generate the golden fixtures under tests/golden/ from the oracle.

    python -m oracle.make_golden          (run from the repo root)

TEST INFRASTRUCTURE.  The reference's TF path cannot run here and has no fixtures of its own
(SURVEY.md 8(c)), so these vectors pin the ORACLE (and, through it, the HIP kernels) against
regressions; they are not outputs of the reference.  Inputs are never stored: they are rebuilt
from Philox streams by `synthetic_*` below, so each .npz holds outputs only and stays small.
"""
import os

import numpy as np

from . import philox
from . import rbm_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- deterministic synthetic inputs (pure functions of their arguments) ----------------
def synthetic_params(n_vis, n_hid, seed):
    """W, b_h, b_v ~ U(-0.05, 0.05) (the Keras 'uniform' range, rbm.py:32-38) from Philox."""
    W = (philox.uniform(n_vis, n_hid, seed, 0x7001, 0) - np.float32(0.5)) * np.float32(0.1)
    b_h = (philox.uniform(1, n_hid, seed, 0x7002, 0)[0] - np.float32(0.5)) * np.float32(0.1)
    b_v = (philox.uniform(1, n_vis, seed, 0x7003, 0)[0] - np.float32(0.5)) * np.float32(0.1)
    return W.astype(np.float32), b_h.astype(np.float32), b_v.astype(np.float32)


def synthetic_binary(rows, cols, seed, p=0.19):
    """MNIST-like binary data: pixel on with probability p."""
    return (philox.uniform(rows, cols, seed, 0x7004, 0) < np.float32(p)).astype(np.float32)


def synthetic_real(rows, cols, seed):
    """Real-valued data in [0, 1): grey levels / hidden probabilities."""
    return philox.uniform(rows, cols, seed, 0x7005, 0).astype(np.float32)


def save(name, **arrays):
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, name)
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def main():
    # (i) Philox known answers + the first uniforms of streams 0..3
    kat_in = np.array([[0, 0, 0, 0, 0, 0],
                       [0xFFFFFFFF] * 6,
                       [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0]], dtype=np.uint64)
    kat_out = np.stack([np.array([w[0] for w in philox.philox4x32_10(
        tuple(np.array([c]) for c in row[:4]), (int(row[4]), int(row[5])))], dtype=np.uint32) for row in kat_in])
    save("philox.npz", kat_in=kat_in, kat_out=kat_out,
         uniforms=np.stack([philox.uniform(8, 8, 0, s, 0) for s in range(4)]),
         uniforms_seed42_row0_8=philox.uniform(8, 8, 42, 3, 7, row0=8),
         words_seed42=philox.block_words(8, 8, 42, 3, 7))

    # (ii) half steps at the config-1 shape (784 x 256), B = 8 and B = 64
    for B in (8, 64):
        W, b_h, b_v = synthetic_params(784, 256, seed=11)
        v = synthetic_binary(B, 784, seed=12)
        rng = O.Rng(seed=42, step=5)
        p_h, u_h, h = O.sample_hidden(v, W, b_h, rng, stream_id=0)
        p_v, u_v, v1 = O.sample_visible(h, W, b_v, rng, stream_id=1)
        sub = 1 if B == 8 else 8   # B = 64: every 8th row of the float planes keeps the file small
        save("half_step_B%d.npz" % B, p_h=p_h[::sub], u_h=u_h[::sub], h=np.packbits(h.astype(np.uint8), axis=1),
             p_v=p_v[::sub], u_v=u_v[::sub], v1=np.packbits(v1.astype(np.uint8), axis=1), row_stride=np.int64(sub))

    # (iii) one fused CD-1 step and one reference_sequential step, small shape stored in full
    nv, nh, B = 64, 48, 24
    W, b_h, b_v = synthetic_params(nv, nh, seed=21)
    v = synthetic_binary(B, nv, seed=22, p=0.3)
    Wf, bhf, bvf, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, lr=0.01, seed=7, step=3)
    Ws, bhs, bvs = O.cd_step_reference_sequential(W, b_h, b_v, v, lr=0.01, seed=7, step=3)
    save("cd_step_small.npz", W_fused=Wf, bh_fused=bhf, bv_fused=bvf, dW=dW, dbh=dbh, dbv=dbv,
         h_pos=ch["h_pos"].astype(np.uint8), v_neg=ch["v_neg"].astype(np.uint8), h_neg=ch["h_neg"],
         W_seq=Ws, bh_seq=bhs, bv_seq=bvs)
    # ... and at the config-1 shape (784 x 256, B = 64) as a strided sub-sample + checksums
    W, b_h, b_v = synthetic_params(784, 256, seed=11)
    v = synthetic_binary(64, 784, seed=12)
    Wf, bhf, bvf, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, lr=1e-3, seed=42, step=0)
    save("cd_step_config1.npz", dW_sub=dW[::49, ::16], dW_sum=np.float64(dW.astype(np.float64).sum()),
         dW_abs_sum=np.float64(np.abs(dW.astype(np.float64)).sum()), dbh=dbh, dbv=dbv,
         W_sub=Wf[::49, ::16], h_pos_rowsum=ch["h_pos"].sum(1), v_neg_rowsum=ch["v_neg"].sum(1))

    # (iv) 3-step fit with a remainder batch (N = 150, bs = 64), both update modes, with scores
    W, b_h, b_v = synthetic_params(nv, nh, seed=31)
    V = synthetic_binary(150, nv, seed=32, p=0.3)
    hps = {"batch_size": 64, "epochs": 1, "lr": 0.01}
    out = {}
    for mode_name in ("fused", "reference_sequential"):
        Wt, bht, bvt, scores, step = O.fit(W, b_h, b_v, V, hps, seed=5, update_mode=mode_name, with_score=True)
        out.update({"W_" + mode_name: Wt, "bh_" + mode_name: bht, "bv_" + mode_name: bvt,
                    "scores_" + mode_name: np.array(scores, dtype=np.float64)})
    save("fit_trajectory.npz", **out)

    # (v) 2-layer DBN, greedy, tiny
    Vd = synthetic_binary(40, 32, seed=41, p=0.4)
    hps = {"batch_size": 16, "epochs": 2, "lr": 0.02}
    layers = [O.OracleLayer(*synthetic_params(32, 24, seed=42), hps, seed=1),
              O.OracleLayer(*synthetic_params(24, 16, seed=43), hps, seed=2)]
    top = O.dbn_fit(layers, Vd)
    feat = O.dbn_transform(layers, Vd)
    back = O.dbn_inv_transform(layers, feat)
    save("dbn_small.npz", W0=layers[0].W, bh0=layers[0].b_h, bv0=layers[0].b_v, W1=layers[1].W,
         bh1=layers[1].b_h, bv1=layers[1].b_v, top=top.astype(np.uint8), feat=feat.astype(np.uint8),
         back=back.astype(np.uint8))

    # (vi) free energy, incl. a large-activation row (naive softplus overflows, stable one does not)
    W, b_h, b_v = synthetic_params(nv, nh, seed=51)
    v = synthetic_real(6, nv, seed=52)
    v[5] *= 4000.0
    W2 = W.copy()
    W2[:, 0] = 0.05  # drives (v.W)_0 past 88 for the scaled row
    save("free_energy.npz", F_stable=O.free_energy(v, W2, b_h, b_v, stable=True),
         F_naive=O.free_energy(v, W2, b_h, b_v, stable=False))

    # (vii) Gaussian-visible mode: one fused CD-1 step (normals by Box-Muller)
    W, b_h, b_v = synthetic_params(nv, nh, seed=61)
    v = synthetic_real(B, nv, seed=62)
    Wg, bhg, bvg, ch, (dW, dbh, dbv) = O.cd_step_fused(W, b_h, b_v, v, lr=0.01, seed=9, step=1,
                                                       mode=O.MODE_VISIBLE_GAUSSIAN)
    save("cd_step_gaussian.npz", W=Wg, bh=bhg, bv=bvg, v_neg=ch["v_neg"], h_pos=ch["h_pos"].astype(np.uint8))


if __name__ == "__main__":
    main()
