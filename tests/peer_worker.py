"""Worker of tests/test_peer_exchange.py: one rank of a world-size-N run of the PEER exchange (kurbm_peer_*), all ranks on GPU 0 --
plain hipIpc between processes that share a device, which RCCL refuses.  argv: output stem.  torch.distributed (gloo) carries the
IPC handles; every rank writes <stem>.rank<r>.npz."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import synthetic_binary, synthetic_params, synthetic_real  # noqa: E402

NV, NH, B, LR, SEED = 784, 256, 640, 1e-3, 5
CASES = {"bern": dict(gauss=False), "gauss": dict(gauss=True)}


def main():
    stem = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                      # EVERY rank on device 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, RBM, dp
    from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM
    dev = torch.device("cuda", 0)
    out = {}
    try:
        x = dp.get_exchange(dev, NV, NH)
        assert isinstance(x, dp.PeerExchange) and x.count() == world
        # ---- the plain all-reduce: ragged sizes, more than one buffer's worth
        for n in (5, 1024, 4099, NV * NH + NH + NV, 2 * (NV * NH) + 7):
            g = np.random.default_rng(100 * n + rank)
            t = torch.from_numpy(g.standard_normal(n).astype(np.float32)).to(dev)
            out["ar_in_%d" % n] = t.cpu().numpy().copy()
            x.allreduce_sum_(t)
            torch.cuda.synchronize()
            out["ar_out_%d" % n] = t.cpu().numpy().copy()
        # ---- one data-parallel step, every piece visible
        for name, c in CASES.items():
            mode = MODE_VISIBLE_GAUSSIAN if c["gauss"] else MODE_VISIBLE_BERNOULLI
            W0 = synthetic_params(NV, NH, 3)
            V = synthetic_real(B, NV, 4) if c["gauss"] else synthetic_binary(B, NV, 4, p=0.3)
            lo, hi = dp.shard_rows(B, world, rank)
            Vd = DeviceMatrix.from_host(V, dev)
            e1 = DeviceRBM(*W0, dev)
            e1.cd_step(Vd, hi - lo, lo, LR, SEED, 7, mode=mode, apply=False, emit_delta=True, row0=lo, compute="x3")
            D = e1.delta_buffer().clone()
            out[name + "_D"] = D.cpu().numpy().copy()
            S = D.clone()
            x.allreduce_sum_(S)
            out[name + "_S"] = S.cpu().numpy().copy()
            e2 = DeviceRBM(*W0, dev)
            e2.cd_step_dp(x, Vd, hi - lo, lo, LR, SEED, 7, mode=mode, row0=lo, compute="x3")
            torch.cuda.synchronize()
            for k, a in zip(("W", "bh", "bv"), e2.get_weights()):
                out[name + "_peer_" + k] = a
            out[name + "_peer_mirror"] = e2._mirrors[3][0].cpu().numpy().copy()
            e3 = DeviceRBM(*W0, dev)
            e3.mirror(3)
            e3.apply_delta(LR, delta=S, compute="x3")
            for k, a in zip(("W", "bh", "bv"), e3.get_weights()):
                out[name + "_ref_" + k] = a
            out[name + "_ref_mirror"] = e3._mirrors[3][0].cpu().numpy().copy()
        # ---- RBM.fit through the exchange (remainder batch: the last rank has no rows of it), persistent chain assembled
        N = 2 * B + 3
        Vfit = synthetic_binary(N, NV, 9, p=0.3)
        r = RBM({"batch_size": B, "epochs": 2, "lr": LR}, NH, mode=MODE_VISIBLE_BERNOULLI, seed=SEED, weights=synthetic_params(NV, NH, 3),
                cd_k=2, persistent=True, compute_dtype="x3", device="cuda:0")
        assert r.fit(Vfit, verbose=0) is None
        chain = r.full_chain()
        for k, a in zip(("W", "bh", "bv"), r.get_weights()):
            out["fit_" + k] = a
        out["fit_chain"] = chain
        np.savez(stem + ".rank%d.npz" % rank, **out)
    finally:
        dp.destroy_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
