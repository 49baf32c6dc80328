"""Worker of tests/test_two_gpus.py: one rank of a world-size-2 RBM.fit on two real GPUs (torch.distributed.run starts two of
these).  argv: case name, output stem.  Every rank writes <stem>.rank<r>.npz with its parameters and the assembled chain."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import synthetic_binary, synthetic_params, synthetic_real  # noqa: E402

CASES = {
    # N, batch, k, persistent, gaussian, compute
    "cd1": dict(N=1200, bs=512, k=1, persistent=False, gauss=False, compute="x3"),
    "pcd2": dict(N=1100, bs=512, k=2, persistent=True, gauss=False, compute="x3"),
    "gauss": dict(N=1024, bs=512, k=1, persistent=False, gauss=True, compute="x3"),
    "idle_rank": dict(N=515, bs=512, k=1, persistent=False, gauss=False, compute="x3"),     # 3-row remainder: rank 1 has no rows
    "bf16": dict(N=1024, bs=512, k=2, persistent=True, gauss=False, compute="bf16"),
    "fp32": dict(N=300, bs=128, k=1, persistent=False, gauss=False, compute="fp32"),
}
NV, NH, LR, SEED, EPOCHS = 784, 256, 1e-3, 5, 2


def data(case):
    c = CASES[case]
    W0 = synthetic_params(NV, NH, 3)
    V = synthetic_real(c["N"], NV, 4) if c["gauss"] else synthetic_binary(c["N"], NV, 4, p=0.3)
    return c, W0, V


def fit(case, device):
    from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN, RBM
    c, W0, V = data(case)
    r = RBM({"batch_size": c["bs"], "epochs": EPOCHS, "lr": LR}, NH, mode=MODE_VISIBLE_GAUSSIAN if c["gauss"] else MODE_VISIBLE_BERNOULLI,
            seed=SEED, weights=W0, cd_k=c["k"], persistent=c["persistent"], compute_dtype=c["compute"], device=device)
    assert r.fit(V, verbose=0) is None
    chain = r.full_chain()                      # (collective under data parallelism)
    W, b_h, b_v = r.get_weights()
    return dict(W=W, b_h=b_h, b_v=b_v, chain=chain if chain is not None else np.zeros(0, np.float32))


if __name__ == "__main__":
    case, stem = sys.argv[1], sys.argv[2]
    rank, local = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=int(os.environ["WORLD_SIZE"]))
    try:
        out = fit(case, "cuda:%d" % local)
        np.savez(stem + ".rank%d.npz" % rank, **out)
    finally:
        from keras_unsupervised_amd.ebm import dp
        dp.destroy_comms()
        dist.destroy_process_group()
