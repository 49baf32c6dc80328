"""Seeded fuzz of the x3 CD-k step against the oracle (float64 statistics of the oracle's chain states) over 70 shapes:
special sizes around the 64 / 128 tile edges, random sizes, CD-1..3, real-valued data, persistent chains."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import rbm_oracle as O
from oracle.make_golden import synthetic_binary, synthetic_params, synthetic_real
from keras_unsupervised_amd.ebm.engine import DeviceRBM, DeviceMatrix
dev = torch.device("cuda", 0)
rs = np.random.RandomState(777)
def rel(a, r): return np.max(np.abs(a.astype(np.float64) - r.astype(np.float64)) / np.maximum(1.0, np.abs(r.astype(np.float64))))
bad = 0
special = [(128, 128, 128), (256, 64, 64), (129, 127, 129), (4, 128, 256), (384, 256, 128), (1, 1, 1), (2, 3, 1), (130, 64, 193), (512, 784, 256), (1024, 100, 300)]
for case in range(70):
    if case < len(special):
        B, nv, nh = special[case]
    else:
        B = int(rs.choice([1, 3, 7, 64, 65, 127, 128, 129, 255, 256, 300, 513]))
        nv = int(rs.choice([rs.randint(1, 600), 64, 128, 192, 256]))
        nh = int(rs.choice([rs.randint(1, 600), 64, 128, 192, 256]))
    k = int(rs.choice([1, 1, 2, 3]))
    real = bool(rs.rand() < 0.3)
    pcd = bool(rs.rand() < 0.3)
    W, b_h, b_v = synthetic_params(nv, nh, seed=3000 + case)
    v = synthetic_real(B, nv, seed=3100 + case) if real else synthetic_binary(B, nv, seed=3100 + case, p=0.4)
    chain0 = synthetic_binary(B, nv, seed=3200 + case, p=0.5) if pcd else None
    e = DeviceRBM(W, b_h, b_v, dev)
    vd = DeviceMatrix.from_host(v, dev)
    cd = DeviceMatrix.from_host(chain0, dev) if pcd else None
    e.cd_step(vd, B, 0, 0.01, case, 3, k=k, apply=False, emit_delta=True, v_chain=cd, compute="x3")
    torch.cuda.synchronize()
    d = e.delta_buffer().cpu().numpy().copy()
    _, _, _, ch, (dW_ref, dbh_ref, dbv_ref) = O.cd_step_fused(W, b_h, b_v, v, 0.01, case, 3, k=k, v_chain=chain0)
    dW, dbh, dbv = d[:nv*nh].reshape(nv, nh), d[nv*nh:nv*nh+nh], d[nv*nh+nh:]
    dbh64 = ch["h_pos"].astype(np.float64).sum(0) - ch["h_neg"].astype(np.float64).sum(0)
    dbv64 = v.astype(np.float64).sum(0) - ch["v_neg"].astype(np.float64).sum(0)
    dW64 = v.astype(np.float64).T @ ch["h_pos"].astype(np.float64) - ch["v_neg"].astype(np.float64).T @ ch["h_neg"].astype(np.float64)
    errs = (rel(dW, dW64), rel(dbh, dbh64), rel(dbv, dbv64), rel(dW_ref, dW64))
    ok = max(errs[:3]) <= 1e-4 and (not pcd or np.array_equal(cd.to_numpy(), ch["v_neg"]))
    if not ok:
        bad += 1
        print("FAIL", case, B, nv, nh, k, real, pcd, errs)
print("cases 70, failures", bad, "(a failure with errors ~0.1 .. 1 is one borderline draw flipping a 0/1 state: expected now and then)")
