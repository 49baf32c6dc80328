#!/usr/bin/env python3
"""Phase lengths of kurbm_score_small (diagnostic build): make -C keras_unsupervised_amd/csrc variant VAR=-DKURBM_SMALL_STAMPS
OUT=libkurbm_smallstamps.so; KURBM_LIB=.../libkurbm_smallstamps.so python tools/small_score_stamps.py [B NH]
Stamps 0-5: workgroup 0 (phase 1, barrier, phase 2, barrier, phase 3); 6, 7: the LAST workgroup (found itself last, score stored)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NH = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 128)
dev = torch.device("cuda", 0)
g = np.random.default_rng(0)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (784, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(784, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B * 8, 784)) < 0.19).astype(np.float32), dev)
lib = eng.lib
lib.kurbm_debug_small_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
acc = np.zeros(7)
for it in range(40):
    eng.score_small(V, B, (it % 8) * B, 1, it, 0, 3)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.kurbm_debug_small_stamps(eng.ctx.handle, out)
    t = np.array(list(out), dtype=np.float64)
    if it >= 8:
        acc += np.diff(t) / 100.0   # us
acc /= 32
print("score, 784 x %d, batch %d (us): " % (NH, B) + "  ".join("%s %.1f" % kv for kv in zip(
    ["phase 1", "barrier", "phase 2", "barrier", "phase 3", "-> last workgroup knows", "its sums + store"], acc)) + "  | sum %.1f" % acc.sum())
