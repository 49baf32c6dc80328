#!/usr/bin/env python3
"""Phase lengths of kurbm_cd_step_small (diagnostic build): make -C keras_unsupervised_amd/csrc variant VAR=-DKURBM_SMALL_STAMPS
OUT=libkurbm_smallstamps.so; KURBM_LIB=.../libkurbm_smallstamps.so python tools/small_stamps.py [B NH]"""
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
fine = np.zeros(4)
has_fine = hasattr(lib, "kurbm_debug_small_fine_stamps")
for it in range(40):
    eng.cd_step(V, B, (it % 8) * B, 1e-3 / B, 1, it, compute="small")
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.kurbm_debug_small_stamps(eng.ctx.handle, out)
    t = np.array(list(out), dtype=np.float64)
    if it >= 8:
        acc += np.diff(t) / 100.0   # us
        if has_fine:
            f8 = (C.c_ulonglong * 8)()
            lib.kurbm_debug_small_fine_stamps(eng.ctx.handle, f8)
            fine += np.diff(np.array(list(f8)[:5], dtype=np.float64)) / 100.0
acc /= 32
fine /= 32
print("784 x %d, batch %d (us, workgroup 0): " % (NH, B) + "  ".join("%s %.1f" % kv for kv in zip(
    ["phase 1", "barrier", "phase 2", "barrier", "phase 3", "barrier", "phase 4"], acc)) + "  | sum %.1f" % acc.sum())
if has_fine:
    print("   phase 2, workgroup 0, first pass (us): loads + MFMAs %.2f  wait for the other waves %.2f  epilogue %.2f  end-of-pass barrier %.2f" % tuple(fine))
