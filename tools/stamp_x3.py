#!/usr/bin/env python3
"""Phase lengths of the x3 GEMM launches of a config-2 CD-1 step (diagnostic build only):

    make -C keras_unsupervised_amd/csrc libkurbm_stamps.so
    KURBM_LIB=keras_unsupervised_amd/csrc/libkurbm_stamps.so python tools/stamp_x3.py [binary|grey] [bern|gauss]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "stamps" in os.environ.get("KURBM_LIB", ""), "set KURBM_LIB to libkurbm_stamps.so"
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
kind = sys.argv[1] if len(sys.argv) > 1 else "binary"
mode = 1 if (len(sys.argv) > 2 and sys.argv[2] == "gauss") else 0
if kind == "binary":
    V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
else:
    V = DeviceMatrix.from_host((np.floor(g.random((B, NV)) * 256.0) / 255.0).astype(np.float32), dev)
lib = eng.lib
lib.kurbm_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.kurbm_debug_set_stamp_buffer.restype = None
buf = torch.zeros(4 * 65536, dtype=torch.int64, device=dev)
NAMES = ["start skew", "prologue", "k loop", "elementwise", "flush", "tail"]


def report(name, s):
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    d = [np.median(s[:, 0] - t0)]
    for q in range(1, 6):
        a, b = s[:, q - 1], s[:, q]
        ok = (a > 0) & (b > 0)
        d.append(np.median((b - a)[ok]) if ok.any() else 0.0)
    end = s[:, 5].max() - t0
    print("%-10s wgs %4d | " % (name, len(s)) + "  ".join("%s %6.0f" % (n, x) for n, x in zip(NAMES[1:], d[1:]))
          + " | tu[0..5]: " + " ".join("%6.0f" % x for x in np.median(s[:, 8:14], axis=0))
          + " | tiles fp8 / 1 piece / 3 pieces: " + " ".join("%6.0f" % np.median(s[:, q]) for q in (6, 7, 14)))


planes = eng.make_planes(V, [(0, B)], mode)
step = lambda: eng.cd_step(V, B, 0, 1e-3 / B, 42, 0, mode=mode, compute="x3", planes=planes)
what = sys.argv[3] if len(sys.argv) > 3 else "step"
if what == "score":   # the score pass of fit(verbose=1): three GEMM launches (kurbm_score_x3)
    step = lambda: eng.score_x3(V, B, 0, 42, 0, mode, 3, planes=planes)
print("data %s, mode %s, %s" % (kind, "gauss" if mode else "bern", what))
for _ in range(3):
    step()
torch.cuda.synchronize()
buf.zero_()
lib.kurbm_debug_set_stamp_buffer(buf.data_ptr())
step()
torch.cuda.synchronize()
lib.kurbm_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().astype(np.float64).reshape(4, 512, 8, 16)
for n, name in enumerate(("vh sample", "hv sample", "vh prob", "statistics") if what == "step" else ("vh + F(v)", "hv -> v'", "F(v')")):
    report(name, s[n, :, 0])
    lw = s[n, :, 7]
    lw = lw[lw[:, 0] == 1]
    if len(lw):   # the first loader wave's loop: cycles summed over its tiles
        print("           loader wave: issue %6.0f  wait for landing %6.0f  wait at the barrier %6.0f"
              % tuple(np.median(lw[:, q]) for q in (1, 2, 3)))
        w0 = s[n, :, 0]
        w0 = w0[w0[:, 0] > 0]
        t0 = w0[:, 0].min()
        if lw[:, 5].max() > 0:
            print("           since the launch's first wave: MFMA wave 0 starts %5.0f, first reads done %5.0f | loader wave starts %5.0f, "
                  "set up %5.0f, first requests issued %5.0f, tile 0 landed %5.0f"
                  % ((np.median(w0[:, 0]) - t0, np.median(w0[:, 1]) - t0) + tuple(np.median(lw[:, q]) - t0 for q in (4, 5, 6, 7))))
            d = lw[:, 4:11] - lw[:, 4:5]
            print("           loader wave, cycles since its own start: arguments warm %5.0f, block mapped %5.0f, role branch %5.0f, set up %5.0f, "
                  "first requests issued %5.0f, tile 0 landed %5.0f" % tuple(np.median(d[:, q]) for q in (4, 5, 6, 1, 2, 3)))
    if n == 0:
        for wv in (1, 4, 5):
            report("  wave %d" % wv, s[n, :, wv])

