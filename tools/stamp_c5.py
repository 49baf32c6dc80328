#!/usr/bin/env python3
"""Phase lengths of the rounded-bf16 path's launches in one persistent CD-10 step of config 5's one-GPU share (diagnostic build):

    make -C keras_unsupervised_amd/csrc libkurbm_stamps.so
    KURBM_LIB=keras_unsupervised_amd/csrc/libkurbm_stamps.so python tools/stamp_c5.py
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

ROWS, NV, NH, K = 1024, 4096, 4096, 10
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((torch.rand(ROWS, NV, device=dev) < 0.19).float(), dev)
chain = DeviceMatrix.from_host(V.view().clone(), dev)
lib = eng.lib
lib.kurbm_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.kurbm_debug_set_stamp_buffer.restype = None
NL = 24   # GEMM launches of a step: 21 sampling half steps + the probability half step + statistics (+ slack)
buf = torch.zeros(NL * 65536, dtype=torch.int64, device=dev)
step = lambda i: eng.cd_step(V, ROWS, 0, 1e-3 / ROWS, 42, i, k=K, v_chain=chain, compute="bf16")
for i in range(3):
    step(i)
torch.cuda.synchronize()
buf.zero_()
lib.kurbm_debug_set_stamp_buffer(buf.data_ptr())
step(3)
torch.cuda.synchronize()
lib.kurbm_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().astype(np.float64).reshape(NL, 512, 8, 16)
NAMES = ["prologue", "k loop", "elementwise", "flush", "tail"]
for n in range(NL):
    w0 = s[n, :, 0]
    w0 = w0[w0[:, 0] > 0]
    if not len(w0):
        continue
    d = []
    for q in range(1, 6):
        a, b = w0[:, q - 1], w0[:, q]
        ok = (a > 0) & (b > 0)
        d.append(np.median((b - a)[ok]) if ok.any() else 0.0)
    line = "launch %2d wgs %4d | " % (n, len(w0)) + "  ".join("%s %6.0f" % (nm, x) for nm, x in zip(NAMES, d))
    line += " | tiles: compute %6.0f  at the barrier %6.0f" % tuple(np.median(w0[:, q]) for q in (6, 7))
    lw = s[n, :, 7]
    lw = lw[lw[:, 0] == 1]
    if len(lw):
        line += " | loader: issue %6.0f  landing %6.0f  barrier %6.0f" % tuple(np.median(lw[:, q]) for q in (1, 2, 3))
        t0 = w0[:, 0].min()   # the first wave of the launch
        line += ("\n            since the launch's first wave: MFMA wave 0 starts %5.0f, first reads done %5.0f | loader wave starts %5.0f, "
                 "set up %5.0f, prologue requests issued %5.0f, tile 0 landed %5.0f"
                 % ((np.median(w0[:, 0]) - t0, np.median(w0[:, 1]) - t0) + tuple(np.median(lw[:, q]) - t0 for q in (4, 5, 6, 7))))
    print(line)
