#!/usr/bin/env python3
"""Where does a k-loop iteration spend its cycles?  (diagnostic build only)

    make -C keras_unsupervised_amd/csrc libkurbm_stamps.so
    KURBM_LIB=keras_unsupervised_amd/csrc/libkurbm_stamps.so python tools/stamp_profile.py

Runs the four kernels of the config-2 CD-1 step from the -DKURBM_STAMPS library and prints, per
kernel, the median over waves of: prologue, k loop, epilogue, and inside the loop the share of
{fetch issue, fragment reads + MFMA, park to LDS, barrier}.  Read the SHARES, not the lengths:
the stamps' fences forbid overlaps the shipped kernels have.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "stamps" in os.environ.get("KURBM_LIB", ""), "set KURBM_LIB to libkurbm_stamps.so"

from keras_unsupervised_amd import _lib  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
eng = DeviceRBM(g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32), np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)
V = DeviceMatrix.from_host((g.random((B, NV)) < 0.19).astype(np.float32), dev)
lib = eng.lib
lib.kurbm_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.kurbm_debug_set_stamp_buffer.restype = None
lib.kurbm_debug_set_off.argtypes = [C.c_int]
lib.kurbm_debug_set_off.restype = None
OFF = int(os.environ.get("STAMP_OFF", "0"))   # ablation mask: 1 fetches, 2 parks, 4 fragment reads, 8 barrier
buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)


def run(name, fn, nblocks, ntiles):
    for _ in range(3):
        fn()
    buf.zero_()
    lib.kurbm_debug_set_stamp_buffer(buf.data_ptr())
    lib.kurbm_debug_set_off(OFF)
    fn()
    torch.cuda.synchronize()
    lib.kurbm_debug_set_stamp_buffer(None)
    lib.kurbm_debug_set_off(0)
    s = buf.cpu().numpy().reshape(-1, 16)[: nblocks * 4].astype(np.float64)
    s = s[s[:, 1] > 0]
    med = np.median(s, axis=0)
    tot = med[0] + med[1] + med[2]
    grp = med[4:12]
    print("off=%2d %-20s waves=%4d total %7.0f cyc | prologue %4.1f%% loop %4.1f%% epilogue %4.1f%% | loop cycles per tile by MFMA group: %s (sum %.0f)"
          % (OFF, name, len(s), tot, 100 * med[0] / tot, 100 * med[1] / tot, 100 * med[2] / tot,
             " ".join("%4.0f" % (x / ntiles) for x in grp), grp.sum() / ntiles))


h_pos = eng.half_step("vh", V, B, 0, 0, 1, 42, 0, 0)["sample"]
v_neg = eng.half_step("hv", h_pos, B, 0, 0, 1, 42, 1, 0)["sample"]
h_neg = eng.half_step("vh", v_neg, B, 0, 0, 0, 42, 0, 0, want_sample=False, want_prob=True)["prob"]
run("half_step_vh_sample", lambda: eng.half_step("vh", V, B, 0, 0, 1, 42, 0, 0), 256, 24)
run("half_step_hv_sample", lambda: eng.half_step("hv", h_pos, B, 0, 0, 1, 42, 1, 0), 224, 32)
run("half_step_vh_prob", lambda: eng.half_step("vh", v_neg, B, 0, 0, 0, 42, 0, 0, want_sample=False, want_prob=True), 256, 24)
run("outer_stats", lambda: eng.outer_delta(V, h_pos, v_neg, h_neg, B), 224, 64)
