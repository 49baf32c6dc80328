#!/usr/bin/env python3
"""Probe: the Gibbs chain of an x3 CD-1 step (conversion + three half steps) over 4096 rows on ONE stream against the
same rows as two 2048-row chains on TWO streams (independent rows: rbm.py:119-124) -- does the second chain fill the
prologue / epilogue bubbles of the first?  Two engines so each half has its own workspace."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM  # noqa: E402

NV, NH, B = 784, 1024, 4096
dev = torch.device("cuda", 0)
g = np.random.default_rng(1)
W = g.uniform(-0.05, 0.05, (NV, NH)).astype(np.float32)
mk = lambda: DeviceRBM(W, np.zeros(NH, np.float32), np.zeros(NV, np.float32), dev)  # noqa: E731
full, ea, eb = mk(), mk(), mk()
V = DeviceMatrix.from_host((torch.rand(B, NV, device=dev) < 0.19).float(), dev)
V.bf16_exact = True
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ITERS = 200


def one_stream(parts):
    def run():
        for (e, lo, n) in parts:
            e.cd_chain_x3(V, n, lo, 1e-6, 42, 0, row0=lo)
    return run


def two_streams():
    main = torch.cuda.current_stream(dev)
    sa.wait_stream(main)
    sb.wait_stream(main)
    with torch.cuda.stream(sa):
        ea.cd_chain_x3(V, B // 2, 0, 1e-6, 42, 0, row0=0)
    with torch.cuda.stream(sb):
        eb.cd_chain_x3(V, B // 2, B // 2, 1e-6, 42, 0, row0=B // 2)
    main.wait_stream(sa)
    main.wait_stream(sb)


def t(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e3


print("chain of %d rows, one stream            : %.1f us" % (B, t(one_stream([(full, 0, B)]))))
print("two chains of %d rows, one stream       : %.1f us" % (B // 2, t(one_stream([(ea, 0, B // 2), (eb, B // 2, B // 2)]))))
print("two chains of %d rows, two streams      : %.1f us" % (B // 2, t(two_streams)))
