#!/usr/bin/env python3
"""Is fit(verbose=1) host-bound?  Host time to enqueue one step + one score (Python + ctypes + launches) against the GPU time of
the same work, config 2; and the pieces: the step, the score call, the pinned-ring push."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from keras_unsupervised_amd.ebm import MODE_VISIBLE_BERNOULLI, RBM  # noqa: E402
from keras_unsupervised_amd.ebm.engine import DeviceMatrix  # noqa: E402
from keras_unsupervised_amd.ebm.rbm import _ScoreRing  # noqa: E402

B, NV, NH = 4096, 784, 1024
dev = torch.device("cuda", 0)
V = DeviceMatrix.from_host((torch.rand(4 * B, NV, device=dev) < 0.19).float(), dev)
r = RBM({"batch_size": B, "epochs": 1, "lr": 1e-3 / B}, NH, mode=MODE_VISIBLE_BERNOULLI, seed=1)
r.build((None, NV))
d = r._dev
r._planes = d.make_planes(V, [(i * B, B) for i in range(4)], r.mode)
ring = _ScoreRing(dev, lambda s, l: None)


def host(fn, n=200):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


step = lambda i: r._update_local(V, (i % 4) * B, B, 1e-3 / B, i)
score = lambda i: r._score(V, (i % 4) * B, B, i)
both = lambda i: (step(i), ring.push(score(i), (i, 0)))
for name, fn in (("step", step), ("score", score), ("step + score + ring", both)):
    h, g = host(fn)
    print("%-20s host enqueue %.1f us, until the GPU is done %.1f us per call" % (name, h, g))
ring.flush()
