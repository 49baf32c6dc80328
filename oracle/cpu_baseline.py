"""CPU baseline of the CD-1 step for bench.py: the reference's op sequence on torch-CPU tensors (MKL sgemm).

TEST / MEASUREMENT INFRASTRUCTURE ONLY: the reference's own TF-CPU path cannot be run here
(TensorFlow absent, path not runnable as written), so what is timed beside the GPU is this restatement of the same
op sequence -- `K.dot`, `K.sigmoid`, `K.random_uniform`, `K.less`, `K.cast`, `K.transpose`, `K.sum`, `K.update_add`
(ku/ebm/rbm.py:46-47, :119-134) -- with the backend a TF-CPU build would bottom out in as well: a multithreaded
sgemm and a native uniform generator (torch.rand stands in for TF's Philox op: this leg measures THROUGHPUT on the
host cores, the parity oracle is oracle/rbm_oracle.py).  Two variants:

  fused                 one chain per step feeds dW, db_h, db_v: 10 B V H flop -- the algorithm the GPU metric counts
  reference_sequential  the six graph executions per step of rbm.py:214-231: three independent chains for the
                        W / b_h / b_v updates applied in sequence (6 B V H each for the half steps, + 4 for dW)
                        plus the free-energy score (fe 2, a fresh chain's v' 4, fe' 2): 28 B V H flop
"""
import os
import time

import torch


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """Cores this process may actually run on: the smallest of the CPU count, the affinity mask and the cgroup quota
    (a container often sees every core of the host but is throttled to a share of them; a thread per visible core then
    thrashes: 256 threads on a 16-core share ran 30x slower than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                quota = int(parts[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def best_thread_count(limit, batch, n_vis, n_hid):
    """The thread count (<= limit) at which the step's sgemm runs fastest here: a CPU share smaller than the visible core
    count does not always show in the affinity mask or the cgroup files, so try a few counts on the actual GEMM."""
    a, b = torch.rand(batch, n_vis), torch.rand(n_vis, n_hid)
    best, best_t = 1, float("inf")
    for n in sorted({c for c in (4, 8, 16, 32, 64, 128, limit) if 1 <= c <= limit}):
        torch.set_num_threads(n)
        a @ b
        t0 = time.perf_counter()
        for _ in range(3):
            a @ b
        dt = time.perf_counter() - t0
        if dt < 0.97 * best_t:       # more threads only when they pay
            best, best_t = n, dt
    return best


def _chain(v, W, b_h, b_v, need_h_neg=True):
    """rbm.py:119-124: v_pos -> h_pos (sample) -> v_neg (sample) -> h_neg (probabilities; a graph execution that does
    not fetch anything depending on h_neg -- the visible-bias update -- never computes it)."""
    h_pos = (torch.rand(v.shape[0], W.shape[1]) < torch.sigmoid(v @ W + b_h)).float()        # rbm.py:46-47, :120
    v_neg = (torch.rand(v.shape[0], W.shape[0]) < torch.sigmoid(h_pos @ W.t() + b_v)).float()  # rbm.py:121-123
    h_neg = torch.sigmoid(v_neg @ W + b_h) if need_h_neg else None                           # rbm.py:124
    return h_pos, v_neg, h_neg


def fused_step(v, W, b_h, b_v, lr):
    h_pos, v_neg, h_neg = _chain(v, W, b_h, b_v)
    W += lr * (v.t() @ h_pos - v_neg.t() @ h_neg)                     # rbm.py:125-128
    b_h += lr * (h_pos.sum(0) - h_neg.sum(0))                         # rbm.py:129-131
    b_v += lr * (v.sum(0) - v_neg.sum(0))                             # rbm.py:132-134


def _free_energy(v, W, b_h, b_v):
    return -(v @ b_v + torch.nn.functional.softplus(v @ W + b_h).sum(1))      # rbm.py:73-75


def reference_sequential_step(v, W, b_h, b_v, lr):
    h_pos, v_neg, h_neg = _chain(v, W, b_h, b_v)                      # K.function #1  rbm.py:214
    W += lr * (v.t() @ h_pos - v_neg.t() @ h_neg)
    h_pos, v_neg, h_neg = _chain(v, W, b_h, b_v)                      # K.function #2  rbm.py:215 (sees the new W)
    b_h += lr * (h_pos.sum(0) - h_neg.sum(0))
    h_pos, v_neg, _ = _chain(v, W, b_h, b_v, need_h_neg=False)        # K.function #3  rbm.py:216
    b_v += lr * (v.sum(0) - v_neg.sum(0))
    fe = _free_energy(v, W, b_h, b_v)                                 # #4  rbm.py:227
    h = (torch.rand(v.shape[0], W.shape[1]) < torch.sigmoid(v @ W + b_h)).float()
    v_p = (torch.rand(v.shape[0], W.shape[0]) < torch.sigmoid(h @ W.t() + b_v)).float()   # #5  rbm.py:230
    fe_p = _free_energy(v_p, W, b_h, b_v)                             # #6  rbm.py:231
    return float((fe - fe_p).abs().mean())                            # rbm.py:233


def run(n_vis, n_hid, batch, budget_s=6.0, seed=1):
    """Times both variants for about `budget_s` seconds each on all host cores.  Returns the bench.py object."""
    cores = int(os.environ.get("BENCH_CPU_THREADS", "0")) or best_thread_count(usable_cpus(), batch, n_vis, n_hid)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(seed)
    v = (torch.rand(batch, n_vis, generator=g) < 0.19).float()
    lr = 1e-3 / batch
    out = {}
    for name, fn in (("fused", fused_step), ("reference_sequential", reference_sequential_step)):
        W = (torch.rand(n_vis, n_hid, generator=g) - 0.5) * 0.1
        b_h, b_v = torch.zeros(n_hid), torch.zeros(n_vis)
        fn(v, W, b_h, b_v, lr)                                        # warm: thread pool, page faults
        fn(v, W, b_h, b_v, lr)
        n, t0 = 0, time.perf_counter()
        while n < 5 or (time.perf_counter() - t0 < budget_s and n < 400):
            fn(v, W, b_h, b_v, lr)
            n += 1
        dt = time.perf_counter() - t0
        out[name] = {"steps_per_sec": n / dt, "steps": n, "seconds": dt}
    flop = {"fused": 10.0, "reference_sequential": 28.0}
    for name in out:
        out[name]["gflops"] = flop[name] * batch * n_vis * n_hid * out[name]["steps_per_sec"] / 1e9
    return {"value": out["fused"]["steps_per_sec"], "unit": "steps/s", "cores": int(torch.get_num_threads()),
            "kind": "port", "cpu_model": cpu_model(), "cpus_visible": os.cpu_count(),
            "sample": "%d fused CD-1 steps (10 B V H flop each) and %d reference_sequential steps (the six graph executions of "
                      "rbm.py:214-231, 28 B V H) of oracle/cpu_baseline.py on torch-CPU (MKL sgemm, torch.rand), %dx%d, B=%d; "
                      "a CPU restatement of ku/ebm, not TF-CPU" % (out["fused"]["steps"], out["reference_sequential"]["steps"],
                                                                  n_vis, n_hid, batch),
            "variants": out}
