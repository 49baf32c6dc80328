#!/usr/bin/env python3
"""Benchmark of the RBM CD-1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): CD-1 Gibbs-steps/sec on a 784-visible x 1024-hidden RBM at batch 4096, fp32 storage /
accumulation / results, synthetic binary data resident in HBM.  One "step" (the unit of `value`) = one full CD-1
parameter update over 4096 rows: h_pos sample, v_neg sample, h_neg probabilities, dW / db_h / db_v applied
(update_mode "fused").  --compute picks how the matrix products run: "x3" (default: fp32 values carried as exact
bf16 triples on the bf16 matrix cores -- held to the fp32 oracle and the fp32 tolerances by the tests) or "fp32"
(fp32 MFMA); rank 0 times BOTH after the timed region and reports them under `paths`, next to the real-valued-data
and Gaussian-mode (the reference's default) variants.

Timing protocol: W untimed warm-up steps, then R blocks of exactly K steps, each bracketed by a barrier +
torch.cuda.synchronize() on both sides (R = max(5, ceil(2000 / K)): at least 2 000 timed steps, ~0.2 s of GPU work, so
that the median block is past the clock ramp of a GPU that was idle when the process started); the block time is the MAX
over ranks.  `value` = N * K / median(block times); `value_first_block` is the first block alone (a cold GPU: the
step time falls ~15 % over the first ~20 ms while the clocks ramp); `value_steady` is the median of R more blocks
after 256 further untimed steps.  All three are on the line, labelled in config.timing.

With N > 1 (one process per GPU, torch.distributed.run) every rank processes its own 4096 rows per step (weak scaling:
global batch 4096 N, config 3 at N = 8) through kurbm_cd_step_x3_dp: the packed [dW|db_h|db_v] sums (3.2 MB) are all-reduced
by RCCL inside libkurbm.so -- one range on the launch stream at this size.  torch.distributed (gloo)
only carries the RCCL unique id and the max-over-ranks of the block times; the barriers are all-reduces on the RCCL
communicator itself.  `rccl_ranks` is ncclCommCount's answer.  UNVERIFIED ON HARDWARE: no box with more than one GPU has been
available to this build, so the N > 1 path has run only at world size 1 (BENCH_FORCE_DP=1), on CPU doubles at world size 2
(tests/test_dp_gloo.py), and in tests/test_two_gpus.py nowhere yet.

The line also carries `roofline` (dominant kernel, HIP-event timed on the launch stream, and the whole step) and
`cpu_baseline` (oracle/cpu_baseline.py: the same op sequence on torch-CPU over all host cores; rank 0, N = 1 only).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

# (RCCL / device-memory sharing across the processes of a node needs dmabuf IPC on this stack; the GPU boxes export it already)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip paths.* beyond x3 / fp32 (real-valued data, Gaussian mode): they launch the SAME kernels with more "
                         "segments, which would blur a rocprofv3 --stats average of the config-2 launches (tools/profile_round.sh)")
    ap.add_argument("--compute", choices=("x3", "fp32"), default=os.environ.get("BENCH_COMPUTE", "x3"))
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start one rank per GPU as CHILD processes
    (python -m torch.distributed.run ... bench.py ...), forward their output (rank 0 prints the JSON line) and exit with the
    launcher's code.  This runs before torch is imported: the parent never touches a GPU, and nothing that has is re-executed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: WORLD_SIZE unset, --gpus %d: launching %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=dict(os.environ))


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _args = parse_args()
    if _args.gpus > 1:
        sys.exit(self_launch(_args))

import numpy as np
import torch
import torch.distributed as dist

N_VIS, N_HID, BATCH = 784, 1024, 4096
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16
PEAK_FP8_MFMA_TFLOPS = 5000.0         # MI355X_MICROARCH.md: dense fp8 (v_mfma_scale_f32_16x16x128_f8f6f4)
BVH2 = 2.0 * BATCH * N_VIS * N_HID                 # one GEMM unit: B x V x H multiply-adds
FLOP_HALF = BVH2                                   # one half-step GEMM
FLOP_OUTER = 2 * BVH2                              # statistics GEMM: k = 2 x batch
FLOP_STEP = 3 * FLOP_HALF + FLOP_OUTER             # 10 B V H, the algorithmic work of a CD-1 step
# bf16 GEMM units the x3 path EXECUTES per step (pieces): binary data 3+3+3 half steps, 1+3 statistics; real-valued data
# 6+3+3 and 3+3; Gaussian visibles on real-valued data 6+3+(3+2+1) and 3+(3+2+1)
X3_UNITS = {"binary": 13, "real": 18, "gaussian_real": 24}
SETTLE_STEPS = 256


def event_time_ms(fn, iters, warm=3):
    """Average duration of fn() in ms by HIP events on torch's current stream (= the launch stream)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def choose_exchange(eng, V, device, rank, world, planes, lr, seed, compute):
    """The exchange of the data-parallel step: KURBM_DP_EXCHANGE = rccl (ncclAllReduce inside libkurbm.so), peer
    (the two-shot exchange over hipIpc peer pointers) or, the default here, auto: RCCL is created first and is what runs unless the
    peer exchange (a) comes up on every rank, (b) reproduces RCCL's sum of a test vector and reports no timeout, and (c) runs the
    data-parallel step faster, by the max over ranks of a short timed probe (every rank takes the same decision: the verdict is
    all-reduced over the control plane).  Returns (exchange, a note for the JSON line)."""
    import torch.distributed as dist
    from keras_unsupervised_amd import _lib
    from keras_unsupervised_amd.ebm import dp
    kind = os.environ.get("KURBM_DP_EXCHANGE", "auto").lower()
    if kind == "peer":
        return dp.PeerExchange(device, rank, world, N_VIS, N_HID), "peer (KURBM_DP_EXCHANGE=peer)"
    if kind == "rccl" or compute != "x3":
        return dp.get_comm(device), "rccl"

    def agree(flag):      # True only if every rank says so
        if world == 1:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def slowest(x):       # max over ranks
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    rccl, rccl_err = None, None
    try:
        rccl = dp.get_comm(device)
    except Exception as e:  # noqa: BLE001  (e.g. BENCH_SHARE_GPU=1: RCCL refuses two ranks on one device)
        rccl_err = str(e)[:160]
    if not agree(rccl is not None):
        if rccl is not None:
            dp.destroy_comms()
        sys.stderr.write("bench.py: RCCL communicator unavailable (%s): trying the peer exchange\n" % rccl_err)
        return dp.PeerExchange(device, rank, world, N_VIS, N_HID), "peer (auto: no RCCL communicator: %s)" % rccl_err
    px, why = None, None
    try:
        os.environ.setdefault("KURBM_PEER_TIMEOUT_MS", "2000")      # (a broken peer path must cost the probe seconds, not minutes)
        px = dp.PeerExchange(device, rank, world, N_VIS, N_HID)
    except Exception as e:  # noqa: BLE001  (on EVERY rank or on none: PeerExchange agrees on that itself)
        why = "peer exchange unavailable (%s)" % str(e)[:120]
    if px is not None:
        # several exchanges of DIFFERENT vectors through the same buffers (a stale read of a peer's buffer -- a cached line of the
        # epoch before -- would show from the second one on), each against RCCL's sum of the same vectors
        g = torch.Generator(device="cpu").manual_seed(77 + rank)
        same = True
        for _ in range(4):
            t0 = torch.randn(N_VIS * N_HID + N_HID + N_VIS, generator=g).to(device)
            a, b = t0.clone(), t0.clone()
            rccl.allreduce_sum_(a)
            px.allreduce_sum_(b)
            torch.cuda.synchronize()
            same = same and bool(torch.allclose(a, b, rtol=1e-5, atol=1e-5))
        same = same and eng.ctx.status() == 0
        if not agree(same):
            why = "peer exchange failed its self-test against RCCL's sums"
    if px is not None and why is None:
        W0 = eng.get_weights()
        times = {}
        for name, x in (("rccl", rccl), ("peer", px)):
            for i in range(5):
                eng.cd_step_dp(x, V, BATCH, 0, lr, seed, i, row0=rank * BATCH, compute="x3", planes=planes)
            x.barrier()
            t = time.perf_counter()
            for i in range(20):
                eng.cd_step_dp(x, V, BATCH, 0, lr, seed, 5 + i, row0=rank * BATCH, compute="x3", planes=planes)
            x.barrier()
            times[name] = slowest((time.perf_counter() - t) / 20)
        ok = agree(eng.ctx.status() == 0)
        eng.set_weights(*W0)
        if not ok:
            why = "peer exchange timed out in the probe"
        elif times["peer"] < times["rccl"]:
            return px, "peer (auto: probe %.1f us per step against %.1f with RCCL's all-reduce)" % (times["peer"] * 1e6, times["rccl"] * 1e6)
        else:
            why = "rccl (auto: probe %.1f us per step against %.1f with the peer exchange)" % (times["rccl"] * 1e6, times["peer"] * 1e6)
    if px is not None:
        px.destroy()
    return rccl, why if why and why.startswith("rccl") else "rccl (auto: %s)" % why


def main():
    args = parse_args()
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner on file descriptor 1 when a communicator comes up,
    # gloo its connection notes -- everything written before the result goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # (WORLD_SIZE unset and --gpus N > 1 never gets here: bench.py then launches its own ranks, see self_launch)
        sys.exit("WORLD_SIZE is %d but --gpus is %d: start bench.py without a launcher (it starts its own ranks), or with "
                 "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py "
                 "--gpus %d ..." % (world, args.gpus, args.gpus, args.gpus))
    if local_rank >= torch.cuda.device_count() and os.environ.get("BENCH_SHARE_GPU", "0") != "1":
        sys.exit("rank %d: local rank %d has no GPU (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    if os.environ.get("BENCH_SHARE_GPU", "0") == "1":
        local_rank = 0       # rehearsal of the N > 1 path on a one-GPU box: every rank on device 0 (RCCL refuses that; the peer exchange does not)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only (unique id, max of times)

    from keras_unsupervised_amd.ebm import dp
    from keras_unsupervised_amd.ebm.engine import MODE_VISIBLE_GAUSSIAN, DeviceMatrix, DeviceRBM

    # synthetic workload, resident in HBM before the timed region
    g = np.random.default_rng(1)
    W = g.uniform(-0.05, 0.05, size=(N_VIS, N_HID)).astype(np.float32)
    eng = DeviceRBM(W, np.zeros(N_HID, np.float32), np.zeros(N_VIS, np.float32), device)
    n_batches = 16
    u = eng.philox_uniform(n_batches * BATCH, N_VIS, 1234 + rank, 0x7004, 0)
    V = DeviceMatrix((u.t < 0.19).to(torch.float32).contiguous(), n_batches * BATCH, N_VIS, u.ld)
    lr = 1e-3 / BATCH            # keeps the weights finite over long runs; throughput does not depend on lr
    seed = 42
    # what RBM.fit does once per call on the x3 path: the bf16 planes of every batch window of the resident data matrix
    # (kurbm_x3_convert_rows); the steps then read them instead of converting their 4096 rows again every epoch
    planes = eng.make_planes(V, [(i * BATCH, BATCH) for i in range(n_batches)]) if args.compute == "x3" else None

    # BENCH_FORCE_DP=1 runs the data-parallel step (1-rank RCCL communicator) on a single-GPU box: a rehearsal of the N > 1 path
    use_dp = world > 1 or os.environ.get("BENCH_FORCE_DP", "0") == "1"
    comm, exchange_note = (choose_exchange(eng, V, device, rank, world, planes, lr, seed, args.compute) if use_dp else (None, None))
    rccl_ranks = comm.count() if comm is not None else None
    if comm is not None:
        assert rccl_ranks == world, "the exchange reports %d ranks, launched %d" % (rccl_ranks, world)

    def step(i):
        lo = (i % n_batches) * BATCH
        if comm is None:
            eng.cd_step(V, BATCH, lo, lr, seed, i, compute=args.compute, planes=planes)
        else:
            eng.cd_step_dp(comm, V, BATCH, lo, lr, seed, i, row0=rank * BATCH, compute=args.compute, planes=planes)

    def fence():
        if comm is not None:
            comm.barrier()       # device idle -> all-reduce on the RCCL communicator -> device idle
        torch.cuda.synchronize()

    counter = [0]

    def timed_block(k):
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step(counter[0])
            counter[0] += 1
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step(counter[0])
        counter[0] += 1
    repeats = max(5, math.ceil(2000 / max(args.steps, 1)))
    blocks = [timed_block(args.steps) for _ in range(repeats)]
    for _ in range(SETTLE_STEPS):
        step(counter[0])
        counter[0] += 1
    steady = [timed_block(args.steps) for _ in range(repeats)]
    assert bool(torch.isfinite(eng.W.t).all().item()), "weights diverged"
    eng.check_status()   # (a device-side wait that timed out -- the peer exchange -- would have skipped updates: not a measurement)
    fence()          # every rank is past its last data-parallel step; from here on only rank 0 touches its GPU

    # ---- the honest variants, under the SAME block protocol as `value` (single GPU, x3) -------------------------------------
    # value_convert_per_step: no resident planes -- every step converts its own 4096 rows, as the FIRST epoch of a fit() does
    #   for every window (the reference's example conf is `epochs: 1`: rbm_softmax_mnist_conf.json:14);
    # value_reference_default: what `RBM(hps, n).fit(V)` is in the reference -- the constructor's default Gaussian-visible mode
    #   (rbm.py:22) on grey-level data (rbm_softmax_mnist.py:103) with the default verbose = 1, i.e. the update AND the score pass
    #   (rbm.py:225-234) every step; planes resident, the score left on the device as fit() leaves it.
    variants = {}
    if world == 1 and args.compute == "x3" and not args.no_variants:
        from keras_unsupervised_amd.ebm.engine import CHAIN_SCORE

        def protocol(step_fn):
            c = [0]
            def block(k):
                fence()
                t0 = time.perf_counter()
                for _ in range(k):
                    step_fn(c[0])
                    c[0] += 1
                fence()
                return time.perf_counter() - t0
            for _ in range(args.warmup):
                step_fn(c[0])
                c[0] += 1
            bl = [block(args.steps) for _ in range(repeats)]
            med = statistics.median(bl)
            return {"value": args.steps / med, "ms_per_step": med / args.steps * 1e3, "value_first_block": args.steps / bl[0],
                    "block_ms": [b * 1e3 for b in bl]}
        Wkeep0 = eng.get_weights()
        variants["value_convert_per_step"] = protocol(
            lambda i: eng.cd_step(V, BATCH, (i % n_batches) * BATCH, lr, seed, i, compute="x3", planes=None))
        ug = eng.philox_uniform(n_batches * BATCH, N_VIS, 99, 0x7005, 0)
        Vg16 = DeviceMatrix((torch.floor(ug.t * 256.0) / 255.0).contiguous(), n_batches * BATCH, N_VIS, ug.ld)
        del ug
        plg16 = eng.make_planes(Vg16, [(i * BATCH, BATCH) for i in range(n_batches)], MODE_VISIBLE_GAUSSIAN)

        def ref_default(i, score=True, mode=MODE_VISIBLE_GAUSSIAN):
            lo = (i % n_batches) * BATCH
            eng.cd_step(Vg16, BATCH, lo, lr, seed, i, mode=mode, compute="x3", planes=plg16)
            if score:
                eng.score_x3(Vg16, BATCH, lo, seed, i, mode, CHAIN_SCORE, planes=plg16)
        variants["value_reference_default"] = protocol(ref_default)
        variants["value_reference_default_quiet"] = protocol(lambda i: ref_default(i, score=False))
        variants["value_grey_level_bernoulli"] = protocol(lambda i: ref_default(i, score=False, mode=0))
        variants["note"] = ("same block protocol as `value` (W = %d warm-up steps, %d blocks of %d steps, median): value_convert_per_step = 0/1 data, "
                            "no resident planes (the epochs: 1 case); value_reference_default = Gaussian-visible mode (rbm.py:22) on grey-level "
                            "data with the score pass of fit(verbose=1) (rbm.py:225-234) in every step; _quiet = the same without the score; "
                            "value_grey_level_bernoulli = Bernoulli mode on the same data, no score" % (args.warmup, repeats, args.steps))
        eng.set_weights(*Wkeep0)
        del Vg16, plg16
        fence()

    out = None
    if rank == 0:
        elapsed = statistics.median(blocks)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        rate = lambda dt: world * args.steps / dt
        import ctypes as C
        from keras_unsupervised_amd import _lib
        # ---- per-kernel durations (HIP events on the launch stream), each launched alone --------------------------
        h_pos = eng.half_step("vh", V, BATCH, 0, 0, 1, seed, 0, 0)["sample"]
        v_neg = eng.half_step("hv", h_pos, BATCH, 0, 0, 1, seed, 1, 0)["sample"]
        h_neg = eng.half_step("vh", v_neg, BATCH, 0, 0, 0, seed, 0, 0, want_sample=False, want_prob=True)["prob"]
        ws = eng.workspace(BATCH)
        rng = _lib.Rng(seed, 0, 0, 0)
        o_h = DeviceMatrix.zeros(BATCH, N_HID, device)
        o_v = DeviceMatrix.zeros(BATCH, N_VIS, device)
        st = lambda: C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        lib, ctx, P = eng.lib, eng.ctx.handle, C.byref(eng.params)
        k_vh = lambda: _lib.check(lib.kurbm_half_step_vh(ctx, P, V.ptr(), BATCH, V.ld, 0, 1, C.byref(rng), o_h.ptr(), None, o_h.ld, st()))
        k_hv = lambda: _lib.check(lib.kurbm_half_step_hv(ctx, P, h_pos.ptr(), BATCH, h_pos.ld, 0, 1, C.byref(rng), o_v.ptr(), None, o_v.ld, st()))
        k_vhp = lambda: _lib.check(lib.kurbm_half_step_vh(ctx, P, v_neg.ptr(), BATCH, v_neg.ld, 0, 0, None, None, o_h.ptr(), o_h.ld, st()))
        k_out = lambda: _lib.check(lib.kurbm_outer_partial(ctx, V.ptr(), h_pos.ptr(), v_neg.ptr(), h_neg.ptr(), BATCH, N_VIS, N_HID,
                                                           V.ld, h_pos.ld, ws.data_ptr(), ws.numel(), st()))
        kern = {}
        for name, fn, flop in (("half_step_vh_sample", k_vh, FLOP_HALF), ("half_step_hv_sample", k_hv, FLOP_HALF),
                               ("half_step_vh_prob", k_vhp, FLOP_HALF), ("outer_stats_gemm", k_out, FLOP_OUTER)):
            ms = event_time_ms(fn, 50)
            kern[name] = {"ms": ms, "tflops": flop / (ms * 1e-3) / 1e12, "frac": flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        # the x3 launches, each alone, replayed on the planes of a complete x3 step (kurbm_cd_step_x3_stage)
        eng.cd_step(V, BATCH, 0, lr, seed, 0, compute="x3")
        # GEMM units a launch EXECUTES: (bf16 units, fp8 units).  0/1 data: the positive statistics v_pos^T h_pos run on the
        # fp8 matrix cores (KURBM_X3_F8POS, default on), the other products as three bf16 pieces of the real-valued operand
        f8pos = os.environ.get("KURBM_X3_F8POS", "1") != "0"
        x3_units = {"x3_half_step_vh_sample": (3, 0), "x3_half_step_hv_sample": (3, 0), "x3_half_step_vh_prob": (3, 0),
                    "x3_stats_gemm": (3, 1) if f8pos else (4, 0)}

        def mfma_roof_ms(u16, u8):   # the launch's MFMAs at the dense peaks of their types
            return (u16 * BVH2 / (PEAK_BF16_MFMA_TFLOPS * 1e12) + u8 * BVH2 / (PEAK_FP8_MFMA_TFLOPS * 1e12)) * 1e3
        x3_algo = {"x3_half_step_vh_sample": FLOP_HALF, "x3_half_step_hv_sample": FLOP_HALF,
                   "x3_half_step_vh_prob": FLOP_HALF, "x3_stats_gemm": FLOP_OUTER}
        kern_x3 = {}
        for name, stage in (("x3_vpos_to_bf16", 0), ("x3_half_step_vh_sample", 1), ("x3_half_step_hv_sample", 2),
                            ("x3_half_step_vh_prob", 3), ("x3_stats_gemm", 4), ("x3_reduce_apply_plus_mirror", 5)):
            ms = event_time_ms(lambda stage=stage: eng.cd_step_x3_stage(V, BATCH, 0, lr, seed, 0, stage), 50)
            kern_x3[name] = {"ms": ms}
            if name in x3_units:
                u16, u8 = x3_units[name]
                ex, roof = (u16 + u8) * BVH2, mfma_roof_ms(u16, u8)
                kern_x3[name].update({"executed_tflops": ex / (ms * 1e-3) / 1e12, "units_bf16": u16, "units_fp8": u8,
                                      "mfma_roof_ms": roof, "frac_of_mfma_roof": roof / ms,
                                      "algorithmic_tflops": x3_algo[name] / (ms * 1e-3) / 1e12})
        # ---- whole-step variants, same engine, HIP events (single GPU, local step) ---------------------------------
        def path(fn, units=None):
            ms = event_time_ms(fn, 100, warm=10)
            tf = FLOP_STEP / (ms * 1e-3) / 1e12
            r = {"ms_per_step": ms, "steps_per_sec": 1e3 / ms, "algorithmic_tflops": tf}
            if units is None:
                r["frac_of_fp32_mfma_peak"] = tf / PEAK_F32_MFMA_TFLOPS
            else:
                r["executed_tflops"] = units * BVH2 / (ms * 1e-3) / 1e12
                r["executed_frac_of_bf16_peak"] = r["executed_tflops"] / PEAK_BF16_MFMA_TFLOPS
            return r
        pl1 = planes if planes is not None else eng.make_planes(V, [(0, BATCH)])
        paths = {"x3": path(lambda: eng.cd_step(V, BATCH, 0, lr, seed, 0, compute="x3", planes=pl1), X3_UNITS["binary"]),
                 "x3_convert_per_step": path(lambda: eng.cd_step(V, BATCH, 0, lr, seed, 0, compute="x3"), X3_UNITS["binary"]),
                 "fp32": path(lambda: eng.cd_step(V, BATCH, 0, lr, seed, 0, compute="fp32"))}
        # grey-level data (k / 255, not bf16-exact: the batch travels as three pieces), Bernoulli mode and the
        # reference's default Gaussian-visible mode (rbm.py:22; relu thresholds, N(loc, 1) visibles)
        if not args.no_variants:
            ug = eng.philox_uniform(BATCH, N_VIS, 99, 0x7005, 0)
            Vg = DeviceMatrix((torch.floor(ug.t * 256.0) / 255.0).contiguous(), BATCH, N_VIS, ug.ld)
            plg = eng.make_planes(Vg, [(0, BATCH)])
            paths["x3_real_valued"] = path(lambda: eng.cd_step(Vg, BATCH, 0, lr, seed, 0, compute="x3", planes=plg), X3_UNITS["real"])
            paths["fp32_real_valued"] = path(lambda: eng.cd_step(Vg, BATCH, 0, lr, seed, 0, compute="fp32"))
            Wkeep = eng.get_weights()
            paths["gaussian_default_mode"] = path(lambda: eng.cd_step(Vg, BATCH, 0, lr, seed, 0, mode=MODE_VISIBLE_GAUSSIAN, compute="x3",
                                                                      planes=plg), X3_UNITS["gaussian_real"])
            paths["gaussian_default_mode_fp32"] = path(lambda: eng.cd_step(Vg, BATCH, 0, lr, seed, 0, mode=MODE_VISIBLE_GAUSSIAN, compute="fp32"))
            eng.set_weights(*Wkeep)
        traffic_all, traffic_file, pmc_all, pmc_file = {}, None, {}, None
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
                traffic_all.update(json.load(open(f)))      # later files win
                traffic_file = os.path.relpath(f, ROOT)
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_x3_gemm.json"))):
                pmc_all.update(json.load(open(f)))
                pmc_file = os.path.relpath(f, ROOT)
        except Exception:
            traffic_all, pmc_all = {}, {}
        if args.compute == "x3":
            # The dominant KERNEL, as `rocprofv3 --stats` sees it: both sampling half steps are launches of one kernel (one row of
            # kernel_stats.csv, its average over both), so they compete as their mean -- not the slower launch of the pair alone
            symbols = {"x3_half_step_vh_sample": "k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true, false>",
                       "x3_half_step_hv_sample": "k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true, false>",
                       "x3_half_step_vh_prob": "k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 0, true, false>",
                       "x3_stats_gemm": "k_gemm_pb<256, 64, 4, 2, 64, 3, 1, 0, true, false>"}
            pair = ["x3_half_step_vh_sample", "x3_half_step_hv_sample"]
            if all(k in kern_x3 for k in pair):
                ms2 = sum(kern_x3[k]["ms"] for k in pair) / 2
                e = dict(kern_x3[pair[0]])
                scale = e["ms"] / ms2
                e.update(ms=ms2, executed_tflops=e["executed_tflops"] * scale, algorithmic_tflops=e["algorithmic_tflops"] * scale,
                         frac_of_mfma_roof=e["mfma_roof_ms"] / ms2, launches=pair)
                kern_x3["x3_half_step_sample_both_launches"] = e
                symbols["x3_half_step_sample_both_launches"] = symbols[pair[0]]
                cand = [k for k in x3_units if k not in pair] + ["x3_half_step_sample_both_launches"]
            else:
                cand = list(x3_units)
            dom = max(cand, key=lambda k: kern_x3[k]["ms"])
            dom_pmc = pair[1] if dom == "x3_half_step_sample_both_launches" else dom
            step16, step8 = (12, 1) if f8pos else (13, 0)
            roofline = {"bound": "mfma", "kernel": dom, "kernel_symbol": symbols.get(dom), "achieved": kern_x3[dom]["executed_tflops"],
                        "peak": kern_x3[dom]["executed_tflops"] / kern_x3[dom]["frac_of_mfma_roof"], "unit": "TFLOP/s",
                        "frac": kern_x3[dom]["frac_of_mfma_roof"],
                        "frac_algorithmic": kern_x3[dom]["algorithmic_tflops"] / PEAK_BF16_MFMA_TFLOPS,
                        "frac_algorithmic_note": "SURVEY 8(d)'s algorithmic flop of the launch (statistics 4 B V H, a half step 2 B V H) over its "
                                                 "duration, against the dense bf16 MFMA peak (the unit that executes it)",
                        "note": "EXECUTED MFMA flop of the launch (3 GEMM units per half step: the three bf16 pieces of W; statistics: "
                                "v_pos^T h_pos, 0/1 x 0/1, one unit on the fp8 matrix cores + three bf16 units for v_neg^T h_neg; one unit = "
                                "2 B V H) over the launch duration (HIP events, this run); peak = that flop over the time its MFMAs "
                                "take at the dense peaks of their types (bf16 2.5, fp8 5.0 PFLOP/s), frac = that time / the duration",
                        "traffic": traffic_all.get(dom_pmc, {}).get("hbm_bytes_per_launch"),
                        "traffic_source": "%s (rocprofv3 --pmc passes of an earlier run of this command; not measured in this run)" % traffic_file,
                        "mfma_busy": pmc_all.get(dom_pmc, {}).get("mfma_busy_share"),
                        "wait_any": pmc_all.get(dom_pmc, {}).get("wait_any_share"),
                        "l2_hit_rate": pmc_all.get(dom_pmc, {}).get("l2_hit_rate"),
                        "counters_source": "%s (SQ / TCC counter passes of tools/profile_round.sh on the same kernels; not measured in this run)" % pmc_file,
                        "step": {"executed_tflops": X3_UNITS["binary"] * BVH2 / (ms_per_step * 1e-3) / 1e12 * world,
                                 "frac": mfma_roof_ms(step16, step8) / ms_per_step,
                                 "note": "all 13 GEMM units of a step (%d bf16 + %d fp8): the time their MFMAs take at the dense peaks "
                                         "over the timed step time, per GPU" % (step16, step8)},
                        "kernels": kern_x3, "kernels_fp32_path": kern}
        else:
            dom = max(kern, key=lambda k: kern[k]["ms"])
            roofline = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": kern[dom]["frac"],
                        "traffic": traffic_all.get(dom_pmc, {}).get("hbm_bytes_per_launch"),
                        "traffic_source": "%s (rocprofv3 --pmc passes of an earlier run; not measured in this run)" % traffic_file,
                        "step": {"algorithmic_tflops": FLOP_STEP / (ms_per_step * 1e-3) / 1e12 * world,
                                 "frac": FLOP_STEP / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS},
                        "kernels": kern, "kernels_x3_path": kern_x3}
        dtype = ("f32 (storage, accumulation, results); products as exact bf16 triples on the bf16 MFMA (x3), the 0/1 x 0/1 "
                 "product of the positive statistics on the fp8 MFMA (exact)" if args.compute == "x3" else "f32")
        out = {
            "metric": "cd1_gibbs_steps_per_sec", "value": value,
            "unit": "steps/s (1 step = CD-1 update over 4096 rows, 784x1024 fp32)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "value_first_block": rate(blocks[0]), "value_steady": rate(statistics.median(steady)),
            "rccl_ranks": rccl_ranks,
            "value_convert_per_step": variants.get("value_convert_per_step", {}).get("value"),
            "value_reference_default": variants.get("value_reference_default", {}).get("value"),
            "variants": variants or None,
            "config": {"workload": "rbm_784x1024_cd1_batch4096_fp32 (BASELINE.json configs[1]%s)" % ("" if world == 1 else "; configs[2] shape: 4096 rows per GPU"),
                       "n_vis": N_VIS, "n_hid": N_HID, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "cd_k": 1, "update_mode": "fused", "lr": "1e-3/4096", "parallelism": "dp%d" % world,
                       "compute": args.compute, "flop_per_step": FLOP_STEP,
                       "data_planes": None if planes is None else "resident: bf16 planes of the 16 batch windows made once before the warm-up "
                                      "(%.0f MB), as RBM.fit does once per call; paths.x3_convert_per_step converts the batch in every step" % (planes.buf.numel() / 1e6),
                       "data_parallel_step": None if comm is None else (
                           "kurbm_cd_step_x3_peer (two-shot all-reduce over hipIpc peer pointers, the apply fused into its second shot)"
                           if hasattr(comm, "capacity") else
                           "kurbm_cd_step_x3_dp (RCCL all-reduce inside libkurbm.so; row ranges: KURBM_DP_CHUNKS, default one)"),
                       "exchange": exchange_note,
                       "timing": {"blocks": repeats, "steps_per_block": args.steps,
                                  "value": "N x K / median block time; W = %d untimed warm-up steps before the first block" % args.warmup,
                                  "value_first_block": "first block alone (cold clocks)",
                                  "value_steady": "median of %d more blocks after %d further untimed steps" % (repeats, SETTLE_STEPS),
                                  "block_ms": [b * 1e3 for b in blocks], "steady_block_ms": [b * 1e3 for b in steady]}},
            "paths": paths,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_baseline
            out["cpu_baseline"] = cpu_baseline.run(N_VIS, N_HID, BATCH)
    if dist.is_initialized():
        dist.barrier()   # (gloo: the other ranks wait here, on the host, while rank 0 measures)
    if comm is not None:
        if hasattr(comm, "capacity"):
            comm.destroy()
        dp.destroy_comms()
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
