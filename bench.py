#!/usr/bin/env python3
"""Benchmark of the RBM CD-1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): CD-1 Gibbs-steps/sec on a 784-visible x 1024-hidden RBM at batch 4096,
fp32, synthetic binary data resident in HBM.  --compute picks how the matrix products run: "x3"
(default: fp32 values carried as exact bf16 triples on the bf16 matrix cores, fp32 accumulate --
results held to the fp32 oracle and the fp32 tolerances by tests/test_gpu_parity.py::test_x3_*) or
"fp32" (fp32 MFMA).  Rank 0 times BOTH paths after the timed region and reports them under `paths`.  One "step" (the unit of `value`) = one full CD-1
parameter update over 4096 rows: h_pos sample, v_neg sample, h_neg probabilities, dW / db_h / db_v
applied (update_mode "fused").  With N > 1 every rank processes its own 4096 rows per step (weak
scaling: global batch 4096 N, config 3 at N = 8), the packed [dW|db_h|db_v] sums are all-reduced
over RCCL, and `value` counts N units per global step.

The line also carries `roofline` (per-launch fp32-MFMA fraction of the dominant kernel, HIP-event
timed on the launch stream) and `cpu_baseline` (the numpy oracle of the same step timed on this
host's cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_VIS, N_HID, BATCH = 784, 1024, 4096
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16
FLOP_HALF = 2.0 * BATCH * N_VIS * N_HID            # one half-step GEMM
FLOP_OUTER = 4.0 * BATCH * N_VIS * N_HID           # statistics GEMM: k = 2 x batch
FLOP_STEP = 3 * FLOP_HALF + FLOP_OUTER             # 10 B V H


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


def cpu_baseline():
    """The oracle's fused CD-1 step (numpy + OpenBLAS) on the host cores, bounded sample."""
    from oracle import rbm_oracle as O
    from oracle.make_golden import synthetic_binary, synthetic_params
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    W, b_h, b_v = synthetic_params(N_VIS, N_HID, seed=1)
    v = synthetic_binary(BATCH, N_VIS, seed=1234)
    O.cd_step_fused(W, b_h, b_v, v, 1e-3 / BATCH, 42, 0)      # warm
    n, t0 = 0, time.perf_counter()
    while n < 5 or (time.perf_counter() - t0 < 10.0 and n < 40):
        W, b_h, b_v, _, _ = O.cd_step_fused(W, b_h, b_v, v, 1e-3 / BATCH, 42, n + 1)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "steps/s", "cores": int(cores), "kind": "port",
            "sample": "%d fused CD-1 steps of oracle/rbm_oracle.py (numpy sgemm + numpy Philox), 784x1024, B=4096" % n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compute", choices=("x3", "fp32"), default=os.environ.get("BENCH_COMPUTE", "x3"))
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from keras_unsupervised_amd.ebm import dp
    from keras_unsupervised_amd.ebm.engine import DeviceMatrix, DeviceRBM

    # synthetic workload, resident in HBM before the timed region
    g = np.random.default_rng(1)
    W = g.uniform(-0.05, 0.05, size=(N_VIS, N_HID)).astype(np.float32)
    eng = DeviceRBM(W, np.zeros(N_HID, np.float32), np.zeros(N_VIS, np.float32), device)
    n_batches = 16
    u = eng.philox_uniform(n_batches * BATCH, N_VIS, 1234 + rank, 0x7004, 0)
    V = DeviceMatrix((u.t < 0.19).to(torch.float32).contiguous(), n_batches * BATCH, N_VIS, u.ld)
    del u
    lr = 1e-3 / BATCH            # keeps the weights finite over long runs; throughput does not depend on lr
    seed = 42

    # BENCH_FORCE_DP=1 runs the data-parallel sequence (emit delta, all-reduce, apply) even on one
    # rank: a plumbing rehearsal of the N > 1 path on a single-GPU box
    force_dp = os.environ.get("BENCH_FORCE_DP", "0") == "1"
    if force_dp and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)

    pipe = dp.X3Pipeline(eng) if (args.compute == "x3" and dp.PRECONVERT and not dp.OVERLAP_ROW_RANGES) else None

    def step(i):
        lo = (i % n_batches) * BATCH
        if world == 1 and not force_dp:
            eng.cd_step(V, BATCH, lo, lr, seed, i, compute=args.compute)
        elif pipe is not None:
            # chain + statistics -> packed sums -> all-reduce (with the NEXT batch's conversion under it) -> apply
            pipe.step(V, BATCH, lo, lr, seed, i, nxt=(((i + 1) % n_batches) * BATCH, BATCH), row0=rank * BATCH)
        else:
            if args.compute == "x3" and dp.OVERLAP_ROW_RANGES:
                dp.x3_sums_overlapped(eng, V, BATCH, lo, lr, seed, i, row0=rank * BATCH)
            else:
                eng.cd_step(V, BATCH, lo, lr, seed, i, apply=False, emit_delta=True, row0=rank * BATCH, compute=args.compute)
                dp.allreduce_sum_(eng.delta_buffer())
            eng.apply_delta(lr, compute=args.compute)

    # set-up, untimed and outside the W warm-up steps: every lazily created buffer (workspace, weight-piece mirror,
    # exactness flag of the data) gets created, and the GPU reaches the clocks it holds under this load -- the kernel
    # trace of a cold start shows the step time falling from 157 to 134 us over the first ~160 steps (20 ms)
    # (profiles/r01_m_kernel_stats.csv's run).  Reported in config.setup_steps.
    SETUP_STEPS = int(os.environ.get("BENCH_SETUP_STEPS", "256"))
    for i in range(SETUP_STEPS):
        step(i)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(eng.W.t).all().item()), "weights diverged"

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        # per-kernel durations (HIP events on the launch stream), each launched alone
        h_pos = eng.half_step("vh", V, BATCH, 0, 0, 1, seed, 0, 0)["sample"]
        v_neg = eng.half_step("hv", h_pos, BATCH, 0, 0, 1, seed, 1, 0)["sample"]
        h_neg = eng.half_step("vh", v_neg, BATCH, 0, 0, 0, seed, 0, 0, want_sample=False, want_prob=True)["prob"]
        import ctypes as C
        from keras_unsupervised_amd import _lib
        ws = eng.workspace(BATCH)
        rng = _lib.Rng(seed, 0, 0, 0)
        o_h = DeviceMatrix.zeros(BATCH, N_HID, device)
        o_v = DeviceMatrix.zeros(BATCH, N_VIS, device)
        dW = torch.empty((N_VIS, N_HID), dtype=torch.float32, device=device)
        st = lambda: C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        lib, ctx, P = eng.lib, eng.ctx.handle, C.byref(eng.params)
        k_vh = lambda: _lib.check(lib.kurbm_half_step_vh(ctx, P, V.ptr(), BATCH, V.ld, 0, 1, C.byref(rng), o_h.ptr(), None, o_h.ld, st()))
        k_hv = lambda: _lib.check(lib.kurbm_half_step_hv(ctx, P, h_pos.ptr(), BATCH, h_pos.ld, 0, 1, C.byref(rng), o_v.ptr(), None, o_v.ld, st()))
        k_vhp = lambda: _lib.check(lib.kurbm_half_step_vh(ctx, P, v_neg.ptr(), BATCH, v_neg.ld, 0, 0, None, None, o_h.ptr(), o_h.ld, st()))
        k_out = lambda: _lib.check(lib.kurbm_outer_partial(ctx, V.ptr(), h_pos.ptr(), v_neg.ptr(), h_neg.ptr(), BATCH, N_VIS, N_HID,
                                                           V.ld, h_pos.ld, ws.data_ptr(), ws.numel(), st()))
        k_outred = lambda: _lib.check(lib.kurbm_outer_delta(ctx, V.ptr(), h_pos.ptr(), v_neg.ptr(), h_neg.ptr(), BATCH, N_VIS, N_HID,
                                                            V.ld, h_pos.ld, dW.data_ptr(), ws.data_ptr(), ws.numel(), st()))
        kern = {}
        for name, fn, flop in (("half_step_vh_sample", k_vh, FLOP_HALF), ("half_step_hv_sample", k_hv, FLOP_HALF),
                               ("half_step_vh_prob", k_vhp, FLOP_HALF), ("outer_stats_gemm", k_out, FLOP_OUTER),
                               ("outer_stats_gemm_plus_reduce", k_outred, FLOP_OUTER)):
            ms = event_time_ms(fn, 50)
            kern[name] = {"ms": ms, "tflops": flop / (ms * 1e-3) / 1e12, "frac": flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
        # the x3 launches, each alone, replayed on the planes of a complete x3 step (kurbm_cd_step_x3_stage)
        eng.cd_step(V, BATCH, 0, lr, seed, 0, compute="x3")
        x3_exec = {"x3_half_step_vh_sample": 3 * FLOP_HALF, "x3_half_step_hv_sample": 3 * FLOP_HALF,
                   "x3_half_step_vh_prob": 3 * FLOP_HALF, "x3_stats_gemm": 4 * FLOP_HALF}     # executed bf16 flop (pieces)
        x3_algo = {"x3_half_step_vh_sample": FLOP_HALF, "x3_half_step_hv_sample": FLOP_HALF,
                   "x3_half_step_vh_prob": FLOP_HALF, "x3_stats_gemm": FLOP_OUTER}
        kern_x3 = {}
        for name, stage in (("x3_vpos_to_bf16", 0), ("x3_half_step_vh_sample", 1), ("x3_half_step_hv_sample", 2),
                            ("x3_half_step_vh_prob", 3), ("x3_stats_gemm", 4), ("x3_reduce_apply_plus_mirror", 5)):
            ms = event_time_ms(lambda stage=stage: eng.cd_step_x3_stage(V, BATCH, 0, lr, seed, 0, stage), 50)
            kern_x3[name] = {"ms": ms}
            if name in x3_exec:
                kern_x3[name].update({"executed_bf16_tflops": x3_exec[name] / (ms * 1e-3) / 1e12,
                                      "frac_of_bf16_peak": x3_exec[name] / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                      "algorithmic_tflops": x3_algo[name] / (ms * 1e-3) / 1e12})
        # both compute paths, same batch, HIP events (single GPU, local step)
        paths = {}
        for c in ("x3", "fp32"):
            ms = event_time_ms(lambda c=c: eng.cd_step(V, BATCH, 0, lr, seed, 0, compute=c), 100, warm=10)
            tf = FLOP_STEP / (ms * 1e-3) / 1e12
            paths[c] = {"ms_per_step": ms, "steps_per_sec": 1e3 / ms, "algorithmic_tflops": tf,
                        "frac_of_fp32_mfma_peak": tf / PEAK_F32_MFMA_TFLOPS}
        traffic_all = {}
        try:
            import glob
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
                traffic_all.update(json.load(open(f)))      # later files win
        except Exception:
            traffic_all = {}
        if args.compute == "x3":
            dom = max(x3_exec, key=lambda k: kern_x3[k]["ms"])
            roofline = {"bound": "mfma", "kernel": dom, "achieved": kern_x3[dom]["executed_bf16_tflops"],
                        "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": kern_x3[dom]["frac_of_bf16_peak"],
                        "note": "executed bf16 MFMA flop (the pieces: 3 per half step, 1 + 3 for the statistics) against the dense "
                                "bf16 peak; algorithmic fp32 flop/s of the same launch in kernels[...].algorithmic_tflops",
                        "traffic": traffic_all.get(dom, {}).get("hbm_bytes_per_launch"),
                        "algorithmic": {"achieved": kern_x3[dom]["algorithmic_tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                                        "unit": "TFLOP/s", "frac": kern_x3[dom]["algorithmic_tflops"] / PEAK_F32_MFMA_TFLOPS,
                                        "note": "SURVEY 8(d) algorithmic fp32 flop of this launch (4 B V H for the statistics, "
                                                "2 B V H per half step) against the fp32 MFMA peak the north star names"},
                        "kernels": kern_x3, "kernels_fp32_path": kern}
        else:
            dom = max((k for k in kern if k != "outer_stats_gemm_plus_reduce"), key=lambda k: kern[k]["ms"])
            roofline = {"bound": "mfma", "kernel": dom, "achieved": kern[dom]["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": kern[dom]["frac"],
                        "traffic": traffic_all.get(dom, {}).get("hbm_bytes_per_launch"),
                        "kernels": kern, "kernels_x3_path": kern_x3}
        step_tflops = FLOP_STEP / (ms_per_step * 1e-3) / 1e12
        roofline["step_algorithmic_tflops"] = step_tflops * world
        roofline["step_frac_of_fp32_mfma_peak"] = step_tflops / PEAK_F32_MFMA_TFLOPS
        dtype = ("f32 (storage, accumulation, results); products as exact bf16 triples on the bf16 MFMA (x3)"
                 if args.compute == "x3" else "f32")
        out = {
            "metric": "cd1_gibbs_steps_per_sec", "value": value,
            "unit": "steps/s (1 step = CD-1 update over 4096 rows, 784x1024 fp32)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "rbm_784x1024_cd1_batch4096_fp32 (BASELINE.json configs[1]%s)" % ("" if world == 1 else "; configs[2] shape: 4096 rows per GPU"),
                       "n_vis": N_VIS, "n_hid": N_HID, "setup_steps": SETUP_STEPS, "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                       "cd_k": 1, "update_mode": "fused", "lr": "1e-3/4096", "parallelism": "dp%d" % world,
                       "compute": args.compute, "flop_per_step": FLOP_STEP},
            "paths": paths,
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
