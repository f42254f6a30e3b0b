#!/usr/bin/env python
"""bench.py — BASELINE.json metric on BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 65,536 synthetic 7-feature x wd21 windows
that are already resident in HBM: dm_predict_windows (BiLSTM classify) + dm_summary_add (per-position
coverage / mod-count accumulate).  N > 1 (launched by torch.distributed.run, one rank per GPU):
windows shard across ranks with no data-path collective (weak scaling: every rank runs its own K
batches); the only collective is one integer all-reduce of the per-position counters at the end,
inside the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 65536
N_BATCHES = 16                # 16 x 65,536 = 1,048,576 windows ("10^6 windows batched 64k")
FLOP_PER_WINDOW = 8.924e6     # 11 live steps x 2 dirs x 3 layers + head
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32
CONTIG_LEN = 4_641_652        # E. coli K-12 sized contig for the synthetic summary


def cpu_baseline(weights, x_sample_src):
    """Oracle (C restatement, OpenMP) on the host cores of this box, bounded to ~10-20 s."""
    from oracle import oracle_np
    cores = os.cpu_count() or 1
    oracle_np.build_c_oracle()
    probe = x_sample_src[:max(256, 32 * cores)]
    t0 = time.perf_counter()
    oracle_np.predict_windows_c(weights, probe, nthreads=cores)
    dt = time.perf_counter() - t0
    rate = len(probe) / max(dt, 1e-6)
    n = int(min(len(x_sample_src), max(len(probe), rate * 12.0)))
    t0 = time.perf_counter()
    oracle_np.predict_windows_c(weights, x_sample_src[:n], nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "base-positions/s", "cores": cores, "kind": "port",
            "sample": "first %d windows of batch 0, oracle/deepmod_oracle.c (fp32 restatement of the TF graph, "
                      "not TensorFlow), %d OpenMP threads, %.1f s" % (n, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    dist = None
    torch = None
    if world > 1:
        import torch  # torch first: its HIP/RCCL runtime is the one the process group uses
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from deepmod_amd import _lib, model, summary, synth

    lib = _lib.load()
    if lib.dm_device_count() < 1:
        raise SystemExit("bench.py: no gfx950 device visible; there is no CPU fallback")
    device = local_rank if world > 1 else 0

    weights = synth.synthetic_weights(seed=7, scale=1.0)
    m = model.BiLSTMModel(weights, device=device)
    m.set_option(_lib.DM_OPT_PROFILE, 1)

    # synthetic windows, distinct per rank and per batch, resident in HBM before the timed region
    n_batches = min(N_BATCHES, max(1, args.steps))
    rng = np.random.default_rng(1234 + rank)
    x_dev, pos_dev, flag_dev = [], [], []
    x0 = None
    for b in range(n_batches):
        xb = synth.synthetic_windows(BATCH, seed=20260928 + 1000 * rank + b)
        if b == 0:
            x0 = xb
        x_dev.append(model.DeviceArray.from_host(xb, device))
        # one aligned base per window: consecutive reference positions of ~8 kb reads, C in ~25 % of rows
        start = rng.integers(0, CONTIG_LEN - BATCH)
        pos_dev.append(model.DeviceArray.from_host((start + np.arange(BATCH)).astype(np.int64), device))
        base_is_c = rng.random(BATCH) < 0.25
        not_gap = rng.random(BATCH) < 0.97
        flag_dev.append((base_is_c.astype(np.uint8) | (not_gap.astype(np.uint8) << 1)))
    prob_dev = model.DeviceArray((BATCH, 2), np.float32, device)
    cls_dev = model.DeviceArray((BATCH,), np.uint8, device)
    flag_dev = [model.DeviceArray.from_host(f, device) for f in flag_dev]
    summ = summary.PositionSummary(CONTIG_LEN, device=device)

    def step(i):
        b = i % n_batches
        m.predict_windows(x_dev[b], prob=prob_dev, cls=cls_dev)
        summ.add_classified(pos_dev[b], flag_dev[b], cls_dev, BATCH)

    def sync_all():
        m.sync()
        summ.sync()
        if torch is not None:
            torch.cuda.synchronize()

    if torch is None:
        try:  # contract: bracket the timed region with torch.cuda.synchronize() as well
            import torch as _t
            if _t.cuda.is_available():
                torch = _t
        except Exception:
            torch = None

    for i in range(args.warmup):
        step(i)
    sync_all()
    m.profile_reset()
    if dist is not None:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        summ.all_reduce_torch(dist)
    sync_all()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms, launches, kwindows = m.profile_get()
    total_windows = BATCH * args.steps * world
    value = total_windows / elapsed

    if rank == 0:
        avg_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        achieved = (kwindows / max(launches, 1)) * FLOP_PER_WINDOW / avg_launch_s / 1e12
        touch, cov, mod = summ.fetch()
        out = {
            "metric": "base-positions/sec (whole node), E. coli 5mC wd21/f7 BiLSTM",
            "value": value, "unit": "base-positions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: rnn_conmodC_P100wd21_f7ne1u0_4 geometry (3x100 BiLSTM, wd21, f7), "
                                   "synthetic weights (real .data shards absent), %d windows/step resident in HBM, "
                                   "%d distinct batches (1,048,576 windows)" % (BATCH, n_batches),
                       "batch": BATCH, "windows_total": total_windows, "parallelism": "window-sharded x%d" % world,
                       "precision": "f32 MFMA (exact)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "lstm32::bilstm_f32_kernel", "avg_launch_ms": avg_launch_s * 1e3,
                         "launches": launches, "flop_per_window": FLOP_PER_WINDOW,
                         "peak_note": "v_mfma_f32_16x16x4_f32 dense fp32, 157.3 TF"},
            "summary_check": {"touch": int(touch.sum()), "cov": int(cov.sum()), "mod": int(mod.sum())},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, x0)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
