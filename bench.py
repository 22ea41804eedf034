#!/usr/bin/env python3
"""Benchmark of the card.io-dmz scan hot path on MI355X.

A "step" = one pass of the full per-frame pipeline (detect edges -> rectify card
-> number-row search -> digit segmentation -> digit categorisation -> expiry
segmentation + categorisation) over one HBM-resident batch of synthetic 640x480
luma frames (BASELINE.json configs[3]).  `value` = frames/s of the whole job (all ranks), inputs resident
in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
one rank per GPU; frames are sharded (weak scaling: B frames per GPU) and the fixed-size
result and expiry records of each step are gathered on rank 0 (RCCL sends over xGMI),
asynchronously: the exchange of step k overlaps the kernels of step k+1 (two alternating
record buffers), and the timed region ends only when the last gather has completed.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEED = 0xCA4D10
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # vector == f32-MFMA peak
# ALGORITHMIC bytes / flops per frame and per kernel (DESIGN.md "Kernels and rooflines").
# The pipeline total is SURVEY 8(d)'s 423,784 B/frame: 307,200 read + 115,560 card written
# + 1,024 result record.
ALGO = {
    #            bytes/frame                flop/frame
    "detect":   (307200 + 64,               2 * 2.2e6 + 1.2e6),
    "geometry": (64 + 152 + 80,             2.0e3),
    # the card quad covers ~427 x 269 source pixels (guide frame at ~1:1 scale): every one is read once,
    # every card pixel written once
    "warp":     (114863 + 115560,           1.2e6 + 0.9e6),
    "vseg":     (103 * 408 + 24,            2 * 103 * (204 * 50 + 150)),
    "hseg":     (27 * 428 + 48,             2.0e5),
    "digits":   (16 * 27 * 19 + 744,        16 * 3 * 2 * (8 * 360 * 9 + 320 * 32 + 320)),
    # expiry: rows below the number (~92 x 428 B) once for the line sums + 3 stripes x 23 rows;
    # the CNN is 4 digits x 1.27 M MAC per group, ~0.6 groups per frame on this corpus
    "expiry_seg": (92 * 428 + 3 * 23 * 428 + 1592, 1.5e5),
    "expiry_cat": (4 * 176 + 160,           0.6 * 4 * 2 * 1.27e6),
}
PIPELINE_BYTES = 307200 + 115560 + 1024 + 1592
# HBM traffic per frame (bytes) from the committed PMC passes profiles/r1_v4_pmc_{FETCH,WRITE}_SIZE_*.txt
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs, KB per dispatch / 4096 frames).
# FETCH_SIZE is NOT doubled: the guide's x2 gfx950 correction is calibrated for 16 B/lane
# streams, these kernels load 4 B/lane ("uncalibrated" there); WRITE_SIZE matched known byte
# counts exactly (k_synth_frames: 307,200 B/frame; k_warp: 112.9 KB vs 115,560 B written).
PMC_TRAFFIC = {
    "detect": (72059.0 + 124234.8 + 256 + 256) * 1024 / 4096,
    "geometry": (166.5 + 596.4 + 384 + 1536) * 1024 / 4096,
    "warp": (361754.3 + 441.5 + 462312.4 + 1413.0) * 1024 / 4096,  # k_warp + k_warp_windows
    "vseg": (258900.1 + 159331.5) * 1024 / 4096,  # includes the register-spill traffic of the 7-workgroup build
    "hseg": (32657.4 + 256.0) * 1024 / 4096,
    "digits": (24702.0 + 2902.4) * 1024 / 4096,
    "expiry_seg": (81005.4 + 62114.3 + 7104.4 + 7748.8) * 1024 / 4096,  # k_expiry_stripes + k_expiry_seg
    "expiry_cat": (5400.2 + 564.5) * 1024 / 4096,
}
# kernels of comparable size per stage timer: the detect stage is two launches (top/bottom boxes,
# left/right boxes); the other stages are one kernel (plus helpers below 1 % of the stage).  The
# roofline object is for the largest single kernel, so stages are ranked by time per launch.
LAUNCHES = {"detect": 2}


def cpu_baseline(orc_mod, frames, budget_s=12.0):
    """The CPU oracle (a port of the reference algorithm, oracle/*.c) timed on ONE host
    core over a bounded sample of the same frames."""
    o = orc_mod.Oracle()
    t0 = time.perf_counter()
    done = 0
    for f in frames:
        res, card = o.scan_frame(f, want_card=True)
        o.scan_card_expiry(card, res)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the benchmark's synthetic 640x480 frames, full pipeline incl. expiry, 1 thread, %.1f s"
                      % (done, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")  # RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    pkg = entry.load_package()
    from dmz_amd import sharding
    ctx = pkg.Context(dev.index)  # raises without the HIP library / a GPU: no fallback
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    B = args.batch
    frames = torch.empty((B, pkg.FRAME_H, pkg.FRAME_W), dtype=torch.uint8, device=dev)
    cards = torch.empty((B, pkg.CARD_H, pkg.CARD_W), dtype=torch.uint8, device=dev)
    XB = pkg.EXPIRY_DTYPE.itemsize
    nbuf = 2 if world > 1 else 1  # alternate record buffers so that a gather can overlap the next step
    results_b = [torch.zeros((B, 1024), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    expiry_b = [torch.zeros((B, XB), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    results, expiry = results_b[0], expiry_b[0]
    gatherer = sharding.RootGatherer(world) if world > 1 else None
    # every rank scans its own contiguous slice of the synthetic corpus (weak scaling:
    # the corpus is world*B frames, rank g owns [g*B, (g+1)*B))
    lo, hi = sharding.shard_range(world * B, rank, world)
    assert hi - lo == B
    ctx.synth_frames(SEED, lo, B, frames)
    torch.cuda.synchronize(dev)

    step_no = [0]

    def step():
        k = step_no[0] % nbuf
        step_no[0] += 1
        if world > 1:
            gatherer.wait(slots=(2 * k, 2 * k + 1))  # only the gathers that still read this pair of buffers
        ctx.pipeline_expiry(frames, B, results_b[k], expiry_b[k], cards)
        if world > 1:
            gatherer.submit(results_b[k], slot=2 * k)
            gatherer.submit(expiry_b[k], slot=2 * k + 1)

    for _ in range(args.warmup):
        step()
    if world > 1:
        gatherer.wait()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    if world > 1:
        gatherer.wait()  # the timed region includes the last exchange
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dev_ms = ev0.elapsed_time(ev1)

    # per-kernel durations with hipEvents on the launch stream (untimed extra steps)
    ctx.set_profiling(True)
    ctx.stage_times(reset=True)
    prof_steps = 2
    for _ in range(prof_steps):
        ctx.pipeline_expiry(frames, B, results, expiry, cards)
    stage = ctx.stage_times(reset=True)
    ctx.set_profiling(False)

    if rank == 0:
        res = results.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(-1)
        gates = {
            "found_all": float((res["found_all"] != 0).mean()),
            "vseg_ok": float(((res["flags"] & pkg.FLAG_VSEG_OK) != 0).mean()),
            "usable": float(((res["flags"] & pkg.FLAG_USABLE) != 0).mean()),
        }
        ex = expiry.cpu().numpy().view(pkg.EXPIRY_DTYPE).reshape(-1)
        gates["expiry_group_found"] = float((ex["n_found"] > 0).mean())
        gates["expiry_categorised"] = float(((ex["categorised"] != 0) & (ex["n_groups"] > 0)).mean())
        gates["expiry_groups_per_frame"] = float(ex["n_groups"].mean())
        per_stage = {}
        for name, (ms, cnt) in stage.items():
            if cnt == 0:
                continue
            avg_ms = ms / prof_steps  # all launches of the stage in one step
            by, fl = ALGO[name]
            per_stage[name] = {
                "ms_per_step": round(avg_ms, 4),
                "GBps": round(by * B / (avg_ms * 1e-3) / 1e9, 2),
                "TFLOPs": round(fl * B / (avg_ms * 1e-3) / 1e12, 3),
            }
        dom = max(per_stage, key=lambda k: per_stage[k]["ms_per_step"] / LAUNCHES.get(k, 1))
        hbm_frac = per_stage[dom]["GBps"] / HBM_PEAK_GBPS
        fl_frac = per_stage[dom]["TFLOPs"] / FP32_PEAK_TFLOPS
        if hbm_frac >= fl_frac:
            roof = {"bound": "hbm", "achieved": per_stage[dom]["GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(hbm_frac, 5), "traffic": None}
        else:
            roof = {"bound": "mfma", "achieved": per_stage[dom]["TFLOPs"], "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fl_frac, 5), "traffic": None}
        roof["kernel"] = dom
        roof["launch_ms"] = round(per_stage[dom]["ms_per_step"] / LAUNCHES.get(dom, 1), 4)
        # bytes per launch, from profiles/ (see PMC_TRAFFIC)
        roof["traffic"] = round(PMC_TRAFFIC[dom] * B / LAUNCHES.get(dom, 1)) if PMC_TRAFFIC[dom] is not None else None
        value = world * B * args.steps / elapsed
        roof["pipeline_GBps"] = round(value / world * PIPELINE_BYTES / 1e9, 2)
        roof["pipeline_frac_of_hbm"] = round(value / world * PIPELINE_BYTES / 1e9 / HBM_PEAK_GBPS, 5)

        out = {
            "metric": "frames/sec full scan pipeline (640x480)",
            "value": round(value, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (int32 accumulators; f32 model scores; f64 warp coordinates)",
            "data": "synthetic",
            "config": {
                "workload": "full pipeline detect->warp->vseg->hseg->digits->expiry (BASELINE configs[3]), "
                            "%d synthetic 640x480 Y frames per GPU resident in HBM" % B,
                "frames_per_gpu": B,
                "parallelism": "frame-sharded x%d, asynchronous gather of the 1 KiB result + 1.6 KiB expiry records on rank 0" % world,
                "gate_pass_rates": gates,
                "device_ms_per_step": round(dev_ms / args.steps, 3),
            },
            "roofline": roof,
            "stages": per_stage,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = frames[:2048].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(entry.load_oracle(), sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
