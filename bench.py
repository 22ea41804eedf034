#!/usr/bin/env python3
"""Benchmark of the card.io-dmz scan hot path on MI355X.

A "step" = one pass of the hot path over one HBM-resident batch of synthetic input.  The default
workload is BASELINE.json configs[3] (`--config 4` in SURVEY 8(d)'s numbering): the full per-frame
pipeline (detect edges -> rectify card -> number-row search -> digit segmentation -> digit
categorisation -> expiry segmentation + categorisation) over 65 536 synthetic 640x480 luma frames.
`value` = frames/s of the whole job (all ranks), inputs resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}] [--batch B]

  --config 2   BASELINE configs[1]: card-edge detection only, 4 096 frames        (dmz_hip_detect_batch)
  --config 3   BASELINE configs[2]: 65 536 pre-warped 428x270 crops, vseg + hseg + digit CNNs
                                                                                (dmz_hip_scan_cards_batch)
  --config 4   BASELINE configs[3]: full pipeline incl. expiry, 65 536 frames    (dmz_hip_pipeline_expiry_batch)

--gpus N > 1 (BASELINE configs[4]): one process per GPU.  Under torch.distributed.run (WORLD_SIZE set)
the script is a rank; run plainly it starts the N ranks itself -- as a CHILD `python -m
torch.distributed.run ...` created before this process has touched the GPU -- and exits with the
child's code.  Frames are sharded by contiguous ranges (weak scaling: B frames per GPU; at N > 1 the
default B is 131 072, i.e. the 1 048 576-frame corpus of configs[4] at N = 8) and the fixed-size
result / expiry records of each step are gathered on rank 0 (RCCL sends over xGMI), asynchronously:
the exchange of step k overlaps the kernels of step k+1 (two alternating record buffers), and the
timed region ends only when the last gather has completed.

--dry-run: no device; the ranks run the same shard / step / gather / timing code on CPU tensors
over gloo with records filled by a rank-and-step pattern, and rank 0 checks what it gathered.  It
exists so that the N > 1 launch and exchange logic is covered by the CPU test suite.
"""
import argparse
import glob
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

SEED = 0xCA4D10
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # vector == f32-MFMA peak
# VALU issue roof, in the counters' own units (no clock assumption).  SQ_ACTIVE_INST_VALU counts one quad-cycle per wave64
# VALU instruction (two for transcendentals), SQ_BUSY_CU_CYCLES the quad-cycles a CU had work.  Calibrated on saturated
# single-opcode loops (tools/dev/valu_counter_calibration.sh, profiles/r4_valu_counter_calibration.txt): the ratio of the
# two reaches 1.807 for the full-rate class (v_mov / add / sub / and / or / xor / ashr, f32 add / mul / fma: a wave64
# instruction every 2 cycles per SIMD) and 0.966 for the half-rate class (everything else incl. fp64, dot4, packed 16-bit,
# conversions, compares, DPP / SDWA forms: one every 4 cycles; profiles/r4_valu_table_gfx950.txt).  `valu_issue_frac` is
# the ratio over the full-rate ceiling, so it can never exceed 1; a kernel made of half-rate instructions saturates at
# 0.966 / 1.807 = 0.535, which `valu_halfrate_saturation` (ratio / 0.966, meaningful for such kernels only) shows.
VALU_RATIO_FULL_RATE = 1.807
VALU_RATIO_HALF_RATE = 0.966
# ALGORITHMIC bytes / flops per frame and per kernel.
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
# kernels behind each stage timer (rocprofv3 names, template arguments stripped); the first one is
# the stage's main kernel, the others are helpers below 1 % of the stage
STAGE_KERNELS = {
    "detect": ("k_detect_walk",), "geometry": ("k_homography", "k_geometry"),
    "warp": ("k_warp", "k_warp_windows"), "vseg": ("k_vseg",), "hseg": ("k_hseg",),
    "digits": ("k_digits", "k_digit_patches"), "expiry_seg": ("k_expiry_seg", "k_expiry_stripes", "k_expiry_slash"),
    "expiry_cat": ("k_expiry_cat",),
}
# algorithmic bytes per unit of each configuration
CONFIGS = {
    2: dict(name="configs[1]", unit="frames", batch=4096, bytes=307200 + 80,
            metric="frames/sec card-edge detection only (640x480)",
            workload="detect only (Sobel-7 + adaptive Canny + gated Hough + corner geometry), BASELINE configs[1]",
            stages=("detect", "geometry")),
    3: dict(name="configs[2]", unit="crops", batch=65536, bytes=115560 + 1024,
            metric="crops/sec digit pass on pre-warped 428x270 cards",
            workload="vseg + hseg + 3 digit CNNs on pre-warped 428x270 crops, BASELINE configs[2]",
            stages=("vseg", "hseg", "digits")),
    4: dict(name="configs[3]", unit="frames", batch=65536, bytes=307200 + 115560 + 1024,
            metric="frames/sec full scan pipeline (640x480)",
            workload="full pipeline detect->warp->vseg->hseg->digits->expiry, BASELINE configs[3]",
            stages=tuple(ALGO)),
}
# BASELINE configs[4]: the 1 048 576-frame corpus, frame-sharded.  Two ways to run it (--scaling):
#   weak   (default; what the driver's N = 1, 2, 4, 8 runs compare): the SAME per-GPU batch at every N -- configs[3]'s 65 536
#          frames per GPU per step -- so value(N) / value(1) is like for like; the corpus is N x 65 536 frames per step;
#   strong the whole corpus every step at every N: each rank owns corpus / N frames and takes them as (corpus / N) / 65 536
#          passes over its resident 65 536-frame batch (one GPU: sixteen passes).
CORPUS_FRAMES = 1048576


# ---------------------------------------------------------------------------------------------
# measured HBM traffic per kernel, from the committed PMC passes in profiles/
# ---------------------------------------------------------------------------------------------
def load_pmc_traffic():
    """{kernel base name: (fetch bytes, write bytes) per frame} from the PMC summaries named by
    profiles/CURRENT (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, KB per dispatch over
    a batch given in the file name).  FETCH_SIZE is doubled: on gfx950 it tallies 128-byte requests
    at 64 bytes (MI355X_MICROARCH.md, HBM section); k_detect_walk, whose reads are known exactly
    (rows x boxes x 128-byte lines), calibrates the doubled figure to 0.2 %.  WRITE_SIZE matches
    known byte counts as reported (k_synth_frames: 307 200 B per frame)."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        tag = open(os.path.join(pdir, "CURRENT")).read().split()[0]
    except OSError:
        return None, None
    out = {}
    for counter, slot, mul in (("FETCH_SIZE", 0, 2.0), ("WRITE_SIZE", 1, 1.0)):
        files = glob.glob(os.path.join(pdir, "%s_pmc_%s_batch*.txt" % (tag, counter)))
        if not files:
            return None, tag
        m = re.search(r"batch(\d+)\.txt$", files[0])
        b0 = int(m.group(1))
        for line in open(files[0]):
            if line.startswith("#") or "dispatches=" not in line:
                continue
            name = line[:line.index("dispatches=")].strip().split("<")[0].split("(")[0]
            mean_kb = float(re.search(r"mean=([0-9.]+)", line).group(1))
            ent = out.setdefault(name, [0.0, 0.0])
            ent[slot] += mul * mean_kb * 1024.0 / b0  # template instances of one kernel add up
    return out, tag


def load_pmc_valu():
    """{kernel base name: wave64 VALU instructions per frame} from the committed SQ_INSTS_VALU pass named by
    profiles/CURRENT (rocprofv3 --pmc, kernel-trace only, tools/profile_round.sh; template instances add up)."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        tag = open(os.path.join(pdir, "CURRENT")).read().split()[0]
    except OSError:
        return None
    files = glob.glob(os.path.join(pdir, "%s_pmc_SQ_insts_batch*.txt" % tag))
    if not files:
        return None
    b0 = int(re.search(r"batch(\d+)\.txt$", files[0]).group(1))
    out, col = {}, None
    for line in open(files[0]):
        if line.startswith("#"):
            continue
        toks = line.split()
        if toks and toks[0] == "kernel":
            col = toks[1:].index("SQ_INSTS_VALU") - len(toks[1:])  # counted from the right: names may contain blanks
            continue
        if col is None or len(toks) < 2:
            continue
        try:
            v = float(toks[col])
        except ValueError:
            continue
        name = line[:34].strip().split("<")[0].split("(")[0]
        out[name] = out.get(name, 0.0) + v / b0
    return out


def stage_valu(valu, stage):
    if not valu:
        return None
    hit = [valu[k] for k in STAGE_KERNELS[stage] if k in valu]
    return sum(hit) if hit else None


def load_pmc_issue():
    """{kernel base name: {counter: sum over the kernel's dispatches of one pipeline pass}} from the committed SQ issue
    pass named by profiles/CURRENT (SQ_BUSY_CU_CYCLES, SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_BUSY_CYCLES, ...; rocprofv3
    --pmc, kernel-trace only, tools/profile_round.sh; template instances add up)."""
    pdir = os.path.join(ROOT, "profiles")
    try:
        tag = open(os.path.join(pdir, "CURRENT")).read().split()[0]
    except OSError:
        return None
    files = glob.glob(os.path.join(pdir, "%s_pmc_SQ_issue_batch*.txt" % tag))
    if not files:
        return None
    out, cols = {}, None
    for line in open(files[0]):
        if line.startswith("#"):
            continue
        toks = line.split()
        if toks and toks[0] == "kernel":
            cols = toks[1:]
            continue
        if cols is None or len(toks) < len(cols) + 1:
            continue
        try:
            vals = [float(t) for t in toks[-len(cols):]]  # counted from the right: names may contain blanks
        except ValueError:
            continue
        name = line[:34].strip().split("<")[0].split("(")[0]
        ent = out.setdefault(name, dict.fromkeys(cols, 0.0))
        for c, v in zip(cols, vals):
            ent[c] += v
    return out


def stage_issue(issue, stage):
    """VALU / matrix-pipe occupancy of the kernels behind a stage timer from the counters: dict or None.
    valu_issue_frac = (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES) / 1.807 (share of the full-rate issue ceiling, <= 1 by
    calibration); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) (cycles over quad-cycles)."""
    if not issue:
        return None
    hit = [issue[k] for k in STAGE_KERNELS[stage] if k in issue]
    busy = sum(h.get("SQ_BUSY_CU_CYCLES", 0.0) for h in hit)
    if not hit or busy <= 0:
        return None
    act = sum(h.get("SQ_ACTIVE_INST_VALU", 0.0) for h in hit)
    mfma = sum(h.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for h in hit)
    return {"valu_active_per_busy": round(act / busy, 4),
            "valu_issue_frac": round(act / busy / VALU_RATIO_FULL_RATE, 4),
            "valu_halfrate_saturation": round(min(1.0, act / busy / VALU_RATIO_HALF_RATE), 4),
            "mfma_busy_frac": round(mfma / (4.0 * busy), 4),
            "_act": act, "_busy": busy, "_mfma": mfma}


def stage_traffic(pmc, stage):
    """(fetch, write) bytes per frame of all kernels behind a stage timer, or None."""
    if not pmc:
        return None
    f = w = 0.0
    hit = False
    for k in STAGE_KERNELS[stage]:
        if k in pmc:
            hit = True
            f += pmc[k][0]
            w += pmc[k][1]
    return (f, w) if hit else None


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a port of the reference algorithm, oracle/*.c), one process per core
# ---------------------------------------------------------------------------------------------
_cpu_barrier = None


def _cpu_init(barrier):
    global _cpu_barrier
    _cpu_barrier = barrier


def _cpu_worker(job):
    wid, config, first, nframes, budget_s = job
    orc = entry.load_oracle()
    o = orc.Oracle()
    if config == 3:
        items = [o.synth_card(SEED, first + i)[0] for i in range(nframes)]
    else:
        items = [o.synth_frame(SEED, first + i)[0] for i in range(nframes)]
    _cpu_barrier.wait()
    t0 = time.perf_counter()
    c0 = time.process_time()
    done = 0
    while True:
        it = items[done % nframes]
        if config == 2:
            o.detect_edges(it)
        elif config == 3:
            o.scan_card_image(it)
        else:
            res, card = o.scan_frame(it, want_card=True)
            o.scan_card_expiry(card, res)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0, time.process_time() - c0


def cpu_quota():
    """(cores the cgroup lets this process use at once or None, where that was read): cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`.  A container with 256 visible cores and a 32-core quota runs 256 busy workers at an
    eighth of a core each -- the per-core figure of such a run says nothing about the CPU."""
    try:
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[1] in ("", "cpu,cpuacct", "cpu"):
                rel = parts[2]
        cands = []
        for base in ("/sys/fs/cgroup" + rel, "/sys/fs/cgroup"):
            cands.append((os.path.join(base, "cpu.max"), None))
        for base in ("/sys/fs/cgroup/cpu,cpuacct" + rel, "/sys/fs/cgroup/cpu" + rel, "/sys/fs/cgroup/cpu,cpuacct", "/sys/fs/cgroup/cpu"):
            cands.append((os.path.join(base, "cpu.cfs_quota_us"), os.path.join(base, "cpu.cfs_period_us")))
        for q, per in cands:
            if not os.path.exists(q):
                continue
            if per is None:
                a, b = open(q).read().split()[:2]
                if a == "max":
                    continue  # (no limit at this level: look further up)
                return float(a) / float(b), q
            quota, period = int(open(q).read()), int(open(per).read())
            if quota > 0 and period > 0:
                return quota / period, q
    except (OSError, ValueError):
        pass
    return None, None


def cpu_baseline(config, budget_s=3.0, frames_per_worker=48):
    """Runs BEFORE this process touches the GPU (the workers are forked).  One worker alone first (1 s: the unloaded
    single-process figure), then min(affinity, cgroup quota) workers at once."""
    import multiprocessing as mp
    affinity = len(os.sched_getaffinity(0))
    quota, quota_src = cpu_quota()
    cores = max(1, min(affinity, int(quota) if quota is not None and quota >= 1 else affinity))
    ctx = mp.get_context("fork")
    with ctx.Pool(1, initializer=_cpu_init, initargs=(ctx.Barrier(1),)) as pool:
        d1, t1, _ = pool.map(_cpu_worker, [(0, config, 0, min(16, frames_per_worker), 1.0)], chunksize=1)[0]
    single = d1 / t1
    barrier = ctx.Barrier(cores)
    with ctx.Pool(cores, initializer=_cpu_init, initargs=(barrier,)) as pool:
        res = pool.map(_cpu_worker, [(w, config, w * frames_per_worker, frames_per_worker, budget_s)
                                     for w in range(cores)], chunksize=1)
    total = sum(d for d, _, _ in res)
    wall = max(t for _, t, _ in res)
    per_core = float(np.mean([d / t for d, t, _ in res]))
    cpu_s = sum(c for _, _, c in res)  # CPU time the workers were actually given (a throttled or shared host gives less than wall x cores)
    unit = CONFIGS[config]["unit"]
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(total / wall, 1), "unit": unit + "/s", "cores": cores, "kind": "port",
            "per_core": round(per_core, 1), "single_process_%s_per_s" % unit: round(single, 1),
            "cpu": model, "cpu_affinity": affinity,
            "cpu_quota": ({"cores": round(quota, 2), "source": quota_src} if quota is not None else None),
            "loaded_over_single": round(per_core / single, 3) if single > 0 else None,
            "effective_cores": round(cpu_s / wall, 1), "per_effective_core": round(total / cpu_s, 1) if cpu_s > 0 else None,
            "sample": "%d %s of the benchmark corpus per process (indices [%d w, %d w + %d), cycled), "
                      "%s, one process per usable host core (%d = min(affinity %d, cgroup quota %s)), %.1f s wall, %.0f s of CPU "
                      "work; before it ONE process alone for 1 s (single_process_*: the unloaded per-core figure)"
                      % (frames_per_worker, unit, frames_per_worker, frames_per_worker, frames_per_worker,
                         CONFIGS[config]["workload"], cores, affinity, "%.1f" % quota if quota is not None else "none",
                         wall, cpu_s)}


# ---------------------------------------------------------------------------------------------
# N > 1 run plainly: start the ranks as a child process before anything here touches the GPU
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def mixed_kind(idx):
    """kind of frame `idx` (an integer array / tensor of corpus indices) in the mixed corpus: 0-3 no card, 4 upside down,
    5-9 a card as generated"""
    return ((idx * 2654435761) % (1 << 32)) % 10


def make_mixed_corpus(torch, frames, first_index, seed):
    """In place: frame i of the corpus (global index first_index + j) by mixed_kind(i) --
    0-3: no card (dark noise only), 4: the card upside down (the frame rotated by 180 degrees), 5-9: as generated.
    tests/test_gpu_full_size.py checks frames of each kind, built by the same rule, against the oracle."""
    dev, n = frames.device, frames.shape[0]
    idx = torch.arange(first_index, first_index + n, device=dev, dtype=torch.int64)
    kind = mixed_kind(idx)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    for c0 in range(0, n, 4096):  # (in chunks: bounded temporaries)
        sl = slice(c0, min(n, c0 + 4096))
        k = kind[sl]
        noise = torch.randint(18, 58, frames[sl].shape, generator=g, device=dev, dtype=torch.uint8)
        frames[sl] = torch.where((k < 4)[:, None, None], noise, frames[sl])
        frames[sl] = torch.where((k == 4)[:, None, None], torch.flip(frames[sl], dims=(1, 2)), frames[sl])
    return kind


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="units (frames / crops) per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the configs[1] / configs[2] lines that follow the headline steps at N = 1 (`side_configs`)")
    ap.add_argument("--corpus", choices=("cards", "mixed"), default="cards",
                    help="cards: every frame shows a card (the metric's corpus); mixed: 40 %% card-less, 10 %% upside-down, "
                         "50 %% cards (config 4 only; a second line for the gated throughput, not the headline metric)")
    ap.add_argument("--gather", choices=("auto", "capi", "torch"), default="auto",
                    help="N > 1: gather of the records on rank 0.  auto (default): through the C-ABI (dmz_hip_gather_records: "
                         "RCCL send / recv on the context's communication queue) when librccl loads on every rank and the "
                         "communicator comes up, VERIFIED after the timed loop against a torch.distributed gather of the same "
                         "records; if the communicator fails or a byte differs the timed loop is REPEATED over the "
                         "torch.distributed gather and the JSON line says so (config.gather = \"torch (capi failed: ...)\").  "
                         "capi: the same, but a failure ends the run.  torch: torch.distributed only.")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the same per-GPU batch at every N (like-for-like curve); strong: the 1 048 576-frame corpus of "
                         "BASELINE configs[4] every step at every N (config 4 only)")
    ap.add_argument("--dry-run", action="store_true", help="no device: shard / gather / timing logic on CPU over gloo")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # nothing above has initialised the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d; run `python bench.py --gpus %d` (it starts its own ranks) or "
                 "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py "
                 "--gpus %d ...`" % (args.gpus, world, args.gpus, args.gpus, args.gpus))

    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]  # resident units per GPU: the same at every N
    passes = 1                      # passes over the resident batch per step
    if args.scaling == "strong":
        if args.config != 4:
            sys.exit("bench.py: --scaling strong goes with --config 4")
        corpus = CORPUS_FRAMES if not args.batch else 16 * args.batch  # (a small corpus for the CPU dry run)
        if corpus % (world * B):
            sys.exit("bench.py: the corpus (%d frames) is not a whole number of %d-frame batches on %d GPUs" % (corpus, B, world))
        passes = corpus // (world * B)

    cpu = None
    if world == 1 and not args.no_cpu_baseline and not args.dry_run:
        cpu = cpu_baseline(args.config)  # forks: must precede the first GPU call of this process

    import torch
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl")  # RCCL on ROCm
    pkg = entry.load_package()
    from dmz_amd import sharding

    XB = pkg.EXPIRY_DTYPE.itemsize
    with_expiry = args.config == 4
    if args.dry_run:
        dev = torch.device("cpu")
        ctx = None
    else:
        torch.cuda.set_device(local_rank if world > 1 else 0)
        dev = torch.device("cuda", local_rank if world > 1 else 0)
        ctx = pkg.Context(dev.index)  # raises without the HIP library / a GPU: no fallback
        stream = torch.cuda.current_stream(dev)
        ctx.set_stream(stream.cuda_stream)

    nbuf = 2 if world > 1 else 1  # alternate record buffers so that a gather can overlap the next step
    results_b = [torch.zeros((B, 1024), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    expiry_b = [torch.zeros((B, XB), dtype=torch.uint8, device=dev) for _ in range(nbuf)] if with_expiry else None
    gatherer = sharding.RootGatherer(world) if world > 1 else None
    # N > 1: the records travel through the C-ABI's gather (dmz_hip_gather_records: ncclSend / ncclRecv into the root on
    # the context's communication queue) when librccl loads and the communicator comes up; the torch.distributed gather
    # (same ranges, same asynchrony) otherwise.  At N = 1 there is nothing to gather.
    use_capi = False
    comm_init_stuck = False  # a helper thread is still inside ncclCommInitRank: leave through os._exit at the end
    capi_failure = None  # why the C-ABI gather is not the one that was timed (auto mode), for the JSON line
    root_dst = None
    if world > 1 and ctx is not None and args.gather in ("auto", "capi"):
        # every rank first agrees that librccl loads everywhere: a rank that entered ncclCommInitRank alone would hang the others
        have = torch.ones(1, dtype=torch.int32, device=dev)
        try:
            probe_uid = pkg.comm_unique_id()  # loads librccl in this process
        except pkg.DmzHipError:
            probe_uid = None
            have.zero_()
        dist.all_reduce(have, op=dist.ReduceOp.MIN)
        uid = [probe_uid if rank == 0 else None]
        ok = torch.zeros(1, dtype=torch.int32, device=dev)
        if bool(have.item()):
            dist.broadcast_object_list(uid, src=0)
            # ncclCommInitRank is collective: a rank that cannot reach its peers would wait in it for ever, and so would the
            # whole run.  It runs on a helper thread with a deadline; the ranks then agree over torch.distributed whether every
            # communicator came up in time (a thread still stuck in the call is abandoned: the process exits through os._exit).
            import threading
            init_err = []

            def _init():
                try:
                    ctx.comm_init(world, rank, uid[0])
                except pkg.DmzHipError as e:
                    init_err.append(str(e))

            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("DMZ_BENCH_COMM_TIMEOUT_S", "120")))
            if th.is_alive():
                comm_init_stuck = True
                print("bench.py: rank %d: ncclCommInitRank did not return in time" % rank, file=sys.stderr)
            elif init_err:
                print("bench.py: rank %d: C-ABI communicator failed (%s)" % (rank, init_err[0]), file=sys.stderr)
            else:
                ok += 1
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not bool(ok.item()):
                capi_failure = "ncclCommInitRank failed or did not return within its deadline on a rank"
        else:
            capi_failure = "librccl did not load on every rank"
        use_capi = bool(ok.item())
        if not use_capi:
            if args.gather == "capi":
                sys.exit("bench.py: --gather capi: no RCCL communicator (%s)" % capi_failure)
            if not comm_init_stuck:
                try:
                    ctx.comm_destroy()
                except pkg.DmzHipError:
                    pass
        elif rank == 0:
            root_dst = [(torch.empty((world * B, 1024), dtype=torch.uint8, device=dev),
                         torch.empty((world * B, XB), dtype=torch.uint8, device=dev) if with_expiry else None)
                        for _ in range(nbuf)]
    # every rank scans its own contiguous slice of the synthetic corpus (weak scaling:
    # the corpus is world*B units, rank g owns [g*B, (g+1)*B))
    lo, hi = sharding.shard_range(world * B, rank, world)
    assert hi - lo == B
    frames = cards = None
    if ctx is not None:
        if args.config != 3:
            frames = torch.empty((B, pkg.FRAME_H, pkg.FRAME_W), dtype=torch.uint8, device=dev)
            ctx.synth_frames(SEED, lo, B, frames)
        if args.config != 2:
            cards = torch.empty((B, pkg.CARD_H, pkg.CARD_W), dtype=torch.uint8, device=dev)
        if args.config == 3:
            ctx.synth_cards(SEED, lo, B, cards)
        torch.cuda.synchronize(dev)
        if args.corpus == "mixed":
            if args.config != 4:
                sys.exit("bench.py: --corpus mixed goes with --config 4")
            make_mixed_corpus(torch, frames, lo, SEED + 17 + rank)
            torch.cuda.synchronize(dev)

    def hot_path(k):
        if ctx is None:  # dry run: a pattern rank 0 can verify after the gather
            results_b[k].fill_((rank * 16 + step_no[0]) & 255)
            if with_expiry:
                expiry_b[k].fill_((rank * 16 + step_no[0] + 7) & 255)
        elif args.config == 2:
            ctx.detect(frames, B, results_b[k])
        elif args.config == 3:
            ctx.scan_cards(cards, B, results_b[k], only_warped=False)
        else:
            ctx.pipeline_expiry(frames, B, results_b[k], expiry_b[k], cards)

    step_no = [0]
    gathered = [None] * (2 * nbuf)

    def step():
        k = step_no[0] % nbuf
        step_no[0] += 1
        if world > 1 and use_capi:  # (read at call time: the auto mode may turn it off after the verification)
            # only the gathers that still read this pair of buffers (slots 2k, 2k + 1): the other pair's stay in flight
            ctx.gather_wait(2 * k, host_sync=False)
            ctx.gather_wait(2 * k + 1, host_sync=False)
            hot_path(k)
            ctx.gather_records(results_b[k], 1024, world * B, 0, root_dst[k][0] if rank == 0 else None, slot=2 * k)
            if with_expiry:
                ctx.gather_records(expiry_b[k], XB, world * B, 0, root_dst[k][1] if rank == 0 else None, slot=2 * k + 1)
            return
        if world > 1:
            gatherer.wait(slots=(2 * k, 2 * k + 1))  # only the gathers that still read this pair of buffers
        hot_path(k)
        if world > 1:
            gathered[2 * k] = gatherer.submit(results_b[k], slot=2 * k)
            if with_expiry:
                gathered[2 * k + 1] = gatherer.submit(expiry_b[k], slot=2 * k + 1)

    def sync():
        if ctx is not None:
            torch.cuda.synchronize(dev)

    def gather_drain():
        if world > 1 and use_capi:
            ctx.gather_wait(-1, host_sync=True)
        elif world > 1:
            gatherer.wait()

    one_pass = step

    def step():  # strong scaling: a rank's share of the corpus is `passes` passes over its resident batch
        for _ in range(passes):
            one_pass()

    def timed_run():
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        for _ in range(args.warmup):
            step()
        gather_drain()
        sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if ctx is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(stream)
        for _ in range(args.steps):
            step()
        gather_drain()  # the timed region includes the last exchange
        if ctx is not None:
            ev1.record(stream)
        sync()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, (ev0.elapsed_time(ev1) if ctx is not None else el * 1e3)

    elapsed, dev_ms = timed_run()

    # The C-ABI gather has no second implementation inside the timed loop to compare with: afterwards the LAST step's records
    # travel once more through torch.distributed and rank 0 compares the two destinations byte for byte.  A wrong offset or
    # count, or a second RCCL beside torch's, must not produce a plausible frames/s figure: with --gather capi the run fails;
    # with auto the timed loop is repeated over the torch.distributed gather and the JSON line names the reason.
    if world > 1 and use_capi:
        last = (step_no[0] - 1) % nbuf
        ref = sharding.RootGatherer(world)
        want_r = ref.submit(results_b[last], slot=0)
        want_x = ref.submit(expiry_b[last], slot=1) if with_expiry else None
        ref.wait()
        torch.cuda.synchronize(dev)
        same = torch.ones(1, dtype=torch.int32, device=dev)
        if rank == 0:
            if not torch.equal(root_dst[last][0].view(-1), want_r.view(-1)):
                same.zero_()
            if with_expiry and not torch.equal(root_dst[last][1].view(-1), want_x.view(-1)):
                same.zero_()
        dist.broadcast(same, src=0)
        if not bool(same.item()):
            if args.gather == "capi":
                sys.exit("bench.py: the C-ABI gather's records differ from the torch.distributed gather of the same step")
            capi_failure = "its records differed from the torch.distributed gather of the same step"
            use_capi = False
            try:
                ctx.comm_destroy()
            except pkg.DmzHipError:
                pass
            elapsed, dev_ms = timed_run()

    if args.dry_run:
        ok = True
        if rank == 0 and world > 1:
            last = (step_no[0] - 1) % nbuf
            got = gathered[2 * last].view(world, B, 1024)
            for g in range(world):
                ok = ok and bool((got[g] == ((g * 16 + step_no[0]) & 255)).all())
            if with_expiry:
                gx = gathered[2 * last + 1].view(world, B, XB)
                for g in range(world):
                    ok = ok and bool((gx[g] == ((g * 16 + step_no[0] + 7) & 255)).all())
        if rank == 0:
            print(json.dumps({"metric": cfg["metric"], "dry_run": True, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "frames_per_gpu": B, "units_per_gpu": B * passes,
                              "corpus_frames": world * B * passes, "scaling": args.scaling, "gather_ok": ok,
                              "shard": [lo, hi], "backend": "gloo" if world > 1 else "none"}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(0 if ok else 1)

    # BASELINE configs[1] and configs[2] under the same clock (N = 1, default workload only): after the headline steps, 20
    # steps each of the detect-only entry on the first 4 096 frames and of the digit pass on 65 536 synthetic pre-warped crops
    # (generated into the card buffer, which the headline steps no longer need), same context, same timing discipline.
    side = None
    if world == 1 and args.config == 4 and args.corpus == "cards" and B >= CONFIGS[2]["batch"] and not args.no_side_configs:
        side = {}
        side_res = torch.zeros((B, 1024), dtype=torch.uint8, device=dev)
        ctx.synth_cards(SEED, lo, B, cards)
        for cid, call, units in ((2, lambda: ctx.detect(frames, CONFIGS[2]["batch"], side_res), CONFIGS[2]["batch"]),
                                 (3, lambda: ctx.scan_cards(cards, B, side_res, only_warped=False), B)):
            for _ in range(2):
                call()
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = time.perf_counter()
            e0.record(stream)
            for _ in range(20):
                call()
            e1.record(stream)
            sync()
            wall = time.perf_counter() - ts
            c = CONFIGS[cid]
            rate = units * 20 / wall
            side[c["name"]] = {"metric": c["metric"], "value": round(rate, 1), "unit": c["unit"] + "/s", "units_per_step": units,
                               "steps": 20, "ms_per_step": round(wall / 20 * 1e3, 4), "device_ms_per_step": round(e0.elapsed_time(e1) / 20, 4),
                               "algorithmic_bytes_per_unit": c["bytes"],
                               "frac_of_hbm": round(rate * c["bytes"] / 1e9 / HBM_PEAK_GBPS, 5)}
        del side_res

    # per-kernel durations with hipEvents on the launch stream (untimed extra steps)
    ctx.set_profiling(True)
    ctx.stage_times(reset=True)
    prof_steps = 2
    for _ in range(prof_steps):
        hot_path(0)
    stage = ctx.stage_times(reset=True)
    # BASELINE configs[3] "bf16 conv with fp32 parity check": the expiry CNN's convolutions in their arithmetic
    # variants on the same frames (untimed extra steps): stage time, score difference and digit agreement
    # against the fp32 variant.  The timed region above ran the default (F16X3: 16-bit matrix core, split operands).
    variants = None
    if with_expiry and rank == 0:
        variants = {}
        ref_scores = None
        for name, mode in (("f32", pkg.EXPIRY_CONV_F32), ("f16x3", pkg.EXPIRY_CONV_F16X3), ("bf16x3", pkg.EXPIRY_CONV_BF16X3),
                           ("bf16", pkg.EXPIRY_CONV_BF16)):
            ctx.set_expiry_conv(mode)
            ctx.stage_times(reset=True)
            hot_path(0)
            ms = ctx.stage_times(reset=True)["expiry_cat"][0]
            ex = expiry_b[0].cpu().numpy().view(pkg.EXPIRY_DTYPE).reshape(-1)
            live = (np.arange(pkg.EXPIRY_MAX_GROUPS)[None, :] < ex["n_groups"][:, None]) & (ex["categorised"][:, None] != 0)
            sc = ex["groups"]["scores"][live]  # [groups, 4, 10]
            if ref_scores is None:
                ref_scores = sc
            variants[name] = {"expiry_cat_ms": round(ms, 4),
                              "max_abs_score_diff_vs_f32": float(np.abs(sc - ref_scores).max()) if sc.size else 0.0,
                              "digit_match_vs_f32": float((sc.argmax(-1) == ref_scores.argmax(-1)).mean()) if sc.size else 1.0,
                              "groups": int(sc.shape[0])}
        ctx.set_expiry_conv(pkg.EXPIRY_CONV_F16X3)
        hot_path(0)  # the records reported below come from the default variant again
    ctx.set_profiling(False)

    if rank == 0:
        results = results_b[0]
        res = results.cpu().numpy().view(pkg.RESULT_DTYPE).reshape(-1)
        gates = {"found_all": float((res["found_all"] != 0).mean())} if args.config != 3 else {}
        if args.config != 2:
            gates["vseg_ok"] = float(((res["flags"] & pkg.FLAG_VSEG_OK) != 0).mean())
            gates["upside_down"] = float(((res["flags"] & pkg.FLAG_UPSIDE_DOWN) != 0).mean())
            gates["usable"] = float(((res["flags"] & pkg.FLAG_USABLE) != 0).mean())
        if with_expiry:
            ex = expiry_b[0].cpu().numpy().view(pkg.EXPIRY_DTYPE).reshape(-1)
            gates["expiry_group_found"] = float((ex["n_found"] > 0).mean())
            gates["expiry_categorised"] = float(((ex["categorised"] != 0) & (ex["n_groups"] > 0)).mean())
            gates["expiry_groups_per_frame"] = float(ex["n_groups"].mean())
        pmc, pmc_tag = load_pmc_traffic()
        per_stage = {}
        for name, (ms, cnt) in stage.items():
            if cnt == 0:
                continue
            avg_ms = ms / prof_steps  # all launches of the stage in one step
            by, fl = ALGO[name]
            ent = {"ms_per_step": round(avg_ms, 4),
                   "GBps": round(by * B / (avg_ms * 1e-3) / 1e9, 2),
                   "TFLOPs": round(fl * B / (avg_ms * 1e-3) / 1e12, 3)}
            tr = stage_traffic(pmc, name)
            if tr is not None:
                ent["hbm_fetch_B_per_unit"] = round(tr[0])
                ent["hbm_write_B_per_unit"] = round(tr[1])
                ent["traffic_over_algorithmic"] = round((tr[0] + tr[1]) / by, 3)
            per_stage[name] = ent
        valu = load_pmc_valu()
        issue = load_pmc_issue()
        for name, ent in per_stage.items():
            sv = stage_valu(valu, name)
            if sv is not None:
                ent["valu_instructions_per_unit"] = round(sv)
            si = stage_issue(issue, name)
            if si is not None:
                ent.update({k: v for k, v in si.items() if not k.startswith("_")})
            # stages whose matrix work runs on bf16 / f16 operand splits (vseg, slash MLP, digit and expiry convolutions): the
            # flop figure counts the fp32 product they reproduce, not the instructions issued -- it is NOT divided by any peak;
            # the matrix pipe's occupancy is mfma_busy_frac (counters)
            ent["TFLOPs_is"] = "fp32-equivalent algorithmic flops (information; no fraction of a peak is derived from it)"
        dom = max(per_stage, key=lambda k: per_stage[k]["ms_per_step"])
        hbm_frac = per_stage[dom]["GBps"] / HBM_PEAK_GBPS
        mf_frac = per_stage[dom].get("mfma_busy_frac", 0.0)
        vi_frac = per_stage[dom].get("valu_issue_frac", 0.0)
        if hbm_frac >= mf_frac:
            roof = {"bound": "hbm", "achieved": per_stage[dom]["GBps"], "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(hbm_frac, 5), "traffic": None}
        else:
            # matrix-bound kernel: achieved = the fp32 product it reproduces per second, peak = the f32 matrix peak, and frac
            # = the matrix pipe's measured busy share (split-operand kernels issue more, narrower instructions than the
            # flop figure counts, so achieved / peak would not be a fraction of anything)
            roof = {"bound": "mfma", "achieved": per_stage[dom]["TFLOPs"], "peak": FP32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(mf_frac, 5), "traffic": None,
                    "frac_is": "SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) of the committed PMC pass"}
        # The contract's two roofs (HBM bytes, matrix pipe) are reported as `achieved / peak / frac`; the roof that actually
        # binds these kernels is VALU instruction issue: reported beside them (`valu_issue`), and named in `bound` when its
        # saturation is the largest of the three.
        if per_stage[dom].get("valu_halfrate_saturation", 0.0) > max(hbm_frac, mf_frac):
            roof["bound_of_contract_roofs"] = roof["bound"]
            roof["bound"] = "valu_issue"
        roof["kernel"] = STAGE_KERNELS[dom][0]
        roof["launch_ms"] = per_stage[dom]["ms_per_step"]
        roof["algorithmic_bytes_per_launch"] = ALGO[dom][0] * B
        tr = stage_traffic(pmc, dom)
        if tr is not None:
            # HBM bytes per launch: FETCH_SIZE x 2 + WRITE_SIZE of the committed PMC passes, scaled to this batch
            roof["traffic"] = round((tr[0] + tr[1]) * B)
            roof["traffic_over_algorithmic"] = round((tr[0] + tr[1]) / ALGO[dom][0], 3)
            roof["traffic_source"] = "profiles/%s_pmc_{FETCH,WRITE}_SIZE_*.txt (FETCH_SIZE doubled: gfx950 correction)" % pmc_tag
        value = world * B * passes * args.steps / elapsed
        roof["pipeline_GBps"] = round(value / world * cfg["bytes"] / 1e9, 2)
        roof["pipeline_frac_of_hbm"] = round(value / world * cfg["bytes"] / 1e9 / HBM_PEAK_GBPS, 5)
        if pmc:
            tot = [stage_traffic(pmc, s) for s in per_stage]
            if all(t is not None for t in tot):
                roof["pipeline_traffic_B_per_unit"] = round(sum(t[0] + t[1] for t in tot))
                roof["pipeline_traffic_over_algorithmic"] = round(sum(t[0] + t[1] for t in tot) / cfg["bytes"], 3)
        if issue:
            si_all = [stage_issue(issue, s) for s in per_stage]
            if all(v is not None for v in si_all):
                act, busy = sum(v["_act"] for v in si_all), sum(v["_busy"] for v in si_all)
                sv_all = [stage_valu(valu, s) for s in per_stage] if valu else []
                roof["valu_issue"] = {
                    "active_per_busy": round(act / busy, 4), "frac": round(act / busy / VALU_RATIO_FULL_RATE, 4),
                    "halfrate_saturation": round(min(1.0, act / busy / VALU_RATIO_HALF_RATE), 4),
                    "instructions_per_unit": round(sum(sv_all)) if sv_all and all(v is not None for v in sv_all) else None,
                    "dominant_kernel": {"kernel": STAGE_KERNELS[dom][0],
                                        **{k: per_stage[dom].get(k) for k in ("valu_active_per_busy", "valu_issue_frac",
                                                                              "valu_halfrate_saturation", "mfma_busy_frac")}},
                    "unit": "SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES, summed over the pipeline's kernels; frac = that over 1.807 "
                            "(the full-rate class's saturated ratio), halfrate_saturation = over 0.966 (the half-rate class's)",
                    "source": "profiles/%s_pmc_SQ_issue_*.txt; calibration profiles/r4_valu_counter_calibration.txt" % pmc_tag}
        out = {
            "metric": cfg["metric"],
            "value": round(value, 1),
            "unit": cfg["unit"] + "/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u8 (int32 accumulators; f32 model scores; f64 warp coordinates)",
            "data": "synthetic",
            "config": {
                "workload": "%s, %d synthetic %s per GPU resident in HBM"
                            % (cfg["workload"], B, "640x480 Y frames" if args.config != 3 else "428x270 card crops"),
                "baseline_config": cfg["name"],
                **({"corpus": "mixed: 40 % card-less frames, 10 % upside-down cards, 50 % cards (not the metric's corpus: the gated "
                              "throughput line of SURVEY section 7)"} if args.corpus == "mixed" else {}),
                **({"gather": "capi (dmz_hip_gather_records: RCCL send / recv; verified against torch.distributed after the timed loop)"
                              if use_capi else ("torch (capi failed: %s)" % capi_failure if capi_failure else "torch")} if world > 1 else {}),
                "units_per_gpu": B * passes,
                "corpus_frames": world * B * passes,
                **({"resident_batch": B, "passes_per_step": passes} if passes > 1 else {}),
                "algorithmic_bytes_per_unit": cfg["bytes"],
                "parallelism": "frame-sharded x%d, asynchronous gather of the 1 KiB result%s records on rank 0"
                               % (world, " + 1.6 KiB expiry" if with_expiry else ""),
                "gate_pass_rates": gates,
                "device_ms_per_step": round(dev_ms / args.steps, 3),
                **({"queues": "timed steps: scan chains forked after vseg on three device queues (dmz_hip_set_two_queues, "
                              "default); `stages` / `roofline` leg: one queue, per-kernel hipEvents"} if with_expiry else {}),
                **({"expiry_conv": "f16x3 (default: f16 matrix core on operands split in two f16 parts, three products, fp32 accumulation)",
                    "expiry_conv_variants": variants} if variants else {}),
            },
            "roofline": roof,
            "stages": per_stage,
        }
        if side:
            out["side_configs"] = side
        if pmc_tag:
            # (which fields are counters of a committed profile, not of this run)
            out["counters_from"] = {"profile": pmc_tag,
                                    "fields": "hbm_*_B_per_unit, traffic*, valu_*, mfma_busy_frac, roofline.traffic / valu_issue and "
                                              "an mfma-bound roofline.frac come from profiles/%s_pmc_* (rocprofv3 --pmc passes of the "
                                              "same kernels at the batch in the file names), not from this run; ms / GBps / TFLOPs / "
                                              "achieved are measured live" % pmc_tag}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if comm_init_stuck:
        sys.stdout.flush()
        os._exit(0)  # (a helper thread is parked inside librccl: no orderly teardown of this context)
    ctx.close()


if __name__ == "__main__":
    main()
