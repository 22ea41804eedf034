"""The N>1 path on CPU: world_size-2 gloo run of the frame sharding + result gather used by
bench.py (RCCL on the GPUs).  Each rank 'scans' its shard with the CPU oracle standing in
for the device (this is the test's checker, not the product path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as entry

SEED, TOTAL = 77, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = entry.load_package()
    from dmz_amd import sharding
    orc = entry.load_oracle()
    o = orc.Oracle()
    lo, hi = sharding.shard_range(TOTAL, rank, world)
    recs = np.zeros(hi - lo, orc.RESULT_DTYPE)
    for k, idx in enumerate(range(lo, hi)):
        y, _ = o.synth_frame(SEED, idx)
        recs[k], _ = o.scan_frame(y, want_card=False)
    local = torch.from_numpy(recs.view(np.uint8).reshape(hi - lo, 1024).copy())
    out = sharding.gather_results(local, world)
    # the asynchronous gather-to-root bench.py uses: two batches in flight on alternating slots
    g = sharding.RootGatherer(world)
    a = g.submit(local, slot=0)
    second = local.flip(0).contiguous()
    b = g.submit(second, slot=1)
    g.wait(slots=(0,))
    g.wait()
    if rank == 0:
        assert torch.equal(a, out)
        assert torch.equal(b, torch.cat([c.flip(0) for c in out.chunk(world, dim=0)]))
        q.put(out.numpy().copy())
    else:
        assert a is None and b is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_tile_the_corpus(pkg):
    from dmz_amd import sharding
    for total in (0, 1, 7, 8, 65536, 1048576):
        for world in (1, 2, 4, 8):
            r = [sharding.shard_range(total, g, world) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_two_rank_gloo_gather_equals_single_process(pkg, orc, oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got = got.view(orc.RESULT_DTYPE).reshape(-1)
    assert got.shape[0] == TOTAL
    for idx in range(TOTAL):
        y, _ = oracle.synth_frame(SEED, idx)
        want, _ = oracle.scan_frame(y, want_card=False)
        assert got[idx].tobytes() == want.tobytes(), idx


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(entry.ROOT, "bench.py")] + args, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_starts_its_own_ranks_when_run_plainly():
    """`python bench.py --gpus 2` without torchrun: the script starts two ranks as a child process and
    they run bench.py's own shard / step / asynchronous-gather / timing code (--dry-run: CPU tensors over
    gloo, records filled with a rank-and-step pattern that rank 0 verifies after the gather)."""
    for config, steps in ((4, 3), (2, 2)):
        rc, line, err = _run_bench(["--gpus", "2", "--dry-run", "--batch", "48", "--steps", str(steps),
                                    "--config", str(config)])
        assert rc == 0, err[-2000:]
        assert line["n_gpus"] == 2 and line["dry_run"] and line["gather_ok"] and line["steps"] == steps
        assert line["frames_per_gpu"] == 48 and line["shard"] == [0, 48] and line["backend"] == "gloo"
        # weak scaling: the per-GPU batch does not depend on the number of ranks (the curve is like for like)
        assert line["scaling"] == "weak" and line["units_per_gpu"] == 48 and line["corpus_frames"] == 96


def test_bench_strong_scaling_runs_the_same_corpus_at_every_n():
    """--scaling strong: the corpus is fixed (16 resident batches), each rank takes corpus / N frames as passes over its batch"""
    seen = {}
    for n in (1, 2):
        rc, line, err = _run_bench(["--gpus", str(n), "--dry-run", "--batch", "16", "--steps", "2", "--scaling", "strong"])
        assert rc == 0, err[-2000:]
        assert line["scaling"] == "strong" and line["gather_ok"]
        seen[n] = (line["units_per_gpu"], line["corpus_frames"])
    assert seen[1] == (256, 256) and seen[2] == (128, 256)


def test_bench_per_gpu_batch_is_the_same_at_every_n():
    import bench
    assert bench.CORPUS_FRAMES == 1048576 and bench.CORPUS_FRAMES % (8 * bench.CONFIGS[4]["batch"]) == 0
    assert bench.CONFIGS[4]["bytes"] == 423784 and bench.CONFIGS[2]["batch"] == 4096 and bench.CONFIGS[3]["batch"] == 65536
    assert not hasattr(bench, "FRAMES_PER_GPU_MULTI")  # (round 3 ran 131 072 frames per GPU at N > 1 against 65 536 at N = 1)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=1" in err


def test_bench_reads_traffic_from_the_committed_profiles():
    import bench
    pmc, tag = bench.load_pmc_traffic()
    assert pmc is not None and tag, "profiles/CURRENT must name a PMC pass"
    for stage in bench.ALGO:
        tr = bench.stage_traffic(pmc, stage)
        assert tr is not None and tr[0] > 0, stage
