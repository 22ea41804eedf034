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
