"""SURVEY 8(e) on hardware with more than one GPU: the C-ABI's own gather (dmz_hip_gather_records: RCCL send / recv on the
context's communication queue) against torch.distributed's, byte for byte, for both record types.  Skipped on the 1-GPU
boxes the round's tests run on; the first multi-GPU node exercises it.  The ranks are child processes started before they
touch the GPU (bench.py starts its own: `python bench.py --gpus 2`)."""
import json
import os
import subprocess
import sys

import pytest

import __graft_entry__ as entry

pytestmark = pytest.mark.gpu


def _bench(args, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(entry.ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_two_rank_capi_gather_equals_the_torch_gather(pkg):
    if pkg.load_library().dmz_hip_device_count() < 2:
        pytest.skip("one GPU: the two-rank RCCL gather needs a second device (world-size-1: test_capi_gather_at_world_size_one)")
    # --gather capi: after the timed loop the last step's result AND expiry records travel once more through torch.distributed
    # and rank 0 compares the two destinations byte for byte; a difference or a failed communicator ends the run with an error
    rc, line, err = _bench(["--gpus", "2", "--batch", "4096", "--steps", "3", "--warmup", "1", "--gather", "capi",
                            "--no-cpu-baseline"])
    assert rc == 0, err[-3000:]
    assert line["n_gpus"] == 2 and line["config"]["gather"].startswith("capi"), line["config"]
    assert line["config"]["corpus_frames"] == 8192
    # the default mode takes the same path and says so
    rc, line, err = _bench(["--gpus", "2", "--batch", "4096", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--config", "2"])
    assert rc == 0, err[-3000:]
    assert line["config"]["gather"].startswith("capi") or line["config"]["gather"].startswith("torch (capi failed"), line["config"]
