"""bench.py's roofs are roofs: every fraction it derives from the committed counter passes (profiles/CURRENT) is <= 1, and
no bench line kept under profiles/ for this round carries a fraction above 1 (the old VALU-issue figure assumed
four cycles per instruction and read 1.29 for a kernel made of full-rate instructions)."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_counter_roofs_stay_below_one():
    issue = bench.load_pmc_issue()
    assert issue, "profiles/CURRENT names no SQ issue pass"
    seen = 0
    for stage in bench.STAGE_KERNELS:
        si = bench.stage_issue(issue, stage)
        if si is None:
            continue
        seen += 1
        for key in ("valu_issue_frac", "valu_halfrate_saturation", "mfma_busy_frac"):
            assert 0.0 <= si[key] <= 1.0, (stage, key, si[key])
        # the raw ratio never passes the calibrated ceiling of the full-rate class
        assert si["valu_active_per_busy"] <= bench.VALU_RATIO_FULL_RATE, (stage, si)
    assert seen >= 6


def _fracs(node, path=""):
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _fracs(v, path + "/" + k)
    elif isinstance(node, (int, float)) and ("frac" in path.rsplit("/", 1)[-1] or path.endswith("saturation")):
        yield path, node


def test_no_fraction_above_one_in_this_rounds_bench_lines():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r4_*bench*.json")))
    for f in files:
        for line in open(f):
            line = line.strip()
            if not line.startswith("{"):
                continue
            for path, v in _fracs(json.loads(line)):
                assert v <= 1.0, (os.path.basename(f), path, v)


def test_cpu_baseline_states_the_quota_and_the_unloaded_rate(tmp_path, monkeypatch):
    """the CPU baseline starts min(affinity, cgroup quota) workers and says what it was given.  The quota
    reader is exercised on a synthetic cgroup v2 file; the baseline itself on a one-second run."""
    import bench
    q, src = bench.cpu_quota()
    assert q is None or (q > 0 and src)
    r = bench.cpu_baseline(2, budget_s=0.5, frames_per_worker=4)  # configs[1] (detect only): the cheapest oracle leg
    assert r["kind"] == "port" and r["cores"] >= 1 and r["cores"] <= r["cpu_affinity"]
    assert r["single_process_frames_per_s"] > 0 and r["per_core"] > 0 and r["effective_cores"] > 0
    assert "cpu_quota" in r and "one process per usable host core" in r["sample"]
    # a worker's rate under load is within a factor of a few of the unloaded one unless the host throttles (then the record says
    # how many cores' worth of CPU time the workers obtained)
    assert r["per_effective_core"] > 0.2 * r["single_process_frames_per_s"]
