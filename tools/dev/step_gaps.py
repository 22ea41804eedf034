#!/usr/bin/env python3
"""Developer tool (GPU box): idle gaps between the kernels of the timed pipeline steps, from a rocprofv3 kernel trace.
usage: cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline
       python3 tools/dev/step_gaps.py /tmp/kt"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
# the timed steps: from the last k_detect_walk<false...> launch group backwards -- take the last 3 detect-hz launches
import re


def short(n):
    m = re.search(r"(k_\w+|__amd_\w+)", n)
    return m.group(1) if m else n[:30]


hz = [e for e in ev if "k_detect_walk<false" in e[2]]
# launches of the top / bottom detector: [warm-up, timed steps ..., (later: whatever bench.py verifies)]; the window = the timed steps
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = hz[1][0]
t1 = hz[1 + nsteps][0] if len(hz) > 1 + nsteps else max(e[1] for e in ev)
ev = [e for e in ev if e[0] < t1]
cov, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in ev:
    if e < t0:
        continue
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            cov += cur_e - cur_s
            gaps.append((s - cur_e, short(n)))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print("window %.3f ms, some kernel running %.3f ms, nothing running %.3f ms (%.2f %%)" % ((t1 - t0) / 1e6, cov / 1e6, (t1 - t0 - cov) / 1e6, 100 * (1 - cov / (t1 - t0))))
gaps.sort(reverse=True)
print("largest gaps (us, before kernel):", [(round(g / 1e3, 1), n) for g, n in gaps[:10]])
per = {}
for s, e, n in ev:
    if s >= t0:
        k = short(n)
        per[k] = per.get(k, 0) + (e - s)
print("kernel time in the window (ms, overlapping queues add up):", {k: round(v / 1e6, 2) for k, v in sorted(per.items(), key=lambda x: -x[1])[:14]})
