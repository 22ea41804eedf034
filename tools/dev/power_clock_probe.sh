#!/bin/bash
# Developer probe (GPU box): socket power and shader clock while the pipeline runs back to back, and idle before / after.
cd "$(dirname "$0")/../.."
sample() { rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|Temperature (Sensor junction)\|mclk" | sed 's/^/   /' | head -8; }
echo "== idle"; sample
python tools/dev/pipe_ab.py 65536 250 > gpurun_out/power_probe_run.log 2>&1 &
PID=$!
sleep 8   # synthesis + first steps
for i in 1 2 3 4 5 6; do echo "== running, sample $i"; sample; sleep 0.5; done
wait $PID
cat gpurun_out/power_probe_run.log | cut -c1-110
echo "== idle again"; sample
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
