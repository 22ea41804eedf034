cd /root/repo
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_r1_v4.json
cat gpurun_out/bench_r1_v4.json | cut -c1-1500
bash tools/profile_round.sh r1v4 > /dev/null 2>&1
ls gpurun_out/profiles_r1v4
true
