#!/bin/bash
# Round 4, GPU call 2: (A) decoder: subtree form (1), + wave-uniform skip (5), + run phase two tables ahead (9), all three (13);
# (B) the same at 8 MiB blocks; (C) what the box gives the container (cpu quota); (D) the restructured bench.py end to end at 8 MiB blocks.
OUT=gpurun_out/c2
mkdir -p $OUT
echo "== C host"; { nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; grep Cpus_allowed_list /proc/self/status; free -g | head -2; df -h /dev/shm | tail -1; lscpu | grep -E "Model name|Socket|Thread|NUMA node\(s\)"; } 2>&1 | tee $OUT/host.txt
echo "== A decoder experiments, 768 x 2 MiB, cycle counters"
timeout 400 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles --exp=0,1,5,9,13 2>&1 | grep variant | tee $OUT/dec_exp_768.txt
echo "== A one per CU"
timeout 200 python tools/cm_coresidency.py 2 256 --only=sync --exp=0,1,9,13 2>&1 | grep variant | tee $OUT/dec_exp_256.txt
echo "== B 768 x 8 MiB"
timeout 400 python tools/cm_coresidency.py 8 768 --only=sync3 --exp=0,1,9,13 2>&1 | grep variant | tee $OUT/dec_exp_768_8MiB.txt
echo "== D bench.py at 8 MiB blocks"
timeout 600 python bench.py --blocks 768 --block-mib 8 --steps 2 --host-api-block-mib 8 --cfg3-bytes 100000000 > $OUT/bench_8MiB.json 2> $OUT/bench_8MiB.err
grep "bench " $OUT/bench_8MiB.err | tail -30
python - <<'P'
import json
d=json.loads(open("gpurun_out/c2/bench_8MiB.json").read().strip().splitlines()[-1])
print(d["value"], d["step_s"], d["step_spread"], d["config"]["compressed_ratio"], d["config"]["bwt_output_repeat_rate_16MiB_sample"])
print(json.dumps(d["cpu_baseline"])[:600])
for k,v in d["configs"].items(): print(k, json.dumps(v)[:400])
P
