#!/bin/bash
# Round 6, LAST call on the round's tree (the driver's command first and last): `python3 bench.py --gpus 1 --steps 20 --warmup 5`, unflagged, no profiler,
# then the whole -m gpu suite.
OUT=gpurun_out/final
mkdir -p $OUT
timeout 1800 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log
echo "bench rc=$?"
grep "^\[bench" $OUT/bench.log > $OUT/bench.progress.txt; cat $OUT/bench.progress.txt
grep -v "^\[bench" $OUT/bench.log | tail -8
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/final/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d.get("step_s"), "complete", d.get("complete"), "error", d.get("error"), "failed_legs", d.get("failed_legs"))
print("config", json.dumps({k: v for k, v in d["config"].items() if k not in ("workload", "sharding_model")})[:2500])
print("roofline", json.dumps(d.get("roofline"))); print("stages", json.dumps(d.get("stages")))
c = d.get("cpu_baseline", {}); print("cpu", c.get("value"), c.get("gpu_over_cpu"), c.get("parity"), c.get("build_probe_1_thread_8MiB_MiBps"))
P
timeout 1500 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
# (what the mRLE kernels take per 256 MiB block after round 6's 16-byte stores: the stage probe's host times bracket them)
timeout 200 python3 tools/stage_probe.py 256 --noise=0.035 2>/dev/null | tee $OUT/stage_probe_256MiB.txt
# SURVEY 8d's "additionally the unmodified CLI wall-clock on tmpfs" (round 2's record refreshed): the reference's main.c linked against the reference and against the product,
# 256 MiB of text at -b 32 -j 8 -- eight blocks per call cannot fill a GPU, which is the point of the number.
[ -n "$R06_SKIP_CLI" ] || timeout 900 python3 tools/cli_time.py 256 32 8 2>&1 | tail -1 | tee $OUT/cli_wall_clock_tmpfs.json
