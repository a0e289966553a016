#!/bin/bash
# Round 6, FIRST GPU call: the driver's exact bench command, unflagged, no profiler, on the tree with the guarded parity sample,
# the headroom rule and a JSON line on every exit path.  Then the round's new -m gpu tests.
OUT=gpurun_out/c1
mkdir -p $OUT
timeout 1800 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log
echo "bench rc=$?"
grep "^\[bench" $OUT/bench.log > $OUT/bench.progress.txt; cat $OUT/bench.progress.txt
grep -v "^\[bench" $OUT/bench.log | tail -20
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/c1/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d.get("step_s"), "complete", d.get("complete"), "error", d.get("error"), "failed_legs", d.get("failed_legs"))
print("config", json.dumps({k: v for k, v in d["config"].items() if k not in ("workload", "sharding_model")})[:1500])
print("roofline", json.dumps(d.get("roofline"))); print("stages", json.dumps(d.get("stages")))
c = d.get("cpu_baseline", {}); print("cpu", c.get("value"), c.get("gpu_over_cpu"), c.get("parity"))
for k, v in d.get("configs", {}).items(): print(k, json.dumps(v)[:300])
P
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headroom or large_lean_batch or rings_on_gpu" 2>&1 | tail -15 | tee $OUT/pytest_new.log
