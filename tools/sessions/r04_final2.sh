#!/bin/bash
# Round 4, last GPU call: the driver's bench command WITHOUT a profiler on HEAD (after the ring / pool fixes), then the stream driver on a file of 256 blocks.
OUT=gpurun_out/f2
mkdir -p $OUT
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log
grep "^\[bench" $OUT/bench.log > $OUT/bench.progress.txt; cat $OUT/bench.progress.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/f2/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "spread", d["step_spread"], "ratio", d["config"]["compressed_ratio"], "given_up", d["config"]["cm_blocks_given_up"])
print("roofline", json.dumps(d["roofline"])); print("stages", json.dumps(d["stages"])); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("gpu_over_cpu"), d["cpu_baseline"].get("parity"))
for k, v in d["configs"].items(): print(k, json.dumps(v)[:300])
P
echo "== stream driver, 2 GiB at -b 8 (256 blocks)"
timeout 400 python tools/stream_time.py 2048 8 768 64 2>&1 | tail -1 | tee $OUT/stream_driver_vs_cli_b8.json
