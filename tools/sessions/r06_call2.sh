#!/bin/bash
# Round 6, call 2: the two measurement legs VERDICT r05 item 7 asks for, each a run of its own WITH the reference beside it --
# X1: 256 uniformly random blocks of 256 MiB (north_star: "synthetic enwik-style and random blocks"); cfg5: a GPU-filling batch of 511 MiB blocks.
# First (10 s): what the suffix sorter's deep path handles in a 256 MiB block of the bench's text (BZ3_BWT_TRACE).
OUT=gpurun_out/c2
mkdir -p $OUT
BZ3_BWT_TRACE=1 timeout 300 python3 tools/stage_probe.py 256 --noise=0.035 > $OUT/stage_probe_256MiB.txt 2> $OUT/bwt_trace_256MiB.txt
cat $OUT/stage_probe_256MiB.txt; grep "^\[bwt\]" $OUT/bwt_trace_256MiB.txt | head -20
timeout 1500 python3 bench.py --kind random --blocks 256 --block-mib 256 --steps 1 --warmup 0 --no-extras > $OUT/random.json 2> $OUT/random.log
echo "random rc=$?"; grep "^\[bench" $OUT/random.log | tee $OUT/random.progress.txt
timeout 1700 python3 bench.py --leg cfg5 --blocks 256 --steps 1 --warmup 0 --cpu-threads 16 > $OUT/cfg5.json 2> $OUT/cfg5.log
echo "cfg5 rc=$?"; grep "^\[bench" $OUT/cfg5.log | tee $OUT/cfg5.progress.txt
python3 - <<'P'
import json
for f in ("random", "cfg5"):
    try:
        d = json.loads(open(f"gpurun_out/c2/{f}.json").read().strip().splitlines()[-1])
        c = d.get("cpu_baseline", {})
        print(f, "value", d["value"], "step_s", d.get("step_s"), "error", d.get("error"), "| cpu", c.get("value"), c.get("gpu_over_cpu"), c.get("parity"), "| stages", json.dumps(d.get("stages")))
    except Exception as e:
        print(f, "no line:", e)
P
