#!/bin/bash
# Round 6, call 5: at what clock do the CM launches run?  The decoder's cycle counters say ~1.7 GHz in a launch of a second (1,026 cycles per byte = 600 ns), the
# single-wave microbenchmark of round 1 ticked at 2.4 GHz.  rocm-smi once a second beside a 768 x 32 MiB batch, read-only (the pool refuses rocm-smi --set*: every job runs with the machine's default settings).
OUT=gpurun_out/c5
mkdir -p $OUT
rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | tee $OUT/smi_idle.txt
rocm-smi --showclkfrq 2>&1 | head -60 | tee -a $OUT/smi_idle.txt
one() {  # tag
  python3 tools/clock_sampler.py $OUT/clocks_$1.jsonl 75 &
  SP=$!
  timeout 300 python3 bench.py --blocks 768 --block-mib 32 --steps 1 --warmup 0 --no-extras --no-cpu-baseline 2> $OUT/bench_$1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 't_enc', d['stages']['t_enc_s'], 't_dec', d['stages']['t_dec_s'], 'cm', d['stages']['enc']['cm'], d['stages']['dec']['cm'])" | tee -a $OUT/summary.txt
  wait $SP
  python3 - <<P | tee -a $OUT/summary.txt
import json
rows=[json.loads(l) for l in open("$OUT/clocks_$1.jsonl")]
def num(v):
    import re
    m=re.search(r"[-+]?\d+(\.\d+)?", str(v)); return float(m.group()) if m else None
ks=sorted({k for r in rows for k in r if k not in ("t","err")})
for k in ks:
    vals=[num(r.get(k)) for r in rows if num(r.get(k)) is not None]
    if vals: print("   $1", k, "min", min(vals), "max", max(vals), "samples", [r.get(k) for r in rows][::4][:20])
P
}
one auto
