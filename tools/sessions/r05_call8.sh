#!/bin/bash
# Round 5, GPU call 8: (1) one full-size step, no profiler, tail ring 8 x 8 on the CU partition (the default now), rings' phase times; (2) the cfg5 leg again (lean states).
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
BZ3_HIP_TRACE_RINGS=1 timeout 900 python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "bz3 rings\|^\[bench" "$OUT/bench.log" | grep -v " 1 blocks" | tail -8
python -c "import json;d=json.load(open('$OUT/bench.json'));s=d['stages'];print('value',d['value'],'front',round(s['t_enc_s']-s['enc']['cm']/1e3,2),'cm',round(s['enc']['cm']/1e3,2),round(s['dec']['cm']/1e3,2),'tail',round(s['t_dec_s']-s['dec']['cm']/1e3,2));print(json.dumps(s))"
echo "== cfg5 leg"
BZ3_HIP_TRACE_RINGS=1 timeout 900 python bench.py --leg cfg5 --blocks 256 --steps 1 > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.log"
grep "bz3 rings\|^\[bench\|HIP failure" "$OUT/bench_cfg5.log" | grep -v " 1 blocks" | tail -8
python -c "import json;d=json.load(open('$OUT/bench_cfg5.json'));print('cfg5 value',d['value'],json.dumps(d['stages']))"
