#!/bin/bash
# Round 5, GPU call 11: the CM encoder's slow path as a search for the first due renormalisation (cm.hip cm_code_from_first_due) against the record
# run's library (bzip3_amd/lib/ab/libbzip3_prev.so = HEAD before the change), same box: bench.py --blocks 768 at 8 and 32 MiB, two steps each.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_parity or block_parity or three_blocks_per_cu" 2>&1 | tail -2
ab() {
  local name=$1 mib=$2; shift 2
  timeout 400 python bench.py --gpus 1 --blocks 768 --block-mib $mib --steps 2 --warmup 0 --no-extras --no-cpu-baseline "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));s=d['stages'];print('$name', 'value',d['value'],'steps',d['step_s'],'cm enc',round(s['enc']['cm'],1),'cm dec',round(s['dec']['cm'],1),'ms')"
}
ab prev_8 8 --lib=bzip3_amd/lib/ab/libbzip3_prev.so
ab head_8 8
ab prev_32 32 --lib=bzip3_amd/lib/ab/libbzip3_prev.so
ab head_32 32
