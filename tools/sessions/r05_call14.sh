#!/bin/bash
# Round 5, GPU call 14 (what was left of the budget): the final tree's library -- stage / block / CM-variant parity, one small step.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_parity or block_parity or three_blocks_per_cu or lean_states" 2>&1 | tail -2
timeout 80 python bench.py --gpus 1 --blocks 768 --block-mib 8 --steps 2 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_final_8.json" 2> "$OUT/bench_final_8.log"
python -c "import json;d=json.load(open('$OUT/bench_final_8.json'));s=d['stages'];print('final tree 768 x 8 MiB: value',d['value'],'steps',d['step_s'],'cm enc',round(s['enc']['cm'],1),'cm dec',round(s['dec']['cm'],1),'ms')"
grep "encode_blocks done" "$OUT/bench_final_8.log" | tr '\n' ' '; echo
