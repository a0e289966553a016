#!/bin/bash
# Round 5, GPU call 9: the CM kernels with their input loaded a chunk / a window ahead against the library one commit earlier, same box:
# bench.py --blocks 768 --block-mib 8 --steps 2 (768 different blocks, three per CU: each block reads its own input from HBM), then at 32 MiB; CM launch times.
# Before that the GPU tests of what the round added last (scan / sorter hooks, eight-slot ring, the large lean batch with kept workspace, CM parity).
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
timeout 500 python -m pytest tests/test_sorter.py tests/test_gpu_parity.py -m gpu -x -q -k "scan or sorter_on or rings_on_gpu or large_lean or stage_parity or block_parity or cm_decode_of_arbitrary or three_blocks_per_cu" 2>&1 | tail -3
ab() {  # name, block MiB, extra args
  local name=$1 mib=$2; shift 2
  timeout 400 python bench.py --gpus 1 --blocks 768 --block-mib $mib --steps 2 --warmup 0 --no-extras --no-cpu-baseline "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));s=d['stages'];print('$name', 'value',d['value'],'steps',d['step_s'],'cm enc',round(s['enc']['cm'],1),'cm dec',round(s['dec']['cm'],1),'ms; front',round(s['t_enc_s']-s['enc']['cm']/1e3,2),'tail',round(s['t_dec_s']-s['dec']['cm']/1e3,2))"
}
ab prev_8 8 --lib=bzip3_amd/lib/ab/libbzip3_prev.so
ab head_8 8
ab prev_32 32 --lib=bzip3_amd/lib/ab/libbzip3_prev.so
ab head_32 32
