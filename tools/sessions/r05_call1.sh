#!/bin/bash
# Round 5, GPU call 1: two full-size steps with BOTH queued switches on (BZ3_HIP_KEEP_WS=1, BZ3_HIP_TAIL_PIPE=8,8); the step's split
# (front end / CM / tail) is compared with profiles/r04_bench_768x256MiB.json (front 58.6, tail 24.8 s).
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
BZ3_HIP_KEEP_WS=1 BZ3_HIP_TAIL_PIPE=8,8 timeout 1100 python bench.py --gpus 1 --steps 2 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_keepws_tail8x8.json" 2> "$OUT/bench_keepws_tail8x8.log"
grep "^\[bench" "$OUT/bench_keepws_tail8x8.log" | tail -12
