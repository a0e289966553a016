#!/bin/bash
# Round 5, GPU call 12: which of a workgroup's two waves codes -- by block index (rounds 3-4: BZ3_CM_NO_CLAIM=1) or by claiming a free SIMD (cm.hip, the default now).
# 768 x 32 MiB, two steps each: the record run's library (block index, old slow path), HEAD with BZ3_CM_NO_CLAIM=1 (block index, slow path as a search for the
# first due renormalisation), HEAD (claims + that slow path).  The encode launch of every step from the stage timer is only kept for the last step, so the
# encode CALL times of both steps are printed too.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage_parity or block_parity or three_blocks_per_cu" 2>&1 | tail -2
ab() {
  local name=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --blocks 768 --block-mib 32 --steps 2 --warmup 0 --no-extras --no-cpu-baseline $LIBARG > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));s=d['stages'];print('$name', 'value',d['value'],'steps',d['step_s'],'cm enc (last step)',round(s['enc']['cm'],1),'cm dec',round(s['dec']['cm'],1),'ms')"
  grep "encode_blocks done" "$OUT/bench_$name.log" | tr '\n' ' '; echo
}
LIBARG="--lib=bzip3_amd/lib/ab/libbzip3_prev.so" ab record_tree X=1
LIBARG="" ab head_block_index BZ3_CM_NO_CLAIM=1
LIBARG="" ab head_claims X=1
LIBARG="" ab head_claims_again X=1
