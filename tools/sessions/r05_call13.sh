#!/bin/bash
# Round 5, GPU call 13 (the round's last): the encoder's slow path alone.  Both libraries claim SIMDs for their coder waves; libbzip3_oldslow.so codes a byte
# whose untested bits crossed a bucket again in halves (rounds 3-4), HEAD searches the first due renormalisation.  768 x 32 MiB, two steps each.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
ab() {
  local name=$1; shift
  timeout 300 python bench.py --gpus 1 --blocks 768 --block-mib 32 --steps 2 --warmup 0 --no-extras --no-cpu-baseline "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));s=d['stages'];print('$name', 'value',d['value'],'steps',d['step_s'],'cm enc (last step)',round(s['enc']['cm'],1),'cm dec',round(s['dec']['cm'],1),'ms')"
  grep "encode_blocks done" "$OUT/bench_$name.log" | tr '\n' ' '; echo
}
ab claims_old_slow_path --lib=bzip3_amd/lib/ab/libbzip3_oldslow.so
ab claims_new_slow_path
