#!/bin/bash
# Where do the front end and the tail of a bench step wait?  768 x 32 MiB (the serial LZP kernels against the whole-GPU kernels as at 256 MiB,
# at an eighth of the time) under rocprofv3 --kernel-trace, then tools/rocpd_summary.py --gaps.   Usage: tools/r03_gaps.sh <outdir> [env assignments...]
OUT=$(realpath -m "$1"); shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ $# -eq 0 ] && set -- ""
for cfg in "$@"; do
  tag=$(echo "${cfg:-default}" | tr -c 'A-Za-z0-9_=,\n' '_')
  rm -rf "$OUT/p"
  env $cfg timeout 900 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --blocks 768 --block-mib 32 --steps 1 --no-extras --no-cpu-baseline > "$OUT/bench_$tag.json" 2> "$OUT/bench_$tag.log"
  db=$(find "$OUT/p" -name "*.db" | head -1)
  echo "== $tag"
  python "$REPO/tools/rocpd_summary.py" --gaps "$db" | tee "$OUT/gaps_$tag.txt"
  python - "$OUT/bench_$tag.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "cm enc", d["stages"]["enc"]["cm"], "cm dec", d["stages"]["dec"]["cm"], "t_enc", d["stages"]["t_enc_s"], "t_dec", d["stages"]["t_dec_s"], d["stages"].get("front_end_ring"))
EOP
  rm -rf "$OUT/p"
done
