#!/bin/bash
# The PMC passes of tools/r03_final.sh alone (its first run left databases behind and nothing came back): CM kernels, then the sorter.
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 200 bash "$REPO/tools/pmc_pass.sh" "$OUT/pmc" 2>&1 | tail -3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/s_$c"
  timeout 60 rocprofv3 --kernel-trace --pmc $c -d "$OUT/s_$c" -o pass -- python "$REPO/tools/stage_probe.py" 256 > "$OUT/sorter_$c.log" 2>&1
  db=$(find "$OUT/s_$c" -name "*.db" | head -1)
  python "$REPO/tools/rocpd_summary.py" --pmc "$db" "rocprofv3 --kernel-trace --pmc $c -- python tools/stage_probe.py 256  (one 256 MiB text block, both repetitions; counter values in KiB; FETCH_SIZE is to be doubled on gfx950)" > "$OUT/sorter_pmc_$c.txt"
  rm -rf "$OUT/s_$c" "$OUT/sorter_$c.log"
done
du -sh "$OUT"
