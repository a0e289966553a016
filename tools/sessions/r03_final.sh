#!/bin/bash
# Round 3, last GPU call: (1) one full-size step of the headline workload under rocprofv3 --kernel-trace (kernel summary + gap analysis),
# (2) the PMC passes behind roofline.traffic (FETCH_SIZE / WRITE_SIZE, separate passes) over the CM kernels, (3) the same two
# counters over the suffix sorter's kernels at 256 MiB.   Usage: tools/r03_final.sh <outdir>
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout 1200 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "^\[bench" "$OUT/bench.log" | tail -6
db=$(find "$OUT/p" -name "*.db" | head -1)
python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline   (MI355X, ROCm 7.2)" > "$OUT/kernels.txt"
python "$REPO/tools/rocpd_summary.py" --gaps "$db" "the same run: where the front end and the tail wait" > "$OUT/gaps.txt"
rm -rf "$OUT/p"
head -12 "$OUT/kernels.txt" | cut -c1-64,100-200
cat "$OUT/gaps.txt"
python - "$OUT/bench.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "roofline", json.dumps(d["roofline"]), "stages", json.dumps(d["stages"]))
EOP
echo "== PMC passes over the CM kernels"
timeout 600 bash "$REPO/tools/pmc_pass.sh" "$OUT/pmc" 2>&1 | tail -4
echo "== PMC passes over the sorter (one 256 MiB text block through the stage hooks, both repetitions)"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/s_$c"
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/s_$c" -o pass -- python "$REPO/tools/stage_probe.py" 256 > "$OUT/sorter_$c.log" 2>&1
  db=$(find "$OUT/s_$c" -name "*.db" | head -1)
  python "$REPO/tools/rocpd_summary.py" --pmc "$db" "rocprofv3 --kernel-trace --pmc $c -- python tools/stage_probe.py 256  (counter values in KiB; FETCH_SIZE is to be doubled on gfx950)" > "$OUT/sorter_pmc_$c.txt"
  rm -rf "$OUT/s_$c"
  grep -E "k_rs_scatter|k_bwt_tail|k_ub_walk" "$OUT/sorter_pmc_$c.txt" | cut -c1-60,90-170 | head -8
done
