#!/bin/bash
# One full-size step of the headline workload (768 x 256 MiB) under rocprofv3 --kernel-trace: the kernel summary for profiles/.
# Usage: tools/r03_bench_profile.sh <outdir> [extra bench.py flags]
OUT=$(realpath -m "$1"); shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout 1500 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "^\[bench" "$OUT/bench.log" | tail -12
db=$(find "$OUT/p" -name "*.db" | head -1)
python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline $*   (MI355X, ROCm 7.2)" > "$OUT/kernels.txt"
rm -rf "$OUT/p"
head -40 "$OUT/kernels.txt" | cut -c1-64,100-200
python - "$OUT/bench.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "roofline", json.dumps(d["roofline"]), "bwt", json.dumps(d["bwt_roofline"]), "stages", json.dumps(d["stages"]))
EOP
