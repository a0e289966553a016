#!/bin/bash
# Round 3, session 2: where HEAD stands.  (a) every whole-GPU stage of one 256 MiB text block under rocprofv3 --kernel-trace,
# (b) the CM kernels at three blocks per CU with their cycle counters, (c) the whole pipeline at 768 x 8 MiB.
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
( cd /tmp && export TMPDIR=/tmp && rm -rf "$OUT/p" && BZ3_BWT_TRACE=1 timeout 400 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/tools/stage_probe.py" 256 > "$OUT/probe.log" 2>&1 ) || tail -5 "$OUT/probe.log"
grep -E "^\[bwt\]|MiB rep" "$OUT/probe.log" | head -20
db=$(find "$OUT/p" -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" "rocprofv3 --kernel-trace -- python tools/stage_probe.py 256   (MI355X, ROCm 7.2; both repetitions)" > "$OUT/probe_kernels.txt"
rm -rf "$OUT/p"
head -45 "$OUT/probe_kernels.txt" | cut -c1-60,100-200
echo "== CM decode sync3, cycle counters"
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles 2>&1 | grep variant | tee "$OUT/cm_dec.txt"
echo "== CM encode split"
timeout 300 python tools/cm_encode_split.py 2 768 2>&1 | grep encoder | tee "$OUT/cm_enc.txt"
echo "== pipeline 768 x 8 MiB"
timeout 600 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline > "$OUT/pipe.json" 2> "$OUT/pipe.log"
tail -3 "$OUT/pipe.log"
python - "$OUT/pipe.json" <<'EOF'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "stages", json.dumps(d["stages"]))
EOF
