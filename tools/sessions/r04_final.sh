#!/bin/bash
# Round 4: the driver's own bench command under rocprofv3 --kernel-trace (kernel summary of the SAME command as the bench line, gap analysis),
# then the PMC passes behind roofline.traffic.   Usage: tools/r04_final.sh <outdir under gpurun_out>
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout 1950 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "^\[bench" "$OUT/bench.log" > "$OUT/bench.progress.txt"; cat "$OUT/bench.progress.txt"
db=$(find "$OUT/p" -name "*.db" | head -1)
python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --gpus 1 --steps 20 --warmup 5   (MI355X, ROCm 7.2; the driver's command: two timed steps + the legs)" > "$OUT/kernels.txt"
python "$REPO/tools/rocpd_summary.py" --gaps "$db" "the same run: where the front end and the tail wait" > "$OUT/gaps.txt" 2>&1
rm -rf "$OUT/p"
head -14 "$OUT/kernels.txt" | cut -c1-64,100-200
head -40 "$OUT/gaps.txt"
python - "$OUT/bench.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "spread", d["step_spread"], "ratio", d["config"]["compressed_ratio"], "given_up", d["config"]["cm_blocks_given_up"])
print("roofline", json.dumps(d["roofline"])); print("stages", json.dumps(d["stages"])); print("cpu", json.dumps(d["cpu_baseline"])[:900])
for k, v in d["configs"].items(): print(k, json.dumps(v)[:500])
EOP
echo "== PMC passes over the CM kernels"
cd "$REPO"
timeout 600 bash "$REPO/tools/pmc_pass.sh" "$OUT/pmc" 2>&1 | tail -4
cp "$REPO/profiles/pmc_traffic.json" "$OUT/pmc_traffic.json"
