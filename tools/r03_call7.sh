#!/bin/bash
# Round 3, session 2: the leaner sync decoder (walker without votes / control-flow moves, model waves two bytes per trip).
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
echo "== parity (CM tests of the GPU suite)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or cm_decode_of_arbitrary or cm_row_cache or three_blocks_per_cu or lean_states or mutated or golden" > "$OUT/parity.log" 2>&1
tail -3 "$OUT/parity.log"
echo "== CM decode, cycle counters"
for v in sync3:768 sync:256; do
  timeout 300 python tools/cm_coresidency.py 2 ${v#*:} --only=${v%:*} --cycles 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
done
for t in 8; do echo "== BZ3_CM_TUNE=$t (walker priority off)"; BZ3_CM_TUNE=$t timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"; done
echo "== pipeline 768 x 8 MiB"
timeout 600 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline > "$OUT/pipe.json" 2> "$OUT/pipe.log"
tail -3 "$OUT/pipe.log"
python - "$OUT/pipe.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "stages", json.dumps(d["stages"]))
EOP
