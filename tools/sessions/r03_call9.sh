#!/bin/bash
# Round 3, session 2: the trimmed encoder coder loop, the vectorised LZP decoder and the 8 x 8 tail ring.
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
OLD=bzip3_amd/lib/ab/libbzip3_f98d2d6.so  # built beforehand (not in git): git worktree add /tmp/wt f98d2d6 && (cd /tmp/wt && python bzip3_amd/build.py) && cp /tmp/wt/bzip3_amd/lib/libbzip3.so $OLD
echo "== parity (subset of the GPU suite)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or block_parity or cm_ or lean or rings or mutated or golden or batch_api or three_blocks" > "$OUT/parity.log" 2>&1
tail -3 "$OUT/parity.log"
for L in $OLD ""; do
  echo "== pipeline 768 x 8 MiB ${L:-HEAD}"
  timeout 600 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline ${L:+--lib=$L} > "$OUT/pipe.json" 2> "$OUT/pipe.log"
  python - "$OUT/pipe.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "cm enc", d["stages"]["enc"]["cm"], "cm dec", d["stages"]["dec"]["cm"], "t_enc", d["stages"]["t_enc_s"], "t_dec", d["stages"]["t_dec_s"], d["stages"]["front_end_ring"])
EOP
done
bash tools/r03_gaps256.sh "$OUT/g" 128 2>&1 | grep -v "^ *[0-9.]* *[0-9.]*  " | tail -12
