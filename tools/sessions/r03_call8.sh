#!/bin/bash
# Round 3, session 2: same-box A/B of the CM kernels -- the library of commit f98d2d6 (round-2 decoder, new sorter) against HEAD.
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
OLD=bzip3_amd/lib/ab/libbzip3_f98d2d6.so  # built beforehand (not in git): git worktree add /tmp/wt f98d2d6 && (cd /tmp/wt && python bzip3_amd/build.py) && cp /tmp/wt/bzip3_amd/lib/libbzip3.so $OLD
echo "== parity (CM tests of the GPU suite)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or cm_decode_of_arbitrary or cm_row_cache or three_blocks_per_cu or lean_states or mutated or golden" > "$OUT/parity.log" 2>&1
tail -3 "$OUT/parity.log"
echo "== CM decode at 2 MiB: f98d2d6, then HEAD (with cycle counters)"
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --lib=$OLD 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
timeout 300 python tools/cm_coresidency.py 2 256 --only=sync --lib=$OLD 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
timeout 300 python tools/cm_coresidency.py 2 256 --only=sync --cycles 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
echo "== CM decode at 32 MiB (longer runs): f98d2d6, then HEAD"
timeout 600 python tools/cm_coresidency.py 32 768 --only=sync3 --lib=$OLD 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
timeout 600 python tools/cm_coresidency.py 32 768 --only=sync3 --cycles 2>&1 | grep variant | tee -a "$OUT/cm_dec.txt"
for L in $OLD ""; do
  echo "== pipeline 768 x 8 MiB ${L:-HEAD}"
  timeout 600 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline ${L:+--lib=$L} > "$OUT/pipe.json" 2> "$OUT/pipe.log"
  python - "$OUT/pipe.json" <<'EOP'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "step_s", d["step_s"], "cm enc", d["stages"]["enc"]["cm"], "cm dec", d["stages"]["dec"]["cm"], "t_enc", d["stages"]["t_enc_s"], "t_dec", d["stages"]["t_dec_s"])
EOP
done
