#!/bin/bash
# Round 3, GPU call 1: kernel-level profile of HEAD (the tree the driver timed in BENCH_r02) and the A/B of every opt-in.
#   bash tools/r03_call1.sh gpurun_out/r03_c1
# = tools/r03_first_call.sh without its parity leg (GPUTEST_r02 ran the whole GPU suite on this tree: 76 passed) plus the CM cycle counters.
set -e
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
sed -i 's/^echo "== parity".*$/echo "== parity skipped" | tee "$OUT\/summary.txt"/; /^timeout 600 python -m pytest/d; /^tail -2 "\$OUT\/parity.log"/d' tools/r03_first_call.sh
bash tools/r03_first_call.sh "$OUT"
echo "== CM decode, cycle counters (2 MiB, 256/512/768 copies)" | tee -a "$OUT/summary.txt"
python tools/cm_coresidency.py 2 256 768 --cycles --only=sync,sync3 2>&1 | tee "$OUT/cm_coresidency.txt" | tee -a "$OUT/summary.txt"
echo "== CM encode split" | tee -a "$OUT/summary.txt"
python tools/cm_encode_split.py 2 256 768 2>&1 | tee "$OUT/cm_encode_split.txt" | tee -a "$OUT/summary.txt"
