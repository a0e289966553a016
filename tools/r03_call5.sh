#!/bin/bash
# Round 3: suffix sorter timing iteration (256 MiB stage probe under rocprofv3) + CM decoder redo-priority A/B.
set -e
OUT=$(realpath -m "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "suffix_sorter" > "$OUT/parity.log" 2>&1 || true
tail -2 "$OUT/parity.log"
( cd /tmp && export TMPDIR=/tmp && rm -rf "$OUT/p" && BZ3_BWT_TRACE=1 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/tools/stage_probe.py" 256 > "$OUT/probe.log" 2>&1 ) || tail -5 "$OUT/probe.log"
grep -E "^\[bwt\]|MiB rep" "$OUT/probe.log" | head -20
db=$(find "$OUT/p" -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" "rocprofv3 --kernel-trace -- python tools/stage_probe.py 256   (MI355X, ROCm 7.2)" > "$OUT/probe_kernels.txt"
rm -rf "$OUT/p"
grep -E "k_bwt|k_big|k_bg|k_rs_|k_scan|k_isa" "$OUT/probe_kernels.txt" | cut -c1-70,100-200
if [ -n "$2" ]; then
for t in 0 2; do echo "== BZ3_CM_TUNE=$t"; BZ3_CM_TUNE=$t python tools/cm_coresidency.py 8 768 --only=sync3 2>&1 | grep variant; done
fi
