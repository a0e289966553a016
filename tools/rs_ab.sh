#!/bin/bash
# A/B of the radix sorter's experiment switches on one full-size block: rocprofv3 kernel trace (no other trace domains) of
# tools/stage_probe.py under  default | BZ3_RS_STAGED=1 | BZ3_RS_NO_XCD=1 | both,  one per-kernel summary each, and one line per
# configuration with the time of the sorter's kernels and of the whole BWT.
#   bash tools/rs_ab.sh <output dir under gpurun_out> [block MiB, default 256]
set -e
OUT=$(realpath "$1")
MIB=${2:-256}
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, env assignments...
    local name=$1; shift
    rm -rf "$OUT/$name"
    env "$@" rocprofv3 --kernel-trace -d "$OUT/$name" -o pass -- python "$REPO/tools/stage_probe.py" "$MIB" > "$OUT/$name.log" 2>&1 || { tail -5 "$OUT/$name.log"; exit 1; }
    local db; db=$(find "$OUT/$name" -name "*.db" | head -1)
    python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- $* python tools/stage_probe.py $MIB   (MI355X, ROCm 7.2)" > "$OUT/rs_ab_$name.txt"
    rm -rf "$OUT/$name"
    python - "$OUT/rs_ab_$name.txt" "$name" <<'PY'
import sys
sort_ms = bwt_ms = 0.0
for line in open(sys.argv[1]):
    f = line.split()
    if line.startswith("#") or len(f) < 6 or not f[-5].replace(".", "").isdigit():
        continue
    ms = float(f[-5])
    name = line[:100]
    if "k_rs_" in name or "k_scan_" in name:
        sort_ms += ms
    if "k_rs_" in name or "k_scan_" in name or "k_bwt_" in name:
        bwt_ms += ms
print(f"{sys.argv[2]:<14} sorter kernels {sort_ms:9.1f} ms   sorter + k_bwt_* {bwt_ms:9.1f} ms   (all calls of the probe: 2 repetitions, LZP and unBWT sorts included)")
PY
}
run default BZ3_RS_DUMMY=0 | tee "$OUT/rs_ab_summary.txt"
run staged BZ3_RS_STAGED=1 | tee -a "$OUT/rs_ab_summary.txt"
run noxcd BZ3_RS_NO_XCD=1 | tee -a "$OUT/rs_ab_summary.txt"
run staged_noxcd BZ3_RS_STAGED=1 BZ3_RS_NO_XCD=1 | tee -a "$OUT/rs_ab_summary.txt"
