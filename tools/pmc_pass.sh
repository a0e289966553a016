#!/bin/bash
# Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, kernel trace only -- no other trace domains) over a small bench.py run that
# uses the same CM kernels as the full-size one (768 blocks = three per CU), then tools/pmc_traffic.py turns them into profiles/pmc_traffic.json.
#   bash tools/pmc_pass.sh <output dir under gpurun_out>
set -e
trap 'rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"' EXIT  # whatever happens, the databases stay on the box: gpurun brings back 64 MiB at most
OUT=$(realpath "$1")
REPO=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--blocks 768 --block-mib ${PMC_BLOCK_MIB:-2} --no-cpu-baseline --no-extras --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/$c"
    # (the exit code is not the test: under rocprofv3 the process may die in its exit handlers AFTER bench.py has printed its line and the database is written -- round 6)
    rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o pass -- python "$REPO/bench.py" $ARGS > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || true
    [ -n "$(find "$OUT/$c" -name '*.db' | head -1)" ] && grep -q '"value"' "$OUT/bench_$c.json" || { tail -5 "$OUT/bench_$c.err"; rm -rf "$OUT/$c"; exit 1; }
done
F=$(find "$OUT/FETCH_SIZE" -name "*.db" | head -1)
W=$(find "$OUT/WRITE_SIZE" -name "*.db" | head -1)
python "$REPO/tools/rocpd_summary.py" --pmc "$F" "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py $ARGS" > "$OUT/pmc_fetch.txt"
python "$REPO/tools/rocpd_summary.py" --pmc "$W" "rocprofv3 --kernel-trace --pmc WRITE_SIZE -- python bench.py $ARGS" > "$OUT/pmc_write.txt"
python "$REPO/tools/pmc_traffic.py" "$F" "$W" "$OUT/bench_FETCH_SIZE.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python bench.py $ARGS; MI355X, ROCm 7.2"
cp "$REPO/profiles/pmc_traffic.json" "$OUT/pmc_traffic.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"  # the databases stay on the box: gpurun brings back 64 MiB at most
