#!/bin/bash
# Round 5, GPU call 3: where do the rings' iterations spend their time?  256 x 64 MiB (one step: CM launches ~16 + ~31 s), BZ3_HIP_TRACE_RINGS=1
# (host wall time per phase of the front-end / tail loops), once plain and once under rocprofv3 --kernel-trace with the gap analysis
# (tools/rocpd_summary.py --gaps: time with no kernel, time with only the one-workgroup-per-block kernels running) and the kernel summary.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
echo "== ring tests"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rings_on_gpu or lean_states" 2>&1 | tail -3
echo "== plain"
BZ3_HIP_TRACE_RINGS=1 timeout 300 python bench.py --gpus 1 --blocks 256 --block-mib 64 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_256x64.json" 2> "$OUT/bench_256x64.log"
grep "bz3 rings\|^\[bench" "$OUT/bench_256x64.log" | tail -8
python -c "import json;d=json.load(open('$OUT/bench_256x64.json'));print(d['value'], d['stages'])"
echo "== traced"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/p"
BZ3_HIP_TRACE_RINGS=1 timeout 400 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --gpus 1 --blocks 256 --block-mib 64 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_256x64_traced.json" 2> "$OUT/bench_256x64_traced.log"
grep "bz3 rings" "$OUT/bench_256x64_traced.log"
db=$(find "$OUT/p" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python "$REPO/tools/rocpd_summary.py" --gaps "$db" "256 x 64 MiB, one step" | tee "$OUT/gaps_256x64.txt"
  python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --blocks 256 --block-mib 64 --steps 1 --no-extras --no-cpu-baseline" > "$OUT/kernels_256x64.txt"
  head -40 "$OUT/kernels_256x64.txt" | cut -c1-70,100-175
fi
rm -rf "$OUT/p"
