#!/bin/bash
# Round 5, GPU call 10: the whole `-m gpu` suite on the round's final tree, then the record run: one full-size step under rocprofv3 --kernel-trace
# (kernel summary, gap analysis, rings' phase times) -> profiles/r05_bench_768x256MiB_rocprof.json, r05_kernel_stats_bench_768x256MiB.txt.
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
cd /tmp; export TMPDIR=/tmp
rm -rf "$OUT/p"
BZ3_HIP_TRACE_RINGS=1 timeout 1000 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python "$REPO/bench.py" --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.log"
grep "bz3 rings\|^\[bench" "$OUT/bench.log" | grep -v " 1 blocks" | tail -8
python -c "import json;d=json.load(open('$OUT/bench.json'));s=d['stages'];print('value',d['value'],'front',round(s['t_enc_s']-s['enc']['cm']/1e3,2),'cm',round(s['enc']['cm']/1e3,2),round(s['dec']['cm']/1e3,2),'tail',round(s['t_dec_s']-s['dec']['cm']/1e3,2));print(json.dumps(s))"
db=$(find "$OUT/p" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline   (MI355X, ROCm 7.2; round 5, final tree)" > "$OUT/kernels.txt"
  python "$REPO/tools/rocpd_summary.py" --gaps "$db" "768 x 256 MiB, one step" > "$OUT/gaps.txt"
  cat "$OUT/gaps.txt"; head -14 "$OUT/kernels.txt" | cut -c1-72,100-176
fi
rm -rf "$OUT/p"
