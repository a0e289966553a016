#!/bin/bash
# Round 6, call 6: (1) one full-size step with 80 reserved CUs and tail windows of 25 (does hiding more LZP decoders still pay for the CUs the walks lose?);
# (2) the record run: one full-size step under rocprofv3 --kernel-trace on the round's tree (kernel summary, gap analysis, rings' phase times);
# (3) the two PMC passes (FETCH_SIZE / WRITE_SIZE, separate, kernel trace only) over a 768 x 16 MiB batch -> profiles/pmc_traffic.json of this round.
REPO=$(cd "$(dirname "$0")/../.." && pwd); OUT=$REPO/gpurun_out/c6; mkdir -p "$OUT"; cd "$REPO"
BZ3_HIP_CU_RESERVE=80 BZ3_HIP_TAIL_PIPE=25,4 BZ3_HIP_TRACE_RINGS=1 timeout 700 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/full_r80_w25.json 2> $OUT/full_r80_w25.log
grep "^\[bench\|bz3 rings" $OUT/full_r80_w25.log | grep -v " 1 blocks" | tee $OUT/full_r80_w25.progress.txt
cd /tmp; export TMPDIR=/tmp
rm -rf "$OUT/p"
BZ3_HIP_TRACE_RINGS=1 timeout 1000 rocprofv3 --kernel-trace -d "$OUT/p" -o pass -- python3 "$REPO/bench.py" --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_rocprof.json" 2> "$OUT/bench_rocprof.log"
grep "bz3 rings\|^\[bench" "$OUT/bench_rocprof.log" | grep -v " 1 blocks" | tee "$OUT/bench_rocprof.progress.txt"
python3 -c "import json;d=json.loads(open('$OUT/bench_rocprof.json').read().strip().splitlines()[-1]);s=d['stages'];print('value',d['value'],'front',round(s['t_enc_s']-s['enc']['cm']/1e3,2),'cm',round(s['enc']['cm']/1e3,2),round(s['dec']['cm']/1e3,2),'tail',round(s['t_dec_s']-s['dec']['cm']/1e3,2));print(json.dumps(s))"
db=$(find "$OUT/p" -name "*.db" | head -1)
if [ -n "$db" ]; then
  python3 "$REPO/tools/rocpd_summary.py" "$db" "rocprofv3 --kernel-trace -- python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline   (MI355X, ROCm 7.2; round 6)" > "$OUT/kernels.txt"
  python3 "$REPO/tools/rocpd_summary.py" --gaps "$db" "768 x 256 MiB, one step" > "$OUT/gaps.txt"
  cat "$OUT/gaps.txt"; head -16 "$OUT/kernels.txt" | cut -c1-72,100-176
fi
rm -rf "$OUT/p"
cd "$REPO"
PMC_BLOCK_MIB=16 timeout 900 bash tools/pmc_pass.sh "$OUT/pmc" 2>&1 | tail -5
cat "$OUT/pmc/pmc_traffic.json" | head -40
