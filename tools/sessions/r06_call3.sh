#!/bin/bash
# Round 6, call 3: configuration experiments (no code change) + the first CM decoder experiment.
#  a. suffix sorter: how many more code windows the big groups get before rank doubling takes what is left (BZ3_BWT_BIG_ROUNDS 1..4), 256 MiB of the bench's text
#  b. decoder tail at 768 x 64 MiB: reserved CUs / window of the LZP decoders' ring (BZ3_HIP_CU_RESERVE, BZ3_HIP_TAIL_PIPE), with the rings' own accounting
#  c. CM decoder, same box A/B: the product against the early-row variant (tools/build_variant.py early_row -DCM_EXP_EARLY_ROW=1), cycle counters
OUT=gpurun_out/c3
mkdir -p $OUT
for k in 1 2 3 4; do
  echo "== BZ3_BWT_BIG_ROUNDS=$k" | tee -a $OUT/bwt_big_rounds.txt
  BZ3_BWT_BIG_ROUNDS=$k BZ3_BWT_TRACE=1 timeout 200 python3 tools/stage_probe.py 256 --noise=0.035 2> $OUT/bwt_trace_$k.txt | tee -a $OUT/bwt_big_rounds.txt
  grep "^\[bwt\]" $OUT/bwt_trace_$k.txt | tail -12 | tee -a $OUT/bwt_big_rounds.txt
done
echo "== CM decoder A/B" | tee $OUT/cm_early_row.txt
for rep in 1 2; do
  timeout 300 python3 tools/cm_coresidency.py 2 768 --only=sync3 --cycles 2>&1 | grep variant | sed 's/^/product   /' | tee -a $OUT/cm_early_row.txt
  timeout 300 python3 tools/cm_coresidency.py 2 768 --only=sync3 --cycles --lib=bzip3_amd/lib/libbzip3_early_row.so 2>&1 | grep variant | sed 's/^/early_row /' | tee -a $OUT/cm_early_row.txt
done
timeout 300 python3 tools/cm_coresidency.py 8 768 --only=sync3 2>&1 | grep variant | sed 's/^/product   8MiB /' | tee -a $OUT/cm_early_row.txt
timeout 300 python3 tools/cm_coresidency.py 8 768 --only=sync3 --lib=bzip3_amd/lib/libbzip3_early_row.so 2>&1 | grep variant | sed 's/^/early_row 8MiB /' | tee -a $OUT/cm_early_row.txt
echo "== tail ring" | tee $OUT/tail_ring.txt
run_tail() {  # reserve window,slots
  echo "-- BZ3_HIP_CU_RESERVE=$1 BZ3_HIP_TAIL_PIPE=$2" | tee -a $OUT/tail_ring.txt
  BZ3_HIP_CU_RESERVE=$1 BZ3_HIP_TAIL_PIPE=$2 BZ3_HIP_TRACE_RINGS=1 timeout 400 python3 bench.py --blocks 768 --block-mib 64 --steps 1 --warmup 0 --no-extras --no-cpu-baseline 2> $OUT/tail_$1_$2.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], 't_enc', d['stages']['t_enc_s'], 't_dec', d['stages']['t_dec_s'], 'cm', d['stages']['enc']['cm'], d['stages']['dec']['cm'])" | tee -a $OUT/tail_ring.txt
  grep "bz3 rings" $OUT/tail_$1_$2.log | tee -a $OUT/tail_ring.txt
}
run_tail 48 16,4
run_tail 64 20,4
run_tail 64 21,4
run_tail 56 18,4
run_tail 48 16,4
