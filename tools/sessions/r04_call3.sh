#!/bin/bash
# Round 4, GPU call 3: (A) decoder experiments again with the run phase's table loads out of the compiler's sight; (B) the front end after
# round 4's changes (code table built on the device, persistent wide / tail kernels, one read-back per pass, no per-block waits) against
# round 3's library: BWT alone at 256 MiB, and a 768 x 8 MiB batch; (C) GPU parity of what changed; (D) how many reference threads the box rewards.
OUT=gpurun_out/c3
mkdir -p $OUT
echo "== A decoder experiments, 768 x 2 MiB, cycle counters"
timeout 400 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles --exp=0,1,5,9,13 2>&1 | grep variant | tee $OUT/dec_exp_768.txt
echo "== A 768 x 8 MiB"
timeout 400 python tools/cm_coresidency.py 8 768 --only=sync3 --exp=0,1,5,9,13 2>&1 | grep variant | tee $OUT/dec_exp_768_8MiB.txt
echo "== A one per CU"
timeout 200 python tools/cm_coresidency.py 2 256 --only=sync --exp=0,1,13 2>&1 | grep variant | tee $OUT/dec_exp_256.txt
echo "== C parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or suffix_sorter or block_parity or lean or golden or cfg1 or batch_api or rings or device_resident" 2>&1 | tail -3 | tee $OUT/parity.txt
echo "== B stages at 256 MiB: round 3 library, then HEAD"
timeout 300 python tools/stage_probe.py 256 --lib=bzip3_amd/lib/ab/libbzip3_r03.so 2>&1 | grep MiB | tee $OUT/stages_r03.txt
timeout 300 python tools/stage_probe.py 256 2>&1 | grep MiB | tee $OUT/stages_head.txt
timeout 300 python tools/stage_probe.py 256 --noise=0.035 2>&1 | grep MiB | tee $OUT/stages_head_noise.txt
echo "== B 768 x 8 MiB batch: round 3 library, then HEAD"
for L in "--lib bzip3_amd/lib/ab/libbzip3_r03.so" ""; do
  timeout 300 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('value', d['value'], 'steps', d['step_s'], 't_enc', s['t_enc_s'], 'cm_enc_ms', s['enc']['cm'], 'front_end_s', round(s['t_enc_s']-s['enc']['cm']/1e3,3), 't_dec', s['t_dec_s'], 'cm_dec_ms', s['dec']['cm'], 'tail_s', round(s['t_dec_s']-s['dec']['cm']/1e3,3))" | tee -a $OUT/batch_8MiB.txt
done
echo "== B kernel trace of the stages at 256 MiB (HEAD)"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o stages -- python $GRAFT_REPO_ROOT/tools/stage_probe.py 256 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
F=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$F" ] && head -45 "$F" | cut -c1-220 | tee $OUT/kernel_stats_stages_256MiB.csv.txt
DB=$(find $OUT/prof -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" "rocprofv3 --kernel-trace -- python tools/stage_probe.py 256" > $OUT/kernel_stats_stages_256MiB.txt 2>/dev/null; head -50 $OUT/kernel_stats_stages_256MiB.txt
rm -rf $OUT/prof
echo "== D reference threads"
timeout 400 python tools/cpu_threads_probe.py 32 2>&1 | tail -1 | tee $OUT/cpu_threads.json
echo "== E LDS / issue counters of the decoder (exp 1, 768 x 2 MiB)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/$OUT/sq_counters_avail.txt
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  rm -rf /tmp/pmcx
  timeout 200 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmcx -o p -- python $GRAFT_REPO_ROOT/tools/cm_coresidency.py 2 768 --only=sync3 --exp=1 > /dev/null 2>/tmp/pmcx.err || tail -3 /tmp/pmcx.err
  DB=$(find /tmp/pmcx -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --pmc "$DB" "rocprofv3 --kernel-trace --pmc $SET -- python tools/cm_coresidency.py 2 768 --only=sync3 --exp=1" | grep -v "k_bwt\|k_rs_\|k_scan\|k_vlc\|at::" | head -12 | tee -a $GRAFT_REPO_ROOT/$OUT/pmc_decoder_exp1.txt
done
cd $GRAFT_REPO_ROOT
