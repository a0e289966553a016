#!/bin/bash
# Round 4, GPU call 4: the C2 cell pairs as two aligned 16-bit LDS accesses instead of one dword access that is misaligned half of the time
# (call 3's counters: the decoder's LDS busy 56 % of all cycles, three quarters of it SQ_LDS_UNALIGNED_STALL).
# Decoder: experiment bit 16 (17 = subtree + halves, 21 = + per-wave skip, 29 = + run phase); encoder: HEAD against round 3's library.
OUT=gpurun_out/c4
mkdir -p $OUT
echo "== A decoder experiments, 768 x 2 MiB, cycle counters"
timeout 400 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles --exp=0,5,17,21,29 2>&1 | grep variant | tee $OUT/dec_exp_768.txt
echo "== A 768 x 8 MiB"
timeout 400 python tools/cm_coresidency.py 8 768 --only=sync3 --exp=0,5,21,29 2>&1 | grep variant | tee $OUT/dec_exp_768_8MiB.txt
echo "== A one / two per CU"
timeout 200 python tools/cm_coresidency.py 2 256 --only=sync --cycles --exp=0,21 2>&1 | grep variant | tee $OUT/dec_exp_256.txt
timeout 200 python tools/cm_coresidency.py 2 512 --only=sync3 --exp=0,21 2>&1 | grep variant | tee $OUT/dec_exp_512.txt
echo "== B encoder: round 3 library, then HEAD (tools/cm_encode_split.py: full / rows kernels)"
timeout 200 python tools/cm_encode_split.py 2 256 512 768 --lib=bzip3_amd/lib/ab/libbzip3_r03.so 2>&1 | grep '"full"\|"rows"' | tee $OUT/enc_split_r03.txt
timeout 200 python tools/cm_encode_split.py 2 256 512 768 2>&1 | grep '"full"\|"rows"' | tee $OUT/enc_split_head.txt
echo "== B 768 x 8 MiB batch (rows3 kernels): round 3 library, then HEAD"
for L in "--lib bzip3_amd/lib/ab/libbzip3_r03.so" ""; do
  timeout 300 python bench.py --blocks 768 --block-mib 8 --steps 2 --no-extras --no-cpu-baseline $L 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('value', d['value'], 'steps', d['step_s'], 't_enc', s['t_enc_s'], 'cm_enc_ms', s['enc']['cm'], 'front_end_s', round(s['t_enc_s']-s['enc']['cm']/1e3,3), 't_dec', s['t_dec_s'], 'cm_dec_ms', s['dec']['cm'], 'tail_s', round(s['t_dec_s']-s['dec']['cm']/1e3,3))" | tee -a $OUT/batch_8MiB.txt
done
echo "== C parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or block_parity or lean or golden or cfg1 or cm_row_cache or three_blocks or cm_decode_of" 2>&1 | tail -3 | tee $OUT/parity.txt
echo "== D LDS counters: decoder exp 21, then the encoder + decoder of a 768 x 2 MiB batch"
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES"; do
  rm -rf /tmp/pmcx
  timeout 200 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmcx -o p -- python $GRAFT_REPO_ROOT/tools/cm_coresidency.py 2 768 --only=sync3 --exp=21 > /dev/null 2>/tmp/pmcx.err || tail -3 /tmp/pmcx.err
  DB=$(find /tmp/pmcx -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --pmc "$DB" "rocprofv3 --kernel-trace --pmc $SET -- python tools/cm_coresidency.py 2 768 --only=sync3 --exp=21" | grep -v "k_bwt\|k_rs_\|k_scan\|k_vlc\|at::\|rocclr" | head -10 | tee -a $GRAFT_REPO_ROOT/$OUT/pmc_decoder_exp21.txt
done
rm -rf /tmp/pmcx
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES -d /tmp/pmcx -o p -- python $GRAFT_REPO_ROOT/bench.py --blocks 768 --block-mib 2 --no-cpu-baseline --no-extras --steps 1 > /dev/null 2>/tmp/pmcx.err || tail -3 /tmp/pmcx.err
DB=$(find /tmp/pmcx -name "*.db" | head -1)
[ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --pmc "$DB" "rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES -- python bench.py --blocks 768 --block-mib 2 --no-cpu-baseline --no-extras --steps 1" | grep "k_cm_\|counter" | head -12 | tee $GRAFT_REPO_ROOT/$OUT/pmc_batch_lds.txt
cd $GRAFT_REPO_ROOT
