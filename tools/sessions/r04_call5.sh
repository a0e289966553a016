#!/bin/bash
# Round 4, GPU call 5: (A) the shipped decoder (subtree-owning waves + aligned C2 halves) against round 3's library on the same box;
# (B) the encoder at one / two / three blocks per CU with counters (why does the third block cost 1.3x?); (C) a 768 x 8 MiB and a
# 768 x 32 MiB batch of the enwik8-calibrated text (nothing may be given up any more); (D) grids of the BWT's wide / tail kernels.
OUT=gpurun_out/c5
mkdir -p $OUT
echo "== A decoder: round 3 library, then HEAD (768 x 2 MiB, 768 x 8 MiB, 256 x 2 MiB)"
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles --lib=bzip3_amd/lib/ab/libbzip3_r03.so 2>&1 | grep variant | tee $OUT/dec_r03.txt
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles 2>&1 | grep variant | tee $OUT/dec_head.txt
timeout 300 python tools/cm_coresidency.py 8 768 --only=sync3 --lib=bzip3_amd/lib/ab/libbzip3_r03.so 2>&1 | grep variant | tee -a $OUT/dec_r03.txt
timeout 300 python tools/cm_coresidency.py 8 768 --only=sync3 2>&1 | grep variant | tee -a $OUT/dec_head.txt
timeout 300 python tools/cm_coresidency.py 2 256 512 --only=sync,sync2 2>&1 | grep variant | tee -a $OUT/dec_head.txt
echo "== B encoder HEAD: full / rows / rows3 at 256 512 768 copies of a 2 MiB block"
timeout 300 python tools/cm_encode_split.py 2 256 512 768 2>&1 | grep encoder | tee $OUT/enc_split_head.txt
echo "== B encoder counters, rows3 kernel at 256 and 768 copies"
cd /tmp && export TMPDIR=/tmp
cat > /tmp/enc_only.py <<'P'
import sys, os, ctypes as C
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests")]
import bzip3_amd, datagen
lib = bzip3_amd.load(); g = bzip3_amd.StageApi(lib)
n = 2 << 20
lib.bz3_hip_set_cm_mode(0)
plain = g.bwt(datagen.text(n, seed=5, chains=2048))[1]
lib.bz3_hip_set_cm_mode(2)
inb = bzip3_amd._cbuf(plain, n); out = (C.c_uint8 * (lib.bz3_bound(n) + 64))(); coded = C.c_int32(0)
k = int(sys.argv[1])
ms = lib.bz3_hip_stage_cm_encode_many(inb, n, out, C.byref(coded), k)
print("copies", k, "ms", ms, "ns/B", ms * 1e6 / n, "coded", coded.value)
P
for K in 256 768; do
 for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH"; do
  rm -rf /tmp/pmcx
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmcx -o p -- python /tmp/enc_only.py $K > /tmp/pmcx.out 2>/tmp/pmcx.err || tail -3 /tmp/pmcx.err
  DB=$(find /tmp/pmcx -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py --pmc "$DB" "copies $K: $SET" | grep "k_cm_encode_rows3\|^# " | tee -a $GRAFT_REPO_ROOT/$OUT/pmc_encoder_rows3.txt
 done
done
cd $GRAFT_REPO_ROOT
echo "== C batches of the calibrated text"
for MIB in 8 32; do
  timeout 400 python bench.py --blocks 768 --block-mib $MIB --steps 1 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('block_mib', d['config']['block_bytes']>>20, 'value', d['value'], 'ratio', d['config']['compressed_ratio'], 'given_up', d['config']['cm_blocks_given_up'], 'repeat', d['config']['bwt_output_repeat_rate_16MiB_sample'], 't_enc', s['t_enc_s'], 'cm_enc_ms', s['enc']['cm'], 'front_end_s', round(s['t_enc_s']-s['enc']['cm']/1e3,3), 't_dec', s['t_dec_s'], 'cm_dec_ms', s['dec']['cm'], 'tail_s', round(s['t_dec_s']-s['dec']['cm']/1e3,3))" | tee -a $OUT/batches.txt
done
timeout 400 python bench.py --blocks 768 --block-mib 32 --steps 1 --no-extras --no-cpu-baseline --noise 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print('NO NOISE block_mib', d['config']['block_bytes']>>20, 'value', d['value'], 'ratio', d['config']['compressed_ratio'], 'given_up', d['config']['cm_blocks_given_up'], 'repeat', d['config']['bwt_output_repeat_rate_16MiB_sample'], 't_enc', s['t_enc_s'], 'cm_enc_ms', s['enc']['cm'], 'front_end_s', round(s['t_enc_s']-s['enc']['cm']/1e3,3), 't_dec', s['t_dec_s'], 'cm_dec_ms', s['dec']['cm'], 'tail_s', round(s['t_dec_s']-s['dec']['cm']/1e3,3))" | tee -a $OUT/batches.txt
echo "== D BWT grids (stage_probe 256: transform alone)"
for G in "" "2048,24576" "16384,262144" "65536,1048576"; do
  echo "BZ3_BWT_GRIDS=$G"; BZ3_BWT_GRIDS=$G timeout 200 python tools/stage_probe.py 256 --noise=0.035 2>&1 | grep rep1 | tee -a $OUT/bwt_grids.txt
done
