#!/bin/bash
# Round 4, GPU call 1: (A) the decoder experiments (subtree-owning model waves, lanes off the guessed path skip) at three / one block per CU,
# (B) the encoder with the LDS-staged sink against round 3's library at one / two / three per CU, (C) GPU parity of the CM paths with the sink,
# (D) PMC passes (WRITE_SIZE per coded byte was 3.2), (E) what a busy host does to the front end, (F) clocks while the CM kernels run.
OUT=gpurun_out/c1
mkdir -p $OUT
(for i in $(seq 1 150); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -2 | tr '\n' ' '; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 2; done) > $OUT/clocks.txt 2>&1 &
SMI=$!
echo "== A decoder experiments, 768 x 2 MiB (three per CU), cycle counters"
timeout 300 python tools/cm_coresidency.py 2 768 --only=sync3 --cycles --exp=0,1,2,3 2>&1 | tee $OUT/dec_exp_768.txt
echo "== A one per CU (whole model)"
timeout 200 python tools/cm_coresidency.py 2 256 --only=sync --cycles --exp=0,3 2>&1 | tee $OUT/dec_exp_256.txt
echo "== A two per CU, rows3 kernel at 512"
timeout 200 python tools/cm_coresidency.py 2 512 --only=sync3 --cycles --exp=0,3 2>&1 | tee $OUT/dec_exp_512.txt
echo "== B encoder: round 3 library"
timeout 200 python tools/cm_encode_split.py 2 256 512 768 --lib=bzip3_amd/lib/ab/libbzip3_r03.so 2>&1 | grep encoder | tee $OUT/enc_split_r03.txt
echo "== B encoder: staged sink"
timeout 200 python tools/cm_encode_split.py 2 256 512 768 2>&1 | grep encoder | tee $OUT/enc_split_sink.txt
kill $SMI 2>/dev/null
echo "== C parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or cm_row_cache or three_blocks or lean or golden or block_parity or in_place" 2>&1 | tail -3 | tee $OUT/parity_sink.txt
echo "== D PMC"
timeout 400 bash tools/pmc_pass.sh $OUT/pmc 2>&1 | tail -3
cp profiles/pmc_traffic.json $OUT/pmc_traffic_new.json 2>/dev/null
echo "== E host contention"
timeout 300 python tools/host_contention.py 8 768 2>$OUT/host_contention.err | tee $OUT/host_contention.json
tail -3 $OUT/host_contention.err
echo "== F clocks"; sort $OUT/clocks.txt | uniq -c | sort -rn | head -8
