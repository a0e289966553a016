#!/bin/bash
# Round 6, call 9: the encoder's two-thread front end (bz3_hip_set_front_end_duo) -- parity on real streams first, then one full-size step with it and one without on the same box.
OUT=gpurun_out/c9
mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_thread_front_end or rings_on_gpu or large_lean_batch" 2>&1 | tail -6 | tee $OUT/pytest_duo.log
for duo in 1 0; do
  BZ3_BENCH_FRONT_DUO=$duo BZ3_HIP_TRACE_RINGS=1 timeout 700 python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/full_duo$duo.json 2> $OUT/full_duo$duo.log
  echo "== BZ3_BENCH_FRONT_DUO=$duo rc=$?" | tee -a $OUT/summary.txt
  grep "^\[bench\|bz3 rings" $OUT/full_duo$duo.log | grep -v " 1 blocks" | tee -a $OUT/summary.txt
  python3 -c "import json;d=json.loads(open('$OUT/full_duo$duo.json').read().strip().splitlines()[-1]);print('value',d.get('value'),'error',d.get('error'),json.dumps(d.get('stages',{}).get('front_end_ring')))" | tee -a $OUT/summary.txt
done
