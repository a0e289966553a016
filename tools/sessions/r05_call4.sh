#!/bin/bash
# Round 5, GPU call 4: the CU partition (api.hip DeviceCtx: the side streams' LZP kernels on 32 CUs of their own, the tail's whole-GPU kernels on the
# other 224).  256 x 64 MiB, one step each, BZ3_HIP_TRACE_RINGS=1: (a) BZ3_HIP_CU_RESERVE=0 (no partition: call 3's figures), (b) the default
# (reserve 32, tail on the rest), (c) reserve 64, (d) reserve 32 + the encoder's front end on the rest as well (BZ3_HIP_FRONT_REST=1).
OUT=$(realpath -m "$1"); REPO=$(cd "$(dirname "$0")/../.." && pwd); mkdir -p "$OUT"; cd "$REPO"
run() {  # name, env...
  local name=$1; shift
  env "$@" BZ3_HIP_TRACE_RINGS=1 timeout 300 python bench.py --gpus 1 --blocks 256 --block-mib 64 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  echo "== $name: $*"
  grep "bz3 rings" "$OUT/bench_$name.log" | grep -v " 1 blocks"
  python -c "import json;d=json.load(open('$OUT/bench_$name.json'));s=d['stages'];print('value',d['value'],'t_enc',s['t_enc_s'],'t_dec',s['t_dec_s'],'cm',round(s['enc']['cm']/1e3,2),round(s['dec']['cm']/1e3,2),'front',round(s['t_enc_s']-s['enc']['cm']/1e3,2),'tail',round(s['t_dec_s']-s['dec']['cm']/1e3,2))"
}
run reserve0 BZ3_HIP_CU_RESERVE=0
run reserve32 BZ3_HIP_CU_RESERVE=32
run reserve64 BZ3_HIP_CU_RESERVE=64
run reserve32_front BZ3_HIP_CU_RESERVE=32 BZ3_HIP_FRONT_REST=1
echo "== parity with the partition on (rings, lean states, batch api)"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rings_on_gpu or lean_states or batch_api or device_resident" 2>&1 | tail -3
