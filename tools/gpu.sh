#!/bin/bash
# Every GPU call of round 6 goes through here: rebuild the product, check the ABI (exports == include/*.h == the ctypes table), then gpurun.
# (Round 6's first call shipped a stale libbzip3.so: a symbol added to the header and the binding after the last build.)
#   tools/gpu.sh <timeout-seconds> <session script>
set -e
cd "$(dirname "$0")/.."
python bzip3_amd/build.py >/dev/null
python -m pytest tests/test_abi.py -x -q -p no:cacheprovider 2>&1 | tail -1
make -C oracle >/dev/null 2>&1 || true
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "bash $2"
