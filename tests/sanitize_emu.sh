#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Builds the kernel sources against the CPU emulation of the HIP execution model (tests/emu) with
# -fsanitize=address,undefined and runs tests/sanitize_run.py on it: out-of-bounds accesses in the workspace carving (ring of
# LZP context slots, interleaved links, splitter lists) and undefined arithmetic show up here, not on the GPU.  ~3 minutes.
#   bash tests/sanitize_emu.sh [build dir, default /tmp/bz3_san]
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/bz3_san}
CS=$REPO/bzip3_amd/csrc; EM=$REPO/tests/emu
mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -DBZ3_EMU -I $EM -I $CS -Wno-unknown-pragmas -Wno-attributes"
for f in sort crc32c mrle lzp bwt unbwt cm api stream; do g++ $FLAGS -x c++ -c "$CS/$f.hip" -o "$OUT/$f.o" & done
g++ $FLAGS -c "$EM/hip_emu.cpp" -o "$OUT/hip_emu.o" &
wait
g++ -shared -fsanitize=address,undefined -o "$OUT/libemu_san.so" "$OUT"/*.o -lpthread
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python "$REPO/tests/sanitize_run.py" "$OUT/libemu_san.so"
