"""Differential fuzzer (CPU, emulator) for the trio CM encoder (cm.hip cm_encode_trio: three blocks per workgroup, one coder wave with a lane per
block, bz3_hip_debug_cm_encode_trio): random batches of 1-8 DIFFERENT blocks -- sizes 1 .. 6000 incl. odd ones, Zipf sources, BWT output of
text, runs, noise (given up by the row cache, coded again by the whole-model kernel) -- through bz3_encode_blocks with the 40-row test cache or the
shipped 52-row one, classic and lean (in-place) states, every coded block against the oracle's.  Not collected by pytest:
    python tests/fuzz_cm_trio.py <seed> <seconds>        (BZ3_EMU_SCHED=<k> for hostile wave scheduling)"""
import sys, os, time, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'emu')]
import numpy as np
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rng = np.random.default_rng(seed)
text = datagen.shakespeare()
bs = 65 * 1024
cap = lib.bz3_bound(bs) + 64


def source():
    kind = rng.integers(0, 5)
    n = int(rng.integers(1, 6000)) if rng.integers(0, 4) else int(rng.integers(1, 80))
    if kind == 0:
        nsym = int(rng.integers(2, 256)); a = float(rng.uniform(0.2, 3.0))
        p = 1.0 / np.arange(1, nsym + 1) ** a; p /= p.sum()
        syms = rng.permutation(256)[:nsym].astype(np.uint8)
        return bytes(syms[rng.choice(nsym, size=n, p=p)])
    if kind == 1:
        off = int(rng.integers(0, len(text) - 7000)); return text[off:off + n]
    if kind == 2:
        vals = rng.integers(0, 256, size=max(1, n // 20), dtype=np.uint8); lens = rng.integers(1, 60, size=len(vals))
        return bytes(np.repeat(vals, lens)[:n]) or b"r"
    if kind == 3:
        return bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    return (text[:int(rng.integers(50, 700))] * 12)[:n] or b"t"


t0 = time.time(); it = 0; bad = 0; nblk = 0
launches0 = lib.bz3_hip_debug_cm_encode_trio(1)
while time.time() - t0 < budget:
    it += 1
    lib.bz3_hip_set_cm_mode(int(rng.choice([9, 9, 2])))
    lib.bz3_hip_set_lean_states(int(rng.integers(0, 2)))
    n = int(rng.integers(1, 9))
    blocks = [source() for _ in range(n)]
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        want = o.encode_block(d, bs)
        if lib.bz3_last_error(states[i]) != 0 or sizes[i] != want[0] or bytes(bufs[i][: sizes[i]]) != want[2]:
            bad += 1
            print('MISMATCH', seed, it, i, n, len(d), flush=True)
            open('/tmp/fuzz_trio_fail_%d_%d_%d.bin' % (seed, it, i), 'wb').write(d)
    nblk += n
    for s in states:
        lib.bz3_free(s)
print('fuzz trio seed', seed, 'batches', it, 'blocks', nblk, 'trio launches', lib.bz3_hip_debug_cm_encode_trio(0) - launches0, 'bad', bad, flush=True)
