"""Differential fuzzer (CPU, emulator): every CM kernel variant (whole model, row caches of 96 / 44 / 56 / 40 slots) against the oracle on random sources (Zipf alphabets of 2-255 symbols, BWT output of text, runs, noise),
truncated streams included, plus block round trips with classic and lean states.  Not collected by pytest:
    python tests/fuzz_cm_variants.py <seed> <seconds>
Round 1: seeds 1-3 x 2400 s = 2,389 stage iterations + 341 block round trips, 0 mismatches.
Round 2 (sync / solo decoders, structure-of-arrays encoder ring added): seeds 61-66 x 2400 s, four of them with BZ3_EMU_SCHED = -3 / -9 / 7 / -21:
4,541 stage iterations + 648 block round trips, 0 mismatches."""
import sys, os, time, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'emu')]
import numpy as np
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle(); g = bzip3_amd.StageApi(lib)
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time(); it = 0; bad = 0
text = datagen.shakespeare()
while time.time() - t0 < budget:
    it += 1
    kind = rng.integers(0, 4)
    n = int(rng.integers(1, 3500))
    if kind == 0:
        nsym = int(rng.integers(2, 256)); a = float(rng.uniform(0.2, 3.0))
        p = 1.0/np.arange(1, nsym+1)**a; p /= p.sum()
        syms = rng.permutation(256)[:nsym].astype(np.uint8)
        d = bytes(syms[rng.choice(nsym, size=n, p=p)])
    elif kind == 1:
        off = int(rng.integers(0, len(text) - 4000)); d = o.bwt(text[off:off+max(n,2)])[1]
    elif kind == 2:  # runs
        vals = rng.integers(0, 256, size=max(1, n // 20), dtype=np.uint8); lens = rng.integers(1, 60, size=len(vals))
        d = bytes(np.repeat(vals, lens)[:n])
    else:
        d = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
    if not d: continue
    mode = int(rng.choice([9, 9, 9, 0, 1, 2]))  # tiny cache (recycles slots all the time), whole model, the shipped caches
    lib.bz3_hip_set_cm_mode(mode)
    c = o.cm_encode(d)
    e = g.cm_encode(d); dd = g.cm_decode(c, len(d))
    cut = c[: int(rng.integers(0, len(c) + 1))]
    tr = g.cm_decode(cut, len(d)) == o.cm_decode(cut, len(d))
    if not (e == c and dd == d and tr):
        bad += 1
        print('MISMATCH', seed, it, kind, mode, len(d), e == c, dd == d, tr, flush=True)
        open('/tmp/fuzz_fail_%d_%d.bin' % (seed, it), 'wb').write(d)
    # block level, lean in-place, now and then
    if it % 7 == 0:
        lib.bz3_hip_set_lean_states(int(rng.integers(0, 2)))
        blk = (text[:int(rng.integers(100, 900))] * int(rng.integers(1, 6)) + d)[:6000]
        bs = 65 * 1024
        with bzip3_amd.State(bs, lib) as st:
            a = st.encode_block(blk); ref = o.encode_block(blk, bs)
            r = st.decode_block(a[2], len(blk)) if a[0] > 0 else (len(blk), 0, blk)
        if not (a == ref and r[2] == blk):
            bad += 1; print('BLOCK MISMATCH', seed, it, mode, len(blk), flush=True)
            open('/tmp/fuzz_failblk_%d_%d.bin' % (seed, it), 'wb').write(blk)
        lib.bz3_hip_set_lean_states(0)
print('fuzz seed', seed, 'iterations', it, 'bad', bad, flush=True)
