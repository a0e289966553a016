"""Differential fuzzer (CPU, emulator): the mRLE encoder and decoder kernels (mrle.hip) against the oracle on run-structured inputs (run lengths around 1, 254-257, 509-511,
flagged and unflagged values, long stretches that pass through unchanged -- the 16-byte fast paths of round 6 -- with single runs inside them), and the decoder on
truncated and arbitrary streams (a length sequence cut off by the end of the stream: the reference's quirk at src/libbz3.c:320-322).  Not collected by pytest:
    python tests/fuzz_mrle.py <seed> <seconds>
Round 6: seeds 1201 / 1202 x 150 s, 0 mismatches."""
import sys, os, time, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'emu')]
import numpy as np
import bzip3_amd
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle(); g = bzip3_amd.StageApi(lib)
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rng = np.random.default_rng(seed)
t0 = time.time(); it = 0; bad = 0
while time.time() - t0 < budget:
    it += 1
    kind = int(rng.integers(0, 4))
    if kind == 0:    # runs of special lengths
        vals = rng.integers(0, 256, size=int(rng.integers(1, 60)), dtype=np.uint8)
        lens = rng.choice([1, 1, 1, 2, 3, 16, 17, 254, 255, 256, 257, 509, 510, 511, 765], size=len(vals))
        d = bytes(np.repeat(vals, lens))
    elif kind == 1:  # mostly plain bytes (the fast paths) with a few runs
        n = int(rng.integers(17, 20000))
        a = rng.integers(32, 127, size=n, dtype=np.uint8)
        a[1:] = np.where(a[1:] == a[:-1], a[1:] ^ 1, a[1:])  # no accidental runs
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, n)); l = int(rng.integers(2, 600)); a[p : p + l] = a[p]
        d = bytes(a)
    elif kind == 2:  # few symbols, many runs
        n = int(rng.integers(1, 9000))
        d = bytes(np.repeat(rng.integers(0, 4, size=n // 3 + 1, dtype=np.uint8) * 85, rng.integers(1, 9, size=n // 3 + 1))[:n])
    else:
        d = bytes(rng.integers(0, 256, size=int(rng.integers(1, 5000)), dtype=np.uint8))
    if not d:
        continue
    e_o, e_g = o.mrle_encode(d), g.mrle_encode(d)
    if e_o != e_g:
        bad += 1; print("ENCODE mismatch", it, kind, len(d)); continue
    # decode: the genuine stream, truncated streams, streams with garbage, and wrong output lengths
    for trial in range(4):
        s = bytearray(e_o)
        outlen = len(d)
        if trial == 1 and len(s) > 33:
            s = s[: int(rng.integers(33, len(s)))]
        elif trial == 2 and len(s) > 40:
            for _ in range(3):
                s[int(rng.integers(0, len(s)))] = int(rng.integers(0, 256))
        elif trial == 3:
            outlen = max(1, len(d) + int(rng.integers(-20, 20)))
        r_o, r_g = o.mrle_decode(bytes(s), outlen), g.mrle_decode(bytes(s), outlen)
        if r_o[0] != r_g[0] or (r_o[0] == 0 and r_o[1] != r_g[1]):
            bad += 1; print("DECODE mismatch", it, kind, trial, len(d), len(s), outlen, r_o[0], r_g[0])
print(f"fuzz seed {seed} iterations {it} bad {bad}")
