"""Differential fuzzer (CPU, emulator) for the LZP decoder (lzp.hip k_lzp_decode) against the checker: sources with long repeats (254-chains in the match
lengths), sprinkled 0xF2 literals, runs; whole, truncated, mutated streams and output caps; 5-70 KB, i.e. several 16 KiB chunks.  Not collected by pytest:
    python tests/fuzz_lzp_decode.py <seed> <seconds>"""
import sys, os, time, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "emu")]
import numpy as np
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o, g = Oracle(), bzip3_amd.StageApi(lib)
seed = int(sys.argv[1]); budget = float(sys.argv[2])
rng = np.random.default_rng(seed)
text = datagen.shakespeare()
t0 = time.time(); it = 0; bad = 0; used = 0
while time.time() - t0 < budget:
    it += 1
    n = int(rng.integers(5000, 70000))
    kind = int(rng.integers(0, 4))
    if kind == 0:   # text with long repeats (matches of several hundred bytes: 254-chains in the length)
        piece = text[int(rng.integers(0, len(text) - 3000)):][: int(rng.integers(60, 2500))]
        d = bytearray()
        while len(d) < n:
            d += piece if rng.integers(0, 3) else text[int(rng.integers(0, len(text) - 500)):][: int(rng.integers(1, 400))]
        d = bytes(d[:n])
    elif kind == 1:  # the same with 0xF2 literals sprinkled in
        piece = text[int(rng.integers(0, len(text) - 3000)):][: int(rng.integers(60, 900))]
        a = np.frombuffer((piece * (n // len(piece) + 1))[:n], dtype=np.uint8).copy()
        a[rng.integers(0, n, size=int(rng.integers(1, 200)))] = 0xF2
        d = bytes(a)
    elif kind == 2:  # long runs
        vals = rng.integers(0, 256, size=max(1, n // 300), dtype=np.uint8); lens = rng.integers(1, 900, size=len(vals))
        d = bytes(np.repeat(vals, lens)[:n])
    else:
        d = (text[int(rng.integers(0, 100000)):][: int(rng.integers(300, 20000))] * 40)[:n]
    m, lz = o.lzp_encode(d)
    if m <= 0:
        continue
    used += 1
    r1 = g.lzp_decode(lz, len(d) + 64)
    ok = r1 == o.lzp_decode(lz, len(d) + 64) and r1[1] == d
    # truncated / corrupted streams decode like the checker (same result code and bytes)
    cut = lz[: int(rng.integers(4, len(lz)))]
    ok2 = g.lzp_decode(cut, len(d) + 64) == o.lzp_decode(cut, len(d) + 64)
    a = np.frombuffer(lz, dtype=np.uint8).copy(); a[rng.integers(4, len(a), size=3)] = rng.integers(0, 256, size=3)
    mut = bytes(a)
    ok3 = g.lzp_decode(mut, len(d) + 64) == o.lzp_decode(mut, len(d) + 64)
    cap = int(rng.integers(8, len(d)))
    ok4 = g.lzp_decode(lz, cap) == o.lzp_decode(lz, cap)
    if not (ok and ok2 and ok3 and ok4):
        bad += 1
        print("MISMATCH", seed, it, kind, len(d), ok, ok2, ok3, ok4, flush=True)
        open("/tmp/fuzz_lzpd_fail_%d_%d.bin" % (seed, it), "wb").write(d)
print("fuzz lzp-decode seed", seed, "iterations", it, "with matches", used, "bad", bad, flush=True)
