"""TEST INFRASTRUCTURE ONLY: body of tests/sanitize_emu.sh -- stage parity and batches through every ring shape of the front end / decode tail
(classic and lean states) against the oracle, on the ASan + UBSan build of the emulator.  python tests/sanitize_run.py <lib.so>"""
import os, sys, ctypes as C
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, 'emu')]
import bzip3_amd, datagen
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(sys.argv[1]))
g = bzip3_amd.StageApi(lib)
o = Oracle()
t = datagen.shakespeare()
# stage parity on a few inputs (LZP links layout, BWT assign, unBWT splitters)
cases = [t[1000:1000 + 4096 * 3], t[5000:5000 + 4096 * 2 + 17], t[:4097], t[:300], b"ab", datagen.random_bytes(9000),
         datagen.low_entropy(12000), datagen.repeats(30000), bytes(8192), bytes(range(256)) * 40, (t[100:400] * 40)]
for d in cases:
    assert g.lzp_encode(d) == o.lzp_encode(d), len(d)
    assert g.bwt(d) == o.bwt(d), len(d)
    idx, u = o.bwt(d)
    assert g.unbwt(u, idx) == (0, d), len(d)
print("stages ok", flush=True)
# round 5: the sorter's hooks (raw-count scatter / scanned offsets, 8- and 9-bit digits), the one-launch scan at ragged sizes, the LZP link build through
# position bins on a block of several tiles with a ragged last one
import numpy as np
rng = np.random.default_rng(5)
for n, kb, db in ((4097, 18, 9), (70001, 18, 9), (70001, 16, 8), (128 * 4096 + 4096 + 3, 9, 9)):
    keys = rng.integers(0, 1 << kb, n, dtype=np.uint64).astype(np.uint32)
    sk = np.empty(n, dtype=np.uint32); si = np.empty(n, dtype=np.uint32)
    assert lib.bz3_hip_debug_sort_u32(keys.ctypes.data_as(C.c_void_p), n, kb, db, sk.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p)) == -(-kb // db)
    assert np.array_equal(si, np.argsort(keys, kind="stable").astype(np.uint32)), (n, kb, db)
for n in (2049, 5003, 65537, 262144, 262147):
    d = rng.integers(0, 1000, n, dtype=np.uint64).astype(np.uint32)
    want = np.zeros(n, dtype=np.uint64); want[1:] = np.cumsum(d.astype(np.uint64))[:-1]
    buf = d.copy(); tot = C.c_uint32(0)
    assert lib.bz3_hip_debug_scan_u32(buf.ctypes.data_as(C.c_void_p), n, C.byref(tot)) == 0 and np.array_equal(buf, want.astype(np.uint32)), n
for d in (t[3000:3000 + 4096 * 9 + 5] + t[3100:3100 + 20000], (t[200:1500] * 30)[:4096 * 6 - 1]):
    assert g.lzp_encode(d) == o.lzp_encode(d), len(d)
    m, lz = o.lzp_encode(d)
    if m > 0:
        assert g.lzp_decode(lz, len(d) + 64) == (len(d), d)
print("sorter / scan / binned links ok", flush=True)
for rounds in (0, 8):  # big groups straight to rank doubling / through several more windows first (test hook of the suffix sorter)
    lib.bz3_hip_debug_bwt_big_rounds(rounds)
    for d in cases + [t[777:777 + n] for n in (511, 512, 513, 1025, 2049, 4097)] + [b"a" * 2049]:
        assert g.bwt(d) == o.bwt(d), (rounds, len(d))
lib.bz3_hip_debug_bwt_big_rounds(-1)
print("sorter paths ok", flush=True)
bs = 65 * 1024
def batch(n, pipe):
    for var in ("BZ3_HIP_LZP_PIPE", "BZ3_HIP_TAIL_PIPE"):
        if pipe: os.environ[var] = pipe
        else: os.environ.pop(var, None)
    blocks = []
    for i in range(n):
        if i % 5 == 3: blocks.append((t[i * 400 : i * 400 + 150] * 3) + t[9000:9100])
        elif i % 7 == 6: blocks.append(b"x" * (20 + i))
        else: blocks.append(t[i * 400 : i * 400 + 260 + 7 * i])
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    cap = lib.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks): C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], i
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    for i, d in enumerate(blocks):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, i
    for s in states: lib.bz3_free(s)
for lean in (0, 1):
    lib.bz3_hip_set_lean_states(lean)
    for pipe in (None, "1,4", "5,3", "3,2"):
        batch(23, pipe)
        print("batch ok", lean, pipe, flush=True)
# round 6: the encoder's front end on two host threads (its second scratch region, the held swap buffers, the per-window events)
lib.bz3_hip_set_front_end_duo(1)
for lean in (0, 1):
    lib.bz3_hip_set_lean_states(lean)
    for pipe in (None, "1,4", "3,3"):
        batch(23, pipe)
        print("two-thread batch ok", lean, pipe, flush=True)
lib.bz3_hip_set_front_end_duo(-1)
print("ok")
