"""TEST INFRASTRUCTURE ONLY: differential fuzzer of the batch paths under the CPU emulator (tests/emu) against the oracle -- random
batches (stored / plain / LZP-coded / run-length-coded / incompressible blocks of random sizes) through random shapes of the
encoder's front-end ring and the decoder's tail ring (BZ3_HIP_LZP_PIPE, BZ3_HIP_TAIL_PIPE), classic and lean states, the suffix sorter's
big groups through 0 / 1 / 8 more windows before the deep path (bz3_hip_debug_bwt_big_rounds), workspaces kept or handed back (BZ3_HIP_WS_KEEP_MB).
    python tests/fuzz_rings.py <seed> <minutes>"""
import ctypes as C
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE, os.path.join(HERE, "emu")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402
from build_emu import build  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def make_block(rng, text):
    kind = rng.random()
    size = int(rng.choice([rng.randint(0, 70), rng.randint(64, 2000), rng.randint(2000, 12000), rng.randint(12000, 40000)]))
    if kind < 0.35:
        o = rng.randint(0, len(text) - size - 1)
        return text[o : o + size]
    if kind < 0.55:  # repeats: LZP applies
        o = rng.randint(0, len(text) - 4000)
        unit = text[o : o + rng.randint(50, 900)]
        return (unit * (size // max(1, len(unit)) + 1))[:size]
    if kind < 0.7:  # runs: mRLE applies
        return b"".join(bytes([rng.randint(0, 255)]) * rng.randint(1, 600) for _ in range(size // 200 + 1))[:size]
    if kind < 0.85:
        return bytes(rng.getrandbits(8) for _ in range(size))
    return bytes(rng.choice(b"ab\xf2\xff\x00") for _ in range(size))


def main():
    seed, minutes = int(sys.argv[1]), float(sys.argv[2])
    rng = random.Random(seed)
    lib = bzip3_amd._declare(C.CDLL(build()))
    o = Oracle()
    text = datagen.shakespeare()
    bs = 65 * 1024
    t_end = time.time() + 60 * minutes
    it = blocks_done = 0
    while time.time() < t_end:
        it += 1
        n = rng.choice([1, 2, 3, 5, 8, 13, 21, rng.randint(1, 30)])
        env = {}
        if rng.random() < 0.7:
            env["BZ3_HIP_LZP_PIPE"] = "%d,%d" % (rng.randint(1, 9), rng.randint(2, 4))
        if rng.random() < 0.7:
            env["BZ3_HIP_TAIL_PIPE"] = "%d,%d" % (rng.randint(1, 9), rng.randint(2, 4))
        lib.bz3_hip_debug_bwt_big_rounds(rng.choice([-1, -1, 0, 8]))
        if rng.random() < 0.3:
            env["BZ3_HIP_WS_KEEP_MB"] = "0"
        for k in ("BZ3_HIP_LZP_PIPE", "BZ3_HIP_TAIL_PIPE", "BZ3_HIP_WS_KEEP_MB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        lean = rng.random() < 0.5
        assert lib.bz3_hip_set_lean_states(1 if lean else 0) == 0
        blocks = [make_block(rng, text) for _ in range(n)]
        states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
        cap = lib.bz3_bound(bs) + 64
        bufs = [(C.c_uint8 * cap)() for _ in range(n)]
        for b, d in zip(bufs, blocks):
            C.memmove(b, d, len(d))
        ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
        sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
        lib.bz3_encode_blocks(states, ptrs, sizes, n)
        for i, d in enumerate(blocks):
            want = o.encode_block(d, bs)
            got = (sizes[i], lib.bz3_last_error(states[i]) if len(d) >= 64 else want[1], bytes(bufs[i][: max(0, sizes[i])]))
            if got != (want[0], want[1], want[2]):
                open("/tmp/fuzz_rings_fail_%d_%d_%d.bin" % (seed, it, i), "wb").write(d)
                raise SystemExit("MISMATCH encode seed %d it %d block %d env %r lean %r" % (seed, it, i, env, lean))
        bsz = (C.c_size_t * n)(*[cap] * n)
        orig = (C.c_int32 * n)(*[len(d) for d in blocks])
        lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
        for i, d in enumerate(blocks):
            if lib.bz3_last_error(states[i]) != 0 and len(d) >= 64 or bytes(bufs[i][: len(d)]) != d:
                open("/tmp/fuzz_rings_fail_%d_%d_%d.bin" % (seed, it, i), "wb").write(d)
                raise SystemExit("MISMATCH decode seed %d it %d block %d env %r lean %r" % (seed, it, i, env, lean))
        for s in states:
            lib.bz3_free(s)
        blocks_done += n
        if it % 10 == 0:
            print("seed %d: %d batches, %d blocks, no mismatch" % (seed, it, blocks_done), flush=True)
    print("seed %d DONE: %d batches, %d block round trips, no mismatch" % (seed, it, blocks_done), flush=True)


if __name__ == "__main__":
    main()
