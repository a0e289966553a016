"""Frame-API parity cases shared by the emulator suite and the GPU suite: a libbz3.h handle under test is compared with
the REAL reference (oracle/_ref/libbz3ref.so) on good and malformed multi-block frames."""
import ctypes as C

import pytest

from oracle_lib import RefLib, require_ref


def frame_calls(L, bs, data, mutate=None, out_room=None):
    """(compress rc, frame bytes, decompress rc, decoded bytes committed) through a libbz3.h handle."""
    out = (C.c_uint8 * (L.bz3_bound(len(data)) + 64))()
    osz = C.c_size_t(len(out))
    rc = L.bz3_compress(bs, data, out, len(data), C.byref(osz))
    frame = bytes(out[: osz.value])
    if rc != 0:
        return rc, frame, None, None
    if mutate:
        frame = mutate(frame)
    room = len(data) + 16 if out_room is None else out_room
    back = (C.c_uint8 * max(room, 1))()
    bsz = C.c_size_t(room)
    rc2 = L.bz3_decompress(frame, back, len(frame), C.byref(bsz))
    return rc, frame, rc2, C.string_at(back, bsz.value)


def check(lib, five, bs, only=None):
    """`five`: data that makes 5 chunks at block size bs (the last one short)."""
    ref = require_ref()
    assert 4 * bs < len(five) < 5 * bs
    exact = five[: 2 * bs]  # multiple of the block size: the (sic) empty last chunk of src/libbz3.c:914
    for data in ((five, exact, five[:100], b"") if only is None else (five, exact)):
        a, b = frame_calls(lib, bs, data), frame_calls(ref.lib, bs, data)
        assert a == b, ("good frame", len(data), a[0], b[0], a[2], b[2])

    def cut(k):
        return lambda f: f[: len(f) - k]

    def flip(pos):
        return lambda f: f[:pos] + bytes([f[pos] ^ 0x40]) + f[pos + 1 :]

    def poke32(pos, v):
        return lambda f: f[:pos] + int(v).to_bytes(4, "little") + f[pos + 4 :]

    good = frame_calls(ref.lib, bs, five)[1]
    n0 = int.from_bytes(good[13:17], "little")  # compressed size of chunk 0
    second = 13 + 8 + n0                          # header of chunk 1
    n1 = int.from_bytes(good[second : second + 4], "little")
    muts = {"cut1": cut(1), "cut9": cut(9), "cut_big": cut(len(good) // 3), "flip_chunk0": flip(13 + 8 + min(30, n0 - 1)),
            "flip_chunk1": flip(second + 8 + min(100, n1 - 1)), "size_huge": poke32(second, 0x7FFFFFFF), "size_over_block": poke32(second, bs + 1),
            "size_plus1": poke32(second, n1 + 1), "orig_negative": poke32(second + 4, 0xFFFFFFFF), "orig_small": poke32(second + 4, 10),
            "n_blocks_9": poke32(9, 9), "n_blocks_max": poke32(9, 0xFFFFFFFF), "n_blocks_2": poke32(9, 2), "block_size_bad": poke32(5, 1000),
            "magic": flip(0)}
    for name, m in muts.items():
        if only is not None and name not in only:
            continue
        a, b = frame_calls(lib, bs, five, m), frame_calls(ref.lib, bs, five, m)
        assert a[2] == b[2] and a[3] == b[3], (name, a[2], b[2], len(a[3]), len(b[3]))
    a, b = frame_calls(lib, bs, five, None, out_room=3 * bs), frame_calls(ref.lib, bs, five, None, out_room=3 * bs)
    assert a[2] == b[2] and a[3] == b[3], ("small output", a[2], b[2], len(a[3]), len(b[3]))
