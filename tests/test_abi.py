"""CPU tests of the drop-in boundary: the product .so loads, exports every symbol include/*.h declares,
its pure host-side entry points behave like the reference's, and it refuses to work without a GPU."""
import ctypes as C
import os
import re

import pytest

import bzip3_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for h in ("libbz3.h", "bz3_hip.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"BZIP3_API\s+[^;(]*?\b(bz3_\w+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    from bzip3_amd.build import build

    build()  # hipcc cross-compiles gfx950 without a GPU
    return bzip3_amd.load()


def test_exports_every_declared_symbol(lib):
    names = _declared_symbols()
    ref14 = ["bz3_version", "bz3_last_error", "bz3_strerror", "bz3_new", "bz3_free", "bz3_bound", "bz3_compress", "bz3_decompress",
             "bz3_min_memory_needed", "bz3_encode_block", "bz3_decode_block", "bz3_encode_blocks", "bz3_decode_blocks",
             "bz3_orig_size_sufficient_for_decode"]  # include/libbz3.h:62-235 of the reference
    assert set(ref14) <= set(names)
    assert len(names) >= 14 + 15
    raw = C.CDLL(bzip3_amd.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"libbzip3.so does not export {n}"


def test_pure_host_entry_points(lib, ref_lib):
    for n in (0, 1, 49, 50, 64, 65 * 1024, 1 << 20, (511 << 20)):
        assert lib.bz3_bound(n) == n + n // 50 + 32 == ref_lib.lib.bz3_bound(n)
    assert lib.bz3_version().startswith(b"1.5.2")
    assert lib.bz3_min_memory_needed(1000) == 0 == ref_lib.lib.bz3_min_memory_needed(1000)
    assert lib.bz3_min_memory_needed((511 << 20) + 1) == 0
    assert lib.bz3_min_memory_needed(1 << 20) > lib.bz3_bound(1 << 20)
    # bz3_orig_size_sufficient_for_decode: header-only logic (src/libbz3.c:1025-1055)
    import struct
    hdrs = [struct.pack("<IiB", 1, -1, 0), struct.pack("<IiBI", 1, 5, 2, 1000), struct.pack("<IiBII", 1, 5, 6, 900, 1200),
            struct.pack("<IiBI", 1, 5, 4, 5000), b"\0" * 8, struct.pack("<IiB", 1, 5, 6) + b"\0" * 7]
    for h in hdrs:
        for orig in (0, 999, 1000, 1200, 10 ** 6):
            buf = (C.c_uint8 * max(1, len(h))).from_buffer_copy(h)
            assert lib.bz3_orig_size_sufficient_for_decode(buf, len(h), orig) == ref_lib.lib.bz3_orig_size_sufficient_for_decode(buf, len(h), orig)


def test_block_size_validation_needs_no_device(lib):
    assert not lib.bz3_new(65 * 1024 - 1)
    assert not lib.bz3_new((511 << 20) + 1)


def test_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.bz3_hip_device_count() == 0
    assert not lib.bz3_new(1 << 20)  # NULL, never a CPU fallback
    with pytest.raises(RuntimeError):
        bzip3_amd.State(1 << 20)
    out = (C.c_size_t)(100)
    src = (C.c_uint8 * 100)()
    dst = (C.c_uint8 * 1000)()
    assert lib.bz3_compress(1 << 20, src, dst, 100, C.byref(out)) == bzip3_amd.BZ3_ERR_INIT


def test_missing_extension_raises(tmp_path):
    with pytest.raises(RuntimeError):
        bzip3_amd.load(str(tmp_path / "nope.so"))


def test_product_never_touches_the_oracle():
    # only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bzip3_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                s = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in s and "oracle_lib" not in s and "libbz3ref" not in s and "libbz3_emu" not in s, f


def test_prototypes_equal_the_reference_header_when_it_is_here():
    """include/libbz3.h is written for this implementation (reference sources stay out of the repository), so its 14 prototypes are
    compared with the reference's own header token for token wherever the reference tree exists (the build container; not the GPU box)."""
    import re

    ref_path = "/root/reference/include/libbz3.h"
    if not os.path.exists(ref_path):
        pytest.skip("the reference tree is not on this machine")

    def protos(path):
        text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        out = {}
        for m in re.finditer(r"BZIP3_API\s+([^;{]+);", text):
            decl = re.sub(r"\s+", " ", m.group(1)).strip()
            decl = re.sub(r"\s*([(),*])\s*", r"\1", decl)
            name = re.search(r"(bz3_\w+)\(", decl).group(1)
            out[name] = decl
        return out

    ours, theirs = protos(os.path.join(ROOT, "include", "libbz3.h")), protos(ref_path)
    assert sorted(ours) == sorted(theirs) and len(theirs) == 14
    for name in theirs:
        a = re.sub(r"\b\w+(?=[,)])", "", ours[name])    # parameter names may differ, types and order may not
        b = re.sub(r"\b\w+(?=[,)])", "", theirs[name])
        assert a == b, (name, ours[name], theirs[name])
    for macro in ("BZ3_OK", "BZ3_ERR_OUT_OF_BOUNDS", "BZ3_ERR_BWT", "BZ3_ERR_CRC", "BZ3_ERR_MALFORMED_HEADER", "BZ3_ERR_TRUNCATED_DATA", "BZ3_ERR_DATA_TOO_BIG",
                  "BZ3_ERR_INIT", "BZ3_ERR_DATA_SIZE_TOO_SMALL"):
        val = lambda p: re.search(r"#define\s+%s\s+(-?\d+)" % macro, open(p).read()).group(1)
        assert val(os.path.join(ROOT, "include", "libbz3.h")) == val(ref_path), macro


def test_cu_partition_masks_are_disjoint_and_balanced(lib):
    """The decoder's CU partition (api.hip DeviceCtx::cu_masks): the reserved CUs and the rest are disjoint, cover the device, and the reserved ones are
    spread evenly whichever way the driver deals mask bits to the eight XCDs -- in blocks of 32 bits or round robin (bit i -> XCD i mod 8)."""
    import ctypes as C

    for cus, reserve in ((256, 48), (256, 32), (256, 64), (256, 16), (128, 32)):
        words = (cus + 31) // 32
        side = (C.c_uint32 * words)()
        rest = (C.c_uint32 * words)()
        assert lib.bz3_hip_debug_cu_masks(cus, reserve, side, rest) == words
        s_bits = {32 * w + b for w in range(words) for b in range(32) if (side[w] >> b) & 1}
        r_bits = {32 * w + b for w in range(words) for b in range(32) if (rest[w] >> b) & 1}
        blocks = min(8, cus // 32)
        per = min(8, max(1, reserve // 8))
        assert not (s_bits & r_bits) and (s_bits | r_bits) == set(range(cus))
        assert len(s_bits) == per * blocks
        assert all(sum(1 for i in s_bits if i // 32 == a) == per for a in range(blocks))          # dealt in blocks of 32
        if blocks == 8:
            assert all(sum(1 for i in s_bits if i % 8 == x) == per for x in range(8))              # dealt round robin


def test_headroom_rule_sizing_arithmetic(lib):
    """the headroom rule (include/bz3_hip.h bz3_hip_set_workspace_headroom; api.hip ring_contexts_for / DeviceCtx::arena_slack).  The ring of
    LZP contexts is sized so that the arena it lives in -- request + slack -- and the swap buffers its blocks borrow leave `headroom` bytes of the device free;
    round 5 had a fixed 6 GiB margin that an unbounded 1/16 slack (5 GB of an 80 GB arena) consumed."""
    GiB = 1 << 30
    assert lib.bz3_hip_workspace_headroom() == 4 * GiB  # the default
    lib.bz3_hip_set_workspace_headroom(3 * GiB)
    assert lib.bz3_hip_workspace_headroom() == 3 * GiB
    lib.bz3_hip_set_workspace_headroom(-1)
    assert lib.bz3_hip_workspace_headroom() == 4 * GiB
    # the slack is a sixteenth, capped at 512 MiB (+ 1 MiB)
    assert lib.bz3_hip_debug_arena_slack(16 << 20) == (1 << 20) + (1 << 20)
    assert lib.bz3_hip_debug_arena_slack(80 * GiB) == (512 << 20) + (1 << 20)
    # the bench's regime: 768 lean states of 256 MiB, 91 GiB of the device free when the encode call starts
    n = 256 << 20
    need, ctx, cap = lib.bz3_hip_debug_workspace_bytes(n, 0), lib.bz3_hip_debug_workspace_bytes(n, 1), lib.bz3_bound(n)
    assert 14 * GiB < need < 17 * GiB and 2 * GiB < ctx < 2.5 * GiB
    fixed = 768 * (64 + 65536 + 256 + 131072 + 512) + (1 << 20)
    for headroom in (0, 4 * GiB, 6 * GiB, 8 * GiB, 40 * GiB):
        for free in (91 * GiB, 60 * GiB, 30 * GiB, 18 * GiB):
            for have in (0, 20 * GiB):
                c = lib.bz3_hip_debug_ring_contexts(free, have, need, fixed, ctx, cap, 1, headroom)
                arena = need + fixed + c * (ctx + 1024)
                held = arena + lib.bz3_hip_debug_arena_slack(arena) + c * cap  # the arena as arena_for allocates it + one borrowed swap buffer per context
                if c > 0:
                    assert free + have - held >= headroom, (headroom, free, have, c)  # the rule holds by construction ...
                    c1 = c + 1
                    arena1 = need + fixed + c1 * (ctx + 1024)
                    assert free + have - (arena1 + lib.bz3_hip_debug_arena_slack(arena1) + c1 * cap) < headroom + (128 << 20)  # ... and nothing much is left unused
    # 91 GiB free, the bench's headroom: a ring of 4 x 7 as in rounds 4-5 (pipeline_shape: window = contexts / 4)
    assert lib.bz3_hip_debug_ring_contexts(91 * GiB, 0, need, fixed, ctx, cap, 1, 8 * GiB) // 4 == 7
    # classic states own their swap buffers: 7/10 of what is free, but never into the headroom
    assert lib.bz3_hip_debug_ring_contexts(10 * GiB, 0, 1 * GiB, 0, 1 * GiB, 0, 0, 0) == 6
    assert lib.bz3_hip_debug_ring_contexts(10 * GiB, 0, 1 * GiB, 0, 1 * GiB, 0, 0, 4 * GiB) == 4
    assert lib.bz3_hip_debug_ring_contexts(1 * GiB, 0, 2 * GiB, 0, 1 * GiB, 0, 1, 0) == 0
