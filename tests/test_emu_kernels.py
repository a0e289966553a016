"""CPU tests of the kernel LOGIC: the sources of bzip3_amd/csrc compiled against the test-only fiber
emulation of the HIP execution model (tests/emu), diffed against the oracle stage by stage and end to
end.  This is test infrastructure for a GPU-less container; the product library has no CPU path and the
real parity tests are the -m gpu ones."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import bzip3_amd
import datagen

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    from build_emu import build

    lib = bzip3_amd._declare(C.CDLL(build()))
    return lib


CASES = dict(datagen.nasty_cases())
CASES["text20k"] = datagen.shakespeare()[300000:320000]
CASES["rand9k"] = datagen.random_bytes(9000)
CASES["lowent"] = datagen.low_entropy(12000)
CASES["repeats"] = datagen.repeats(30000)


@pytest.mark.parametrize("name", sorted(CASES))
def test_stage_parity(emu, oracle, name):
    d = CASES[name]
    g = bzip3_amd.StageApi(emu)
    for k in {len(d), max(0, len(d) - 1), max(0, len(d) - 3), min(len(d), 5)}:
        assert g.crc32c(d[:k]) == oracle.crc32c(d[:k])
    e = oracle.mrle_encode(d)
    assert g.mrle_encode(d) == e
    assert g.mrle_decode(e, len(d)) == (0, d)
    for cut in (len(e) - 1, len(e) - 2, 40):
        if 32 <= cut <= len(e):
            a, b = g.mrle_decode(e, len(d), cut), oracle.mrle_decode(e, len(d), cut)
            assert a[0] == b[0] and (a[0] != 0 or a[1] == b[1])
    assert g.lzp_encode(d) == oracle.lzp_encode(d)
    n, z = oracle.lzp_encode(d)
    if n > 0:
        assert g.lzp_decode(z, len(d) + 100) == (len(d), d)
        assert g.lzp_decode(z, len(d) // 2) == oracle.lzp_decode(z, len(d) // 2)
    if len(d) > 1:
        assert g.bwt(d) == oracle.bwt(d)
        idx, u = oracle.bwt(d)
        assert g.unbwt(u, idx) == (0, d)
        dd = u[:2500]
        c = oracle.cm_encode(dd)
        assert g.cm_encode(dd) == c
        assert g.cm_decode(c, len(dd)) == dd
        assert g.cm_decode(c[: len(c) // 2], len(dd)) == oracle.cm_decode(c[: len(c) // 2], len(dd))


@pytest.mark.parametrize("name", ["empty", "one", "63", "64", "65", "runs", "f2", "nearmiss", "text20k", "rand9k", "repeats"])
def test_block_parity(emu, oracle, name):
    d = CASES[name]
    bs = 65 * 1024
    with bzip3_amd.State(bs, emu) as st:
        a = st.encode_block(d)
        assert a == oracle.encode_block(d, bs)
        assert st.decode_block(a[2], len(d))[:2] == (len(d), 0) or len(d) == 0
        assert st.decode_block(a[2], len(d))[2] == d


def test_cm_decode_of_truncated_stream_matches_reference_semantics(emu, oracle):
    # read_in() returns -1 past the end (src/libbz3.c:345), which can push `code` below `low`; the decoder must
    # keep comparing absolute values (caught on the GPU in round 1 with the low-entropy input)
    g = bzip3_amd.StageApi(emu)
    idx, u = oracle.bwt(datagen.low_entropy(10000))
    c = oracle.cm_encode(u)
    for frac in (2, 3, 7):
        cc = c[: len(c) // frac]
        assert g.cm_decode(cc, len(u)) == oracle.cm_decode(cc, len(u))


def test_cm_decode_of_arbitrary_bytes_matches_reference(emu, oracle):
    # The decoder must mirror decode_bytes (src/libbz3.c:436-494) on ANY input: random bytes drive it through
    # improbable symbols, long renormalisation runs and (after the end of the stream) the `code < low` states.
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(11)
    for size, n in ((0, 300), (3, 500), (400, 1500), (3000, 2500)):
        junk = bytes(rng.integers(0, 256, size=size, dtype=np.uint8))
        assert g.cm_decode(junk, n) == oracle.cm_decode(junk, n)
    skew = bytes(rng.choice(np.array([0, 255, 1, 128], dtype=np.uint8), size=1500, p=[0.6, 0.3, 0.05, 0.05]))
    assert g.cm_decode(skew, 4000) == oracle.cm_decode(skew, 4000)


def test_unbwt_of_arbitrary_bytes_matches_reference(emu, oracle):
    # libsais_unbwt is defined (up to one corner that reads unset memory) on ANY (bytes, index); what it returns for a
    # block whose payload is corrupt decides which error the caller reports (oracle pinned in test_oracle.py)
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(21)
    for trial in range(24):
        n = int(rng.integers(2, 60)) if trial % 2 else int(rng.integers(60, 9000))
        k = [1, 2, 3, 256][trial % 4]
        u = bytes(rng.integers(0, k, size=n, dtype=np.uint8))
        for idx in sorted({1, n, int(rng.integers(1, n + 1))}):
            assert g.unbwt(u, idx) == oracle.unbwt(u, idx), (n, k, idx)


def test_decoder_error_codes(emu, oracle):
    bs = 65 * 1024
    blk = oracle.encode_block(datagen.shakespeare()[:6000], bs)[2]
    muts = [blk[: len(blk) // 2], blk[:4] + b"\0\0\0\0" + blk[8:], blk[:8] + b"\x7f" + blk[9:], blk[:20] + bytes([blk[20] ^ 1]) + blk[21:],
            blk[:4] + b"\xff\xff\xff\x7f" + blk[8:], blk[:4] + b"\xfb\xff\xff\xff" + blk[8:], b"\0" * 9]
    with bzip3_amd.State(bs, emu) as st:
        for m in muts:
            assert st.decode_block(m, 6000)[:2] == oracle.decode_block(m, 6000, bs)[:2]
        for bsz, cs, osz in [(5, len(blk), 6000), (len(blk) - 1, len(blk), 6000), (70000, -5, 6000), (70000, len(blk), -1),
                             (70000, len(blk), 10 ** 9), (3000, len(blk), 6000), (70000, len(blk), 5999)]:
            assert st.decode_block(blk, osz, buffer_size=bsz, comp_size=cs)[:2] == oracle.decode_block(blk, osz, bs, buffer_size=bsz, comp_size=cs)[:2]
        n, err, _ = st.encode_block(b"x" * (bs + 1))
        assert (n, err) == (-1, bzip3_amd.BZ3_ERR_DATA_TOO_BIG)


def test_batch_api_and_frame_api(emu, oracle):
    bs = 65 * 1024
    t = datagen.shakespeare()
    blocks = [t[i * 4000 : (i + 1) * 4000] for i in range(3)] + [b"tiny"]
    n = len(blocks)
    states = (C.c_void_p * n)(*[emu.bz3_new(bs) for _ in range(n)])
    cap = emu.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        assert emu.bz3_last_error(states[i]) == 0
        assert bytes(bufs[i][: sizes[i]]) == oracle.encode_block(d, bs)[2]
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    for i, d in enumerate(blocks):
        assert emu.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d
    for s in states:
        emu.bz3_free(s)
    # frame API round trip (src/libbz3.c:876-997)
    data = (t[:3000] * 30)[: 65 * 1024 + 5000]  # two chunks; repetitive, so LZP keeps the emulated CM stage small
    out = (C.c_uint8 * (emu.bz3_bound(len(data)) + 64))()
    osz = C.c_size_t(len(out))
    assert emu.bz3_compress(bs, data, out, len(data), C.byref(osz)) == 0
    assert bytes(out[:5]) == b"BZ3v1"
    back = (C.c_uint8 * (len(data) + 16))()
    bsz2 = C.c_size_t(len(back))
    assert emu.bz3_decompress(out, back, osz.value, C.byref(bsz2)) == 0
    assert bytes(back[: bsz2.value]) == data


def test_frame_api_multi_block_matches_reference(emu):
    """bz3_compress / bz3_decompress (src/libbz3.c:876-997) run the blocks of a frame as one batch here; the frame
    bytes, the return codes and the bytes committed before an error must equal the reference's sequential loop.
    (Repetitive data: LZP collapses it, so the emulated CM stage stays small; the GPU suite runs the same cases on text.)"""
    import frame_cases

    rng = np.random.default_rng(4)
    unit = bytes(rng.integers(0, 256, size=997, dtype=np.uint8))
    frame_cases.check(emu, (unit * 400)[: 4 * 65 * 1024 + 1234], 65 * 1024,
                      only=("cut9", "flip_chunk1", "orig_small", "n_blocks_2", "magic"))


# ---- row-cache CM kernels (cm.hip, R > 0): same bytes as the full-model kernels and the oracle ---------------------
def _skewed(rng, nsym, n, a=1.3):
    p = 1.0 / np.arange(1, nsym + 1) ** a
    syms = rng.permutation(256)[:nsym].astype(np.uint8)
    return bytes(syms[rng.choice(nsym, size=n, p=p / p.sum())])


@pytest.fixture()
def cm_mode(emu):
    yield lambda m: emu.bz3_hip_set_cm_mode(m)
    emu.bz3_hip_set_cm_mode(-1)


def test_cm_row_cache_kernels_match_oracle(emu, oracle, cm_mode):
    """Mode 9 = the emulator-only 40-row instantiation (slots are recycled all the time on these inputs: eviction to
    the spill area, reload, pinning of the rows in flight); modes 1 / 2 = the shipped 96/112-row and 44/56-row kernels.  Inputs whose
    working set does not fit are given up by the kernel and coded again by the full-model kernel: same bytes."""
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(5)
    cases = {
        "text": (oracle.bwt(datagen.shakespeare()[100000:104000])[1], False),
        "skew60": (_skewed(rng, 60, 3500), False),          # 60 live rows > 40 slots: recycles, but within the miss budget
        "flat200": (_skewed(rng, 200, 2500, 0.3), True),    # thrashes: given up
        "tiny": (b"ab" * 20, False),
        "one": (b"x", False),
    }
    assert cm_mode(9) == 0
    for name, (d, gives_up) in cases.items():
        c = oracle.cm_encode(d)
        n0 = emu.bz3_hip_cm_blocks_given_up()
        assert g.cm_encode(d) == c, name
        n1 = emu.bz3_hip_cm_blocks_given_up()
        assert g.cm_decode(c, len(d)) == d, name
        n2 = emu.bz3_hip_cm_blocks_given_up()
        assert (n1 - n0, n2 - n1) == ((1, 1) if gives_up else (0, 0)), name
        assert g.cm_decode(c[: len(c) // 2], len(d)) == oracle.cm_decode(c[: len(c) // 2], len(d)), name  # truncated stream
    assert cm_mode(1) == 0
    d = _skewed(rng, 130, 4000)  # 130 live rows > 96 / 112 slots
    c = oracle.cm_encode(d)
    n0 = emu.bz3_hip_cm_blocks_given_up()
    assert g.cm_encode(d) == c and g.cm_decode(c, len(d)) == d
    assert emu.bz3_hip_cm_blocks_given_up() == n0
    assert cm_mode(2) == 0
    d = _skewed(rng, 70, 3000)  # 70 live rows > 44 / 56 slots
    c = oracle.cm_encode(d)
    assert g.cm_encode(d) == c and g.cm_decode(c, len(d)) == d
    assert emu.bz3_hip_cm_blocks_given_up() == n0
    junk = bytes(rng.integers(0, 256, size=900, dtype=np.uint8))  # arbitrary input: 256 live rows, handed back
    for mode in (9, 0):
        assert cm_mode(mode) == 0
        assert g.cm_decode(junk, 2000) == oracle.cm_decode(junk, 2000), mode
    # modes 3.. were the polling, lock-step and single-wave decoders of rounds 1-2 (removed in round 3)
    for mode in (3, 5, 7, 8, 12, 100):
        assert emu.bz3_hip_set_cm_mode(mode) == -1
    assert emu.bz3_hip_set_cm_mode(14) == -1


def test_batch_api_through_row_cache_kernels(emu, oracle, cm_mode):
    """bz3_encode_blocks / bz3_decode_blocks with the row-cache variant forced: the blocks the kernel gives up (random
    bytes) go through a second, full-model launch of the same batch call; every block equals the oracle's."""
    assert cm_mode(9) == 0
    bs = 65 * 1024
    t = datagen.shakespeare()
    blocks = [t[:3000], datagen.random_bytes(2500), t[5000:8000], b"tiny", datagen.random_bytes(1500, seed=9)]
    n = len(blocks)
    states = (C.c_void_p * n)(*[emu.bz3_new(bs) for _ in range(n)])
    cap = emu.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    n0 = emu.bz3_hip_cm_blocks_given_up()
    emu.bz3_encode_blocks(states, ptrs, sizes, n)
    assert emu.bz3_hip_cm_blocks_given_up() - n0 == 2
    for i, d in enumerate(blocks):
        assert emu.bz3_last_error(states[i]) == 0
        assert bytes(bufs[i][: sizes[i]]) == oracle.encode_block(d, bs)[2]
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    assert emu.bz3_hip_cm_blocks_given_up() - n0 == 4
    for i, d in enumerate(blocks):
        assert emu.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d
    for s in states:
        emu.bz3_free(s)


# ---- lean states (bz3_hip_set_lean_states): no per-state swap buffer, in-place CM encode, staged CM decode -----------
@pytest.fixture()
def lean(emu):
    emu.bz3_hip_set_lean_states(1)
    yield
    emu.bz3_hip_set_lean_states(0)
    emu.bz3_hip_set_cm_mode(-1)


@pytest.mark.parametrize("name", ["empty", "63", "65", "runs", "f2", "nearmiss", "repeats"])
def test_lean_block_parity(emu, oracle, lean, name):
    d = CASES[name]
    bs = 65 * 1024
    for mode in (-1, 9):  # full-model kernels; row-cache kernels (in-place coding restricts when a block may be given up)
        assert emu.bz3_hip_set_cm_mode(mode) == 0
        with bzip3_amd.State(bs, emu) as st:
            a = st.encode_block(d)
            assert a == oracle.encode_block(d, bs)
            r = st.decode_block(a[2], len(d))
            assert (r[:2] == (len(d), 0) or len(d) == 0) and r[2] == d


def test_lean_decoder_error_codes_and_small_buffers(emu, oracle, lean):
    bs = 65 * 1024
    t = datagen.shakespeare()
    plain = (t[:1200] * 3) + t[5000:6400]  # LZP applies (model & 2): the lean LZP decoder writes into the caller's buffer
    blk = oracle.encode_block(plain, bs)[2]
    assert blk[8] & 2
    n = len(plain)
    muts = [blk[:30] + bytes([blk[30] ^ 1]) + blk[31:], blk[:9] + (n + 40000).to_bytes(4, "little") + blk[13:]]
    with bzip3_amd.State(bs, emu) as st:
        assert st.decode_block(blk, n)[2] == plain
        for m in muts:
            assert st.decode_block(m, n)[:2] == oracle.decode_block(m, n, bs)[:2]
        # buffers smaller than the reference's swap buffer: same verdicts (DATA_SIZE_TOO_SMALL vs CRC)
        for bsz, cs, osz in [(n, len(blk), n), (n // 2, len(blk), n // 2), (5, len(blk), n)]:
            assert st.decode_block(blk, osz, buffer_size=bsz, comp_size=cs)[:2] == oracle.decode_block(blk, osz, bs, buffer_size=bsz, comp_size=cs)[:2], (bsz, cs, osz)
        n2, err, _ = st.encode_block(b"x" * (bs + 1))
        assert (n2, err) == (-1, bzip3_amd.BZ3_ERR_DATA_TOO_BIG)


def test_lean_batch_and_frame(emu, oracle, lean):
    assert emu.bz3_hip_set_cm_mode(9) == 0
    bs = 65 * 1024
    t = datagen.shakespeare()
    blocks = [t[:3000], datagen.random_bytes(2500), b"tiny", t[5000:8000] * 3, b"", datagen.low_entropy(3000)]
    n = len(blocks)
    states = (C.c_void_p * n)(*[emu.bz3_new(bs) for _ in range(n)])
    cap = emu.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: sizes[i]]) == oracle.encode_block(d, bs)[2], i
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: len(d)]) == d, i
    for s in states:
        emu.bz3_free(s)
    emu.bz3_hip_release_cached_memory()
    import frame_cases

    rng = np.random.default_rng(4)
    unit = bytes(rng.integers(0, 256, size=997, dtype=np.uint8))
    frame_cases.check(emu, (unit * 400)[: 4 * 65 * 1024 + 1234], 65 * 1024, only=("flip_chunk1", "orig_small"))


def test_cm_in_place_sink(emu, oracle):
    """cm.hip CmSink through the stage hook: the coder's output starts `gap` bytes below its input in ONE buffer; when the
    coded bytes would reach input that is not in registers yet they go to the side buffer instead (stitched by the caller);
    an exhausted side buffer is an error, never a silent overwrite."""
    d = datagen.random_bytes(2600)
    c = oracle.cm_encode(d)
    try:
        for gap, side, ok in ((0, 65536, True), (7, 65536, True), (90, 65536, True), (2590, 65536, True), (10, 100, False)):
            os.environ["BZ3_CM_TEST_GAP"], os.environ["BZ3_CM_TEST_SIDE"] = str(gap), str(side)
            out = (C.c_uint8 * (len(d) + 200))()
            n = emu.bz3_hip_stage_cm_encode(bzip3_amd._cbuf(d, len(d)), len(d), out)
            assert (n == len(c) and C.string_at(out, n) == c) if ok else n == -1, (gap, side, n)
    finally:
        os.environ.pop("BZ3_CM_TEST_GAP", None)
        os.environ.pop("BZ3_CM_TEST_SIDE", None)


def test_cm_protocols_survive_stalled_waves(emu):
    """Liveness of the LDS hand-off protocols under hostile wave scheduling (BZ3_EMU_SCHED < 0: whole waves of the emulated
    workgroup are put to sleep for dozens of scheduler sweeps at random).  Round 1 found this way that a model wave of the
    CM decoder which is held up for longer than the walker needs for one byte missed the walker's verdict in the one-word
    mailbox and waited for ever (on the GPU: two stalled runs with several workgroups per CU); the verdict is now recovered
    from the following one.  Round 2 found the same way that the barrier-synchronised decoder needs TWO words for the decoded
    byte (the walker stores byte i+1 while a stalled model wave has not read byte i yet).  Runs in a subprocess because the scheduler mode is fixed when the library is loaded."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o, g = Oracle(), bzip3_amd.StageApi(lib)
d = o.bwt(datagen.shakespeare()[100000:101500])[1]
c = o.cm_encode(d)
for mode in (0, 9, 1, 2):  # whole model, tiny cache (slots recycled all the time), the shipped 96- and 44/56-row caches
    assert lib.bz3_hip_set_cm_mode(mode) == 0
    assert g.cm_encode(d) == c and g.cm_decode(c, len(d)) == d
    assert g.cm_decode(c[: len(c) // 2], len(d)) == o.cm_decode(c[: len(c) // 2], len(d))
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    for seed in ("-3", "-7"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_EMU_SCHED=seed), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (seed, r.stdout[-300:], r.stderr[-800:])


def test_cm_policy_by_batch_size_routes_and_hands_blocks_back(oracle):
    """The default policy picks the CM kernels by batch size (api.hip cm_variant_for): with BZ3_HIP_CUS=2 a batch of 4 blocks takes the
    two-per-CU row-cache kernels, a batch of 5 the three-per-CU kernels.  Round 5: blocks that cannot fit a row cache go STRAIGHT to the
    whole-model kernel inside the same call -- on encode by the BWT's histogram (more than a quarter of the bytes outside the 40 most frequent values: the random
    block and the 112-value one), on decode by a payload that hardly shrank (the random block); the 112-value block, which shrinks by 14 %, still reaches
    the row-cache decoder, is handed back by it and decoded again.  Subprocess: the CU count is read when the device context is created."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import numpy as np
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
assert lib.bz3_hip_set_cm_mode(-1) == 0
assert [lib.bz3_hip_cm_variant_for(0, k, e) for k in (1, 2, 3, 4, 5) for e in (0, 1)] == [0, 0, 0, 0, 1, 1, 1, 1, 2, 2]
bs = 65 * 1024
t = datagen.shakespeare()
walk = (np.random.default_rng(3).integers(0, 112, 30000)).astype(np.uint8).tobytes()  # 112 equally likely byte values: shrinks to ~0.86, but no row cache holds 112 rows
for n in (4, 5):
    blocks = [t[i * 900 : i * 900 + 700 + i] for i in range(n - 2)] + [datagen.random_bytes(6000, seed=n), walk]
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    cap = lib.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    g0, r0 = lib.bz3_hip_cm_blocks_given_up(), lib.bz3_hip_cm_blocks_routed_full()
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    g1, r1 = lib.bz3_hip_cm_blocks_given_up(), lib.bz3_hip_cm_blocks_routed_full()
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], (n, i)
    assert sizes[n - 1] * 10 < len(walk) * 9 and sizes[n - 2] * 10 >= 6000 * 9  # the 112-value block shrank by more than a tenth, the random one did not
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    g2, r2 = lib.bz3_hip_cm_blocks_given_up(), lib.bz3_hip_cm_blocks_routed_full()
    for i, d in enumerate(blocks):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, (n, i)
    assert (g1 - g0, r1 - r0) == (0, 2), (n, g1 - g0, r1 - r0)  # encode: both routed by their histograms, nothing handed back
    assert (g2 - g1, r2 - r1) == (1, 1), (n, g2 - g1, r2 - r1)  # decode: the random block routed, the 112-value one handed back by the row-cache decoder
    for s in states:
        lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_HIP_CUS="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


def test_a_batch_of_incompressible_blocks_takes_one_whole_model_launch(oracle):
    """bench.py's `random` leg in small: EVERY block of a batch that would take a row-cache variant is routed (encode: histogram, decode: payload size),
    so the CM stage is one launch of the whole-model kernel and nothing is given up; classic and lean states."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
bs = 65 * 1024
for lean in (0, 1):
    lib.bz3_hip_set_lean_states(lean)
    n = 5
    blocks = [datagen.random_bytes(7000 + 100 * i, seed=20 + i) for i in range(n)]
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    cap = lib.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    g0, r0 = lib.bz3_hip_cm_blocks_given_up(), lib.bz3_hip_cm_blocks_routed_full()
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], (lean, i)
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    for i, d in enumerate(blocks):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, (lean, i)
    assert (lib.bz3_hip_cm_blocks_given_up() - g0, lib.bz3_hip_cm_blocks_routed_full() - r0) == (0, 2 * n), lean
    for s in states:
        lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_HIP_CUS="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


def test_a_small_mixed_batch_is_never_split(oracle):
    """routing by histogram / payload size only pays when the WHOLE batch would take a row-cache variant.  A batch of at most one block
    per CU is whole-model work anyway: a text block beside a random one is ONE CM launch per direction, nothing routed, nothing given up (round 5's decoder
    split such a batch into two serial launches, each as long as its slowest block)."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
bs = 65 * 1024
t = datagen.shakespeare()
for lean in (0, 1):
    lib.bz3_hip_set_lean_states(lean)
    n = 2
    assert lib.bz3_hip_cm_variant_for(0, n, 0) == 0
    blocks = [t[5000:5900], datagen.random_bytes(6000, seed=77)]
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    cap = lib.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    g0, r0 = lib.bz3_hip_cm_blocks_given_up(), lib.bz3_hip_cm_blocks_routed_full()
    lib.bz3_hip_debug_cm_launches(1)
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    assert lib.bz3_hip_debug_cm_launches(1) == 1, lean
    for i, d in enumerate(blocks):
        assert bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], (lean, i)
    assert sizes[1] * 10 >= 6000 * 9  # the random block's payload did not shrink: round 5's decoder routed it
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    assert lib.bz3_hip_debug_cm_launches(1) == 1, lean
    for i, d in enumerate(blocks):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, (lean, i)
    assert (lib.bz3_hip_cm_blocks_given_up() - g0, lib.bz3_hip_cm_blocks_routed_full() - r0) == (0, 0), lean
    for s in states:
        lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_HIP_CUS="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


# ---- streaming file driver (stream.hip) ------------------------------------------------------------------
def _bz3_file(oracle, data, bs):
    """The reference CLI's file for `data` at block size bs (-j 1 layout: doc/bzip3_format.md, src/main.c:173-180, :243-256)."""
    out = [b"BZ3v1", bs.to_bytes(4, "little")]
    for off in range(0, len(data), bs):
        chunk = data[off : off + bs]
        n, err, blk = oracle.encode_block(chunk, bs)
        assert err == 0
        out += [n.to_bytes(4, "little"), len(chunk).to_bytes(4, "little"), blk]
    return b"".join(out)


def test_stream_driver_writes_and_reads_the_cli_format(emu, oracle, tmp_path):
    bs = 65 * 1024
    rng = np.random.default_rng(6)
    unit = bytes(rng.integers(0, 256, size=911, dtype=np.uint8))
    data = (unit * 300)[: 2 * bs + 7777]  # three blocks, the last one short; repetitive, so the emulated CM stage stays small

    def run(fn, src_bytes, *args):
        src, dst = tmp_path / "in.bin", tmp_path / "out.bin"
        src.write_bytes(src_bytes)
        fi, fo = os.open(src, os.O_RDONLY), os.open(dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        try:
            rc = fn(fi, fo, *args)
        finally:
            os.close(fi)
            os.close(fo)
        return rc, dst.read_bytes()

    want = _bz3_file(oracle, data, bs)
    for per_batch in (2, 5):
        rc, enc = run(emu.bz3_hip_encode_stream, data, bs, per_batch)
        assert rc == 0 and enc == want
    assert run(emu.bz3_hip_decode_stream, want, 2) == (0, data)
    # an input that is a multiple of the block size: no empty chunk (like -j 1); the decoder accepts the one -j N appends
    exact = data[: 2 * bs]
    rc, enc = run(emu.bz3_hip_encode_stream, exact, bs, 3)
    assert rc == 0 and enc == _bz3_file(oracle, exact, bs)
    empty_chunk = oracle.encode_block(b"", bs)[2]
    quirk = enc + len(empty_chunk).to_bytes(4, "little") + (0).to_bytes(4, "little") + empty_chunk
    assert run(emu.bz3_hip_decode_stream, quirk, 2) == (0, exact)
    assert run(emu.bz3_hip_encode_stream, b"", bs, 2) == (0, b"BZ3v1" + bs.to_bytes(4, "little"))
    assert run(emu.bz3_hip_decode_stream, b"BZ3v1" + bs.to_bytes(4, "little"), 2) == (0, b"")
    # malformed files: what has been decoded before the failing block is committed, the code says why
    second = 9 + 8 + int.from_bytes(want[9:13], "little")
    rc, out = run(emu.bz3_hip_decode_stream, want[: second + 20], 2)
    assert rc == bzip3_amd.BZ3_ERR_TRUNCATED_DATA and out == data[:bs]
    flipped = want[: second + 40] + bytes([want[second + 40] ^ 0x10]) + want[second + 41 :]
    rc, out = run(emu.bz3_hip_decode_stream, flipped, 1)
    assert rc in (bzip3_amd.BZ3_ERR_CRC, bzip3_amd.BZ3_ERR_BWT, bzip3_amd.BZ3_ERR_MALFORMED_HEADER) and out == data[:bs]
    assert run(emu.bz3_hip_decode_stream, b"BZ3v2" + want[5:], 2)[0] == bzip3_amd.BZ3_ERR_MALFORMED_HEADER
    assert run(emu.bz3_hip_decode_stream, want[:9] + (2 ** 31 - 1).to_bytes(4, "little") + want[13:], 2)[0] == bzip3_amd.BZ3_ERR_MALFORMED_HEADER
    assert emu.bz3_hip_encode_stream(0, 1, 1000, 2) == bzip3_amd.BZ3_ERR_INIT


def test_stream_driver_returns_on_a_write_error_mid_stream(emu, oracle, tmp_path):
    """A write error in a later batch (the reader of a pipe goes away: EPIPE; ENOSPC / EFBIG behave the same) must end the
    pipeline with BZ3_HIP_ERR_IO -- the coder used to wait for ever for a batch the reader no longer produced."""
    import threading

    bs = 65 * 1024
    rng = np.random.default_rng(8)
    unit = bytes(rng.integers(0, 256, size=733, dtype=np.uint8))
    data = (unit * 1200)[: 9 * bs + 100]  # ten blocks, one block per batch: the failure hits batch 3 or later
    want = _bz3_file(oracle, data, bs)
    third = 9
    for _ in range(3):
        third += 8 + int.from_bytes(want[third : third + 4], "little")
    for fn, src_bytes, args, keep in ((emu.bz3_hip_encode_stream, data, (bs, 1), third), (emu.bz3_hip_decode_stream, want, (1,), 3 * bs)):
        src = tmp_path / "in.bin"
        src.write_bytes(src_bytes)
        rd, wr = os.pipe()
        got = []

        def drain():
            n = 0
            while n < keep:
                b = os.read(rd, keep - n)
                if not b:
                    break
                got.append(b)
                n += len(b)
            os.close(rd)  # every later write fails with EPIPE (Python ignores SIGPIPE)

        t = threading.Thread(target=drain)
        t.start()
        fi = os.open(src, os.O_RDONLY)
        res = {}
        call = threading.Thread(target=lambda: res.setdefault("rc", fn(fi, wr, *args)))
        call.start()
        call.join(timeout=120)
        assert not call.is_alive(), "stream driver hangs after a write error"
        os.close(fi)
        os.close(wr)
        t.join()
        assert res["rc"] == -100  # BZ3_HIP_ERR_IO
        assert b"".join(got) == (want if fn is emu.bz3_hip_encode_stream else data)[:keep]  # what was committed before the failure is intact


def test_device_groups_of_a_batch_run_concurrently(oracle):
    """A batch whose states live on several GPUs (bz3_new round-robins over the visible devices) is split into one group per
    GPU and the groups run at the same time, one host thread each (api.hip for_each_device_group; the reference forks a
    thread per block, src/libbz3.c:845-856).  Two emulated devices (BZ3_EMU_DEVICES=2; their kernel launches take turns on the
    one emulated GPU, the host sides overlap): same bytes as the oracle both ways, a failing block does not disturb the
    others, and both groups were inside their group function at the same time."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
assert lib.bz3_hip_device_count() == 2
bs = 65 * 1024
t = datagen.shakespeare()
n = 30  # 15 blocks per device: each group runs its front end through a ring of four context slots (api.hip pipeline_shape)
blocks = [t[i * 3000 : i * 3000 + 2500 + 7 * i] for i in range(n - 1)] + [b"tiny"]
states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
assert sorted(lib.bz3_hip_state_device(s) for s in states) == [0] * 15 + [1] * 15
cap = lib.bz3_bound(bs) + 64
bufs = [(C.c_uint8 * cap)() for _ in range(n)]
for b, d in zip(bufs, blocks):
    C.memmove(b, d, len(d))
ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
sizes[2] = bs + 1  # too much data for its state: that block fails (BZ3_ERR_DATA_TOO_BIG), the others are coded
lib.bz3_hip_debug_peak_concurrent_groups(1)
lib.bz3_encode_blocks(states, ptrs, sizes, n)
assert lib.bz3_hip_debug_peak_concurrent_groups(1) == 2
for i, d in enumerate(blocks):
    if i == 2:
        assert sizes[i] == -1 and lib.bz3_last_error(states[i]) == bzip3_amd.BZ3_ERR_DATA_TOO_BIG
        continue
    assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], i
enc2 = o.encode_block(blocks[2], bs)[2]
C.memmove(bufs[2], enc2, len(enc2))
sizes[2] = len(enc2)
bsz = (C.c_size_t * n)(*[cap] * n)
orig = (C.c_int32 * n)(*[len(d) for d in blocks])
lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
assert lib.bz3_hip_debug_peak_concurrent_groups(1) == 2
for i, d in enumerate(blocks):
    assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, i
for s in states:
    lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_EMU_DEVICES="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


def test_front_end_ring_follows_the_memory_and_shrinks_when_the_arena_does_not_fit():
    """Shape of the encoder's front-end ring (api.hip pipeline_shape / encode_group): four context slots when the memory holds at
    least three blocks per slot and the batch is large enough, two otherwise; when the workspace allocation fails although
    hipMemGetInfo promised the room (BZ3_EMU_MALLOC_LIMIT: allocations above the limit fail with hipErrorOutOfMemory), the ring
    shrinks -- first to two slots, then window by window -- instead of failing the batch, and the blocks still equal the oracle's."""
    import subprocess

    code = r'''
import os, sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
bs = 65 * 1024
t = datagen.shakespeare()
def run(n, size=300):
    blocks = [t[i * 700 : i * 700 + size + 11 * i] for i in range(n)]
    states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
    cap = lib.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    for i, d in enumerate(blocks):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: sizes[i]]) == o.encode_block(d, bs)[2], (n, i)
    for s in states:
        lib.bz3_free(s)
    ring = lib.bz3_hip_debug_front_end_ring()
    released.append(bool(ring >> 30))
    return ring & 0xFFFF, (ring >> 16) & 0xFF
released = []
assert lib.bz3_hip_debug_front_end_ring() == 0
assert run(5) == (5, 2)            # small batch: one window, two slots
assert run(13) == (8, 4)           # 16 GiB "free": windows of 8 through four slots
assert released == [False, False]  # workspaces of this size are kept for the next call
os.environ["BZ3_HIP_WS_KEEP_MB"] = "0"  # ... unless they grew beyond twice the sorter's needs + this slack (1 GiB by default): the contexts of
assert run(13, 60000) == (8, 4) and released[-1]    # 4 x 8 blocks of 60 KB did; the workspace is handed back when the call ends,
assert run(13, 60000) == (8, 4) and released[-1]    # and the next call allocates again
del os.environ["BZ3_HIP_WS_KEEP_MB"]
lib.bz3_hip_release_cached_memory()  # drop the workspace: the next call has to allocate again
os.environ["BZ3_EMU_MALLOC_LIMIT"] = str(16 << 20)
w, ns = run(13, 60000)             # an LZP context is ~8.6 bytes per input byte: 4 x 8 of them + the sorter's workspace need ~28 MiB
assert ns == 2 and 1 <= w < 8, (w, ns)   # ... which did not fit 16 MiB: two slots, smaller windows
os.environ["BZ3_EMU_MALLOC_LIMIT"] = str(1 << 20)
lib.bz3_hip_release_cached_memory()
blocks = [t[:300]]
st = lib.bz3_new(bs)
buf = (C.c_uint8 * (lib.bz3_bound(bs) + 64))()
C.memmove(buf, blocks[0], 300)
assert lib.bz3_encode_block(st, buf, 300) == -1 and lib.bz3_last_error(st) != 0   # nothing fits: an error, not a crash
lib.bz3_free(st)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


def test_keep_workspace_mode_reuses_the_arena_across_a_round_trip():
    """BZ3_HIP_KEEP_WS=1 (experiment for round 5, api.hip keep_workspace): a lean batch's workspace is NOT handed back when the encode call ends, the
    decode call that follows reuses it and carves the swap buffers of its tail windows from it (the pool serves what does not fit); blocks and
    error codes are what they are without the switch.  Two round trips of 7 small lean blocks through windows of 2 x 3 (BZ3_HIP_TAIL_PIPE), a
    failing block among them on the second.  (That the workspace of a LARGE batch stays is the `released` bit of
    test_front_end_ring_follows_the_memory_and_shrinks_when_the_arena_does_not_fit; here the arena is small and stays either way.)"""
    import subprocess

    code = r'''
import os, sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
bs = 65 * 1024
t = datagen.shakespeare()
lib.bz3_hip_set_lean_states(1)
os.environ["BZ3_HIP_WS_KEEP_MB"] = "0"
n = 7
blocks = [t[i * 700 : i * 700 + 2500 + 11 * i] for i in range(n)]
states = (C.c_void_p * n)(*[lib.bz3_new(bs) for _ in range(n)])
cap = lib.bz3_bound(bs) + 64
for trip in range(2):
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_encode_blocks(states, ptrs, sizes, n)
    want = [o.encode_block(d, bs)[2] for d in blocks]
    for i in range(n):
        assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: sizes[i]]) == want[i], (trip, i)
    if trip == 1:
        bufs[4][sizes[4] // 2] ^= 0x55  # a corrupted payload: this block fails its CRC, its neighbours do not
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    lib.bz3_hip_debug_arena_swap_buffers(1)
    lib.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    carved = lib.bz3_hip_debug_arena_swap_buffers(0)
    assert (carved > 0) == (os.environ.get("BZ3_HIP_KEEP_WS") == "1"), carved
    for i, d in enumerate(blocks):
        if trip == 1 and i == 4:
            assert lib.bz3_last_error(states[i]) != 0
        else:
            assert lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, (trip, i)
for s in states:
    lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    for keep in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_HIP_KEEP_WS=keep, BZ3_HIP_TAIL_PIPE="2,3"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (keep, r.stdout[-300:], r.stderr[-1200:])


def test_auto_cm_policy_and_encode_many_hook(oracle):
    """The automatic CM policy by batch size (api.hip cm_variant_for; BZ3_HIP_CUS=2 pretends the GPU has two CUs): up to one block per CU
    the whole-model kernels (0), up to two per CU the 96-row pair (1), beyond that the three-per-CU pair (2); a forced mode wins.  And the profiling hook that launches N copies of one CM encode job returns the oracle's bytes."""
    import subprocess

    code = r'''
import sys, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
assert lib.bz3_hip_cm_variant_for(0, 1, 0) == 0 and lib.bz3_hip_cm_variant_for(0, 2, 1) == 0
assert lib.bz3_hip_cm_variant_for(0, 3, 0) == 1 and lib.bz3_hip_cm_variant_for(0, 4, 1) == 1
assert lib.bz3_hip_cm_variant_for(0, 5, 0) == 2 and lib.bz3_hip_cm_variant_for(0, 700, 1) == 2
assert lib.bz3_hip_cm_variant_for(9, 5, 0) == -1
assert lib.bz3_hip_set_cm_mode(2) == 0 and lib.bz3_hip_cm_variant_for(0, 1, 0) == 2
assert lib.bz3_hip_set_cm_mode(-1) == 0
o = Oracle()
d = o.bwt(datagen.shakespeare()[200000:201200])[1]
want = o.cm_encode(d)
out = (C.c_uint8 * (lib.bz3_bound(len(d)) + 64))()
coded = C.c_int32(0)
for copies in (1, 3, 5):  # whole-model, rows and rows3 encoders
    ms = lib.bz3_hip_stage_cm_encode_many(bzip3_amd._cbuf(d, len(d)), len(d), out, C.byref(coded), copies)
    assert ms >= 0 and bytes(out[: coded.value]) == want, copies
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, BZ3_HIP_CUS="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1200:])


@pytest.mark.parametrize("pipe", [None, "1,4", "5,3", "40,2"], ids=["auto", "w1s4", "w5s3", "w40s2"])
def test_pipelined_windows_of_a_large_batch(emu, oracle, pipe, monkeypatch):
    """A batch larger than the encoder's front-end windows (the LZP drivers of window k run on a side stream of the device beside the
    preparation of the next windows and the completion of the previous ones, over a ring of 2-4 context slots: api.hip pipeline_shape /
    encode_group) and than the decoder's tail windows (32 blocks: LZP decoders beside the next window's inverse BWTs): every block equals
    the oracle's both ways, with LZP applied, declined and skipped.  BZ3_HIP_LZP_PIPE = "window,slots" forces the ring's shape: windows of
    one block (36 windows through 4 slots), a ragged last window through 3 slots, a single window larger than the batch."""
    for var in ("BZ3_HIP_LZP_PIPE", "BZ3_HIP_TAIL_PIPE"):  # the decoder's tail windows take the same shape (decode_group)
        if pipe:
            monkeypatch.setenv(var, pipe)
        else:
            monkeypatch.delenv(var, raising=False)
    bs = 65 * 1024
    t = datagen.shakespeare()
    blocks = []
    for i in range(36):
        if i % 5 == 3:
            blocks.append((t[i * 400 : i * 400 + 150] * 3) + t[9000:9100])   # repeats: LZP applies (model & 2)
        elif i % 7 == 6:
            blocks.append(b"x" * (20 + i))                                   # stored (< 64 bytes)
        else:
            blocks.append(t[i * 400 : i * 400 + 260 + 7 * i])
    n = len(blocks)
    states = (C.c_void_p * n)(*[emu.bz3_new(bs) for _ in range(n)])
    cap = emu.bz3_bound(bs) + 64
    bufs = [(C.c_uint8 * cap)() for _ in range(n)]
    for b, d in zip(bufs, blocks):
        C.memmove(b, d, len(d))
    ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
    sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_encode_blocks(states, ptrs, sizes, n)
    models = set()
    for i, d in enumerate(blocks):
        want = oracle.encode_block(d, bs)[2]
        assert bytes(bufs[i][: sizes[i]]) == want, i
        models.add(want[8] if len(d) >= 64 else -1)
    assert {-1, 0, 2} <= models  # stored, plain and LZP-coded blocks all went through the windows
    bsz = (C.c_size_t * n)(*[cap] * n)
    orig = (C.c_int32 * n)(*[len(d) for d in blocks])
    emu.bz3_decode_blocks(states, ptrs, bsz, sizes, orig, n)
    for i, d in enumerate(blocks):
        assert emu.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, i
    for s in states:
        emu.bz3_free(s)


@pytest.mark.parametrize("pipe", [None, "1,4", "3,3", "5,2"], ids=["auto", "w1s4", "w3s3", "w5s2"])
def test_two_thread_front_end(emu, oracle, pipe, monkeypatch):
    """Round 6: bz3_hip_set_front_end_duo(1) -- phase A of the encoder's front end (CRC, mRLE, LZP preparation) on a second host thread and stream, up to
    slots - 1 windows ahead of phase B (LZP emission, BWT, header) on the calling thread, swap buffers of lean states handed back behind events (api.hip
    encode_group).  36 blocks (LZP applied, declined, stored; one with an invalid size, which fails alone) through forced ring shapes, classic and lean
    states, twice in a row (the second call reuses pool and arena): the oracle's bytes, and the ring reports the two-thread form."""
    if pipe:
        monkeypatch.setenv("BZ3_HIP_LZP_PIPE", pipe)
    else:
        monkeypatch.delenv("BZ3_HIP_LZP_PIPE", raising=False)
    bs = 65 * 1024
    t = datagen.shakespeare()
    blocks = []
    for i in range(36):
        if i % 5 == 3:
            blocks.append((t[i * 400 : i * 400 + 150] * 3) + t[9000:9100])
        elif i % 7 == 6:
            blocks.append(b"y" * (20 + i))
        elif i == 10:
            blocks.append(datagen.random_bytes(900, seed=5))
        else:
            blocks.append(t[i * 400 : i * 400 + 260 + 7 * i])
    want = [oracle.encode_block(d, bs)[2] for d in blocks]
    n = len(blocks)
    try:
        assert emu.bz3_hip_set_front_end_duo(1) == 0
        for lean in (0, 1):
            assert emu.bz3_hip_set_lean_states(lean) == 0
            states = (C.c_void_p * n)(*[emu.bz3_new(bs) for _ in range(n)])
            cap = emu.bz3_bound(bs) + 64
            for trip in range(2):
                bufs = [(C.c_uint8 * cap)() for _ in range(n)]
                for b, d in zip(bufs, blocks):
                    C.memmove(b, d, len(d))
                ptrs = (C.c_void_p * n)(*[C.addressof(b) for b in bufs])
                sizes = (C.c_int32 * n)(*[len(d) for d in blocks])
                if trip == 1:
                    sizes[17] = bs + 1  # larger than the state's block size: BZ3_ERR_DATA_TOO_BIG for this block alone
                emu.bz3_encode_blocks(states, ptrs, sizes, n)
                assert (emu.bz3_hip_debug_front_end_ring() >> 29) & 1 == 1, (lean, trip)
                for i in range(n):
                    if trip == 1 and i == 17:
                        assert sizes[i] == -1 and emu.bz3_last_error(states[i]) == bzip3_amd.BZ3_ERR_DATA_TOO_BIG
                    else:
                        assert bytes(bufs[i][: sizes[i]]) == want[i], (lean, trip, i)
            for s_ in states:
                emu.bz3_free(s_)
        assert emu.bz3_hip_set_front_end_duo(0) == 0
    finally:
        emu.bz3_hip_set_front_end_duo(-1)
        emu.bz3_hip_set_lean_states(0)


SORTER_CASES = datagen.suffix_sorter_cases()


@pytest.mark.parametrize("name", sorted(SORTER_CASES))
def test_suffix_sorter_paths(emu, oracle, name):
    d = SORTER_CASES[name]
    g = bzip3_amd.StageApi(emu)
    assert g.bwt(d) == oracle.bwt(d), (name, len(d))


def test_suffix_sorter_deep_path_for_big_groups(emu, oracle):
    """bz3_hip_debug_bwt_big_rounds(0) (test hook): groups too large for the resolve kernel go straight to rank doubling,
    so the deep path also runs on inputs that normally finish on the big path; and a block through the whole encoder."""
    g = bzip3_amd.StageApi(emu)
    try:
        emu.bz3_hip_debug_bwt_big_rounds(0)
        for name in ("phrase1", "phrase3", "text300k"):
            d = SORTER_CASES[name][:120000]
            assert g.bwt(d) == oracle.bwt(d), name
        emu.bz3_hip_debug_bwt_big_rounds(8)  # several windows for the big groups, then the deep path for what the resolve kernel handed back
        for name in ("mixdeep", "phrase3"):
            assert g.bwt(SORTER_CASES[name]) == oracle.bwt(SORTER_CASES[name]), name
    finally:
        emu.bz3_hip_debug_bwt_big_rounds(-1)
    t = datagen.shakespeare()
    d = t[40000:70000] + t[40000:52000]
    assert bzip3_amd.encode_block(d, 65 * 1024, emu)[2] == oracle.encode_block(d, 65 * 1024)[2]


def test_single_block_calls_from_many_threads_are_collected(oracle):
    """Eight host threads call bz3_encode_block / bz3_decode_block at once on states of their own (what the reference's batch API does
    with pthreads, src/libbz3.c:831-856, and what a threaded binding does): the calls are collected into ONE batch per direction --
    one CM launch, not eight one after the other --, every thread gets the oracle's bytes and return values, and a failing block
    (too much data) fails alone.  A process of its own: the collector's window is widened so that the test does not depend on timing."""
    import subprocess

    code = r'''
import sys, threading, ctypes as C
sys.path[:0] = [%r, %r, %r]
import bzip3_amd, datagen
from build_emu import build
from oracle_lib import Oracle
lib = bzip3_amd._declare(C.CDLL(build()))
o = Oracle()
bs = 65 * 1024
t = datagen.shakespeare()
n = 8
blocks = [t[i * 5000 : i * 5000 + 3000 + 11 * i] for i in range(n)]
states = [lib.bz3_new(bs) for _ in range(n)]
cap = lib.bz3_bound(bs) + 64
bufs = [(C.c_uint8 * cap)() for _ in range(n)]
for b, d in zip(bufs, blocks):
    C.memmove(b, d, len(d))
sizes = [len(d) for d in blocks]
sizes[5] = bs + 1  # too much data for its state
lib.bz3_hip_set_collect_window_us(300000)
lib.bz3_hip_debug_collected_batches(1, None)
ret = [None] * n
def enc(i):
    ret[i] = lib.bz3_encode_block(states[i], bufs[i], sizes[i])
th = [threading.Thread(target=enc, args=(i,)) for i in range(n)]
[x.start() for x in th]; [x.join() for x in th]
big = C.c_uint(0)
assert lib.bz3_hip_debug_collected_batches(1, C.byref(big)) == 1 and big.value == n, (big.value,)
for i, d in enumerate(blocks):
    if i == 5:
        assert ret[i] == -1 and lib.bz3_last_error(states[i]) == bzip3_amd.BZ3_ERR_DATA_TOO_BIG
        continue
    want = o.encode_block(d, bs)
    assert ret[i] == want[0] and lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: ret[i]]) == want[2], i
enc5 = o.encode_block(blocks[5], bs)[2]
C.memmove(bufs[5], enc5, len(enc5)); ret[5] = len(enc5)
back = [None] * n
def dec(i):
    back[i] = lib.bz3_decode_block(states[i], bufs[i], cap, ret[i], len(blocks[i]))
th = [threading.Thread(target=dec, args=(i,)) for i in range(n)]
[x.start() for x in th]; [x.join() for x in th]
assert lib.bz3_hip_debug_collected_batches(1, C.byref(big)) == 1 and big.value == n
for i, d in enumerate(blocks):
    assert back[i] == len(d) and lib.bz3_last_error(states[i]) == 0 and bytes(bufs[i][: len(d)]) == d, i
# one caller alone: a batch of one, the same bytes
lib.bz3_hip_set_collect_window_us(-1)
C.memmove(bufs[0], blocks[0], len(blocks[0]))
assert lib.bz3_encode_block(states[0], bufs[0], len(blocks[0])) == o.encode_block(blocks[0], bs)[0]
for s in states:
    lib.bz3_free(s)
print("ok")
''' % (os.path.dirname(HERE), HERE, os.path.join(HERE, "emu"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-300:], r.stderr[-1500:])


def test_cm_decoder_fast_walk_rarely_falls_back(emu, oracle, cm_mode, monkeypatch):
    """The walker's fast walk names the surviving lane by `d <= range` instead of per-level votes (cm.hip).  A lane on an improbable wrong
    path can reach range 0 and wrap back into validity; tested every other level that does not happen, tested once at the end it sent
    14 % of the bytes of text through the checked walk for nothing (round 3).  The cycle-counter build counts the bytes of the checked
    walk: they must stay close to the bytes during which the coder really renormalises (at most one per coded byte)."""
    n = 40 << 10
    plain = oracle.bwt(datagen.text(n, seed=5, chains=64))[1]
    coded = oracle.cm_encode(plain)
    assert cm_mode(0) == 0
    out = (C.c_uint8 * n)()
    cnt = (C.c_uint64 * 16)()
    monkeypatch.setenv("BZ3_CM_DEBUG", "3")
    ms = emu.bz3_hip_stage_cm_decode_many(bzip3_amd._cbuf(coded, len(coded)), len(coded), out, n, 1, cnt)
    monkeypatch.delenv("BZ3_CM_DEBUG")
    assert ms >= 0
    assert bytes(out)[128:] == plain[128:]  # (the counters replace the first bytes of the output)
    slow, wrong = cnt[2] / n, cnt[3] / n
    assert 0.5 * len(coded) / n < slow <= len(coded) / n + 0.01, (slow, len(coded) / n)
    rep = float((np.frombuffer(plain, dtype=np.uint8)[1:] == np.frombuffer(plain, dtype=np.uint8)[:-1]).mean())
    assert abs(wrong - (1.0 - rep)) < 0.01, (wrong, rep)  # wrong guesses = bytes that do not repeat their predecessor


def test_lzp_decoder_chunks_alignments_and_caps(emu, oracle):
    """k_lzp_decode works in trips of 16 KiB with one 16-byte access per lane each way (lzp.hip): several trips, 0xF2 bytes at every
    alignment inside the 16 bytes of a lane, a ragged last lane, outputs capped at awkward places (the decoder stops AT the cap,
    :211), and streams that are not LZP output at all -- always the oracle's (count, bytes)."""
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(17)
    t = datagen.shakespeare()
    body = bytearray((t[20000:26000] * 4 + t[40000:52000] + t[20000:23000]) * 2)  # repeats: LZP applies
    for k in rng.integers(0, len(body), size=400):
        body[int(k)] = 0xF2  # escapes all over, at every offset modulo 16
    data = bytes(body)
    nlz, lz = oracle.lzp_encode(data)
    assert 0 < nlz == len(lz) < len(data) - 8
    for cap in (len(data) + 100, len(data), len(data) - 1, 32768 + 7, 16384 + 3, 16384, 16383, 4000, 17, 5):
        assert g.lzp_decode(lz, cap) == oracle.lzp_decode(lz, cap), cap
    for cut in (len(lz) - 1, len(lz) // 2, 16384 + 5, 4):  # truncated streams
        assert g.lzp_decode(lz[:cut], len(data) + 100) == oracle.lzp_decode(lz[:cut], len(data) + 100), cut
    for seed in range(3):  # arbitrary bytes with many escapes: matches into garbage, lengths running past the cap
        r2 = np.random.default_rng(100 + seed)
        junk = bytes(r2.choice(np.frombuffer(b"ab\xf2\xf2\xff\xfe\x00q", dtype=np.uint8), size=40000))
        for cap in (60000, 33000, 1000):
            assert g.lzp_decode(junk, cap) == oracle.lzp_decode(junk, cap), (seed, cap)


def test_unbwt_single_walk_with_strided_splitters(emu, oracle):
    """Blocks of more than 2^17 rows cut the psi chain at hashed splitter rows (one in 2, 4, 8 ... rows: unbwt.hip); every splitter
    walks its segment once, parks the bytes in a slab of 4 x the mean segment length and leaves the rest of a longer segment to
    k_ub_walk_long.  Genuine transforms come back as the text; arbitrary (bytes, index) pairs as what the reference makes of them."""
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(23)
    t = datagen.shakespeare()
    for n in (140000, 300000, 600000):  # one splitter per 1 / 2 / 4 rows
        src = t[1000 : 1000 + n]
        idx, u = oracle.bwt(src)
        assert g.unbwt(u, idx) == (0, src), n
        junk = bytes(rng.integers(0, 5, size=n, dtype=np.uint8))
        jidx = int(rng.integers(1, n + 1))
        assert g.unbwt(junk, jidx) == oracle.unbwt(junk, jidx), n
        assert g.unbwt(u, jidx) == oracle.unbwt(u, jidx), n


def test_cm_decoder_on_runs_of_every_length_and_every_top_bit_pair(emu, oracle, cm_mode):
    """The guess-ahead decoder's model waves own subtrees of the byte's decision tree (round 4: levels 2..7 below the node of the byte's two
    top bits; the root, the level-1 nodes and one displaced leaf sit in spare lanes): same bytes as the oracle on runs of 1..8 and 1..70
    equal bytes, on bytes of all four top-bit pairs including 0x7E / 0x7F (the displaced leaf), BWT output of text, noise, truncated
    streams -- with the whole model, the tiny row cache and the three-per-CU cache."""
    g = bzip3_amd.StageApi(emu)
    rng = np.random.default_rng(41)
    runs = bytes(np.repeat(rng.integers(0x20, 0x80, size=400, dtype=np.uint8), rng.integers(1, 9, size=400)))
    longruns = bytes(np.repeat(rng.integers(0, 256, size=60, dtype=np.uint8), rng.integers(1, 70, size=60)))
    cases = [runs, longruns, oracle.bwt(datagen.shakespeare()[200000:204000])[1], bytes(rng.integers(0, 256, size=1500, dtype=np.uint8)),
             b"a" * 3000, b"ab" * 700 + b"~" * 40 + b"\x7f" * 40 + b"\xff" * 30 + b"\x00" * 30 + b"~\x7f" * 50, b"x"]
    for mode in (0, 9, 2):
        assert cm_mode(mode) == 0
        for d in cases:
            c = oracle.cm_encode(d)
            assert g.cm_decode(c, len(d)) == d, (mode, len(d))
            cut = c[: len(c) * 2 // 3]
            assert g.cm_decode(cut, len(d)) == oracle.cm_decode(cut, len(d)), (mode, len(d))


def test_calibrated_text_through_the_three_per_cu_kernels(emu, oracle, cm_mode):
    """bench.py's workload (tests/datagen.py text with ENWIK_NOISE: digits and random identifiers among the words) through the
    44 / 56-row caches: oracle bytes both ways and nothing given up (the GPU twin of this test runs 6 MiB blocks)."""
    g = bzip3_amd.StageApi(emu)
    d = oracle.bwt(datagen.text(9000, seed=71, chains=16, noise=datagen.ENWIK_NOISE))[1]
    c = oracle.cm_encode(d)
    assert cm_mode(2) == 0
    n0 = emu.bz3_hip_cm_blocks_given_up()
    assert g.cm_encode(d) == c
    assert g.cm_decode(c, len(d)) == d
    assert emu.bz3_hip_cm_blocks_given_up() == n0
