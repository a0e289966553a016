"""CPU tests: the plain-C oracle against the reference's golden material and,
when oracle/_ref is built, against the reference's own stage functions."""
import hashlib
import os
import struct

import pytest

import datagen
from oracle_lib import ORACLE_DIR

GOLDEN = datagen.GOLDEN


def test_fixture_decodes_to_known_plaintext(oracle):
    # `make test` of the reference (Makefile.am:81-83): bzip3 -d < shakespeare.txt.bz3 == shakespeare.txt.
    # The fixture's blocks carry model=6 with rle_size > orig_size (legacy encoder, src/libbz3.c:731-733).
    raw = open(os.path.join(GOLDEN, "shakespeare.txt.bz3"), "rb").read()
    bs, chunks = datagen.parse_chunks(raw)
    assert bs == 4 << 20 and len(chunks) == 2
    assert [(c, o) for c, o, _ in chunks] == [(955552, 4194304), (307105, 1263895)]
    out = b""
    for comp, orig, blk in chunks:
        assert blk[8] == 6
        n, err, dec = oracle.decode_block(blk, orig, bs)
        assert (n, err) == (orig, 0)
        assert oracle.crc32c(dec) == struct.unpack("<I", blk[:4])[0]
        out += dec
    assert len(out) == 5458199
    assert hashlib.md5(out).hexdigest() == datagen.SHAKESPEARE_MD5


def test_encoder_known_answer_cfg1(oracle, text):
    # BASELINE config 1: shakespeare.txt, -b 8, one block -> 1,229,814-byte file, md5 from SURVEY.md 8c
    n, err, blk = oracle.encode_block(text, 8 << 20)
    assert (n, err) == (1229797, 0)
    crc, idx = struct.unpack("<Ii", blk[:8])
    assert (crc, idx, blk[8]) == (0x18A1405E, 1980452, 2)
    assert struct.unpack("<I", blk[9:13])[0] == 5314513
    f = b"BZ3v1" + struct.pack("<I", 8 << 20) + struct.pack("<II", len(blk), len(text)) + blk
    assert hashlib.md5(f).hexdigest() == "90bb3148f6a5bf00be8d682458dd15dd"


@pytest.mark.parametrize("name", ["63_byte_file.bin", "65_byte_file.bin"])
def test_threshold_seeds(oracle, name):
    # examples/standard_test_files: both sides of the 64-byte stored/coded threshold (src/libbz3.c:596)
    d = open(os.path.join(GOLDEN, name), "rb").read()
    n, err, blk = oracle.encode_block(d, 65 * 1024)
    assert err == 0
    if len(d) < 64:
        assert n == len(d) + 8 and blk[4:8] == b"\xff\xff\xff\xff" and blk[8:] == d
    else:
        assert blk[4:8] != b"\xff\xff\xff\xff"
    assert oracle.decode_block(blk, len(d), 65 * 1024) == (len(d), 0, d)


def _inputs():
    c = dict(datagen.nasty_cases())
    c["text100k"] = datagen.shakespeare()[:100000]
    c["rand5k"] = datagen.random_bytes(5000)
    c["lowent"] = datagen.low_entropy(40000)
    c["repeats"] = datagen.repeats(60000)
    return c


@pytest.mark.parametrize("name", sorted(_inputs().keys()))
def test_oracle_matches_reference_stages(oracle, ref_stages, ref_lib, name):
    d = _inputs()[name]
    assert oracle.crc32c(d) == ref_stages.crc32c(d)
    e = oracle.mrle_encode(d)
    assert e == ref_stages.mrle_encode(d)
    assert oracle.mrle_decode(e, len(d)) == ref_stages.mrle_decode(e, len(d)) == (0, d)
    # truncated RLE streams exercise the reference's stale-length-byte quirk (src/libbz3.c:320-322)
    for cut in (len(e) - 1, len(e) - 2, 33, 34):
        if 32 <= cut <= len(e):
            assert oracle.mrle_decode(e, len(d), cut) == ref_stages.mrle_decode(e, len(d), cut)
    assert oracle.lzp_encode(d) == ref_stages.lzp_encode(d)
    n, z = oracle.lzp_encode(d)
    if n > 0:
        assert oracle.lzp_decode(z, len(d) + 100) == ref_stages.lzp_decode(z, len(d) + 100) == (len(d), d)
        assert oracle.lzp_decode(z, len(d) // 2) == ref_stages.lzp_decode(z, len(d) // 2)
    if len(d) > 0:
        assert oracle.bwt(d) == ref_stages.bwt(d)
        idx, u = oracle.bwt(d)
        assert oracle.unbwt(u, idx) == (0, d)
        c = oracle.cm_encode(u)
        assert c == ref_stages.cm_encode(u)
        assert oracle.cm_decode(c, len(u)) == u
        assert oracle.cm_decode(c[: len(c) // 2], len(u)) == ref_stages.cm_decode(c[: len(c) // 2], len(u))
    bs = max(65 * 1024, len(d))
    a = oracle.encode_block(d, bs)
    assert a == ref_lib.encode_block(d, bs)
    assert oracle.decode_block(a[2], len(d), bs)[:2] == ref_lib.decode_block(a[2], len(d), bs)[:2] == (len(d), 0)


def test_decoder_error_codes_match_reference(oracle, ref_lib, text):
    bs = 65 * 1024
    blk = oracle.encode_block(text[:30000], bs)[2]
    muts = [blk[: len(blk) // 2], blk[:4] + b"\0\0\0\0" + blk[8:], blk[:8] + b"\x7f" + blk[9:], blk[:20] + bytes([blk[20] ^ 1]) + blk[21:],
            blk[:4] + b"\xff\xff\xff\x7f" + blk[8:], blk[:4] + b"\xfb\xff\xff\xff" + blk[8:], b"\0" * 9, blk[:9]]
    for m in muts:
        assert oracle.decode_block(m, 30000, bs)[:2] == ref_lib.decode_block(m, 30000, bs)[:2]
    for bsz, cs, osz in [(5, len(blk), 30000), (len(blk) - 1, len(blk), 30000), (70000, -5, 30000), (70000, len(blk), -1),
                         (70000, len(blk), 10 ** 9), (20000, len(blk), 30000), (70000, len(blk), 29999), (70000, len(blk), 30001)]:
        a = oracle.decode_block(blk, osz, bs, buffer_size=bsz, comp_size=cs)
        b = ref_lib.decode_block(blk, osz, bs, buffer_size=bsz, comp_size=cs)
        assert a[:2] == b[:2], (bsz, cs, osz)


def test_oracle_matches_reference_on_hostile_stage_inputs(oracle, ref_stages):
    """Input no encoder would produce (what a corrupted block feeds into the later stages): the restatement must still
    equal the reference, because the bytes these stages leave decide which error code the block API reports."""
    import numpy as np

    rng = np.random.default_rng(5)
    # inverse BWT of arbitrary bytes with an arbitrary (valid-range) primary index, small and > 2^17 (fastbits shift >= 1)
    for trial in range(400):
        n = int(rng.integers(2, 50)) if trial % 3 else int(rng.integers(50, 4000))
        k = [1, 2, 3, 256][trial % 4]
        u = bytes(rng.integers(0, k, size=n, dtype=np.uint8))
        for idx in sorted({1, n, int(rng.integers(1, n + 1)), int(rng.integers(1, n + 1))}):
            assert oracle.unbwt(u, idx) == ref_stages.unbwt(u, idx), (n, k, idx)
    for trial in range(8):
        n = int(rng.integers(140000, 500000))
        u = bytes(rng.integers(0, [2, 3, 7, 256][trial % 4], size=n, dtype=np.uint8))
        for idx in sorted({1, n, int(rng.integers(1, n + 1))}):
            assert oracle.unbwt(u, idx) == ref_stages.unbwt(u, idx), (n, idx)
    # CM decoder and LZP decoder on arbitrary bytes
    for size, n in ((0, 300), (3, 2000), (5000, 20000), (60000, 90000)):
        junk = bytes(rng.integers(0, 256, size=size, dtype=np.uint8))
        assert oracle.cm_decode(junk, n) == ref_stages.cm_decode(junk, n)
    for trial in range(40):
        n = int(rng.integers(5, 30000))
        a = rng.choice(np.array([0xF2, 0xFF, 0xFE, 0, 1, 65, 66], dtype=np.uint8), size=n, p=[.15, .05, .1, .2, .1, .2, .2]) if trial % 2 else \
            rng.integers(0, 256, size=n, dtype=np.uint8)
        z = bytes(a)
        for max_out in (n + 100, 70000, 3000):
            assert oracle.lzp_decode(z, max_out) == ref_stages.lzp_decode(z, max_out), (trial, max_out)


def test_corrupted_payload_error_codes_match_reference(oracle, ref_lib, text):
    """Flip bytes inside the coded payload (after the header): CM -> unBWT -> LZP -> mRLE -> CRC all run on garbage."""
    bs = 65 * 1024
    for data in (text[:60000], (text[:500] * 200)[:66000]):
        blk = oracle.encode_block(data, bs)[2]
        for pos in (20, 25, 40, 100, len(blk) // 2, len(blk) - 3):
            for bit in (1, 0x40):
                m = blk[:pos] + bytes([blk[pos] ^ bit]) + blk[pos + 1 :]
                assert oracle.decode_block(m, len(data), bs)[:2] == ref_lib.decode_block(m, len(data), bs)[:2], (pos, bit)


def test_mutated_blocks_decode_like_the_reference(oracle, ref_lib):
    """1500 mutated blocks (tests/mutants.py): return value, last_error and the decoded bytes equal the reference's."""
    import mutants

    data = mutants.seeds()
    blocks = [oracle.encode_block(d, mutants.BS)[2] for d in data]
    for m, osz in mutants.mutants(blocks, [len(d) for d in data], 1500):
        a, b = oracle.decode_block(m, osz, mutants.BS), ref_lib.decode_block(m, osz, mutants.BS)
        assert a[:2] == b[:2] and (a[0] < 0 or a[2] == b[2]), (len(m), osz, a[:2], b[:2])


def test_oracle_builds_without_reference_tree():
    # the restatement itself must not depend on /root/reference (absent on the GPU box)
    src = open(os.path.join(ORACLE_DIR, "bz3_oracle.c")).read()
    assert '#include "src/libbz3.c"' not in src and "/root/reference/" not in src.split("*/", 1)[1]
