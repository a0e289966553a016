"""-m gpu, run last: the streaming file driver (bzip3_amd/csrc/stream.hip) against the real reference CLI
(oracle/_ref/bzip3, when it travelled): bz3_hip_encode_stream writes the bytes of `bzip3 -e -b 1`, bz3_hip_decode_stream reads
the reference's file back."""
import os
import subprocess

import pytest

import bzip3_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "bzip3")


def _run(fn, src, dst, *args):
    fi, fo = os.open(src, os.O_RDONLY), os.open(dst, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    try:
        return fn(fi, fo, *args)
    finally:
        os.close(fi)
        os.close(fo)


def test_stream_driver_against_the_reference_cli(gpu_lib, oracle, text, tmp_path):
    src, enc, back = tmp_path / "s.txt", tmp_path / "s.bz3", tmp_path / "s.out"
    src.write_bytes(text)
    bs = 1 << 20
    assert _run(gpu_lib.bz3_hip_encode_stream, src, enc, bs, 4) == 0
    mine = enc.read_bytes()
    # the file format, chunk by chunk, from the oracle's blocks
    want = [b"BZ3v1", bs.to_bytes(4, "little")]
    for off in range(0, len(text), bs):
        n, err, blk = oracle.encode_block(text[off : off + bs], bs)
        want += [n.to_bytes(4, "little"), len(text[off : off + bs]).to_bytes(4, "little"), blk]
    assert mine == b"".join(want)
    assert _run(gpu_lib.bz3_hip_decode_stream, enc, back, 3) == 0 and back.read_bytes() == text
    if os.path.exists(REF):
        ref_enc = subprocess.run([REF, "-e", "-b", "1", "-c", str(src)], capture_output=True, check=True).stdout
        assert ref_enc == mine
        assert subprocess.run([REF, "-d", "-c", str(enc)], capture_output=True, check=True).stdout == text
        # a file written by `-j 4` (one more, empty chunk when the size is a multiple of the block size)
        exact = tmp_path / "exact.txt"
        exact.write_bytes(text[: 3 * bs])
        j4 = tmp_path / "j4.bz3"
        j4.write_bytes(subprocess.run([REF, "-e", "-b", "1", "-j", "4", "-c", str(exact)], capture_output=True, check=True).stdout)
        assert _run(gpu_lib.bz3_hip_decode_stream, j4, back, 8) == 0 and back.read_bytes() == text[: 3 * bs]
    # truncated file: the blocks before the cut are committed
    cut = tmp_path / "cut.bz3"
    cut.write_bytes(mine[: len(mine) * 2 // 3])
    rc = _run(gpu_lib.bz3_hip_decode_stream, cut, back, 2)
    got = back.read_bytes()
    assert rc == bzip3_amd.BZ3_ERR_TRUNCATED_DATA and len(got) % bs == 0 and got == text[: len(got)] and len(got) >= bs
