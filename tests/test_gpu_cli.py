"""-m gpu: the drop-in boundary end to end.  The reference's CLI translation unit (src/main.c, compiled where it
lies into oracle/_ref/bzip3_main.o by oracle/Makefile) is linked against bzip3_amd/lib/libbzip3.so instead of the
reference's libbz3.c, and must produce / accept the reference's exact files (`make test` and `make roundtrip` of the
reference, Makefile.am:70-83)."""
import hashlib
import os
import subprocess

import pytest

import bzip3_amd
import datagen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN_O = os.path.join(ROOT, "oracle", "_ref", "bzip3_main.o")


@pytest.fixture(scope="module")
def cli(tmp_path_factory, gpu_lib):
    if not os.path.exists(MAIN_O):
        pytest.skip("oracle/_ref/bzip3_main.o did not travel (reference tree absent at build time)")
    exe = str(tmp_path_factory.mktemp("cli") / "bzip3_hip")
    libdir = os.path.dirname(bzip3_amd.LIB_PATH)
    subprocess.check_call(["gcc", MAIN_O, "-L" + libdir, "-lbzip3", "-Wl,-rpath," + libdir, "-lpthread", "-o", exe])
    return exe


def test_cli_decodes_reference_fixture(cli):
    out = subprocess.run([cli, "-d", "-c", os.path.join(datagen.GOLDEN, "shakespeare.txt.bz3")], capture_output=True, check=True).stdout
    assert hashlib.md5(out).hexdigest() == datagen.SHAKESPEARE_MD5


def test_cli_encode_is_byte_identical_to_reference(cli, tmp_path, text):
    src = tmp_path / "shakespeare.txt"
    src.write_bytes(text)
    one = subprocess.run([cli, "-e", "-b", "8", "-c", str(src)], capture_output=True, check=True).stdout
    assert len(one) == 1229814 and hashlib.md5(one).hexdigest() == "90bb3148f6a5bf00be8d682458dd15dd"
    # batch path of main.c (-j 4 -> bz3_encode_blocks / bz3_decode_blocks), 1 MiB blocks
    enc = tmp_path / "s.bz3"
    subprocess.run([cli, "-e", "-b", "1", "-j", "4", "-f", str(src), str(enc)], check=True)
    back = subprocess.run([cli, "-d", "-j", "4", "-c", str(enc)], capture_output=True, check=True).stdout
    assert back == text
    ref = os.path.join(ROOT, "oracle", "_ref", "bzip3")
    if os.path.exists(ref):
        ref_enc = subprocess.run([ref, "-e", "-b", "1", "-j", "4", "-c", str(src)], capture_output=True, check=True).stdout
        assert ref_enc == enc.read_bytes()
        assert subprocess.run([ref, "-d", "-c", str(enc)], capture_output=True, check=True).stdout == text
