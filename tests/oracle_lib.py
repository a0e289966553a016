"""ctypes loaders for the CPU checkers (oracle/liboracle.so and, when built, oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Never imported by bzip3_amd/.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
u8p = C.POINTER(C.c_uint8)


def _buf(b):
    return (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) + (b"\0" if len(b) == 0 else b""))


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


class Oracle:
    """The plain-C restatement (oracle/bz3_oracle.c)."""

    def __init__(self):
        p = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(p):
            build_oracle()
        self.lib = L = C.CDLL(p)
        L.orc_bound.restype = C.c_size_t
        L.orc_bound.argtypes = [C.c_size_t]
        L.orc_crc32c.restype = C.c_uint32
        L.orc_crc32c.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_mrle_encode.restype = C.c_int32
        L.orc_mrle_encode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_mrle_decode.restype = C.c_int
        L.orc_mrle_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.orc_lzp_encode.restype = C.c_int32
        L.orc_lzp_encode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_lzp_decode.restype = C.c_int32
        L.orc_lzp_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.orc_bwt.restype = C.c_int32
        L.orc_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.orc_unbwt.restype = C.c_int32
        L.orc_unbwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.orc_cm_encode.restype = C.c_int32
        L.orc_cm_encode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_cm_decode.restype = None
        L.orc_cm_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.orc_encode_block.restype = C.c_int32
        L.orc_encode_block.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        L.orc_decode_block.restype = C.c_int32
        L.orc_decode_block.argtypes = [C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]

    def bound(self, n):
        return self.lib.orc_bound(n)

    def crc32c(self, data, init=1):
        return self.lib.orc_crc32c(init, _buf(data), len(data))

    def mrle_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.orc_mrle_encode(_buf(data), len(data), out)
        return C.string_at(out, n)

    def mrle_decode(self, data, outlen, maxin=None):
        maxin = len(data) if maxin is None else maxin
        out = (C.c_uint8 * max(1, outlen))()
        rc = self.lib.orc_mrle_decode(_buf(data), out, outlen, maxin)
        return rc, C.string_at(out, outlen)

    def lzp_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.orc_lzp_encode(_buf(data), len(data), out)
        return n, (C.string_at(out, n) if n > 0 else b"")

    def lzp_decode(self, data, maxout):
        out = (C.c_uint8 * max(8, maxout))()
        n = self.lib.orc_lzp_decode(_buf(data), len(data), out, maxout)
        return n, (C.string_at(out, n) if n > 0 else b"")

    def bwt(self, data):
        out = (C.c_uint8 * max(1, len(data)))()
        idx = self.lib.orc_bwt(_buf(data), out, len(data))
        return idx, C.string_at(out, len(data))

    def unbwt(self, data, idx):
        out = (C.c_uint8 * max(1, len(data)))()
        rc = self.lib.orc_unbwt(_buf(data), out, len(data), idx)
        return rc, C.string_at(out, len(data))

    def cm_encode(self, data):
        out = (C.c_uint8 * (len(data) + len(data) // 50 + 64))()
        n = self.lib.orc_cm_encode(_buf(data), len(data), out)
        return C.string_at(out, n)

    def cm_decode(self, data, n):
        out = (C.c_uint8 * max(1, n))()
        self.lib.orc_cm_decode(_buf(data), len(data), out, n)
        return C.string_at(out, n)

    def encode_block(self, data, block_size):
        cap = self.bound(max(len(data), 64)) + 64
        buf = (C.c_uint8 * cap)()
        C.memmove(buf, bytes(data), len(data))
        err = C.c_int32(0)
        n = self.lib.orc_encode_block(buf, len(data), block_size, C.byref(err))
        return n, err.value, (C.string_at(buf, n) if n > 0 else b"")

    def decode_block(self, data, orig_size, block_size, buffer_size=None, comp_size=None):
        cap = self.bound(block_size) + 64
        buffer_size = cap if buffer_size is None else buffer_size
        comp_size = len(data) if comp_size is None else comp_size
        buf = (C.c_uint8 * max(cap, buffer_size, len(data) + 1))()
        C.memmove(buf, bytes(data), len(data))
        err = C.c_int32(0)
        n = self.lib.orc_decode_block(buf, buffer_size, comp_size, orig_size, block_size, C.byref(err))
        return n, err.value, (C.string_at(buf, n) if n > 0 else b"")


class RefStages:
    """The REAL reference's static stage functions (oracle/_ref/libbz3ref_stages.so), if built."""

    def __init__(self):
        self.lib = L = _load(os.path.join(ORACLE_DIR, "_ref", "libbz3ref_stages.so"))
        if L is None:
            return
        L.ref_crc32.restype = C.c_uint32
        L.ref_crc32.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.ref_mrlec.restype = C.c_int32
        L.ref_mrlec.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ref_mrled.restype = C.c_int
        L.ref_mrled.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ref_lzp_compress.restype = C.c_int32
        L.ref_lzp_compress.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.ref_lzp_decompress.restype = C.c_int32
        L.ref_lzp_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ref_bwt.restype = C.c_int32
        L.ref_bwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.ref_unbwt.restype = C.c_int32
        L.ref_unbwt.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ref_cm_encode.restype = C.c_int32
        L.ref_cm_encode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.ref_cm_decode.restype = None
        L.ref_cm_decode.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]

    @property
    def available(self):
        return self.lib is not None

    def crc32c(self, data, init=1):
        return self.lib.ref_crc32(init, _buf(data), len(data))

    def mrle_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.ref_mrlec(_buf(data), len(data), out)
        return C.string_at(out, n)

    def mrle_decode(self, data, outlen, maxin=None):
        maxin = len(data) if maxin is None else maxin
        out = (C.c_uint8 * max(1, outlen))()
        rc = self.lib.ref_mrled(_buf(data), out, outlen, maxin)
        return rc, C.string_at(out, outlen)

    def lzp_encode(self, data):
        out = (C.c_uint8 * (len(data) + 64))()
        n = self.lib.ref_lzp_compress(_buf(data), out, len(data))
        return n, (C.string_at(out, n) if n > 0 else b"")

    def lzp_decode(self, data, maxout):
        out = (C.c_uint8 * max(8, maxout))()
        n = self.lib.ref_lzp_decompress(_buf(data), out, len(data), maxout)
        return n, (C.string_at(out, n) if n > 0 else b"")

    def bwt(self, data):
        out = (C.c_uint8 * max(1, len(data)))()
        idx = self.lib.ref_bwt(_buf(data), out, len(data))
        return idx, C.string_at(out, len(data))

    def unbwt(self, data, idx):
        out = (C.c_uint8 * max(1, len(data)))()
        rc = self.lib.ref_unbwt(_buf(data), out, len(data), idx)
        return rc, C.string_at(out, len(data))

    def cm_encode(self, data):
        out = (C.c_uint8 * (len(data) + len(data) // 50 + 64))()
        n = self.lib.ref_cm_encode(_buf(data), len(data), out)
        return C.string_at(out, n)

    def cm_decode(self, data, n):
        out = (C.c_uint8 * max(1, n))()
        self.lib.ref_cm_decode(_buf(data), len(data), out, n)
        return C.string_at(out, n)


class RefLib:
    """The REAL reference library (oracle/_ref/libbz3ref.so) through its public libbz3.h API."""

    def __init__(self, path=None):
        self.lib = L = _load(path or os.path.join(ORACLE_DIR, "_ref", "libbz3ref.so"))
        if L is None:
            return
        bind_libbz3(L)

    @property
    def available(self):
        return self.lib is not None


def require_ref():
    """The real reference for a parity test that must not pass without comparing (every -m gpu test that checks bytes against
    oracle/_ref): fails -- never skips -- when oracle/_ref/libbz3ref.so did not travel to the box (`make -C oracle` builds it
    where /root/reference exists; gpurun ships the built file)."""
    import pytest

    r = RefLib()
    if not r.available:
        pytest.fail("oracle/_ref/libbz3ref.so is missing: this test compares with the REAL reference and does not pass without it "
                    "(run `make -C oracle` in the build container before the snapshot is taken)")
    return r


def bind_libbz3(L):
    """Declare the libbz3.h prototypes (include/libbz3.h) on a loaded library handle."""
    L.bz3_version.restype = C.c_char_p
    L.bz3_new.restype = C.c_void_p
    L.bz3_new.argtypes = [C.c_int32]
    L.bz3_free.restype = None
    L.bz3_free.argtypes = [C.c_void_p]
    L.bz3_last_error.restype = C.c_int8
    L.bz3_last_error.argtypes = [C.c_void_p]
    L.bz3_strerror.restype = C.c_char_p
    L.bz3_strerror.argtypes = [C.c_void_p]
    L.bz3_bound.restype = C.c_size_t
    L.bz3_bound.argtypes = [C.c_size_t]
    L.bz3_min_memory_needed.restype = C.c_size_t
    L.bz3_min_memory_needed.argtypes = [C.c_int32]
    L.bz3_encode_block.restype = C.c_int32
    L.bz3_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.bz3_decode_block.restype = C.c_int32
    L.bz3_decode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32]
    L.bz3_encode_blocks.restype = None
    L.bz3_encode_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bz3_decode_blocks.restype = None
    L.bz3_decode_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bz3_compress.restype = C.c_int
    L.bz3_compress.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bz3_decompress.restype = C.c_int
    L.bz3_decompress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.bz3_orig_size_sufficient_for_decode.restype = C.c_int
    L.bz3_orig_size_sufficient_for_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int32]
    return L


class Bz3:
    """Convenience wrapper over any library exporting the libbz3.h API (reference or bzip3_amd)."""

    def __init__(self, lib):
        self.lib = lib

    def bound(self, n):
        return self.lib.bz3_bound(n)

    def encode_block(self, data, block_size):
        st = self.lib.bz3_new(block_size)
        assert st, "bz3_new failed"
        try:
            cap = self.bound(max(len(data), block_size)) + 64
            buf = (C.c_uint8 * cap)()
            C.memmove(buf, bytes(data), len(data))
            n = self.lib.bz3_encode_block(st, buf, len(data))
            err = self.lib.bz3_last_error(st)
            return n, err, (C.string_at(buf, n) if n > 0 else b"")
        finally:
            self.lib.bz3_free(st)

    def decode_block(self, data, orig_size, block_size, buffer_size=None, comp_size=None):
        st = self.lib.bz3_new(block_size)
        assert st, "bz3_new failed"
        try:
            cap = self.bound(block_size) + 64
            buffer_size = cap if buffer_size is None else buffer_size
            comp_size = len(data) if comp_size is None else comp_size
            buf = (C.c_uint8 * max(cap, buffer_size, len(data) + 1))()
            C.memmove(buf, bytes(data), len(data))
            n = self.lib.bz3_decode_block(st, buf, buffer_size, comp_size, orig_size)
            err = self.lib.bz3_last_error(st)
            return n, err, (C.string_at(buf, n) if n > 0 else b"")
        finally:
            self.lib.bz3_free(st)
