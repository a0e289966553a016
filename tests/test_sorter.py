"""sort.hip's stable LSD radix sort, directly (bz3_hip_debug_sort_u32): both digit widths (8 bits; 9 bits = round 5's two-pass LZP hash
sort), and both ways a pass finds its offsets -- the scatter that reads the hist kernel's raw count table (passes of up to 128 tiles =
512 Ki keys: no scan launches) and the device-wide scan (larger passes).  Checked against numpy's stable argsort.  The emulator runs the
same kernel sources on the CPU; the GPU twin runs them on the MI355X."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import bzip3_amd

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    from build_emu import build

    return bzip3_amd._declare(C.CDLL(build()))

RAW_KEYS = 128 * 4096  # sort.hip RS_RAW_TILES * RS_TILE: the largest pass that takes the raw-count scatter


def _sort(lib, keys, key_bits, digit_bits):
    n = len(keys)
    sk = np.empty(n, dtype=np.uint32)
    si = np.empty(n, dtype=np.uint32)
    passes = lib.bz3_hip_debug_sort_u32(keys.ctypes.data_as(C.c_void_p), n, key_bits, digit_bits, sk.ctypes.data_as(C.c_void_p), si.ctypes.data_as(C.c_void_p))
    assert passes == -(-key_bits // digit_bits)
    return sk, si


def _check(lib, n, key_bits, digit_bits, seed, skew=False):
    rng = np.random.default_rng(seed)
    keys = rng.integers(0, 1 << key_bits, n, dtype=np.uint64).astype(np.uint32)
    if skew:  # most keys in a handful of digits: long per-digit runs in a tile, empty rows in the count table
        keys = np.where(rng.random(n) < 0.9, keys & np.uint32(0x3), keys).astype(np.uint32)
    sk, si = _sort(lib, keys, key_bits, digit_bits)
    covered = digit_bits * -(-key_bits // digit_bits)  # key bits the passes look at (all of them set bits of the keys: keys < 2^key_bits)
    assert covered >= key_bits
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(si, order.astype(np.uint32)), (n, key_bits, digit_bits)
    assert np.array_equal(sk, keys[order])


CASES = [(1, 18, 9), (63, 18, 9), (4096, 18, 9), (4097, 18, 8), (70000, 18, 9), (70000, 18, 8), (300001, 24, 8), (300001, 27, 9)]


@pytest.mark.parametrize("n,key_bits,digit_bits", CASES)
def test_sorter_on_the_emulator(emu, n, key_bits, digit_bits):
    _check(emu, n, key_bits, digit_bits, seed=n + digit_bits)
    _check(emu, n, key_bits, digit_bits, seed=n, skew=True)


def test_sorter_across_the_raw_table_bound_on_the_emulator(emu):
    """one tile below, at and above the bound between the two offset paths (one 9-bit and one 8-bit pass each: the emulator is slow)"""
    for n in (RAW_KEYS - 5, RAW_KEYS, RAW_KEYS + 4096 + 17):
        _check(emu, n, 9, 9, seed=n)
        _check(emu, n, 8, 8, seed=n + 1, skew=True)


SCAN_SIZES = [1, 2047, 2048, 2049, 5003, 65537, 262143, 262144, 262145, 300000]  # one tile | one launch (up to 262144 words) | the recursive scan


def _check_scan(lib, n, seed):
    rng = np.random.default_rng(seed)
    d = rng.integers(0, 4000, n, dtype=np.uint64).astype(np.uint32)
    want = np.zeros(n, dtype=np.uint64)  # (uint64 throughout: [0] + uint64 would promote to float64)
    want[1:] = np.cumsum(d.astype(np.uint64))[:-1]
    want = (want & np.uint64(0xFFFFFFFF)).astype(np.uint32)  # the device sums wrap modulo 2^32
    tot = C.c_uint32(0)
    buf = d.copy()
    assert lib.bz3_hip_debug_scan_u32(buf.ctypes.data_as(C.c_void_p), n, C.byref(tot)) == 0
    assert np.array_equal(buf, want), n
    assert tot.value == int(d.astype(np.uint64).sum()) & 0xFFFFFFFF


@pytest.mark.parametrize("n", SCAN_SIZES)
def test_scan_on_the_emulator(emu, n):
    _check_scan(emu, n, seed=n)


@pytest.mark.gpu
def test_scan_on_the_gpu(gpu_lib):
    for n in SCAN_SIZES + [1 << 24, (1 << 24) + 7]:
        _check_scan(gpu_lib, n, seed=n)


@pytest.mark.gpu
def test_sorter_on_the_gpu(gpu_lib):
    for n, kb, db in CASES + [(RAW_KEYS - 5, 18, 9), (RAW_KEYS, 18, 8), (RAW_KEYS + 4096 + 17, 18, 9), (1024 * 4096, 8, 8), (1024 * 4096 + 4096, 8, 8), (5_000_003, 18, 9), (5_000_003, 32, 8), (40_000_000, 18, 9)]:
        _check(gpu_lib, n, kb, db, seed=n + db)
    _check(gpu_lib, 3_000_000, 18, 9, seed=7, skew=True)
