"""Deterministic input generators shared by the tests and bench.py.

  text(n, seed)     "enwik-style" stand-in: parallel word-bigram Markov chains over the tokens of
                    shakespeare.txt (the plaintext of the reference's only fixture, recovered by decoding
                    tests/golden/shakespeare.txt.bz3 with the oracle -- /root/reference is never read).
  random_bytes      xorshift-free numpy PCG stream (incompressible: LZP declines, RLE declines).
  low_entropy       skewed order-1 source over 16 symbols with short repeat units (LZP/RLE decline,
                    BWT sees long LCPs): the cfg5 stand-in.
  repeats           a 4 KiB paragraph repeated with a few random edits per copy (LZP collapses it).
"""
import hashlib
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SHAKESPEARE_MD5 = "d2028225a89d8b0b3093dddb720da91f"
_cache = {}


def shakespeare():
    """Plaintext of tests/golden/shakespeare.txt.bz3 decoded with the ORACLE (2 chunks, 4 MiB blocks)."""
    if "txt" in _cache:
        return _cache["txt"]
    cache_file = "/tmp/bz3_shakespeare_%s.txt" % SHAKESPEARE_MD5[:8]
    if os.path.exists(cache_file):
        data = open(cache_file, "rb").read()
        if hashlib.md5(data).hexdigest() == SHAKESPEARE_MD5:
            _cache["txt"] = data
            return data
    from oracle_lib import Oracle

    o = Oracle()
    raw = open(os.path.join(GOLDEN, "shakespeare.txt.bz3"), "rb").read()
    assert raw[:5] == b"BZ3v1"
    (bs,) = struct.unpack("<I", raw[5:9])
    pos, out = 9, []
    while pos < len(raw):
        comp, orig = struct.unpack("<II", raw[pos : pos + 8])
        n, err, dec = o.decode_block(raw[pos + 8 : pos + 8 + comp], orig, bs)
        assert n == orig and err == 0
        out.append(dec)
        pos += 8 + comp
    data = b"".join(out)
    assert hashlib.md5(data).hexdigest() == SHAKESPEARE_MD5
    try:
        open(cache_file, "wb").write(data)
    except OSError:
        pass
    _cache["txt"] = data
    return data


def parse_chunks(raw):
    """Split a .bz3 file (CLI format, doc/bzip3_format.md) into (block_size, [(comp, orig, block bytes)])."""
    assert raw[:5] == b"BZ3v1"
    (bs,) = struct.unpack("<I", raw[5:9])
    pos, chunks = 9, []
    while pos < len(raw):
        comp, orig = struct.unpack("<II", raw[pos : pos + 8])
        chunks.append((comp, orig, raw[pos + 8 : pos + 8 + comp]))
        pos += 8 + comp
    return bs, chunks


def bigram_tables():
    """Token table of shakespeare.txt for the Markov text generator (numpy arrays)."""
    if "tab" in _cache:
        return _cache["tab"]
    words = shakespeare().split()
    vocab, inv = np.unique(np.array(words, dtype=object), return_inverse=True)
    inv = inv.astype(np.int64)
    order = np.argsort(inv[:-1], kind="stable")  # successors grouped by predecessor token
    succ = inv[1:][order]
    counts = np.bincount(inv[:-1], minlength=len(vocab)).astype(np.int64)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    # tokens without a successor (only the last word) restart from token 0
    counts_safe = np.where(counts == 0, 1, counts)
    lens = np.array([len(w) + 1 for w in vocab], dtype=np.int64)  # word + one space
    blob = np.frombuffer(b"".join(w + b" " for w in vocab), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    tab = dict(succ=succ, start=start, counts=counts_safe, lens=lens, blob=blob, off=off, nvocab=len(vocab))
    _cache["tab"] = tab
    return tab


ENWIK_NOISE = 0.035  # calibrated in round 4: 100,000,000 B at -b 16 -> 22,681,833 B with the reference (enwik8: 22,677,651)
NOISE_ALPHABET = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789", dtype=np.uint8)


def text(n, seed=1, chains=2048, noise=0.0):
    """n bytes of Markov text.  `chains` independent chains are generated in lockstep and concatenated.
    noise > 0: that fraction of the tokens has its letters replaced by random characters of [a-z0-9] (identifiers, numbers, markup:
    what enwik8 has and Shakespeare has not); ENWIK_NOISE is the fraction at which 100,000,000 bytes compress like enwik8 does
    with the reference's default block size (etc/BENCHMARKS.md:45-47: 4.41 : 1)."""
    t = bigram_tables()
    rng = np.random.Generator(np.random.PCG64(seed))
    avg = float(t["lens"][t["succ"]].mean())  # frequency-weighted word length (+1 space)
    steps = int(n / (avg * chains) * 1.15) + 16
    state = rng.integers(0, t["nvocab"], size=chains)
    toks = np.empty((steps, chains), dtype=np.int64)
    for s in range(steps):
        r = rng.integers(0, 1 << 30, size=chains)
        state = t["succ"][(t["start"][state] + r % t["counts"][state]) % len(t["succ"])]
        toks[s] = state
    toks = toks.T.reshape(-1)  # chain-major: each chain's words are contiguous
    lens = t["lens"][toks]
    ends = np.cumsum(lens)
    total = int(ends[-1])
    noisy = rng.random(len(toks)) < noise if noise > 0 else None
    if total < n:
        return np.tile(_emit(t, toks, lens, ends, noisy, rng), n // total + 1)[:n].tobytes()
    return _emit(t, toks, lens, ends, noisy, rng)[:n].tobytes()


def _emit(t, toks, lens, ends, noisy=None, rng=None):
    starts = ends - lens
    total = int(ends[-1])
    tok_of_byte = np.repeat(np.arange(len(toks)), lens)
    within = np.arange(total) - starts[tok_of_byte]
    out = t["blob"][t["off"][toks[tok_of_byte]] + within]
    if noisy is not None:
        mask = noisy[tok_of_byte] & (within < lens[tok_of_byte] - 1)  # the token's letters, not its space
        out = out.copy()
        out[mask] = NOISE_ALPHABET[rng.integers(0, len(NOISE_ALPHABET), size=int(mask.sum()))]
    return out


def random_bytes(n, seed=2):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=n, dtype=np.uint8).tobytes()


def low_entropy(n, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    units = [bytes(rng.integers(97, 113, size=int(rng.integers(3, 24))).astype(np.uint8)) for _ in range(64)]
    p = rng.dirichlet(np.full(64, 0.3))
    out, size = [], 0
    while size < n:
        picks = rng.choice(64, size=4096, p=p)
        chunk = b"".join(units[k] for k in picks)
        out.append(chunk)
        size += len(chunk)
    return b"".join(out)[:n]


def repeats(n, seed=4):
    rng = np.random.Generator(np.random.PCG64(seed))
    para = bytearray(text(4096, seed=seed + 100, chains=4))
    out, size = [], 0
    while size < n:
        p = bytearray(para)
        for _ in range(4):
            p[int(rng.integers(0, len(p)))] = int(rng.integers(97, 123))
        out.append(bytes(p))
        size += len(p)
    return b"".join(out)[:n]


def nasty_cases():
    """Small adversarial inputs for the byte filters (runs around 255/256/510, 0xF2 escapes, near-misses)."""
    rng = np.random.Generator(np.random.PCG64(7))
    cases = {
        "empty": b"",
        "one": b"x",
        "63": bytes(range(63)),
        "64": bytes(range(64)),
        "65": bytes(range(65)),
        "zeros300": b"\0" * 300,
        "runs": b"".join(bytes([65 + (i % 5)]) * L for i, L in enumerate([1, 2, 3, 254, 255, 256, 257, 509, 510, 511, 512, 765, 766, 1, 1, 2])),
        "f2": (b"a" * 1000 + b"\xf2" * 600 + b"xyz" * 200) * 3,
        "f2text": bytes(rng.choice(np.frombuffer(b"ab\xf2 \xf2cd", dtype=np.uint8), size=5000)),
        "ab": bytes(rng.choice(np.frombuffer(b"ab", dtype=np.uint8), size=6000)),
        "banana": b"banana" * 50,
        "tailzeros": b"q" * 100 + b"\0" * 9,
        "period7": b"abcdefg" * 900,
    }
    t = shakespeare()
    cases["nearmiss"] = (t[:3000] * 5) + t[5000:9000] + t[:3000]
    holes = bytearray(t[10000:14000] * 6)
    for k in range(0, len(holes), 53):
        holes[k] = 35 + (k % 7)
    cases["holes"] = bytes(holes)
    cases["ff"] = b"\xff" * 700 + b"\xfe" * 300 + b"\xff" * 255
    # alphabet sizes around the limits of the suffix sorter's alphabet compaction (ranks of the byte values present: 127 values still
    # fit 7 bits next to the past-the-end rank 0, 128 do not; 255 / 256 values take the plain-byte path), zero bytes next to the end
    for sigma in (127, 128, 255, 256):
        vals = np.sort(rng.permutation(256)[:sigma]).astype(np.uint8)
        body = vals[rng.integers(0, sigma, size=3000)]
        cases["alpha%d" % sigma] = bytes(np.concatenate([vals, body, vals[::-1]]))
    cases["zerotail"] = b"ab\0ab\0\0ab" * 40 + b"ab\0\0\0\0\0\0\0"
    return cases


def _fib_string(n):
    a, b = b"a", b"ab"
    while len(b) < n:
        a, b = b, b + a
    return b[:n]


def suffix_sorter_cases():
    """Inputs for every path of the round-3 suffix sorter (bwt.hip): groups the resolve kernel finishes in LDS (text, small
    alphabets: counting for groups <= 64, the bitonic network above), groups of > 512 suffixes that take the big path once or
    several times (a phrase repeated in random surroundings), lists that overflow it or repeat too deeply and fall back to rank
    doubling (runs, periods, Fibonacci strings, a passage repeated verbatim), suffixes that end inside a window (zero-coded tails),
    and sizes around the tile geometry (round 3's first kernel: 1536-slot anchors, 2048-slot windows; the shipped kernels: 512-slot anchor
    tiles, groups of <= 64 to the tail list, <= 256 to the wide kernel)."""
    rng = np.random.default_rng(5)
    t = shakespeare()
    letters = lambda k: bytes(rng.integers(97, 123, size=k, dtype=np.uint8))
    c = {
        "text300k": t[:300000], "two": bytes(rng.integers(0, 2, size=20000, dtype=np.uint8)), "four": bytes(rng.integers(0, 4, size=30000, dtype=np.uint8) + 65),
        "aaaa": b"a" * 5000, "ab": b"ab" * 4000, "abc": b"abc" * 3000 + b"x", "fib": _fib_string(20000), "zeros+text": b"\0" * 3000 + t[1000:9000] + b"\0" * 2000,
        "deeprep": t[5000:9000] + t[100000:103000] + t[5000:9000] + b"#" + t[5000:9000], "allbytes": bytes(range(256)) * 40,
        "tail0": t[2000:6000] + b"\0" * 10, "n2": b"ab", "n2b": b"aa", "n3": b"aba", "n15": b"aaaaaaaaaaaaaab", "n17": t[:17],
        # 1200 copies of a 22-letter phrase among 150 k random letters: groups of 1200 that one more window resolves
        "phrase1": b"".join(b"quickbrownfoxjumpsover" + letters(110) for _ in range(1200)),
        # a 40-letter phrase: its first positions need several big rounds (11 letters per 56-bit window)
        "phrase3": b"".join(b"thequickbrownfoxjumpsoverthelazydogagain" + letters(400) for _ in range(700)),
    }
    # groups of 100 that share up to 45 letters (the resolve kernel hands them back after its three steps: more than 64 members)
    # beside groups of 700 that go through big rounds: the deep path must start from the SHALLOWER of the two depths
    long_phrase = b"thequickbrownfoxjumpsoverthelazydogagainandagainuntilthecowscomehomeandthenoncemorewithfeelingforgoodmeasure"
    c["mixdeep"] = (b"".join(long_phrase + letters(500) for _ in range(600)) +
                    b"".join(b"3141592653589793238462643383279502884197169399375" + letters(300) for _ in range(100)))  # digits: these groups sort far from the big ones
    for n in (1535, 1536, 1537, 2047, 2048, 2049, 3071, 3072, 3073, 4097):
        c["text%d" % n] = t[777 : 777 + n]
        c["pair%d" % n] = (t[9000:9000 + n // 2] * 2)[:n]
    # the geometry of the shipped kernels (round 4): groups of exactly 64 / 65 members (tail list | wide kernel), 256 / 257 (wide
    # kernel | big list), blocks around one and two anchor tiles of 512 slots, and a group that straddles two anchor tiles
    for members in (64, 65, 256, 257):
        c["group%d" % members] = b"".join(b"zyxwvutsrqponm" + letters(30) for _ in range(members)) + letters(3000)
    for n in (511, 512, 513, 1023, 1024, 1025):
        c["tile%d" % n] = t[4242 : 4242 + n]
    c["straddle"] = letters(400) + b"".join(b"mmmmmmmmmmmmmm" + bytes([97 + (k % 26)]) + letters(5) for k in range(200)) + letters(300)
    return c
