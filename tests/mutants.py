"""Mutated blocks for decoder-hardening parity (cf. the reference's examples/fuzz-decode-block.c):
deterministic bit flips, header garbage, truncations, random 32-bit fields and payload splices of valid blocks."""
import numpy as np

import datagen

BS = 65 * 1024


def seeds():
    text = datagen.shakespeare()
    return [text[:20000], (text[:700] * 100)[:60000], datagen.low_entropy(30000), datagen.random_bytes(5000), text[100000 : 100000 + 66000],
            b"a" * 3000 + text[:2000]]


def mutants(blocks, sizes, count, seed=2024):
    """Yields (mutated block, orig_size argument).  `blocks` are valid encoded blocks, `sizes` their original sizes."""
    rng = np.random.default_rng(seed)
    for it in range(count):
        k = it % len(blocks)
        blk = bytearray(blocks[k])
        n = sizes[k]
        kind = it % 5
        if kind == 0:  # payload bit flips
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(9, len(blk)))
                blk[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # header field garbage
            p = int(rng.integers(0, min(17, len(blk))))
            blk[p] = int(rng.integers(0, 256))
        elif kind == 2:  # truncate
            blk = blk[: int(rng.integers(0, len(blk)))]
        elif kind == 3:  # random 32-bit field
            f = int(rng.integers(0, 4)) * 4 + (0 if rng.integers(0, 2) else 1)
            f = min(f, max(0, len(blk) - 4))
            blk[f : f + 4] = int(rng.integers(0, 2 ** 32)).to_bytes(4, "little")
        else:  # splice the payload of another block
            other = blocks[(k + 1) % len(blocks)]
            p = int(rng.integers(9, len(blk)))
            blk[p:] = other[p : p + len(blk) - p] + bytes(max(0, len(blk) - p - len(other[p:])))
        yield bytes(blk), (n if it % 7 else int(rng.integers(0, 70000)))
