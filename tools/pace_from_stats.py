#!/usr/bin/env python3
"""What paces a bench step, from the per-kernel summary of a profiled run (tools/rocpd_summary.py format, e.g.
profiles/r02_kernel_stats_bench_768x256MiB.txt):

    python tools/pace_from_stats.py <kernel_stats.txt> <blocks coded by the main step> [<all blocks that went through the library>]

Prints the whole-GPU kernel time of the encoder's front end and of the decoder's tail per block, and beside them the chains of the
serial single-workgroup kernels (LZP drivers / decoders: calls x average duration), which only stay hidden while they are shorter.
The second count (default: the first) includes the extra legs of bench.py (cfg3: 4 blocks, random: 64 small blocks)."""
import sys

FRONT = ("k_rs_", "k_scan_", "k_bwt_", "k_bg_", "k_lzp_links", "k_lzp_static", "k_lzp_hash", "k_lzp_emit", "k_mrle_", "k_crc_", "k_store_small", "k_write_header")
TAIL = ("k_ub_", "k_mrd_", "k_unstore_small")
SERIAL = {"k_lzp_driver": "LZP drivers (encode front end)", "k_lzp_decode": "LZP decoders (decode tail)"}
CM = ("k_cm_encode", "k_cm_decode")


def rows(path):
    for line in open(path):
        f = line.split()
        if line.startswith("#") or len(f) < 7 or not f[-5].replace(".", "").isdigit():
            continue
        name = line[: line.index(f[-6], 40) if f[-6] in line[40:] else 100].strip()
        yield name, int(f[-6]), float(f[-5]), float(f[-3])  # calls, total ms, average us


def main():
    path, nblk = sys.argv[1], int(sys.argv[2])
    nall = int(sys.argv[3]) if len(sys.argv) > 3 else nblk
    front = tail = 0.0
    serial, cm = {}, {}
    inv_sort = 0.0
    for name, calls, total_ms, avg_us in rows(path):
        short = name.split("(")[0].replace("void ", "").replace("bz3::", "")
        if any(k in name for k in SERIAL):
            key = next(k for k in SERIAL if k in name)
            serial[key] = (calls, total_ms, avg_us)
        elif any(k in name for k in CM):
            cm[short] = (calls, total_ms)
        elif "unsigned char" in name and "k_rs_" in name:  # the inverse BWT's one radix pass over bytes
            inv_sort += total_ms
        elif any(k in name for k in TAIL):
            tail += total_ms
        elif any(k in name for k in FRONT):
            front += total_ms
    tail += inv_sort
    print(f"whole-GPU kernels, encoder front end : {front / 1e3:8.1f} s = {front / nall:7.1f} ms per block  ({nall} blocks)")
    print(f"whole-GPU kernels, decoder tail      : {tail / 1e3:8.1f} s = {tail / nall:7.1f} ms per block")
    for key, (calls, total_ms, avg_us) in serial.items():
        print(f"{SERIAL[key]:<37}: {calls} launches x {avg_us / 1e6:.3f} s = {total_ms / 1e3:.1f} s in a row on one stream = {total_ms / nall:.1f} ms per block"
              f"  (hidden only while the launches in flight together bring this below the kernels' figure)")
    for name, (calls, total_ms) in sorted(cm.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:<37}: {calls} launch(es), {total_ms / 1e3:.1f} s")


if __name__ == "__main__":
    main()
