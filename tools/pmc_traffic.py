"""HBM traffic factors of the CM kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950:
/opt/skills/guides/MI355X_MICROARCH.md, "rocprofv3 PMC slots"), written into profiles/pmc_traffic.json, where bench.py picks them up
for `roofline.traffic`.

    python tools/pmc_traffic.py <fetch results.db> <write results.db> <bench line .json> [source text]

Both passes and the bench line must come from the same bench.py command (small blocks are fine: the factors are per byte).
FETCH_SIZE is doubled as the guide prescribes for gfx950 (128-byte requests tallied at 64 bytes); WRITE_SIZE is taken as reported.
Counter values are KiB per dispatch."""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(db_path, counter):
    con = sqlite3.connect(db_path)
    rows = con.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main():
    fetch_db, write_db, bench_json = sys.argv[1:4]
    source = sys.argv[4] if len(sys.argv) > 4 else ""
    line = json.load(open(bench_json))
    nblk, bs = line["config"]["blocks_per_gpu"], line["config"]["block_bytes"]
    steps = line["steps"] + line["warmup"]
    plain = nblk * bs * steps
    coded = plain / line["config"]["compressed_ratio"]
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out["units"] = "counter values are KiB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B); WRITE_SIZE as reported"
    if source:
        out["source_latest"] = source  # (the key was "source_r04" up to round 5)
    for name in sorted(set(fetch) | set(write)):
        short = name.split("(")[0].replace("bz3::", "").replace("void ", "")
        if not short.startswith("k_cm_"):
            continue
        f = fetch.get(name, (0, 0.0))[1] * 1024 * 2
        w = write.get(name, (0, 0.0))[1] * 1024
        rec = {"plain_bytes": int(plain), "coded_bytes": int(coded), "fetch_size_kib_raw": round(fetch.get(name, (0, 0.0))[1], 2), "write_size_kib_raw": round(write.get(name, (0, 0.0))[1], 2)}
        if "decode" in short:
            rec["fetch_bytes_per_coded_byte"] = round(f / coded, 4)
            rec["write_bytes_per_decoded_byte"] = round(w / plain, 4)
        else:
            rec["fetch_bytes_per_input_byte"] = round(f / plain, 4)
            rec["write_bytes_per_coded_byte"] = round(w / coded, 4)
        out[short] = rec
        print(short, rec)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
