"""The streaming driver (bz3_hip_encode_stream / bz3_hip_decode_stream: SURVEY.md 8f/N1, the reference CLI's process() loop of
src/main.c:351-407 as a read || code || write pipeline over large batches) beside the reference CLI at -j 64 on the same file on
tmpfs.  GPU box, no torch:
    python tools/stream_time.py [file MiB=8192] [-b MiB=32] [blocks_per_batch=768] [ref -j=64]
Both must write the same .bz3 bytes and restore the input.  Prints one JSON line."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bzip3_amd  # noqa: E402
import datagen  # noqa: E402


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(64 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    per_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    j = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    ref = os.path.join(ROOT, "oracle", "_ref", "bzip3")
    assert os.path.exists(ref), "oracle/_ref did not travel"
    lib = bzip3_amd.load()
    assert lib.bz3_hip_device_count() > 0
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        src = os.path.join(d, "in.txt")
        piece = datagen.text(min(mib, 256) << 20, seed=43, chains=65536)
        with open(src, "wb") as f:  # the file: the 256 MiB text with its 1 MiB pieces rotated per repetition (blocks differ, statistics do not)
            for r in range((mib + 255) // 256):
                k = (r * 37) % 256
                f.write(piece[k << 20 :] + piece[: k << 20])
        if mib < 256:
            os.truncate(src, mib << 20)
        del piece
        md5 = md5_of(src)
        rec = {"file_mib": mib, "block_mib": b, "tmpfs": base is not None}
        enc, back = os.path.join(d, "ref.bz3"), os.path.join(d, "ref.out")
        t0 = time.perf_counter(); subprocess.run([ref, "-e", "-b", str(b), "-j", str(j), "-f", src, enc], check=True); te = time.perf_counter() - t0
        t0 = time.perf_counter(); subprocess.run([ref, "-d", "-j", str(j), "-f", enc, back], check=True); td = time.perf_counter() - t0
        assert md5_of(back) == md5
        os.remove(back)
        rec["reference_cli"] = {"jobs": j, "t_enc_s": round(te, 2), "t_dec_s": round(td, 2), "round_trip_MiBps": round(mib / (te + td), 2), "bz3_bytes": os.path.getsize(enc)}
        ref_md5 = md5_of(enc)
        ours, back2 = os.path.join(d, "hip.bz3"), os.path.join(d, "hip.out")
        fi, fo = os.open(src, os.O_RDONLY), os.open(ours, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        t0 = time.perf_counter(); rc = lib.bz3_hip_encode_stream(fi, fo, b << 20, per_batch); te = time.perf_counter() - t0
        os.close(fi); os.close(fo)
        assert rc == 0, rc
        fi, fo = os.open(ours, os.O_RDONLY), os.open(back2, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        t0 = time.perf_counter(); rc = lib.bz3_hip_decode_stream(fi, fo, per_batch); td = time.perf_counter() - t0
        os.close(fi); os.close(fo)
        assert rc == 0, rc
        assert md5_of(back2) == md5, "stream driver: round trip changed the data"
        rec["stream_driver"] = {"blocks_per_batch": per_batch, "t_enc_s": round(te, 2), "t_dec_s": round(td, 2), "round_trip_MiBps": round(mib / (te + td), 2),
                                "bz3_bytes": os.path.getsize(ours)}
        # the CLI at -j N appends an empty chunk when the input is a multiple of the block size (main.c:352-362); -j 1 and the stream driver do not
        rec["identical_files"] = md5_of(ours) == ref_md5
        rec["stream_over_reference"] = round(rec["stream_driver"]["round_trip_MiBps"] / rec["reference_cli"]["round_trip_MiBps"], 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
