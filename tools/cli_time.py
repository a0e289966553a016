"""Wall clock of the UNMODIFIED reference CLI (src/main.c) on tmpfs, linked once against the reference's own libbz3.c (oracle/_ref/bzip3)
and once against bzip3_amd/lib/libbzip3.so ("additionally the unmodified CLI wall-clock on tmpfs").  GPU box, no torch:
    python tools/cli_time.py [file MiB=256] [-b MiB=32] [-j N=8]
File = synthetic text (tests/datagen.py) under /dev/shm; both CLIs must write identical .bz3 files and restore the input.
Prints one JSON line."""
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


def timed(cmd, **kw):
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True, **kw)
    return time.perf_counter() - t0


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    j = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ref = os.path.join(ROOT, "oracle", "_ref", "bzip3")
    main_o = os.path.join(ROOT, "oracle", "_ref", "bzip3_main.o")
    assert os.path.exists(ref) and os.path.exists(main_o), "oracle/_ref did not travel"
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as d, tempfile.TemporaryDirectory() as bindir:  # (/dev/shm is mounted noexec: the binary lives elsewhere)
        hip = os.path.join(bindir, "bzip3_hip")
        libdir = os.path.dirname(bzip3_amd.LIB_PATH)
        subprocess.check_call(["gcc", main_o, "-L" + libdir, "-lbzip3", "-Wl,-rpath," + libdir, "-lpthread", "-o", hip])
        src = os.path.join(d, "in.txt")
        data = datagen.text(mib << 20, seed=41, chains=65536)
        open(src, "wb").write(data)
        md5 = hashlib.md5(data).hexdigest()
        del data
        rec = {"file_mib": mib, "block_mib": b, "jobs": j, "tmpfs": base is not None}
        for name, exe in (("reference", ref), ("bzip3_amd", hip)):
            enc, back = os.path.join(d, name + ".bz3"), os.path.join(d, name + ".out")
            te = timed([exe, "-e", "-b", str(b), "-j", str(j), "-f", src, enc])
            td = timed([exe, "-d", "-j", str(j), "-f", enc, back])
            assert hashlib.md5(open(back, "rb").read()).hexdigest() == md5, name + ": round trip changed the data"
            rec[name] = {"t_enc_s": round(te, 2), "t_dec_s": round(td, 2), "round_trip_MiBps": round(mib / (te + td), 2), "bz3_bytes": os.path.getsize(enc),
                         "bz3_md5": hashlib.md5(open(enc, "rb").read()).hexdigest()}
            os.remove(back)
        rec["identical_files"] = rec["reference"]["bz3_md5"] == rec["bzip3_amd"]["bz3_md5"]
        assert rec["identical_files"], "the two CLIs wrote different files"
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
