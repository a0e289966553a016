"""TEST INFRASTRUCTURE ONLY.  Compiles the kernel sources of bzip3_amd/csrc with g++ against the fiber
emulation of the HIP execution model (tests/emu/hip_emu.hpp, -DBZ3_EMU) into
tests/emu/libbz3_emu_TESTONLY.so, so kernel logic can be diffed against the oracle on a machine
without a GPU.  The bzip3_amd package never loads this library."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "bzip3_amd", "csrc")
OUT = os.path.join(HERE, "libbz3_emu_TESTONLY.so")
SOURCES = ["sort.hip", "crc32c.hip", "mrle.hip", "lzp.hip", "bwt.hip", "unbwt.hip", "cm.hip", "api.hip", "stream.hip"]


def build():
    """BZ3_EMU_DEFS="-DX=1 ...": a variant build (round 6's compile-time kernel experiments under the fuzzers), in a library and object directory of its own."""
    global OUT
    defs = os.environ.get("BZ3_EMU_DEFS", "").split()
    tag = ("_" + hashlib.sha256(" ".join(defs).encode()).hexdigest()[:8]) if defs else ""
    OUT = os.path.join(HERE, f"libbz3_emu{tag}_TESTONLY.so")
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "hip_emu.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(HERE, "hip_emu.hpp")]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    h = hashlib.sha256()
    h.update(os.environ.get("BZ3_EMU_WATCH", "").encode())
    for p in sorted(deps):
        h.update(open(p, "rb").read())
    stamp = OUT + ".sha"
    if os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return OUT
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build" + tag), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build" + tag, os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DBZ3_EMU", *defs, *(["-DBZ3_EMU_WATCH"] if os.environ.get("BZ3_EMU_WATCH") else []), "-I", HERE, "-I", CSRC, "-x", "c++", "-c", s, "-o", o,
               "-Wno-unknown-pragmas", "-Wno-attributes"]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"emu build failed for {s}:\n{out}")
    subprocess.check_call(["g++", "-shared", "-o", OUT, *objs, "-lpthread"])
    open(stamp, "w").write(h.hexdigest())
    return OUT


if __name__ == "__main__":
    print(build())
