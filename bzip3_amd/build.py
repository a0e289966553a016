"""Builds bzip3_amd/lib/libbzip3.so: every HIP source under bzip3_amd/csrc compiled for gfx950.

`hipcc` cross-compiles without a GPU, so this runs in the build container as well as on the MI355X
box.  The shared object is kept in-tree (git-ignored, shipped by gpurun) and is the ONLY artefact the
package loads; there is no fallback of any kind.
"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libbzip3.so")
SOURCES = ["sort.hip", "crc32c.hip", "mrle.hip", "lzp.hip", "bwt.hip", "unbwt.hip", "cm.hip", "api.hip", "stream.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unknown-pragmas", "-Wno-pass-failed"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _headers():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    inc = os.path.join(os.path.dirname(HERE), "include")
    out += [os.path.join(inc, f) for f in os.listdir(inc)]
    return out


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest([os.path.join(CSRC, src)] + _headers())
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs, "-Wl,-soname,libbzip3.so"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(verbose=True)
