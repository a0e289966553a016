"""Builds another copy of the product library with extra preprocessor definitions, for same-box A/B runs of an experiment:
    python tools/build_variant.py <name> -DCM_EXP_EARLY_ROW=1 ...   ->  bzip3_amd/lib/libbzip3_<name>.so
(the tools take it with --lib=<path>; it travels with gpurun like the product's own .so and is git-ignored)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bzip3_amd import build as B  # noqa: E402


def main():
    name, defs = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(B.HERE, "build", "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(B.LIB_DIR, exist_ok=True)
    objs, procs = [], []
    for src in B.SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([B._hipcc(), *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, src), "-o", obj]))
    assert all(p.wait() == 0 for p in procs), "compile failed"
    out = os.path.join(B.LIB_DIR, f"libbzip3_{name}.so")
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", out, *objs, "-Wl,-soname,libbzip3.so"])
    print(out)


if __name__ == "__main__":
    main()
