"""Build libdfepe_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python pytorch-deepfepe_amd/build.py [--force]

hipcc cross-compiles gfx950 without a GPU.  Objects are cached under csrc/build/ (git-ignored) and
only rebuilt when a source or header is newer; the shared library lands next to this file so it
travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.abspath(os.path.join(HERE, "..", "include"))
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libdfepe_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-ffp-contract=on", f"-I{INCLUDE}", f"-I{CSRC}"]
FLAGS += os.environ.get("DFEPE_EXTRA_FLAGS", "").split()  # experiment builds (A/B timing of a -D switch); empty for the product


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "dfepe.h")]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[dfepe build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _newer(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
