"""Build libdfepe_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python pytorch-deepfepe_amd/build.py [--force]

hipcc cross-compiles gfx950 without a GPU.  Objects are cached under csrc/build/ (git-ignored), each next to a
stamp holding the SHA-256 of its source, of every header and of the compiler flags: an object is reused only when
that content hash matches, so a build() provably compiles what is in the tree (file times play no role; the
prebuilt objects that travel to the GPU box are rebuilt there if anything differs).  The shared library lands next
to this file so it travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
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
CFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-fast-math", "-ffp-contract=on",
          "-mllvm", "-amdgpu-kernarg-preload-count=16"]  # leading scalar kernel arguments arrive in SGPRs (w8pt16.hip)
CFLAGS += os.environ.get("DFEPE_EXTRA_FLAGS", "").split()  # experiment builds (A/B timing of a -D switch); empty for the product
FLAGS = CFLAGS + [f"-I{INCLUDE}", f"-I{CSRC}"]
# what the stamps hash: the flags with the include directories relative to the repository, so that a relocated tree (the
# snapshot on the GPU box, a scratch copy) reuses its prebuilt objects instead of recompiling all of them
STAMP_FLAGS = " ".join(CFLAGS + ["-Iinclude", "-Icsrc"])


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _digest(paths, extra: str) -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_matches(stamp: str, digest: str) -> bool:
    try:
        return open(stamp).read().strip() == digest
    except OSError:
        return False


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = sorted([os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "dfepe.h")])
    jobs = []
    objs = []
    stamps = {}
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        objs.append(obj)
        digest = _digest([src] + hdrs, STAMP_FLAGS)
        if force or not os.path.exists(obj) or not _stamp_matches(obj + ".sha256", digest):
            jobs.append([hipcc, *FLAGS, "-c", src, "-o", obj])
            stamps[obj] = digest

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
    for obj, digest in stamps.items():  # only after the compile succeeded
        with open(obj + ".sha256", "w") as f:
            f.write(digest + "\n")
    link_digest = _digest(objs, "link")
    if force or jobs or not os.path.exists(LIB) or not _stamp_matches(os.path.join(OBJDIR, "lib.sha256"), link_digest):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB])
        with open(os.path.join(OBJDIR, "lib.sha256"), "w") as f:
            f.write(link_digest + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
