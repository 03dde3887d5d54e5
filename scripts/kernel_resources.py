#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel the library is linked from (code-object metadata of csrc/build/*.o, no GPU needed):
    python scripts/kernel_resources.py [substring ...]
Prints one line per kernel whose demangled name contains every given substring; marks spilling kernels."""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "pytorch-deepfepe_amd", "libdfepe_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(objdir=os.path.join(REPO, "pytorch-deepfepe_amd", "csrc", "build")):
    """One record per kernel of every object under csrc/build (the objects the library is linked from)."""
    out = []
    with tempfile.TemporaryDirectory() as d:
        for o in sorted(f for f in os.listdir(objdir) if f.endswith(".o")):
            co, fat = f"{d}/{o}.co", f"{d}/{o}.fat"
            subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", os.path.join(objdir, o), fat],
                           capture_output=True, text=True)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", f"--input={fat}",
                                f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: re.search(rf"\.{k}:\s+(\S+)", blk)
                name = g("name").group(1)
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                out.append(dict(obj=o, name=dem, vgpr=int(g("vgpr_count").group(1)), agpr=int(g("agpr_count").group(1)), sgpr=int(g("sgpr_count").group(1)),
                                scratch=int(g("private_segment_fixed_size").group(1)), lds=int(g("group_segment_fixed_size").group(1)),
                                spill=int(g("vgpr_spill_count").group(1))))
    return out


if __name__ == "__main__":
    pats = sys.argv[1:]
    for k in sorted(kernels(), key=lambda k: k["name"]):
        if all(p in k["name"] for p in pats):
            flag = "  <-- SCRATCH" if k["scratch"] or k["spill"] else ""
            print(f"{k['name'][:110]:110s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} scratch {k['scratch']:4d} lds {k['lds']:6d}{flag}")
