"""Per-phase static instruction counts of the row-per-pair forward/backward kernels.
   python scripts/isa_phases.py fwd|bwd [IT] [RAW]     (compiles one instantiation with -DDFEPE_ISA_MARKS to assembly)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(REPO, "pytorch-deepfepe_amd", "csrc")

SRC = {
    "fwd": """#include "dfepe_common.h"
#include "w8pt16_body.h"
__global__ void __launch_bounds__(256) k(const W8Args A) {
  __shared__ double xch[16 * 36];
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * 16 + row;
  if (pair >= A.B) return;
  w8pt16_fwd_pair<%d, %s, true>(A, pair, xch + row * 36);
}
""",
    "bwd": """#include "dfepe_common.h"
#include "w8pt16_bwd_body.h"
__global__ void __launch_bounds__(256) k(const W8BwdArgs A) {
  const int row = (int)(threadIdx.x >> 4);
  const int pair = (int)blockIdx.x * 16 + row;
  if (pair >= A.B) return;
  w8pt16_bwd_pair_impl<%d, %s, false>(A, pair, nullptr);
}
""",
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    raw = sys.argv[3] if len(sys.argv) > 3 else "true"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "one.hip")
        open(src, "w").write(SRC[which] % (it, raw))
        out = os.path.join(d, "one.s")
        subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-DDFEPE_ISA_MARKS", *os.environ.get("DFEPE_EXTRA_DEFS", "").split(),
                        f"-I{REPO}/include", f"-I{CSRC}", "-S", "--cuda-device-only", src, "-o", out,
                        "-Rpass-analysis=kernel-resource-usage"], check=True, stderr=subprocess.PIPE)
        lines = open(out).read().split("\n")
    cur, c = "start", collections.OrderedDict()
    tot = collections.Counter()
    for line in lines:
        m = re.search(r"; MARK (\w+)", line)
        if m:
            cur = m.group(1)
            continue
        t = line.strip()
        if ".Lfunc_end" in t:
            break
        if not line.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        d = c.setdefault(cur, collections.Counter())
        for dd in (d, tot):
            dd["all"] += 1
            if op.startswith("v_"):
                dd["valu"] += 1
            if op.startswith(("v_mov", "v_accvgpr")):
                dd["mov"] += 1
            if op.startswith("v_cndmask"):
                dd["cnd"] += 1
            if op.startswith("s_"):
                dd["salu"] += 1
            if "dpp" in t:
                dd["dpp"] += 1
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log")):
                dd["trans"] += 1
    for k, v in c.items():
        print(f"{k:22s}", dict(v))
    print("TOTAL", dict(tot))


if __name__ == "__main__":
    main()
