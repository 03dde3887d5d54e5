"""Static instruction statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel, instructions by
class.  Straight-line kernels (the row-per-pair solver is fully unrolled) make this a fair proxy for issue cycles."""
import collections
import re
import sys


def stats(path, pattern=""):
    txt = open(path).read()
    parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        name = parts[i]
        if pattern and pattern not in name:
            continue
        body = parts[i + 1].split(".Lfunc_end")[0]
        c = collections.Counter()
        for line in body.split("\n"):
            t = line.strip()
            if not line.startswith("\t") or not t or t[0] in ".;":
                continue
            op = t.split()[0]
            c["total"] += 1
            if "dpp" in t:
                c["dpp"] += 1
            if op.startswith("v_"):
                c["valu"] += 1
                if "_f64" in op:
                    c["f64"] += 1
                if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log")):
                    c["trans"] += 1
                if op.startswith("v_cndmask"):
                    c["cndmask"] += 1
                if op.startswith(("v_mov", "v_accvgpr")):
                    c["mov"] += 1
                if op.startswith("v_cmp"):
                    c["cmp"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
                if op.startswith("s_nop"):
                    c["nop"] += 1
                if op.startswith("s_waitcnt"):
                    c["waitcnt"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
        print(name[:90])
        print("   ", dict(c))


if __name__ == "__main__":
    stats(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
