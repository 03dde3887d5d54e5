"""gpurun_out/prof_other_<tag>/ (scripts/profile_other.sh) -> profiles/<tag>_other_configs.md: per (kernel, grid) durations from the
kernel trace, FETCH_SIZE / WRITE_SIZE (KiB; gfx950 FETCH_SIZE counts half the bytes of a wide coalesced stream, MI355X_MICROARCH.md)
and the SQ counters, averaged per launch.   usage: python scripts/profile_other_summary.py r03"""
import collections, csv, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, "gpurun_out", f"prof_other_{tag}")
KEEP = ("w8pt16", "cheirality", "est_")


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


def mean(xs):
    xs = list(xs)
    return sum(xs) / max(len(xs), 1)


def counters(sub):
    path = os.path.join(O, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        per.setdefault((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), int(r["Grid_Size"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = collections.OrderedDict()
    for (d, k, g), v in sorted(per.items()):
        out.setdefault((k, g), []).append(v)
    return out


trace = list(csv.DictReader(open(os.path.join(O, "trace", "p_kernel_trace.csv"))))
dur = collections.OrderedDict()
for r in trace:
    k = short(r["Kernel_Name"])
    if any(s in k for s in KEEP):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        dur.setdefault((k, g), []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
fetch, write, sq, mf = counters("pmc_fetch"), counters("pmc_write"), counters("pmc_sq"), counters("pmc_mfma")
md = [f"# {tag} — kernels outside the bench line (rocprofv3 on one MI355X)", "",
      "`scripts/profile_other.sh` (kernel trace, then FETCH_SIZE / WRITE_SIZE / SQ counters in separate `--pmc` passes) over "
      "`scripts/pmc_probe_other.py`: BASELINE config 5 (N = 1000: one fit + E-from-F + cheirality) at 4096 and at 512 pairs, and one "
      "split-bf16 estimator call forward + backward at B = 4096, N = 100 (fused epilogue) and one at 12 pairs x 2000 points (plain product + `est_norm_fwd_n` / `est_in_bwd_n`; the GEMM rows of the two calls differ by their grids).  HBM-side bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB.  `est_gemm_nt_kernel<planes of A, planes of B, order, epilogue, format>`: epilogue 0 = plain fp32 store, 1 = InstanceNorm + LeakyReLU + split (the forward layer), 2 = the adjoint of the layer below (the fused data gradient); format 0 = bf16 planes, 1 = fp16 planes.", "",
      "| kernel | grid (threads) | launches | avg us | min us | HBM-side MB / launch | GB/s | VALU inst / wave | active % | wait % | MFMA busy % of CU-busy | LDS conflict % |",
      "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for (k, g), ds in dur.items():
    f = fetch.get((k, g)); w = write.get((k, g)); s = sq.get((k, g)); m = mf.get((k, g))
    hb = (2 * mean(v["FETCH_SIZE"] for v in f) + mean(v["WRITE_SIZE"] for v in w)) * 1024 if f and w else None
    avg = mean(ds)
    row = f"| `{k}` | {g} | {len(ds)} | {avg/1e3:.1f} | {min(ds)/1e3:.1f} | " + (f"{hb/1e6:.1f} | {hb/avg:.0f} | " if hb else "– | – | ")
    if s:
        wv = mean(v["SQ_WAVES"] for v in s); wc = mean(v["SQ_WAVE_CYCLES"] for v in s)
        row += f"{mean(v['SQ_INSTS_VALU'] for v in s)/wv:.0f} | {100*mean(v['SQ_ACTIVE_INST_ANY'] for v in s)/wc:.0f} | {100*mean(v['SQ_WAIT_ANY'] for v in s)/wc:.0f} | "
    else:
        row += "– | – | – | "
    if m and mean(v.get("SQ_BUSY_CU_CYCLES", 0) for v in m) > 0:
        busy = mean(v["SQ_BUSY_CU_CYCLES"] for v in m)
        row += f"{100*mean(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for v in m)/busy:.0f} | "
        idx = mean(v.get("SQ_LDS_IDX_ACTIVE", 0) for v in m)
        row += (f"{100*mean(v.get('SQ_LDS_BANK_CONFLICT', 0) for v in m)/idx:.0f} |" if idx > 0 else "– |")
    else:
        row += "– | – |"
    md.append(row)
out = os.path.join(REPO, "profiles", f"{tag}_other_configs.md")
open(out, "w").write("\n".join(md) + "\n")
print("\n".join(md))
