"""Turn gpurun_out/prof_<tag>/ (written by scripts/profile_round.sh on the GPU box) into the tracked files under profiles/:
<tag>_rocprof_summary.md, <tag>_bench_kernel_stats.csv, <tag>_bench_line_profiled.json and traffic.json.
usage: python scripts/profile_summary.py r01"""
import collections, csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
P = os.path.join(REPO, "profiles")
B, N, L, M = 4096, 100, 5, 100


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0][:70]
    n = n.replace("w8pt_fwd_kernel<true, 1>", "w8pt_fwd_kernel<true>")  # RAW, one wavefront per pair
    return n.replace("w8pt_bwd_kernel<true, false, false>", "w8pt_bwd_kernel<true, false>")  # RAW, no point gradients, one wavefront per pair


def counters(sub):
    """{short kernel name: [ {counter: value} per dispatch, in dispatch order ]}"""
    rows = list(csv.DictReader(open(os.path.join(O, sub, "p_counter_collection.csv"))))
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault((int(r["Dispatch_Id"]), short(r["Kernel_Name"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = collections.OrderedDict()
    for (d, k), v in sorted(per.items()):
        out.setdefault(k, []).append(v)
    return out


def mean(xs):
    xs = list(xs)
    return sum(xs) / max(len(xs), 1)


line = json.loads(open(os.path.join(O, "bench_line.json")).read().strip().splitlines()[-1])
stats = list(csv.DictReader(open(os.path.join(O, "trace", "bench_kernel_stats.csv"))))
shutil.copy(os.path.join(O, "trace", "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
json.dump(line, open(os.path.join(P, f"{tag}_bench_line_profiled.json"), "w"), indent=1)

# per-(kernel, grid) statistics from the per-dispatch trace: bench.py also times two informational variants after the
# timed region (all layers in one launch = grid 5x larger; the whole DeepFNet with the estimator GEMMs), which the
# plain per-name --stats table (copied verbatim next to this file) lumps together with the hot-path launches
trace = list(csv.DictReader(open(os.path.join(O, "trace", "bench_kernel_trace.csv"))))
groups = collections.OrderedDict()
tot_ns = 0.0
for r in trace:
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    tot_ns += d
    groups.setdefault((short(r["Kernel_Name"]), int(r["Grid_Size_X"])), []).append(d)
HOT = ("w8pt_fwd", "w8pt_bwd", "floss_kernel", "pose_fwd", "pose_bwd", "loss_head")
md = [f"# {tag} — rocprofv3 summaries (MI355X, B={B}/GPU, N={N}, depth {L}, fused step in one hipGraph)", "",
      "Produced by `scripts/profile_round.sh` (GPU box) + `scripts/profile_summary.py` (here).", "",
      "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 50 --warmup 10`", "",
      f"The verbatim per-name `--stats` table is `{tag}_bench_kernel_stats.csv`.  bench.py runs, after the timed region, two informational "
      "variants (all layers' fits in one launch; the whole DeepFNet with its estimator GEMMs), so the table below is grouped per "
      "(kernel, grid) from the per-dispatch trace of the same run; only the hot-path kernels are listed.", "",
      "| kernel | grid (threads) | what | calls | avg us | min us | max us |", "|---|---|---|---|---|---|---|"]
fwd_avg_us = None
for (k, g), ds in groups.items():
    if not k.startswith(HOT):
        continue
    what = "per-layer launch" + (" (in-step + roofline probe)" if k.startswith("w8pt_fwd") else "")
    if k.startswith(("w8pt_fwd", "w8pt_bwd")):
        if g == B * 64:
            if k.startswith("w8pt_fwd"):
                fwd_avg_us = mean(ds) / 1e3
        elif g == B * 64 * L:
            what = "layers-batched variant (informational)"
        else:
            what = "other batch size (full-model / CPU-baseline sample checks)"
    else:
        what = "once per step"
    md.append(f"| `{k}` | {g} | {what} | {len(ds)} | {mean(ds)/1e3:.2f} | {min(ds)/1e3:.2f} | {max(ds)/1e3:.2f} |")
md += ["", f"bench line of the same (profiled) run: `value` = {line['value']:.0f} pairs/s, ms_per_step = {line['ms_per_step']}, "
       f"roofline.avg_kernel_us = {line['roofline']['avg_kernel_us']} (HIP events around 50 back-to-back stand-alone launches).  The rocprof "
       f"average of the `w8pt_fwd` launches with the hot-path grid ({B} pairs) is {fwd_avg_us:.2f} us — in-step launches (fused softmax, "
       "`weights_out` written) and probe launches together.", ""]

fetch, write = counters("pmc_fetch"), counters("pmc_write")
alg = {  # algorithmic bytes per launch in the fused step (SURVEY.md 8d + what the step additionally writes)
    "w8pt_fwd_kernel<true>": B * (28 * N + 36 + 512 + 4 * N),
    "w8pt_bwd_kernel<true, false>": B * (16 * N + 4 * N + 512 + 36 + 36 + 4 * N),
    "floss_kernel<false, true>": B * (24 * M + 36 * L + 36 + 72 + 4 * L + 36 * L),
    "floss_kernel<true, true>": B * (24 * M + 36 * L + 36 + 72 + 36 * L + 36 * L),
    "pose_fwd_kernel": B * (36 * L + 64 + 28 + 20 * L),
    "pose_bwd_kernel": B * (36 * L + 28 + 36 * L),
}
md += ["## PMC (separate passes, `rocprofv3 --kernel-trace --pmc <counters> -- python scripts/pmc_probe.py`, averages per launch)", "",
       "FETCH_SIZE / WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md (HBM / rocprofv3 section) the gfx950 FETCH_SIZE of a wide "
       "coalesced stream is half the bytes actually fetched, so `read_bytes = 2 * FETCH_SIZE * 1024`; WRITE_SIZE is taken as is.", "",
       "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM-side bytes/launch (2F+W)*1024 | algorithmic bytes/launch | ratio |", "|---|---|---|---|---|---|"]
traffic = {}
for k in fetch:
    if k not in alg:
        continue
    fs, ws = [v["FETCH_SIZE"] for v in fetch[k]], [v["WRITE_SIZE"] for v in write[k]]
    if k.startswith("w8pt_fwd"):  # the last 4 dispatches are the stand-alone probe configuration
        pf, pw = mean(fs[-4:]), mean(ws[-4:])
        fs, ws = fs[:-4], ws[:-4]
        pb = (2 * pf + pw) * 1024
        traffic[f"w8pt_fwd_B{B}_N{N}"] = int(pb)
        a = B * (28 * N + 36)
        md.append(f"| `{k}` stand-alone (bench.py roofline probe: weights in, epi + 512-B save out) | {pf:.0f} | {pw:.0f} | {pb/1e6:.2f} MB | "
                  f"{a/1e6:.2f} MB (28N+36; +2.10 MB save = {(a + B*512)/1e6:.2f} MB) | {pb/(a + B*512):.2f} (vs incl. save) |")
    f_, w_ = mean(fs), mean(ws)
    hb = (2 * f_ + w_) * 1024
    md.append(f"| `{k}`{' in the fused step' if k.startswith('w8pt_fwd') else ''} | {f_:.0f} | {w_:.0f} | {hb/1e6:.2f} MB | {alg[k]/1e6:.2f} MB | {hb/alg[k]:.2f} |")
md += ["", "(`w8pt_fwd` in the fused step also writes the softmax `weights_out` (4N) and the 512-B `save` record per pair; the "
       "stand-alone row is the launch `roofline.achieved` is computed from, its `traffic` goes to `traffic.json`.)", ""]
traffic["_note"] = ("HBM-side bytes per stand-alone w8pt_fwd launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from rocprofv3 --pmc (separate passes), same launch "
                    "configuration as bench.py's roofline probe (writes the 512-B save record on top of the 28N+36 algorithmic bytes)")
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)

sq, lds = counters("pmc_sq"), counters("pmc_lds")
md += ["| kernel | waves | VALU inst/wave | LDS inst/wave | SALU inst/wave | wave-cycles/wave (x4 = clk) | active % | issue-stall % | wait % | "
       "LDS bank-conflict cycles / LDS active |", "|---|---|---|---|---|---|---|---|---|---|"]
for k in sq:
    if k not in alg:
        continue
    v = sq[k][:-4] if k.startswith("w8pt_fwd") else sq[k]
    g = lambda c: mean(x[c] for x in v)
    wv = g("SQ_WAVES")
    wc = g("SQ_WAVE_CYCLES")
    act, wait_any, wait_inst = g("SQ_ACTIVE_INST_ANY"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY")
    lv = lds.get(k, [])
    conf = mean(x["SQ_LDS_BANK_CONFLICT"] for x in lv) / max(mean(x["SQ_LDS_IDX_ACTIVE"] for x in lv), 1.0) if lv else 0.0
    md.append(f"| `{k}` | {wv:.0f} | {g('SQ_INSTS_VALU')/wv:.0f} | {g('SQ_INSTS_LDS')/wv:.0f} | {g('SQ_INSTS_SALU')/wv:.0f} | {wc/wv:.0f} | "
              f"{100*act/wc:.0f} | {100*wait_inst/wc:.0f} | {100*wait_any/wc:.0f} | {conf:.3f} |")
clk_ghz = None
if os.path.exists(os.path.join(O, "pmc_clk", "p_counter_collection.csv")):
    rows = list(csv.DictReader(open(os.path.join(O, "pmc_clk", "p_counter_collection.csv"))))
    busy = [(float(r["Counter_Value"]) / 256.0, (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3) for r in rows
            if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES" and "w8pt_fwd" in r["Kernel_Name"]]
    if busy:
        clk_ghz = mean(b / d for b, d in busy) / 1e3
        md += ["", f"Sustained shader clock during `w8pt_fwd` (SQ_BUSY_CU_CYCLES / 256 CUs / kernel duration, a lower bound because the CUs are not "
               f"busy during the launch ramp): **{clk_ghz:.2f} GHz** (mean of {len(busy)} launches) - not the 2.4 GHz peak clock the MI355X tables quote."]
fv = sq.get("w8pt_fwd_kernel<true>")
if fv:
    v = fv[:-4]
    valu = mean(x["SQ_INSTS_VALU"] for x in v) / mean(x["SQ_WAVES"] for x in v)
    floor_us = 4 * valu * 4 / 2.4e3  # 4 waves per SIMD, 4 cycles per wave64 VALU instruction, 2.4 GHz
    md += ["", f"VALU-issue floor of `w8pt_fwd` at this occupancy: 4096 waves / 1024 SIMDs = 4 waves per SIMD x {valu:.0f} VALU instructions x "
           f"4 cycles = {4*valu*4:.0f} cycles = {floor_us:.1f} us at 2.4 GHz, against {fwd_avg_us:.1f} us measured: the kernel runs at "
           f"{100*floor_us/fwd_avg_us:.0f} % of the vector-issue bound of its own instruction stream; HBM is idle most of the time."]
    if clk_ghz:
        md[-1] += (f"  At the {clk_ghz:.2f} GHz the counters show, the same bound is {4*valu*4/clk_ghz/1e3:.1f} us, i.e. the launch runs at "
                   f"{100*(4*valu*4/clk_ghz/1e3)/fwd_avg_us:.0f} % of it: the kernel is vector-issue bound, and only fewer instructions make it faster (the Jacobi rounds, which are LDS-pipe bound, excepted).")
if fv:
    traffic[f"w8pt_fwd_valu_insts_per_wave_B{B}_N{N}"] = round(valu, 1)
if clk_ghz:
    traffic["w8pt_fwd_sustained_clock_ghz"] = round(clk_ghz, 3)
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
