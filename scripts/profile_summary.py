"""Turn gpurun_out/prof_<tag>/ (written by scripts/profile_round.sh on the GPU box) into the tracked files under profiles/:
<tag>_rocprof_summary.md, <tag>_bench_kernel_stats.csv, <tag>_bench_line_profiled.json and traffic.json.
usage: python scripts/profile_summary.py r02"""
import collections, csv, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(REPO, "gpurun_out", f"prof_{tag}")
P = os.path.join(REPO, "profiles")
B, N, L, M = 4096, 100, 5, 100
FIT_FWD, FIT_BWD, TAIL, HEAD = "w8pt16_fwd_kernel<7, true, true>", "w8pt16_bwd_kernel<7, true, false, true, false>", "loss_tail_kernel<7, false>", "loss_tail_head_kernel"
BWD_HEAD = "w8pt16_bwd_head_kernel<7, true, false>"  # the first backward fit of the step, with the deferred loss head in spare wavefronts
# kernels only the reference's call sequence runs (bench.py: api_path), not the timed step
API = ("loss_tail_kernel<7, true>", "loss_tail_bwd_kernel", "loss_stats_kernel", "row_dot_kernel", "deepf_input_kernel")
HOT = (FIT_FWD, FIT_BWD, BWD_HEAD, TAIL, HEAD) + API


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


def counters(sub):
    """{short kernel name: [ {counter: value} per dispatch, in dispatch order ]}"""
    path = os.path.join(O, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return {}
    rows = list(csv.DictReader(open(path)))
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
shutil.copy(os.path.join(O, "trace", "bench_kernel_stats.csv"), os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
json.dump(line, open(os.path.join(P, f"{tag}_bench_line_profiled.json"), "w"), indent=1)

trace = list(csv.DictReader(open(os.path.join(O, "trace", "bench_kernel_trace.csv"))))
groups = collections.OrderedDict()
for r in trace:
    d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    groups.setdefault((short(r["Kernel_Name"]), int(r["Grid_Size_X"])), []).append(d)
md = [f"# {tag} — rocprofv3 summaries (MI355X, B={B}/GPU, N={N}, depth {L}, fused step = 11 launches in one hipGraph)", "",
      "Produced by `scripts/profile_round.sh` (GPU box) + `scripts/profile_summary.py` (here).", "",
      "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 200 --warmup 20`", "",
      f"The verbatim per-name `--stats` table is `{tag}_bench_kernel_stats.csv`.  bench.py runs, besides the timed region, the roofline probe "
      "(a hipGraph of 50 stand-alone fits), the informational variants (layers batched into one launch; the recurrent-shaped backward; "
      "the whole DeepFNet with its estimator GEMMs; descriptor matching), so the table below is grouped per (kernel, grid) from the "
      "per-dispatch trace of the same run; only the hot-path kernels are listed.", "",
      "| kernel | grid (threads) | what | calls | avg us | median us | min us | max us |", "|---|---|---|---|---|---|---|---|"]
fwd_avg_us = None
step_sum = 0.0
for (k, g), ds in groups.items():
    if k not in HOT:
        continue
    ds_sorted = sorted(ds)
    if k in (FIT_FWD, FIT_BWD):
        if g == B * 16:
            what = "per-layer launch (one 16-lane row per pair: 4096 pairs = 1024 wavefronts)" + (" (in-step + roofline probe)" if k == FIT_FWD else "")
            if k == FIT_FWD:
                fwd_avg_us = mean(ds) / 1e3
            step_sum += (L if k == FIT_FWD else L - 1) * mean(ds) / 1e3  # the first backward launch is BWD_HEAD
        elif g == B * 16 * L:
            what = "layers-batched variant (informational)"
        else:
            what = "other batch size (full-model / sample checks)"
    elif k == BWD_HEAD:
        what = "first backward fit of the step + the deferred loss head in three spare wavefronts of workgroup 0 (448-thread workgroups)"
        step_sum += mean(ds) / 1e3
    elif k == HEAD:
        what = "loss head as a launch of its own: only the informational layers-batched variant (the timed step defers it)"
    elif k in API:
        what = "`api_path` only (behind DeepFNet.forward / get_all_loss_DeepF / get_Rt_loss; not in the timed step)"
    else:
        what = "once per step"
        step_sum += mean(ds) / 1e3
    md.append(f"| `{k}` | {g} | {what} | {len(ds)} | {mean(ds)/1e3:.2f} | {ds_sorted[len(ds)//2]/1e3:.2f} | {min(ds)/1e3:.2f} | {max(ds)/1e3:.2f} |")
md += ["", f"bench line of the same (profiled) run: `value` = {line['value']:.0f} pairs/s, ms_per_step = {line['ms_per_step']}, "
       f"roofline.avg_kernel_us = {line['roofline']['avg_kernel_us']} (HIP events around a hipGraph of 50 back-to-back stand-alone fits, "
       f"i.e. kernel + the dispatch gap between dependent launches).  The rocprof average of the forward fit at the hot-path grid is "
       f"{fwd_avg_us:.2f} us.  Sum of the step's kernel averages: {step_sum:.1f} us against the {1e3*line['ms_per_step']:.1f} us step"
       + ("; the rest is dispatch gaps between the 11 dependent launches." if step_sum < 1e3 * line['ms_per_step'] else
          " -- the averages pool every launch of the run (roofline probe, block repeats, the informational variants, all under the profiler's "
          "per-dispatch instrumentation), the step is timed on the graph replays alone: since round 5 (HIP graph packet capture off, "
          "pytorch-deepfepe_amd/__init__.py) the replayed step shows no measurable gap between its 11 dependent launches."), ""]

fetch, write = counters("pmc_fetch"), counters("pmc_write")
alg = {  # algorithmic bytes per launch in the fused step (SURVEY.md 8d + what the step additionally writes)
    FIT_FWD: B * (28 * N + 36 + 512 + 4 * N),
    FIT_BWD: B * (16 * N + 4 * N + 512 + 36 + 36 + 4 * N),
    TAIL: B * (24 * M + 36 * L + 36 + 72 + L * (4 + 36 + 16 + 4 + 36)),
}
md += ["## PMC (separate passes, `rocprofv3 --kernel-trace --pmc <counters> -- python scripts/pmc_probe.py`, averages per launch)", "",
       "FETCH_SIZE / WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md (HBM / rocprofv3 section) the gfx950 FETCH_SIZE of a wide "
       "coalesced stream is half the bytes actually fetched, so `read_bytes = 2 * FETCH_SIZE * 1024`; WRITE_SIZE is taken as is.", "",
       "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM-side bytes/launch (2F+W)*1024 | algorithmic bytes/launch | ratio |", "|---|---|---|---|---|---|"]
traffic = {}
for k in fetch:
    if k not in alg or k not in write:
        continue
    fs, ws = [v["FETCH_SIZE"] for v in fetch[k]], [v["WRITE_SIZE"] for v in write[k]]
    if k == FIT_FWD:  # the last 4 dispatches are the stand-alone probe configuration (pmc_probe.py)
        pf, pw = mean(fs[-4:]), mean(ws[-4:])
        fs, ws = fs[:-4], ws[:-4]
        pb = (2 * pf + pw) * 1024
        traffic[f"fit_fwd_B{B}_N{N}"] = int(pb)
        a = B * (28 * N + 36)
        md.append(f"| `{k}` stand-alone (bench.py roofline probe: weights in, epi + 512-B save out) | {pf:.0f} | {pw:.0f} | {pb/1e6:.2f} MB | "
                  f"{a/1e6:.2f} MB (28N+36; +2.10 MB save = {(a + B*512)/1e6:.2f} MB) | {pb/(a + B*512):.2f} (vs incl. save) |")
    f_, w_ = mean(fs), mean(ws)
    hb = (2 * f_ + w_) * 1024
    md.append(f"| `{k}` in the fused step | {f_:.0f} | {w_:.0f} | {hb/1e6:.2f} MB | {alg[k]/1e6:.2f} MB | {hb/alg[k]:.2f} |")
md += ["", "(the forward fit in the fused step also writes the softmax `weights_out` (4N) and the 512-B `save` record per pair; the "
       "stand-alone row is the launch `roofline.achieved` is computed from, its `traffic` goes to `traffic.json`.)", ""]
traffic["_note"] = ("HBM-side bytes per stand-alone forward-fit launch = (2*FETCH_SIZE + WRITE_SIZE) KiB from rocprofv3 --pmc (separate passes), same launch "
                    "configuration as bench.py's roofline probe (writes the 512-B save record on top of the 28N+36 algorithmic bytes)")

sq = counters("pmc_sq")
md += ["| kernel | waves | VALU inst/wave | LDS inst/wave | SALU inst/wave | wave-cycles/wave (x4 = clk) | active % | issue-stall % | wait % |",
       "|---|---|---|---|---|---|---|---|---|"]
valu = None
for k in sq:
    if k not in HOT:
        continue
    v = sq[k][:-4] if k == FIT_FWD else sq[k]
    g = lambda c: mean(x[c] for x in v)
    wv = g("SQ_WAVES")
    wc = g("SQ_WAVE_CYCLES")
    act, wait_any, wait_inst = g("SQ_ACTIVE_INST_ANY"), g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY")
    if k == FIT_FWD:
        valu = g("SQ_INSTS_VALU") / wv
    md.append(f"| `{k}` | {wv:.0f} | {g('SQ_INSTS_VALU')/wv:.0f} | {g('SQ_INSTS_LDS')/wv:.0f} | {g('SQ_INSTS_SALU')/wv:.0f} | {wc/wv:.0f} | "
              f"{100*act/wc:.0f} | {100*wait_inst/wc:.0f} | {100*wait_any/wc:.0f} |")
clk_ghz = None
path = os.path.join(O, "pmc_clk", "p_counter_collection.csv")
if os.path.exists(path):
    rows = list(csv.DictReader(open(path)))
    busy = [(float(r["Counter_Value"]) / 256.0, (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3) for r in rows
            if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES" and "w8pt16_fwd" in r["Kernel_Name"]]
    if busy:
        clk_ghz = mean(b / d for b, d in busy) / 1e3
        md += ["", f"Sustained shader clock during the forward fit (SQ_BUSY_CU_CYCLES / 256 CUs / kernel duration, a lower bound because the CUs are "
               f"not busy during the launch ramp): **{clk_ghz:.2f} GHz** (mean of {len(busy)} launches)."]
if valu and fwd_avg_us:
    floor_us = valu * 4 / 2.4e3  # ONE wavefront (four pairs) per SIMD, 4 issue cycles per wave64 VALU instruction, 2.4 GHz
    md += ["", f"VALU-issue floor of the forward fit at this occupancy: 4096 pairs = 1024 wavefronts = ONE per SIMD x {valu:.0f} VALU instructions x "
           f"4 cycles = {valu*4:.0f} cycles = {floor_us:.1f} us at 2.4 GHz, against {fwd_avg_us:.1f} us measured ({100*floor_us/fwd_avg_us:.0f} %): with a single "
           "wavefront per SIMD every dependent-issue bubble, DPP wait state, scalar instruction, branch and memory wait is exposed "
           "(scripts/ubench/lat.hip, lat2.hip: a lone wavefront issues a dependent fp64 FMA every 5.2 cycles, an independent one every 4.1; a scalar "
           "instruction or a branch costs it ~9).  Round 3 showed which of these matter: removing 14 % of the vector instructions moved the "
           "kernel by 3 %, removing 13 scalar instructions and 3 branches from each of the ~10 rounds of the one loop moved it by 8 %.  "
           "This is a fraction of the kernel's own instruction stream, not a roofline."]
    traffic[f"fit_fwd_valu_insts_per_wave_B{B}_N{N}"] = round(valu, 1)
if clk_ghz:
    traffic["fit_fwd_sustained_clock_ghz"] = round(clk_ghz, 3)
if fwd_avg_us:
    traffic[f"fit_fwd_rocprof_avg_us_B{B}_N{N}"] = round(fwd_avg_us, 3)
    traffic["fit_fwd_rocprof_source"] = f"profiles/{tag}_bench_kernel_stats.csv (rocprofv3 --kernel-trace average at the hot-path grid)"
json.dump(traffic, open(os.path.join(P, "traffic.json"), "w"), indent=1)
open(os.path.join(P, f"{tag}_rocprof_summary.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
