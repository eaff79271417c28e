#!/usr/bin/env python3
"""profiles/<round>/README.md from the files of the round's one lease (tools/round_r06.sh) -- every figure below is read
from a file in the directory, none is typed in.

    python tools/round_readme.py gpurun_out/r06 > gpurun_out/r06/README.md
"""
import csv
import glob
import json
import os
import re
import sys


def load_line(path):
    try:
        lines = [l for l in open(path).read().splitlines() if l.strip().startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:
        return None


def fmt(v, spec="%.4g"):
    return "n/a" if v is None else spec % v


def main():
    d = sys.argv[1]
    out = []
    p = out.append
    p("# Round 6 measurements (one MI355X; every file here was written by `tools/round_r06.sh` on one lease, commit `%s`)"
      % os.environ.get("HGMM_COMMIT", "unrecorded"))
    p("")
    p("Figures in this file are generated from the files named beside them (`tools/round_readme.py`).")
    p("")
    t = open(os.path.join(d, "pytest_gpu.log")).read() if os.path.exists(os.path.join(d, "pytest_gpu.log")) else ""
    m = re.findall(r"\d+ passed[^\n]*", t)
    p("* parity suite: %s (`pytest_gpu.log`); smoke: `smoke.log`" % (m[-1] if m else "n/a"))
    b = load_line(os.path.join(d, "bench_n1.json"))
    if b:
        r = b["roofline"]
        p("* headline (`bench_n1.json`): **%s it/s**, %.4f ms per step; materialising E-step %.1f GB/s = **%.3f of 8 TB/s** "
          "(patterns %s ms; store pacer %s GB/s, %s steps down, %s probes held); traffic %s B per launch (%s algorithmic)"
          % (fmt(b["value"], "%.1f"), b["ms_per_step"], r["achieved"], r["frac"],
             ", ".join("%.4f" % v for v in r["patterns_ms"].values()), fmt(r["store_pacer"].get("offered_GBs"), "%.0f"),
             r["store_pacer"].get("controller_steps_down"), r["store_pacer"].get("controller_probes_held"),
             fmt(r.get("traffic"), "%.4g"), fmt(r.get("algorithmic_bytes_per_launch"), "%.4g")))
        s = b.get("summary", {})
        p("* side legs (`bench_legs_n1.json`, summary of the line): " + ", ".join("%s %s" % (k, json.dumps(v)) for k, v in s.items()
                                                                                 if not k.startswith("chart_")))
    stats = os.path.join(d, "rocprofv3_kernel_stats.csv")
    if os.path.exists(stats):
        p("")
        p("## rocprofv3 --kernel-trace --stats of the bench command (`rocprofv3_kernel_stats.csv`; its JSON line: "
          "`bench_n1_under_rocprofv3.json`)")
        p("")
        p("| kernel | calls | mean us | min us | max us |")
        p("|---|---|---|---|---|")
        rows = list(csv.DictReader(open(stats)))
        for row in rows[:14]:
            name = row["Name"].split("(")[0].replace("void ", "").replace("hgmm::", "")
            p("| `%s` | %s | %.1f | %.1f | %.1f |" % (name, row["Calls"], float(row["AverageNs"]) / 1e3, float(row["MinNs"]) / 1e3,
                                                   float(row["MaxNs"]) / 1e3))
    pt = os.path.join(d, "pmc_traffic.json")
    if os.path.exists(pt):
        t = json.load(open(pt))
        p("")
        p("## HBM traffic counters and the E-step runs of this lease (`pmc_traffic.json`, `rocprofv3_pmc_*.csv`)")
        p("")
        p("* E-step: %s B written + %s B fetched per launch; rocprofv3 mean %s us"
          % (fmt(t.get("flat_estep_write_bytes"), "%.5g"), fmt(t.get("flat_estep_fetch_bytes"), "%.4g"),
             fmt(t.get("flat_estep_rocprof_mean_us"), "%.1f")))
        p("")
        p("| run | hipEvent mean ms | frac of 8 TB/s | pacer GB/s | steps down | probes held |")
        p("|---|---|---|---|---|---|")
        for tag, r in (t.get("estep_runs_of_this_lease") or {}).items():
            sp = r.get("store_pacer", {})
            p("| %s | %s | %s | %s | %s | %s |" % (tag, fmt(r.get("avg_launch_ms"), "%.4f"), fmt(r.get("frac"), "%.3f"),
                                                 fmt(sp.get("offered_GBs"), "%.0f"), sp.get("controller_steps_down"),
                                                 sp.get("controller_probes_held")))
    files = sorted(glob.glob(os.path.join(d, "bench_pairs_n1_c*_b*.json")))
    if files:
        p("")
        p("## Independent scan pairs: contexts per GPU x pairs per launch set (`bench_pairs_n1_c<C>_b<B>.json`; default line: "
          "`bench_pairs_n1.json`)")
        p("")
        p("| contexts | batch | pairs/s, float32 scans (float32 pdfs in the stop rule) | pairs/s, float64 scans | ms per step | "
          "max misalignment mm |")
        p("|---|---|---|---|---|---|")
        rows = []
        for f in files:
            r = load_line(f)
            if r:
                both = {r.get("scan_dtype", "float64"): r["value"]}
                o = r.get("other_scan_dtype")
                if o:
                    both[o["scan_dtype"]] = o["pairs_per_s"]
                rows.append((r["config"]["contexts_per_gpu"], r["config"]["batch"], both.get("float32"), both.get("float64"),
                             r["ms_per_step"], r["accuracy"]["max_misalignment_after_mm"]))
        for c, bt, v32, v64, ms, acc in sorted(rows):
            p("| %d | %d | %s | %s | %.2f | %.2f |" % (c, bt, fmt(v32, "%.0f"), fmt(v64, "%.0f"), ms, acc))
        dflt = load_line(os.path.join(d, "bench_pairs_n1.json"))
        if dflt:
            o = dflt.get("other_scan_dtype") or {}
            p("")
            p("default (`--contexts-per-gpu %d --batch %d --scan-dtype %s`): **%.0f pairs/s**; %s scans: %s pairs/s; host phases "
              "per batch, ms: %s" % (dflt["config"]["contexts_per_gpu"], dflt["config"]["batch"], dflt.get("scan_dtype"),
                                     dflt["value"], o.get("scan_dtype"), fmt(o.get("pairs_per_s"), "%.0f"),
                                     json.dumps({k: round(v, 2) for k, v in (dflt.get("host_phases_ms_per_batch") or {}).items()})))
        busy = os.path.join(d, "gpu_busy_pairs_default.txt")
        if os.path.exists(busy):
            p("")
            p("GPU under the default line (`gpu_busy_pairs_default.txt`, rocprofv3 kernel trace of the same command):")
            p("")
            p("```")
            out.extend(open(busy).read().strip().splitlines()[:6])
            p("```")
    for name, title in (("pair_batch_probe.log", "float64 scans"), ("pair_batch_probe_f32.log", "float32 scans"),
                        ("pair_batch_probe_f32_device_solve.log", "float32 scans, reg_device_solve = 1")):
        probe = os.path.join(d, name)
        if os.path.exists(probe):
            p("")
            p("## Where a batch's time goes, one context, %s (`%s`)" % (title, name))
            p("")
            p("```")
            out.extend(open(probe).read().strip().splitlines())
            p("```")
    for name in ("forest_levels_batch32_f32.txt", "forest_levels_batch32_f64.txt"):
        fl = os.path.join(d, name)
        if os.path.exists(fl):
            durs = [float(l.split()[6]) for l in open(fl) if "forest_ll_estep" in l or ("forest_estep_kernel" in l)]
            span = [l for l in open(fl) if l.startswith("build span")]
            p("")
            p("* `%s`: the E-step / log-likelihood launches of the last build of 32 pairs, us, in launch order: first 3 %s ... "
              "last 3 %s; %s" % (name, durs[:3], durs[-3:], span[0].strip() if span else ""))
    kt = os.path.join(d, "kernel_trace_batch32.txt")
    if os.path.exists(kt):
        p("")
        p("## Kernel trace of batches of 32 pairs (`kernel_trace_batch32.txt`; SQ counters: `pmc_sq_batch32.txt`)")
        p("")
        p("```")
        out.extend(open(kt).read().strip().splitlines()[:16])
        p("```")
    fp = os.path.join(d, "fullcov_f32_probe.log")
    if os.path.exists(fp):
        p("")
        p("## Flat full-covariance EM, N = 10^6, J = 800: float64 tile vs float32 tile (`fullcov_f32_probe.log`)")
        p("")
        p("```")
        out.extend(l for l in open(fp).read().strip().splitlines() if "tile" in l or l.startswith("after"))
        p("```")
    for n in (2, 8):
        r = load_line(os.path.join(d, "bench_n%d_rehearsal_one_gpu_peer_exchange.json" % n))
        if r and "sharded_tree" in r:
            st = r["sharded_tree"]
            p("")
            p("* N = %d rehearsal on ONE GPU (flow check, `bench_n%d_rehearsal_one_gpu_peer_exchange.json`): sharded tree level "
              "iterations %s, %s collectives of which %s surplus" % (n, n, st.get("level_iterations"), st.get("collectives"),
                                                                     st.get("surplus_collectives")))
    print("\n".join(out))


if __name__ == "__main__":
    main()
