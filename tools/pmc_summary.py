"""Condenses the rocprofv3 outputs of tools/round_r06.sh (earlier rounds: profile_round.sh):
  <dir>/kt/**/kt_kernel_stats.csv           -> <dir>/rocprofv3_kernel_stats.csv (copied)
  <dir>/pmc_<COUNTER>/**/pmc_counter_collection.csv -> <dir>/rocprofv3_pmc_<counter>.csv (mean per kernel)
  -> <dir>/pmc_traffic.json (bytes per launch of the three flat kernels; FETCH_SIZE / WRITE_SIZE are KB;
     MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streaming reads by 2x on gfx950 --
     applied to the M-step kernel's resp stream only)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def per_kernel_mean(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    out = sys.argv[1]
    stats = find(os.path.join(out, "kt"), "*kernel_stats.csv")
    if stats:
        shutil.copy(stats, os.path.join(out, "rocprofv3_kernel_stats.csv"))
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        path = find(os.path.join(out, "pmc_" + counter), "*counter_collection.csv")
        if not path:
            continue
        means = per_kernel_mean(path, counter)
        with open(os.path.join(out, "rocprofv3_pmc_%s.csv" % counter.lower()), "w") as f:
            f.write("kernel,launches,mean_%s_KB\n" % counter)
            for k in sorted(means):
                f.write('"%s",%d,%g\n' % (k, means[k][1], means[k][0]))
        res[counter] = means

    def kb(counter, needle):
        for k, (mean, _) in res.get(counter, {}).items():
            if needle in k:
                return mean * 1024.0
        return None

    if res:
        e_f, e_w = kb("FETCH_SIZE", "flat_estep_rows_pk_kernel"), kb("WRITE_SIZE", "flat_estep_rows_pk_kernel")
        m_f, m_w = kb("FETCH_SIZE", "flat_mstep_kernel"), kb("WRITE_SIZE", "flat_mstep_kernel")
        f_f, f_w = kb("FETCH_SIZE", "flat_fused_pk_kernel<13>"), kb("WRITE_SIZE", "flat_fused_pk_kernel<13>")
        summary = {
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 4 "
                      "--warmup 1 --estep-reps 3 --no-cpu-baseline` (tools/profile_round.sh); counters are KB; per "
                      "MI355X_MICROARCH.md FETCH_SIZE under-reports wide (16 B/lane) coalesced streaming reads by 2x on "
                      "gfx950 -- applied to the M-step kernel's resp stream only (the E-step's reads are scalar loads "
                      "of X, uncorrected)",
            "measured_on": __import__("datetime").date.today().isoformat(),
            "commit": os.environ.get("HGMM_COMMIT", "unrecorded"),
            "flat_estep_bytes_per_launch": (e_f or 0) + (e_w or 0) if e_w else None,
            "flat_estep_fetch_bytes": e_f, "flat_estep_write_bytes": e_w,
            "flat_mstep_bytes_per_launch": 2 * (m_f or 0) + (m_w or 0) if m_f else None,
            "flat_fused_bytes_per_launch": (f_f or 0) + (f_w or 0) if f_f is not None else None,
        }
        # the SAME lease's kernel statistics and store-pacer state beside the counters (a 480 vs 500 us
        # difference between leases must be attributable): mean launch durations from the --kernel-trace --stats pass,
        # the pacer's rate / steps / probes from the JSON line each profiled bench run printed
        stats_csv = os.path.join(out, "rocprofv3_kernel_stats.csv")
        if os.path.exists(stats_csv):
            means = {}
            with open(stats_csv) as f:
                for row in csv.DictReader(f):
                    means[row["Name"]] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]), float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3)

            def pick(needle):
                for k, v in means.items():
                    if needle in k:
                        return {"mean_us": v[0], "calls": v[1], "min_us": v[2], "max_us": v[3]}
                return None
            summary["rocprof_kernel_stats"] = {"flat_estep_rows_pk_kernel": pick("flat_estep_rows_pk_kernel<3, 1, 4, true>"),
                                               "flat_fused_pk_kernel<13>": pick("flat_fused_pk_kernel<13>"),
                                               "flat_mstep_kernel": pick("flat_mstep_kernel"),
                                               "full_fused_kernel<2>": pick("full_fused_kernel<2>"),
                                               "source": "rocprofv3 --kernel-trace --stats of `python bench.py --skip published_charts,"
                                                         "replica_pairs` on the same lease (rocprofv3_kernel_stats.csv)"}
            e = summary["rocprof_kernel_stats"]["flat_estep_rows_pk_kernel"]
            summary["flat_estep_rocprof_mean_us"] = e["mean_us"] if e else None
        pacer = {}
        for tag, name in (("kernel_trace_run", "bench_n1_under_rocprofv3.json"), ("pmc_FETCH_SIZE_run", "pmc_FETCH_SIZE.stdout"),
                          ("pmc_WRITE_SIZE_run", "pmc_WRITE_SIZE.stdout"), ("unprofiled_run", "bench_n1.json")):
            path = os.path.join(out, name)
            try:
                line = [l for l in open(path).read().splitlines() if l.strip().startswith("{")][-1]
                r = json.loads(line)["roofline"]
                pacer[tag] = {"store_pacer": r.get("store_pacer", {}), "patterns_ms": r.get("patterns_ms"),
                              "avg_launch_ms": r.get("avg_launch_ms"), "frac": r.get("frac")}
                for k in ("rule",):
                    pacer[tag]["store_pacer"].pop(k, None)
            except Exception:
                continue
        summary["estep_runs_of_this_lease"] = pacer
        with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
