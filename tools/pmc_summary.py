"""Condenses the rocprofv3 outputs of tools/profile_round.sh:
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
        with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
