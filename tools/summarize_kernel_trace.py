#!/usr/bin/env python3
"""Kernel-trace statistics of bench.py's TIMED REGION only.

rocprofv3 --kernel-trace --stats averages every launch of a kernel in the process: clock-ramp launches, warm-ups and the
isolated diagnostic launches included, which is not a steady-state figure (round-1 verdict, "profile hygiene").  bench.py
prints `profile_window` = {kernel label, launches before the timed region, timed launches}; this script takes the dispatches
of the headline kernel from the trace CSV in start order and keeps exactly that window.

    python tools/summarize_kernel_trace.py <rocprof-out-dir> <bench.json> <out.csv>

Writes one CSV row in rocprofv3's own --stats column layout (Name, Calls, TotalDurationNs, AverageNs, MinNs, MaxNs, StdDev)
for the window, one for all launches of that kernel (what --stats would have printed), and the back-to-back span
(first start -> last end) / calls, which is what bench.py's HIP-event pair measures."""
import csv
import glob
import json
import statistics
import sys


def main(prof_dir, bench_json, out_csv):
    bench = None
    for line in open(bench_json):
        if line.startswith("{"):
            bench = json.loads(line)
    win = bench["profile_window"]
    short = win["kernel"].split("<")[0]
    rows = []
    for f in glob.glob(prof_dir + "/**/*kernel_trace.csv", recursive=True):
        with open(f, newline="") as fh:
            rows += [r for r in csv.DictReader(fh) if short in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n0, k = win["launches_before_timed_region"], win["timed_launches"]
    if len(rows) < n0 + k:
        raise SystemExit(f"trace holds {len(rows)} launches of {short}, window needs {n0 + k}")
    window = rows[n0:n0 + k]

    def stats(sel):
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
        return len(d), sum(d), sum(d) / len(d), min(d), max(d), statistics.pstdev(d)
    w, a = stats(window), stats(rows)
    span = int(window[-1]["End_Timestamp"]) - int(window[0]["Start_Timestamp"])
    name = window[0]["Kernel_Name"]
    with open(out_csv, "w", newline="") as fh:
        wr = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        wr.writerow(["Name", "Selection", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        wr.writerow([name, f"timed region of bench.py (launches {n0}..{n0 + k - 1} of this kernel)", *w])
        wr.writerow([name, "all launches in the process (what rocprofv3 --stats averages: ramp, warm-up, diagnostics included)", *a])
        wr.writerow([name, "timed region, back-to-back span (first start -> last end) / calls", k, span, span / k, "", "", ""])
    print(f"{short}: timed-region average {w[2] / 1e3:.2f} us over {w[0]} launches (all {a[0]} launches: {a[2] / 1e3:.2f} us); "
          f"span/calls {span / k / 1e3:.2f} us; bench kernel_ms_mean {bench['roofline']['kernel_ms_mean'] * 1e3:.2f} us, "
          f"ms_per_step {bench['ms_per_step'] * 1e3:.2f} us")


if __name__ == "__main__":
    main(*sys.argv[1:4])
