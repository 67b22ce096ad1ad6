#!/usr/bin/env python3
"""Turn rocprofv3 outputs of `tools/bench_configs.py` runs into one table per configuration.

bench_configs.py launches every configuration's kernel a known number of times back to back (each row prints its 'launches': a
size-dependent warm-up + 60 timed; rows of older runs: LAUNCHES), so the dispatches that match
write_px|read_px|write_rgb32|write_rgb16|... (in dispatch order) split into consecutive chunks of LAUNCHES, one per printed configuration.
Inputs: the kernel-trace CSV and the two counter-collection CSVs (FETCH_SIZE, WRITE_SIZE: separate passes, as the MI355X guide
prescribes) plus the JSON lines bench_configs.py printed in the kernel-trace run.
gfx950 correction: FETCH_SIZE counts 64-byte units in KiB/2 -> doubled; WRITE_SIZE is KiB as is (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import re
import sys

LAUNCHES = 210
PAT = re.compile(r"avifgpu::(write_|read_px)")     # every conversion kernel of the library (build_read_tables is not one)


def rows(path_glob):
    files = glob.glob(path_glob, recursive=True)
    if not files:
        return []
    out = []
    for f in files:
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def chunks(seq, counts=None):
    """The dispatch stream cut into one chunk per configuration: by the 'launches' each row printed, else LAUNCHES each."""
    if not counts:
        return [seq[i:i + LAUNCHES] for i in range(0, len(seq) - LAUNCHES + 1, LAUNCHES)]
    out, i = [], 0
    for n in counts:
        if i + n > len(seq):
            break
        out.append(seq[i:i + n])
        i += n
    return out


def main(prof_dir, configs_jsonl, out_json):
    cfgs = [json.loads(l) for l in open(configs_jsonl) if l.startswith("{")]
    kt = [r for r in rows(prof_dir + "/kt/**/*kernel_trace.csv") if PAT.search(r["Kernel_Name"])]
    kt.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in kt]
    per_counter = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"):
        rs = [r for r in rows(prof_dir + "/%s/**/*counter_collection.csv" % name.lower()) if PAT.search(r["Kernel_Name"]) and r["Counter_Name"] == name]
        rs.sort(key=lambda r: int(r["Dispatch_Id"]))
        per_counter[name] = [float(r["Counter_Value"]) for r in rs]
    table = []
    counts = [c["launches"] for c in cfgs] if all("launches" in c for c in cfgs) else None
    dch = chunks(dur, counts)
    fch, wch = chunks(per_counter["FETCH_SIZE"], counts), chunks(per_counter["WRITE_SIZE"], counts)
    for i, c in enumerate(cfgs):
        e = {"config": c["config"], "kernel": c["kernel"], "algorithmic_bytes": c["bytes_per_px"] * c["Mpx_s"] * c["ms_mean"] * 1e3}
        if i < len(dch):
            timed = dch[i][-60:]                      # the 60 timed launches follow the 150-launch clock ramp
            e["kernel_trace_avg_us"] = round(sum(d for _, d in timed) / len(timed) / 1e3, 2)
            e["hip_event_avg_us"] = round(c["ms_mean"] * 1e3, 2)
        if i < len(fch) and i < len(wch):
            fetch = 2.0 * 1024.0 * sum(fch[i][-60:]) / 60
            write = 1024.0 * sum(wch[i][-60:]) / 60
            e["hbm_read_bytes"], e["hbm_write_bytes"] = round(fetch), round(write)
            e["traffic_over_algorithmic"] = round((fetch + write) / e["algorithmic_bytes"], 4)
        # Vector-ALU issue (round 4).  SQ_ACTIVE_INST_VALU counts the quad-cycles the SIMDs spent issuing vector instructions (it
        # equals SQ_INSTS_VALU + one more per transcendental on these kernels); a kernel that kept all 1024 SIMDs issuing every cycle
        # for its whole duration at 2.4 GHz would show valu_issue_frac = 1.  The streaming kernels saturate at ~0.74-0.82 of that.
        for name, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_ACTIVE_INST_VALU", "valu_active_quads")):
            ch = chunks(per_counter.get(name, []), counts)
            if i < len(ch) and ch[i]:
                e[key] = round(sum(ch[i][-60:]) / len(ch[i][-60:]))
        if "valu_active_quads" in e and "kernel_trace_avg_us" in e:
            e["valu_issue_frac"] = round(e["valu_active_quads"] * 4.0 / (1024 * 2.4e9 * e["kernel_trace_avg_us"] * 1e-6), 3)
            px = c["Mpx_s"] * c["ms_mean"] * 1e3
            e["valu_insts_per_px"] = round(e.get("valu_insts", 0) * 64.0 / px, 1) if px else None
        table.append(e)
    json.dump(table, open(out_json, "w"), indent=1)
    for e in table:
        print("%-78s kt %8s us  ev %8s us  traffic x%s  valu %s" % (e["config"][:78], e.get("kernel_trace_avg_us"), e.get("hip_event_avg_us"),
                                                                       e.get("traffic_over_algorithmic"), e.get("valu_issue_frac")))


if __name__ == "__main__":
    main(*sys.argv[1:4])
