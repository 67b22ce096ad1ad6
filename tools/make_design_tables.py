#!/usr/bin/env python3
"""Regenerate the measured tables of DESIGN.md (between the BEGIN/END markers) from profiles/<round>/bench_configs.jsonl (fresh-data
launches + the one-set loop + the read rows' math-free twins), bench_configs_math_only_build.jsonl (the AG_MATH_ONLY library: every row's
compute side alone) and configs_traffic.json (kernel-trace durations + FETCH_SIZE x 2 + WRITE_SIZE), so that the document cannot drift from
the committed measurements.      python tools/make_design_tables.py [round directory, default r05]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
D = os.path.join(ROOT, "profiles", RND)
rows = [json.loads(l) for l in open(os.path.join(D, "bench_configs.jsonl")) if l.startswith("{")]
mpath, tpath = os.path.join(D, "bench_configs_math_only_build.jsonl"), os.path.join(D, "configs_traffic.json")
mathonly = {json.loads(l)["config"]: json.loads(l) for l in open(mpath) if l.startswith("{")} if os.path.exists(mpath) else {}
traffic = {e["config"]: e for e in json.load(open(tpath))} if os.path.exists(tpath) else {}


def table(pred):
    out = ["| Config | kernel variant | B/px | ms | Gpx/s | GB/s | of 8 TB/s (fresh data) | one-set loop, of 8 TB/s | math-free twin, of 8 TB/s | math only ÷ kernel | HBM traffic ÷ algorithmic (PMC) | bound |",
           "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if not pred(r):
            continue
        k = r["kernel"]
        t = traffic.get(r["config"], {}).get("traffic_over_algorithmic")
        mo = mathonly.get(r["config"], {}).get("ms_mean")
        mof = mo / r["ms_mean"] if mo else None
        # which side a row is read against: the compute side alone (the AG_MATH_ONLY build) takes >= 0.8 of the kernel's time -> `valu`
        # (the vector ALUs, LDS and issue slots set the time, the HBM fraction says little); else `hbm`
        bound = "—" if mof is None else ("valu" if mof >= 0.8 else "hbm")
        twin = r.get("twin_frac_of_8TBs")
        tag = ("" if "icc=" not in k else " icc=" + k.split("icc=")[1].rstrip(">").split()[0]) + (" tables=none" if "tables=none" in k else "") + (" out=ref" if "out=ref" in k else "")
        out.append("| %s | `%s` | %g | %.4f | %.0f | %.0f | **%.2f** | %s | %s | %s | %s | %s |" % (
            r["config"], k.split("<")[0] + tag, r["bytes_per_px"], r["ms_mean"], r["Mpx_s"] / 1e3, r["GB_s"], r["frac_of_8TBs"],
            ("%.2f" % r["frac_same"]) if "frac_same" in r else "—", ("%.2f" % twin) if twin else "—", ("%.2f" % mof) if mof else "—",
            ("%.4f" % t) if t else "—", bound))
    return "\n".join(out)


def replace(doc, tag, text):
    a, b = "<!-- BEGIN:%s -->" % tag, "<!-- END:%s -->" % tag
    i, j = doc.index(a) + len(a), doc.index(b)
    return doc[:i] + "\n" + text + "\n" + doc[j:]


p = os.path.join(ROOT, "DESIGN.md")
doc = open(p).read()
doc = replace(doc, "write_table", table(lambda r: r["kernel"].startswith("write_")))
doc = replace(doc, "read_table", table(lambda r: r["kernel"].startswith("read_")))
open(p, "w").write(doc)
print("DESIGN.md tables regenerated from", len(rows), "configurations of", RND)
