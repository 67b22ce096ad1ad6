#!/usr/bin/env python3
"""Regenerate the measured tables of DESIGN.md (between the BEGIN/END markers) from profiles/<round>/bench_configs.jsonl and
profiles/<round>/configs_traffic.json, so the document cannot drift from the committed measurements.
    python tools/make_design_tables.py [round directory, default r03]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", RND, "bench_configs.jsonl")) if l.startswith("{")]
tpath = os.path.join(ROOT, "profiles", RND, "configs_traffic.json")
traffic = {e["config"]: e for e in json.load(open(tpath))} if os.path.exists(tpath) else {}


def table(pred):
    out = ["| Config | kernel variant | B/px | ms | Gpx/s | GB/s | of 8 TB/s | HBM traffic ÷ algorithmic (PMC) | VALU issue (PMC) | bound | math-free twin, of 8 TB/s |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if not pred(r):
            continue
        k = r["kernel"]
        m = re.match(r"(\w+)<(.*)>", k)
        short = m.group(1) + " " + re.sub(r"(depth|planes|out|dst16|transfer|aligned|pxl|nt|prefetch|xcdmap|cs|alpha)=", lambda x: x.group(1)[0] + "", m.group(2)) if m else k
        t = traffic.get(r["config"], {}).get("traffic_over_algorithmic")
        vi = traffic.get(r["config"], {}).get("valu_issue_frac")
        # the roof a row is priced against: whichever of the two resources it keeps busier (HBM at 8 TB/s, the vector ALUs of 1024 SIMDs at 2.4 GHz)
        bound = "—" if vi is None else ("valu" if vi > r["frac_of_8TBs"] else "hbm")
        twin = r.get("twin_frac_of_8TBs")
        tag = ("" if "icc=" not in k else " icc=" + k.split("icc=")[1].rstrip(">").split()[0]) + (" tables=none" if "tables=none" in k else "")
        out.append("| %s | `%s` | %g | %.4f | %.0f | %.0f | %.2f | %s | %s | %s | %s |" % (r["config"], k.split("<")[0] + tag,
                                                                         r["bytes_per_px"], r["ms_mean"], r["Mpx_s"] / 1e3, r["GB_s"], r["frac_of_8TBs"], ("%.4f" % t) if t else "—",
                                                                         ("%.2f" % vi) if vi is not None else "—", bound, ("%.2f" % twin) if twin else "—"))
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
print("DESIGN.md tables regenerated from", len(rows), "configurations")
