#!/usr/bin/env python3
"""Regenerate the measured tables of DESIGN.md (between the BEGIN/END markers) from profiles/r02/bench_configs.jsonl and
profiles/r02/configs_traffic.json, so the document cannot drift from the committed measurements."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles/r02/bench_configs.jsonl")) if l.startswith("{")]
traffic = {e["config"]: e for e in json.load(open(os.path.join(ROOT, "profiles/r02/configs_traffic.json")))}


def table(pred):
    out = ["| Config | kernel variant | B/px | ms | GB/s | of 8 TB/s | HBM traffic ÷ algorithmic (PMC) |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        if not pred(r):
            continue
        k = r["kernel"]
        m = re.match(r"(\w+)<(.*)>", k)
        short = m.group(1) + " " + re.sub(r"(depth|planes|out|dst16|transfer|aligned|pxl|nt|prefetch|xcdmap|cs|alpha)=", lambda x: x.group(1)[0] + "", m.group(2)) if m else k
        t = traffic.get(r["config"], {}).get("traffic_over_algorithmic")
        out.append("| %s | `%s` | %g | %.4f | %.0f | %.2f | %s |" % (r["config"], k.split("<")[0] + ("" if "icc=" not in k else " icc=" + k.split("icc=")[1].rstrip(">")),
                                                                    r["bytes_per_px"], r["ms_mean"], r["GB_s"], r["frac_of_8TBs"], ("%.4f" % t) if t else "—"))
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
