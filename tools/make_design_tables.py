#!/usr/bin/env python3
"""Regenerate the measured tables -- ALL rows into profiles/<round>/TABLES.md, the rows a reader looks for first (BASELINE configurations, the
plug-in's default saves and opens, the hand-offs, one row per ICC stage) into DESIGN.md between its BEGIN/END markers -- from profiles/<round>/bench_configs.jsonl (fresh-data
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


def table(pred, short=False):
    if short:
        return compact(pred)
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
            (r["config"] if not short or len(r["config"]) <= 96 else r["config"][:93] + "..."), k.split("<")[0] + tag, r["bytes_per_px"], r["ms_mean"], r["Mpx_s"] / 1e3, r["GB_s"], r["frac_of_8TBs"],
            ("%.2f" % r["frac_same"]) if "frac_same" in r else "—", ("%.2f" % twin) if twin else "—", ("%.2f" % mof) if mof else "—",
            ("%.4f" % t) if t else "—", bound))
    return "\n".join(out)


def compact(pred):
    """DESIGN.md's form: the columns a reader prices a row with (all columns: TABLES.md)."""
    out = ["| Config | kernel | B/px | ms | of 8 TB/s (fresh) | one-set loop | twin | math only ÷ kernel | traffic ÷ algorithmic | bound |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        if not pred(r):
            continue
        k = r["kernel"]
        t = traffic.get(r["config"], {}).get("traffic_over_algorithmic")
        mo = mathonly.get(r["config"], {}).get("ms_mean")
        mof = mo / r["ms_mean"] if mo else None
        twin = r.get("twin_frac_of_8TBs")
        tag = ("" if "icc=" not in k else " icc=" + k.split("icc=")[1].rstrip(">").split()[0]) + (" out=ref" if "out=ref" in k else "")
        name = r["config"]
        for a, b in ((" (the plug-in's default HDR save, fused hand-off)", " [default HDR save]"), (" (the plug-in's default save: 8-bit, 4:2:2, AvifFormat.cpp:89)", " [default SDR save]"),
                     (" (what the default HDR save decodes to)", ""), (" (what the default save decodes to)", ""), (" (reference hand-off, what integration/ uses by default)", " [reference hand-off]"),
                     (" (reference hand-off = copy)", " (= copy)"), (" (fused hand-off)", ""), ("linear Display-P3 doc -> Rec.2020", "linear P3 -> Rec.2020"), (" 8192^2", ""),
                     (" (default HDR save of a linear-profile document)", ""), (" (smooth + noise)", ""), (", photograph-like input", ", photo-like"),
                     (" (reference hand-off behind the document's profile: integration/'s default for a 32-bit document)", " [reference hand-off]"),
                     (" (transparent SDR 16-bit document at the plug-in's defaults: straight alpha, AvifFormat.cpp:99)", ""), (" (transparent SDR 16-bit document, premultiplied-alpha option on)", "")):
            name = name.replace(a, b)
        out.append("| %s | `%s` | %g | %.4f | **%.2f** | %s | %s | %s | %s | %s |" % (
            name if len(name) <= 84 else name[:81] + "...", k.split("<")[0].replace("write_", "w_").replace("read_", "r_") + tag, r["bytes_per_px"], r["ms_mean"], r["frac_of_8TBs"],
            ("%.2f" % r["frac_same"]) if "frac_same" in r else "—", ("%.2f" % twin) if twin else "—", ("%.2f" % mof) if mof else "—",
            ("%.3f" % t) if t else "—", "—" if mof is None else ("valu" if mof >= 0.8 else "hbm")))
    return "\n".join(out)


def replace(doc, tag, text):
    a, b = "<!-- BEGIN:%s -->" % tag, "<!-- END:%s -->" % tag
    i, j = doc.index(a) + len(a), doc.index(b)
    return doc[:i] + "\n" + text + "\n" + doc[j:]


KEY = ("C2 ", "C3 ", "C4 8192^2 RGB f32 -> 10-bit PQ 4:4:4", "C4 8192^2 RGB f32 -> 10-bit PQ 4:2:0", "C5 ", "D12 8192^2 RGB f32 -> 12-bit PQ 4:2:2", "D12 8192^2 RGB f32 -> 12-bit PQ interleaved",
       "D12 8192^2 RGBA f32 -> 12-bit PQ 4:2:2", "D12 8192^2 RGBA16 premult", "W8 8192^2 RGB8 -> 8-bit 4:2:2", "W8 8192^2 RGB8 -> 10-bit 4:2:0", "REF RGB8",
       "D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ 4:2:2", "D12 + ICC (linear Display-P3 doc -> Rec.2020) 8192^2 RGB f32 -> 12-bit PQ interleaved",
       "C4 + ICC (sampled-curve doc profile -> Rec.2020),", "16-bit doc + ICC, photograph-like input (smooth", "8-bit doc + ICC, photograph-like", "8-bit doc + ICC (LUT",
       "R8 8192^2 8-bit 4:2:0 BT.709", "R8 8192^2 8-bit 4:2:2", "R16 8192^2 12-bit 4:4:4", "R32 8192^2 10-bit 4:4:4", "D12 8192^2 12-bit 4:2:2", "BIG 16384^2 8-bit")


def key(r):
    return any(r["config"].startswith(k) for k in KEY)


full = ("# Measured tables, round %s -- generated by tools/make_design_tables.py from bench_configs.jsonl, bench_configs_math_only_build.jsonl and\n"
        "# configs_traffic.json in this directory; column meanings: DESIGN.md section 6.4\n\n## Write direction\n\n%s\n\n## Read direction\n\n%s\n") % (
    RND, table(lambda r: r["kernel"].startswith("write_")), table(lambda r: r["kernel"].startswith("read_")))
open(os.path.join(D, "TABLES.md"), "w").write(full)
p = os.path.join(ROOT, "DESIGN.md")
doc = open(p).read()
doc = replace(doc, "write_table", table(lambda r: r["kernel"].startswith("write_") and key(r), short=True))
doc = replace(doc, "read_table", table(lambda r: r["kernel"].startswith("read_") and key(r), short=True))
open(p, "w").write(doc)
print("DESIGN.md (key rows) and profiles/%s/TABLES.md (all rows) regenerated from" % RND, len(rows), "configurations")
