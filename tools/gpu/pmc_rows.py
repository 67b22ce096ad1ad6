#!/usr/bin/env python3
"""Memory-hierarchy counters of chosen tools/bench_configs.py rows, one rocprofv3 --pmc pass per counter group (no trace domains
mixed in).  For every row: the mean counter value per launch over the 60 timed launches (bench_configs launches each row's kernel
210 times back to back: 150 ramp + 60 timed).

    python tools/gpu/pmc_rows.py OUT.json "row substring" ["row substring" ...]

Used for the question VERDICT r02 asked of the 16-bit ICC kernel: where do the 2.42x "HBM bytes" of uniformly random input come
from -- HBM, or the Infinity Cache behind the L2 (FETCH_SIZE counts L2 -> fabric requests, MALL hits included)?"""
import csv
import glob
import json
import os
import re
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LAUNCHES = 210
PAT = re.compile(r"avifgpu::(write_|read_px)")
CANDIDATES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCP_TCC_READ_REQ_sum"],
              ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_EA0_RDREQ_DRAM_sum"], ["TCC_EA0_RD_UNCACHED_32B_sum"],
              ["TCC_REQ_sum", "TCC_READ_sum"], ["TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum"], ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"],
              ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"]]


def main():
    global CANDIDATES
    out_json, rows = sys.argv[1], sys.argv[2:]
    if os.environ.get("PMC_GROUPS"):                            # "A,B;C,D": other counters than the memory-hierarchy set, one pass per group
        CANDIDATES = [g.split(",") for g in os.environ["PMC_GROUPS"].split(";") if g]
    env = dict(os.environ, TMPDIR="/tmp")
    listing = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, env=env, cwd="/tmp").stdout
    have = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", listing))
    table = {}
    notes = []
    for group in CANDIDATES:
        g = [c for c in group if c in have]
        if not g:
            notes.append("not on this device: " + " ".join(group))
            continue
        d = "/tmp/pmc_" + g[0]
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc", *g, "-d", d, "-o", "p", "--output-format", "csv", "--", "bash", "-c",
               "cd %s && python tools/bench_configs.py %s > %s/cfg.jsonl 2>/dev/null" % (ROOT, " ".join(shlex.quote(r) for r in rows), d)]
        os.makedirs(d, exist_ok=True)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=int(os.environ.get("PMC_PASS_TIMEOUT", "180")))
        if r.returncode != 0:
            notes.append("pass failed: " + " ".join(g) + ": " + r.stderr[-300:])
            continue
        cfgs = [json.loads(l) for l in open(d + "/cfg.jsonl") if l.startswith("{")]
        recs = []
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            recs += [x for x in csv.DictReader(open(f, newline="")) if PAT.search(x["Kernel_Name"])]
        for name in g:
            vals = sorted(((int(x["Dispatch_Id"]), float(x["Counter_Value"])) for x in recs if x["Counter_Name"] == name))
            vals = [v for _, v in vals]
            start = 0
            for i, c in enumerate(cfgs):
                n = c.get("launches", LAUNCHES)
                chunk = vals[start:start + n][-60:]
                start += n
                if chunk:
                    e = table.setdefault(c["config"], {"kernel": c["kernel"], "algorithmic_bytes": round(c["bytes_per_px"] * c["Mpx_s"] * c["ms_mean"] * 1e3)})
                    e[name] = sum(chunk) / len(chunk)
        subprocess.run(["rm", "-rf", d])
    for e in table.values():
        if "FETCH_SIZE" in e:
            e["l2_to_fabric_read_bytes (FETCH_SIZE KiB x 2 x 1024: gfx950 correction)"] = round(e["FETCH_SIZE"] * 2048)
        if "WRITE_SIZE" in e:
            e["l2_to_fabric_write_bytes (WRITE_SIZE KiB x 1024)"] = round(e["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
    json.dump({"rows": table, "notes": notes}, open(out_json, "w"), indent=1)
    print(json.dumps({"rows": table, "notes": notes}, indent=1))


if __name__ == "__main__":
    main()
