#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_hashes.json: SHA-256 of the CPU oracle's output on every parity case
(tests/cases.py, seed 1234).  These are regression pins of the oracle itself -- the reference has no vectors of its
own for this path and cannot be built here; the values that tie the oracle to the reference are in
survey_kat.json.  Float-tier hashes depend on glibc 2.35's powf/expf/logf (this image)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import harness  # noqa: E402

pkg = harness.pkg


def compute():
    out = {"write": {}, "read": {}}
    for cid, kw in cases.write_cases():
        d = pkg.WriteDesc(**kw)
        out["write"][cid] = harness.digest(harness.oracle_write(d, harness.make_write_source(d)))
    for cid, kw in cases.read_cases():
        d = pkg.ReadDesc(**kw)
        out["read"][cid] = harness.digest(harness.oracle_read(d, harness.make_read_source(d)))
    return out


if __name__ == "__main__":
    json.dump(compute(), open(os.path.join(HERE, "oracle_hashes.json"), "w"), indent=0, sort_keys=True)
    print("written")
