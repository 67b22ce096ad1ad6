"""The oracle's output on every parity case is frozen (tests/golden/oracle_hashes.json, made by make_golden.py):
an accidental edit of oracle/avif_oracle.c or of the input generator shows up here, on CPU, before any GPU run."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden  # noqa: E402


def test_oracle_outputs_match_committed_hashes():
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_hashes.json")))
    got = make_golden.compute()
    assert set(got["write"]) == set(want["write"]) and set(got["read"]) == set(want["read"])
    bad = [k for k in want["write"] if want["write"][k] != got["write"][k]] + \
          [k for k in want["read"] if want["read"][k] != got["read"][k]]
    assert not bad, bad[:10]
