#!/usr/bin/env python3
"""Does a row's position in the process change its time?  The same two configurations, timed repeatedly in one process, fresh buffers
each time and then the SAME buffers again (tools/bench_configs.py machinery)."""
import json, os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "__none__"]
import bench_configs as bc
P = bc.pkg
c4 = dict(width=8192, height=8192, depth=32, planes=3, bit_depth=10, transfer=0, peak_nits=80, alpha_state=0, output=1, chroma=P.CHROMA_444, matrix_coefficients=9, color_primaries=9)
d12 = dict(c4, bit_depth=12)
bc.ONLY[:] = []
for name, kw in [("C4 #1", c4), ("C4 #2", c4), ("D12 #1", d12), ("C4 #3", c4), ("D12 #2", d12), ("D12 #3", d12), ("C4 #4", c4)]:
    bc._bw(name, **kw)
