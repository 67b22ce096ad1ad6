#!/usr/bin/env python3
"""Median per variant of a tools/gpu/ab_fresh.sh log (several interleaved passes of tools/bench_configs.py rows under different library builds):
    python tools/ab_summary.py LOG BASE_VARIANT
prints, per row, every variant's median ms, its gain over BASE_VARIANT's median, and the passes' values in 0.1 us -- the form of the A/B files
of profiles/r05 that are marked LAST SERIES."""
import re,collections,sys
d=collections.defaultdict(lambda: collections.defaultdict(list))
v=None; base=sys.argv[2]
for l in open(sys.argv[1]):
    m=re.match(r'== (\w+) \(pass',l)
    if m: v=m.group(1); continue
    m=re.match(r'(.{88}) +([\d.]+) ms',l)
    if m: d[m.group(1).strip()][v].append(float(m.group(2)))
for k,x in d.items():
    b=sorted(x[base])[len(x[base])//2]
    print('%-80s'%k[:80], ' '.join('%s %.4f (%+.1f%%) [%s]'%(n, sorted(t)[len(t)//2], (b/sorted(t)[len(t)//2]-1)*100, ' '.join('%.0f'%(1e4*u) for u in t)) for n,t in x.items()))
