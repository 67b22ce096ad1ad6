import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import cases, harness
pkg=harness.pkg
gpu=pkg.AvifGpu(0)
rows=[]
for cid,kw in cases.write_cases():
    if not cases.is_float_tier_write(kw): continue
    d=pkg.WriteDesc(**kw); src=harness.make_write_source(d)
    st=harness.compare_write(d, harness.oracle_write(d,src), harness.gpu_write(gpu,d,src,mem="device"))
    rows.append((st["exact_frac"], st["n"], kw["bit_depth"], kw.get("transfer"), cid))
for r in sorted(rows)[:25]: print("%.5f n=%d bits=%d tr=%s %s"%r)
print("count",len(rows))
