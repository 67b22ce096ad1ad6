#!/bin/bash
# tools/bench_geometry.py under several library variants
for v in "$@"; do echo "== $v"; AVIFGPU_LIB=$PWD/avif-format_amd/variants/libavifgpu_$v.so python tools/bench_geometry.py ${DEPTH:-16} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-12s generic %.4f (%.3f)  streaming %.4f (%.3f)' % (d['geometry'], d['generic']['ms'], d['generic']['frac'], d['streaming']['ms'], d['streaming']['frac']))"; done
