for lanes in 2 3; do for tile in 8 16; do
  AVIFGPU_LANES=$lanes AVIFGPU_TILE_MB=$tile timeout 200 python - 2>/dev/null <<'PY'
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench_host_shim as b
pkg = b.pkg
for planes in ("pinned", "pageable"):
    b.run(8192, 8192, 1 << 30, pkg.OUT_YCBCR, planes=planes, reps=4)
    b.run(8192, 8192, 1 << 30, pkg.OUT_YCBCR, planes=planes, nofill=True, reps=4)
PY
done; done | python -c "
import sys, json
for i, l in enumerate(sys.stdin):
    r = json.loads(l); print('lanes', 2 + i // 8, 'tileMB', (8, 16)[(i // 4) % 2], r['planes'], 'nofill' if 'skipped' in r['config'] else 'fill  ', 'tiles', r['tiles'], 'ms', round(r['seconds'] * 1e3, 1))"
