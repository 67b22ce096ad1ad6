timeout 300 python -m pytest tests/test_gpu_extremes.py -m gpu -q -x -k "non_finite and 512-1-0" 2>&1 | grep -E "assert|Error|error|^E" | head -20
python - <<'PY'
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, harness
pkg = harness.pkg
gpu = pkg.AvifGpu(0)
d = pkg.WriteDesc(width=16, height=1, depth=32, planes=3, bit_depth=10, transfer=pkg.TRANSFER_PQ, peak_nits=80, alpha_state=pkg.ALPHA_NONE, output=pkg.OUT_REFERENCE)
src = np.zeros((1, 48), np.float32)
src[0, :6] = [np.nan, np.inf, -np.inf, 1e38, 3.4e38, -1.0]
src[0, 6] = np.array([0x7fa00000], dtype=np.uint32).view(np.float32)[0]
print(harness.gpu_write(gpu, d, src)[0][0, :8])
PY
