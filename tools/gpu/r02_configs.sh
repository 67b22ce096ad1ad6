mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -5
timeout 300 ./tools/hostregister_repro 300 96 2>&1 | tail -3
timeout 300 ./tools/hostregister_repro 100 403 2>&1 | tail -3
bash tools/gpu/profile_configs.sh > gpurun_out/profile_configs.log 2>&1; tail -5 gpurun_out/profile_configs.log
python - <<'PY'
import json
rows = json.load(open("gpurun_out/prof_cfg/configs_traffic.json"))
rows = rows if isinstance(rows, list) else rows.get("configs", rows)
for r in rows:
    print("%-92s %-48s %8.4f %6.3f %s" % (r.get("config", "")[:92], r.get("kernel", "")[:48], r.get("ms_mean", 0), r.get("frac_of_8TBs", 0), r.get("traffic_over_algorithmic", r.get("hbm_over_algorithmic", ""))))
PY
