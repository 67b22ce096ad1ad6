# Recipe of profiles/r06 headline files (run through gpurun: "bash tools/gpu/profile_r06.sh"; summaries are then copied into profiles/r06/).
#  1. kernel trace of the default bench command, reduced to the timed-region launches (tools/summarize_kernel_trace.py) -- the timed region
#     ROTATES over disjoint buffer sets since round 5, so the trace is a fresh-data trace
#  2. HBM traffic counters in their own passes (FETCH_SIZE, WRITE_SIZE: --pmc only, no trace domains)
#  3. the plain bench line (with its own live traffic passes, C5, PCIe, CPU baseline), smoke
set -x
out=gpurun_out/prof_r06; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-c5 --no-pcie --no-live-traffic --no-extra-configs"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/kt -o kt --output-format csv -- bash -c "cd $R && $P > $out/bench_under_kernel_trace.json" > $R/$out/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "write_rgb32_ycbcr444_hot" -d $R/$out/fetch -o f --output-format csv -- bash -c "cd $R && $P" > $R/$out/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "write_rgb32_ycbcr444_hot" -d $R/$out/write -o w --output-format csv -- bash -c "cd $R && $P" > $R/$out/write.log 2>&1
cd $R
python tools/summarize_kernel_trace.py $out/kt $out/bench_under_kernel_trace.json $out/kernel_stats_c4_444_timed_region.csv
find $out/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_c4_444_all_launches.csv
python - <<'PY'
import csv, glob, json
def mean(pat, col):
    v = []
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == col and "write_rgb32_ycbcr444_hot" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
f, nf = mean("gpurun_out/prof_r06/fetch/**/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean("gpurun_out/prof_r06/write/**/*counter_collection.csv", "WRITE_SIZE")
out = {"workload": "8192x8192-444-10", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported",
       "hbm_read_bytes_per_launch": f * 2048 if f else None, "hbm_write_bytes_per_launch": w * 1024 if w else None,
       "hbm_bytes_per_launch": (f * 2048 + w * 1024) if f and w else None, "algorithmic_bytes_per_launch": 8192 * 8192 * 18,
       "source": "profiles/r06: separate rocprofv3 --pmc passes over the default bench command (timed region rotating over disjoint buffer sets), mean over all launches of the kernel"}
json.dump(out, open("gpurun_out/prof_r06/traffic.json", "w"), indent=1)
print(out)
PY
rm -rf $out/kt $out/fetch $out/write
python bench.py --steps 200 --warmup 20 > $out/bench_r06.json 2> $out/bench_r06.err; cat $out/bench_r06.json; tail -3 $out/bench_r06.err
python -c "import __graft_entry__ as g; g.smoke()"
