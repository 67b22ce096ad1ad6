# Recipe of profiles/r02 (run through gpurun: "bash tools/gpu/profile_r02.sh"; summaries are then copied into profiles/r02/).
#  1. kernel trace of the default bench command, reduced to the timed-region launches (tools/summarize_kernel_trace.py)
#  2. HBM traffic counters in their own passes (FETCH_SIZE, WRITE_SIZE: --pmc only, no trace domains)
#  3. the plain bench line, the per-configuration table, PCIe-path numbers
set -x
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-c5 --no-pcie --no-rotate"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt --output-format csv -- bash -c "cd $R && $P > gpurun_out/prof/bench_under_kernel_trace.json" > $R/gpurun_out/prof/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "write_rgb32" -d $R/gpurun_out/prof/fetch -o f --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "write_rgb32" -d $R/gpurun_out/prof/write -o w --output-format csv -- bash -c "cd $R && $P" > $R/gpurun_out/prof/write.log 2>&1
cd $R
python tools/summarize_kernel_trace.py gpurun_out/prof/kt gpurun_out/prof/bench_under_kernel_trace.json gpurun_out/prof/kernel_stats_c4_444_timed_region.csv
find gpurun_out/prof/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/kernel_stats_c4_444_all_launches.csv
python - <<'PY'
import csv, glob, json
def mean(pat, col):
    v = []
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == col and "write_rgb32" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v), len(v)) if v else (None, 0)
f, nf = mean("gpurun_out/prof/fetch/**/*counter_collection.csv", "FETCH_SIZE")
w, nw = mean("gpurun_out/prof/write/**/*counter_collection.csv", "WRITE_SIZE")
out = {"workload": "8192x8192-444-10", "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "correction": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported",
       "hbm_read_bytes_per_launch": f * 2048 if f else None, "hbm_write_bytes_per_launch": w * 1024 if w else None,
       "hbm_bytes_per_launch": (f * 2048 + w * 1024) if f and w else None, "algorithmic_bytes_per_launch": 8192 * 8192 * 18,
       "source": "profiles/r02: separate rocprofv3 --pmc passes over the default bench command, mean over all launches of the kernel"}
json.dump(out, open("gpurun_out/prof/traffic.json", "w"), indent=1)
print(out)
PY
rm -rf gpurun_out/prof/kt gpurun_out/prof/fetch gpurun_out/prof/write
python bench.py --steps 200 --warmup 20 > gpurun_out/prof/bench_r02.json 2> gpurun_out/prof/bench_r02.err; cat gpurun_out/prof/bench_r02.json; tail -3 gpurun_out/prof/bench_r02.err
python -c "import __graft_entry__ as g; g.smoke()"
