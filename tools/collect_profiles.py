"""Turn a tools/measure.sh pass (gpurun_out/meas) into the committed summaries under profiles/<round>/:
kernel-stats CSVs, bench JSON lines and traffic.json (HBM bytes per rollout_kernel launch from the FETCH_SIZE / WRITE_SIZE
passes, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, section HBM)."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "meas")
dst = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r2")
tag = sys.argv[2] if len(sys.argv) > 2 else "a"
os.makedirs(dst, exist_ok=True)


def counter_per_launch(d, name):
    vals = []
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "rollout_kernel" in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append((row["Kernel_Name"], float(row["Counter_Value"])))
        shutil.copy(f, os.path.join(dst, f"{tag}_{d}_counter_collection.csv"))
    return vals


traffic = []
for wl, fd, wd in (("quadruped 256x100", "pmc_fetch", "pmc_write"), ("hand 256x64", "pmc_fetch_hand", "pmc_write_hand"),
                   ("humanoid 1024x128", "pmc_fetch_humanoid", "pmc_write_humanoid")):
    fe, wr = counter_per_launch(fd, "FETCH_SIZE"), counter_per_launch(wd, "WRITE_SIZE")
    if fe and wr:
        # one plan step may launch the kernel twice (capacity tiers); counters are summed per dispatch, reported per launch of the main kernel
        def per_plan(v):      # all rollout dispatches of a plan step, per launch of the flavour that carries the load
            tot = {}
            for k, x in v:
                tot[k] = tot.get(k, 0.0) + x
            main = max(tot, key=tot.get)
            return sum(tot.values()) / sum(1 for k, _ in v if k == main)
        fm, wm = per_plan(fe), per_plan(wr)
        traffic.append(dict(workload=wl, FETCH_SIZE_mean_KB=fm, WRITE_SIZE_mean_KB=wm, launches=len(fe),
                            traffic_bytes_per_launch=1024.0 * (2 * fm + wm),
                            note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py`, rollout_kernel dispatches; "
                                 "FETCH_SIZE doubled per the gfx950 correction"))
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
for d, name in (("stats", "kernel_stats"), ("stats512", "kernel_stats_512"), ("stats_hand", "kernel_stats_hand"),
                ("stats_humanoid", "kernel_stats_humanoid")):
    for f in glob.glob(os.path.join(src, d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}.csv"))
for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "fingers_time.log")):
    shutil.copy(f, os.path.join(dst, f"{tag}_" + os.path.basename(f)))
print(json.dumps(traffic, indent=1))
print(sorted(os.listdir(dst)))
