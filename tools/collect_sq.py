"""profiles/<round>/sq_counters_512.json from a tools/sq_counters.sh pass: per-launch means of the SQ counters of the rollout kernels."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "sq")
out = {}
for tier in ("B", "A"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, f"t{tier}_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "rollout_kernel" in k:
                acc[k.split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out["dense_tier_then_retry" if tier == "B" else "full_capacity_only"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for name, kernels in out.items():
    for k, c in kernels.items():
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            c["wait_any_share"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
            c["valu_active_share"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
            c["busy_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
out["note"] = "quadruped 512 x 100 on one GPU, per launch (mean over the launches of a 3-step bench run); cycle counters count quad-cycles per wave"
dst = os.path.join(ROOT, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r2", "sq_counters_512.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
