"""Condenses rocprofv3 CSV output under gpurun_out/ into the small summaries committed
under profiles/.  usage: summarize_prof.py <tag> <kernel-trace dir> [<fetch pmc dir> <write pmc dir>]
A launch of eskf_run_kernel is a whole horizon of pair steps (bench.py --loop-in-kernel 1: STEPS_PER_RUN_LAUNCH = its
--horizon, every launch of the profiled command a full horizon): its durations and counters are divided by that to
give per-step figures, next to the per-launch ones."""
import collections, csv, glob, json, os, sys

tag, kdir = sys.argv[1], sys.argv[2]
STEPS_PER_RUN_LAUNCH = int(os.environ.get("STEPS_PER_RUN_LAUNCH", "16"))
KEYS = ("eskf_run_kernel", "eskf_step_kernel", "eskf_pre_kernel", "eskf_pre4_kernel", "eskf_nominal_kernel", "eskf_jac_kernel", "eskf_cov_kernel",
        "eskf_cov_rows_kernel", "eskf_post_kernel")


def steps_of(key):
    return STEPS_PER_RUN_LAUNCH if key == "eskf_run_kernel" else 1
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
stats = glob.glob(os.path.join(kdir, "**", "*_kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    for r in rows[:12]:
        f.write(",".join(['"%s"' % r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                          r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
trace = glob.glob(os.path.join(kdir, "**", "*_kernel_trace.csv"), recursive=True)
res = {"tag": tag}
if trace:
    allrows = list(csv.DictReader(open(trace[0])))
    res["step_kernels"] = {}
    for key in KEYS:
        t = [r for r in allrows if key in r["Kernel_Name"]]
        if not t:
            continue
        r0 = t[0]
        info = {k: r0.get(k) for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                         "Accum_VGPR_Count", "SGPR_Count")}
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in t]
        info["launches"] = len(d)
        info["avg_ns"] = sum(d) / len(d)
        info["steps_per_launch"] = steps_of(key)
        info["avg_ns_per_step"] = info["avg_ns"] / steps_of(key)
        res["step_kernels"][key] = info
    # per step: kernels that do not run every step (the flush) are weighted by their launch count
    ref = max(v["launches"] * v["steps_per_launch"] for v in res["step_kernels"].values())
    res["step_avg_ns_sum_of_kernels"] = sum(v["avg_ns"] * v["launches"] / ref for v in res["step_kernels"].values())
if len(sys.argv) >= 5:
    pm = {}
    calib = {}
    for name, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
        fn = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pm[name] = {}
        for k, v in acc.items():
            for key in KEYS:
                if key in k:
                    pm[name][key] = {"launches": len(v), "mean_KiB": sum(v) / len(v), "min_KiB": min(v), "max_KiB": max(v)}
            if "copyBuffer" in k:
                calib[name] = max(v)
    res["pmc_step_kernel"] = pm
    res["pmc_copyBuffer_max_KiB"] = calib
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        # MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a coalesced
        # streaming read (calibrated in the same run on the runtime's copyBuffer of known size);
        # WRITE_SIZE is exact.  Units are KiB.
        def per_step(d):
            ref = max(v["launches"] * steps_of(k) for k, v in d.items())
            return sum(v["mean_KiB"] * v["launches"] / ref for v in d.values())
        rd = per_step(pm["FETCH_SIZE"]) * 1024 * 2
        wr = per_step(pm["WRITE_SIZE"]) * 1024
        res["hbm_bytes_per_launch"] = rd + wr   # per step: all kernels of one predict+update pair
        res["hbm_read_bytes_per_launch_corrected_x2"] = rd
        res["hbm_write_bytes_per_launch"] = wr
json.dump(res, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
