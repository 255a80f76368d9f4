"""Summarise the rocprofv3 passes written by tools/profile_round.sh into one JSON (per-dispatch averages of the dominant
kernels): python tools/pmc_summary.py gpurun_out/<tag> > profiles/<tag>_pmc_summary.json"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_sha16          # fingerprint of csrc/: bench.py quotes this pass only while it describes the current kernels

root = sys.argv[1]
T, N = 64, 4096


def rows(sub):
    for f in glob.glob(os.path.join(root, sub, "**", "*_counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def short(name):
    n = name.replace("void ", "").replace("so100::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "fetch", "write"):
    per = collections.defaultdict(lambda: collections.defaultdict(float))       # (kernel, dispatch) -> counter -> value
    for r in rows(sub):
        k = short(r["Kernel_Name"])
        if "so100_rollout" not in k:                    # the bench's large-batch leg runs the other kernels at a different N
            continue
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        for name, v in c.items():
            acc[k][name].append(v)
out = {"note": "rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over `python bench.py --steps 256 --warmup 64 "
               "--no-cpu-baseline` on one MI355X; per-dispatch averages. FETCH_SIZE / WRITE_SIZE in KB; no gfx950 x2 correction applied "
               "(dword-per-lane accesses, an uncalibrated width); the x2 value is given alongside.",
       "workload": (sys.argv[2] if len(sys.argv) > 2 else "env01_free") + f", {N} envs, T = {T} steps per launch",
       "bench_workload": (sys.argv[2] if len(sys.argv) > 2 else "env01_free"), "envs": N, "policy": "persistent", "source_sha16": source_sha16(),
       "kernels": {}}
for k, c in acc.items():
    d = {name: sum(v) / len(v) for name, v in c.items()}
    d["dispatches"] = {name: len(v) for name, v in c.items()}
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        steps = T * N if "rollout" in k else N
        d["hbm_traffic_bytes_per_launch"] = (d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        d["hbm_traffic_bytes_per_launch_fetch_x2"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        d["hbm_traffic_bytes_per_env_step"] = d["hbm_traffic_bytes_per_launch"] / steps
    if "SQ_INSTS_VALU" in d:
        steps = T * N if "rollout" in k else N
        d["valu_wave_insts_per_env_step_x64"] = d["SQ_INSTS_VALU"] * 64 / steps
        if d.get("SQ_WAVE_CYCLES"):
            d["wait_fraction"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
    out["kernels"][k] = d
print(json.dumps(out, indent=1))
