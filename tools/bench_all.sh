#!/bin/bash
# Every bench workload once, on the GPU box:  tools/bench_all.sh <outdir under gpurun_out/>
# (default line with cpu_baseline and sb3_vecenv_path; the driver's argument set; the other workloads; the forced-RCCL-gather run)
set -e
OUT=gpurun_out/${1:-bench_all}
mkdir -p "$OUT"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_args.json" 2>/dev/null
for w in env01_nopads env01_reference env01_contact; do
  python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
done
for w in env01_reference_links env01_reference_proxies; do
  python bench.py --workload $w --no-cpu-baseline > "$OUT/bench_$w.json" 2>/dev/null
done
python bench.py --workload env02_reference --envs 16384 --no-cpu-baseline > "$OUT/bench_env02_16384.json" 2>/dev/null
python bench.py --workload env05_reference --envs 8192 --no-cpu-baseline > "$OUT/bench_env05_8192.json" 2>/dev/null
SO100_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-sb3-path > "$OUT/bench_force_dist.json" 2>/dev/null
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try: d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(os.path.basename(f), "unreadable", e); continue
    r = d.get("roofline", {}); s = d.get("sb3_vecenv_path") or {}; c = d.get("cpu_baseline") or {}
    print(f"{os.path.basename(f):32s} {d['value']/1e6:8.2f} M env-steps/s  {d['ms_per_step']*1e3:7.1f} us/step  kernel {r.get('kernel_ms')}  frac {r.get('frac')}"
          f"  sb3 {s.get('us_per_step')}  cpu {c.get('value')}")
PY
