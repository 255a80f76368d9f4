#!/bin/bash
# static instruction counts of the RNEA / CRBA code (compile only, no GPU): tools/rnea_count.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -S --cuda-device-only -I so100_mujoco_rl_amd/csrc -o gpurun_out/isa/rnea.s tools/micro/rnea_isa.hip
python3 - <<'PY'
import re
txt = open("gpurun_out/isa/rnea.s").read()
for f in re.split(r"\n(?=_Z\w+:)", txt):
    name = f.split(":")[0]
    if not name.startswith("_Z"): continue
    ins = [l.strip() for l in f.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    v = [l for l in ins if l.startswith("v_")]
    print(f"{name[:20]:22s} total {len(ins):5d}  VALU {len(v):5d}  v_pk {sum(l.startswith('v_pk_') for l in v):4d}  v_mov {sum(l.startswith('v_mov') for l in v):4d}  s_mov {sum(l.startswith('s_mov') for l in ins):4d}")
PY
