#!/bin/bash
# static instruction counts of the pad-contact code (compile only, no GPU): tools/contact_count.sh
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -S --cuda-device-only -I so100_mujoco_rl_amd/csrc -o gpurun_out/isa/contact.s tools/micro/contact_isa.hip
python3 - <<'PY'
import re
txt = open("gpurun_out/isa/contact.s").read()
for f in re.split(r"\n(?=_Z\w+:)", txt):
    name = f.split(":")[0]
    if not name.startswith("_Z"): continue
    ins = [l.strip() for l in f.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    v = sum(l.startswith("v_") for l in ins)
    m = re.search(r"\.vgpr_count:\s+(\d+)", f) or re.search(r"; NumVgprs: (\d+)", f)
    sc = re.search(r"; ScratchSize: (\d+)", f)
    print(f"{name[:24]:26s} total {len(ins):5d}  VALU {v:5d}  scratch-ops {sum('scratch_' in l for l in ins)}  ds-ops {sum(l.startswith('ds_') for l in ins)}  branches {sum(l.startswith('s_cbranch') for l in ins)}  "
          + (f"vgprs {m.group(1)}" if m else "") + (f" scratch {sc.group(1)} B" if sc else ""))
PY
