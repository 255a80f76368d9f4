"""Build a side library with extra preprocessor defines (experiments; the product library is untouched):
    python tools/side_build.py <suffix> -DNAME=VALUE ...     ->  so100_mujoco_rl_amd/libso100sim_<suffix>.so
Use it through SO100_LIB=<path> (so100_mujoco_rl_amd/lib.py)."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
suffix, defs = sys.argv[1], sys.argv[2:]
import torch
tl = os.path.join(os.path.dirname(torch.__file__), "lib")
csrc = os.path.join(ROOT, "so100_mujoco_rl_amd", "csrc")
odir = os.path.join(ROOT, "gpurun_out", "side_obj_" + suffix); os.makedirs(odir, exist_ok=True)
base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=fast",
        "-fno-slp-vectorize", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + defs + ["-c"]
jobs = [base + ["-o", os.path.join(odir, "sim.o"), os.path.join(csrc, "so100_sim.hip")]]
jobs += [base + [f"-DSO100_KIND={k}", "-o", os.path.join(odir, f"kind{k}.o"), os.path.join(csrc, "so100_kind.hip")] for k in range(1, 7)]
with ThreadPoolExecutor(7) as ex:
    list(ex.map(subprocess.check_call, jobs))
out = os.path.join(ROOT, "so100_mujoco_rl_amd", f"libso100sim_{suffix}.so")
subprocess.check_call(["g++", "-shared", "-o", out, os.path.join(odir, "sim.o")] + [os.path.join(odir, f"kind{k}.o") for k in range(1, 7)]
                      + ["-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl, "-Wl,-rpath,/opt/rocm/lib", "-lstdc++", "-lm"])
print("built", out)
