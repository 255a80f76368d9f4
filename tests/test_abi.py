"""The C-ABI library loads and exports every symbol include/so100_sim.h declares (no compute without a GPU), its
metadata calls answer, and the product path fails LOUDLY (no CPU fallback) when no HIP device is usable.  CPU only."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from so100_mujoco_rl_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "so100_sim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(so100_[a-z_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(L):
    from so100_mujoco_rl_amd import lib
    syms = declared_symbols()
    assert sorted(syms) == sorted(lib.EXPORTS), (syms, lib.EXPORTS)
    for s in syms:
        assert hasattr(L, s), s


def test_metadata_calls(L):
    assert L.so100_abi_version() == 3
    assert [L.so100_obs_dim(k) for k in (1, 2, 3, 4, 5, 6)] == [15, 15, 8, 8, 8, 15]
    assert L.so100_obs_dim(0) == -1 and L.so100_obs_dim(7) == -1
    n = L.so100_num_state_fields()
    assert n == 98
    assert L.so100_state_field_index(b"q0") == 0 and L.so100_state_field_index(b"cube_x") == 6
    assert L.so100_state_field_index(b"v0") == 13            # 13 qpos rows then 12 qvel rows: get/set_state rely on it
    assert L.so100_state_field_index(b"nope") == -1
    L.so100_state_field_name.restype = C.c_char_p
    assert [L.so100_state_field_index(L.so100_state_field_name(i)) for i in range(n)] == list(range(n))
    assert L.so100_state_field_name(-1) is None and L.so100_state_field_name(n) is None


def test_create_validates_arguments_and_has_no_cpu_fallback(L):
    from so100_mujoco_rl_amd import lib
    import torch
    h = C.c_void_p()
    bad = lib.Config(9, 16, 0, 8, 3, 4, 16, 0, 0, 0, 0)
    assert L.so100_create(C.byref(bad), C.byref(h)) == -1 and b"env_kind" in L.so100_last_error()
    bad = lib.Config(1, 0, 0, 8, 3, 4, 16, 0, 0, 0, 0)
    assert L.so100_create(C.byref(bad), C.byref(h)) == -1 and b"num_envs" in L.so100_last_error()
    bad = lib.Config(1, 16, 0, 4 | 8, 3, 4, 16, 0, 0, 0, 0)
    assert L.so100_create(C.byref(bad), C.byref(h)) == -1 and b"mutually exclusive" in L.so100_last_error()
    if not torch.cuda.is_available():
        ok = lib.Config(1, 16, 0, 8, 3, 4, 16, 0, 0, 0, 0)
        assert L.so100_create(C.byref(ok), C.byref(h)) == -2
        assert b"no CPU fallback" in L.so100_last_error()
        with pytest.raises(lib.So100Error):
            lib.So100Sim(1, 16)


def test_library_binds_one_hip_runtime(L):
    """libso100sim.so must resolve libamdhip64 to the copy PyTorch already loaded (two runtimes in one process break)."""
    maps = open("/proc/self/maps").read()
    libs = sorted(set(re.findall(r"(/\S*libamdhip64\S*)", maps)))
    assert len(libs) == 1, libs


def test_integration_doc_matches_abi():
    """The ctypes stub printed in INTEGRATION.md must list exactly the fields of so100_config / so100_step_io."""
    from so100_mujoco_rl_amd import lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    io_names = re.findall(r'"(\w+_dev)"', doc.split("class IO")[1].split("lib.so100_last_error")[0])
    assert io_names == [f[0] for f in lib.StepIO._fields_]
    cfg_names = re.findall(r'\("(\w+)", C\.c_', doc.split("class Cfg")[1].split("class IO")[0])
    assert cfg_names == [f[0] for f in lib.Config._fields_]
    hdr = open(os.path.join(ROOT, "include", "so100_sim.h")).read()
    step_io = hdr.split("typedef struct {", 3)[2].split("} so100_step_io;")[0]
    assert re.findall(r"(\w+_dev);", step_io) == io_names


def test_library_is_not_older_than_its_sources():
    """`make -q`: the in-tree libso100sim.so (what travels to the GPU box) was built from the kernel sources as they are now.  A stale library
    silently tests yesterday's kernels against today's oracle (it happened: a narrowphase changed in oracle + host check, the .so was not rebuilt)."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "so100_mujoco_rl_amd", "csrc")
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no build tools here")
    assert subprocess.call(["make", "-q", "-C", csrc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0, \
        "so100_mujoco_rl_amd/libso100sim.so is older than csrc/*: run `make -C so100_mujoco_rl_amd/csrc -j7` (or __graft_entry__.build())"
