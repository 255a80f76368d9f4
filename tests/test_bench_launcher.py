"""bench.py --gpus N must start N ranks itself (torch.distributed.run, one process per GPU) when it is not already
running under a launcher, and relay exactly one JSON line with n_gpus = N.  Driven here on CPU with a stub worker (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r'''
import json, os, sys
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
ones = torch.ones(1); dist.all_reduce(ones)
open(os.path.join(os.environ["STUB_OUT"], f"rank{rank}.txt"), "w").write(" ".join(sys.argv[1:]))
print("noise line that is not the result")
if rank == 0:
    print(json.dumps({"n_gpus": world, "rccl_ranks": int(ones.item()), "argv": sys.argv[1:]}))
dist.destroy_process_group()
'''


def test_gpus_flag_launches_that_many_ranks(tmp_path):
    stub = tmp_path / "stub_worker.py"; stub.write_text(STUB)
    env = dict(os.environ, STUB_OUT=str(tmp_path)); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "7", "--warmup", "2", "--worker", str(stub)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                       # ONE JSON line, nothing else on stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 3 and res["rccl_ranks"] == 3
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("rank")) == ["rank0.txt", "rank1.txt", "rank2.txt"]
    assert "--steps" in res["argv"] and "7" in res["argv"] and "--gpus" in res["argv"]   # the ranks see the same arguments


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)
