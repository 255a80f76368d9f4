cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export PYTHONPATH=$PWD:$PYTHONPATH
mkdir -p gpurun_out/e2e && cd gpurun_out/e2e
for e in Env01-v1 Env02-v1 Env05-v1 Env06-v1; do
  timeout -k 10 150 python -m so100_mujoco_rl_amd.main -a PPO train -e $e --envs 4096 --iters 300 2>&1 | grep -E "iter +(10|100|200|300) |done:|Stopping|Error|error" | sed "s/^/$e  /"
done
