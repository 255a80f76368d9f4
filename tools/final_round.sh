#!/bin/bash
# End-of-round evidence on the GPU box (run through gpurun from the repo root, in two calls: the whole set exceeds one call's limit):
#   tools/final_round.sh <tag> a      bench lines (all workloads) + rocprofv3 passes of env01_free / env01_reference
#   tools/final_round.sh <tag> b      rocprofv3 passes of env01_nopads / env01_contact + phase profiles (needs `python tools/rollout_prof.py build` here first)
# then HERE: tools/copy_profiles.sh <tag> free nopads reference contact
set -e
TAG=${1:-r03b_final}; PART=${2:-a}
mkdir -p gpurun_out/$TAG
if [ "$PART" = a ]; then
  tools/bench_all.sh $TAG > gpurun_out/$TAG/bench_all.txt 2>&1
  for w in free reference; do tools/profile_round.sh ${TAG}_$w env01_$w > gpurun_out/$TAG/prof_$w.log 2>&1; done
else
  for w in nopads contact; do tools/profile_round.sh ${TAG}_$w env01_$w > gpurun_out/$TAG/prof_$w.log 2>&1; done
  for w in free nopads ref c5; do SO100_BALANCE=0 python tools/rollout_prof.py $w random > gpurun_out/$TAG/phase_${w}_random.txt 2>&1; done
  SO100_BALANCE=0 python tools/rollout_prof.py ref resting > gpurun_out/$TAG/phase_ref_resting.txt 2>&1
fi
echo FINAL_ROUND_${PART}_DONE
