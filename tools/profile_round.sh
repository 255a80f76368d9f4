#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag> [workload, e.g. env01_reference]   -> gpurun_out/<tag>/{stats,sq,fetch,write}/...
# Passes are separate (kernel-trace --stats; then one --pmc group each), as the MI355X guide prescribes.
set -e
TAG=${1:-round}
shift || true
WL=${1:-env01_free}
EXTRA="--workload $WL --no-sb3-path"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-large-batch $EXTRA"
PMCB="python3 $PWD/bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-large-batch $EXTRA"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/sq" -- $PMCB > "$OUT/sq.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $PMCB > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $PMCB > "$OUT/write.log" 2>&1
cd - > /dev/null
python3 tools/pmc_summary.py "$OUT" "$WL" > "$OUT/pmc_summary.json"
tail -1 "$OUT/stats.log"
