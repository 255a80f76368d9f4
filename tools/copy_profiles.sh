#!/bin/bash
# Copy the judged summaries of tools/profile_round.sh runs from gpurun_out/ (scratch) into profiles/ (tracked):
#   tools/copy_profiles.sh r02_final free nopads reference contact
set -e
cd "$(dirname "$0")/.."
P=$1; shift
for w in "$@"; do
  d=gpurun_out/${P}_$w
  rm -f profiles/${P}_${w}_kernel_stats.csv
  cp "$(ls -t $d/stats/*/*_kernel_stats.csv | head -1)" profiles/${P}_${w}_kernel_stats.csv
  cp $d/pmc_summary.json profiles/${P}_${w}_pmc_summary.json
  grep "^{" $d/stats.log | tail -1 > profiles/${P}_${w}_bench_under_rocprof.json
done
ls profiles | grep "^$P" | head -40
