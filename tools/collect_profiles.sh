#!/bin/bash
# After tools/profile_all.sh <tag> ran through gpurun: copy the summaries from gpurun_out/ (scratch) to profiles/ (tracked).
TAG=${1:-r05}
cd "$(dirname "$0")/.."
for f in summary.md summary.json kernel_stats.csv marg_summary.md marg_summary.json marg_kernel_stats.csv marg_stats.json bench_line.json \
         bench_line_driver_shape.json config_sweep.md walker_scaling.md marg_instances.txt time_step.txt time_logpost.txt soak.txt gantt_marg.txt; do
    [ -s gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
python3 tools/kernel_resources.py > profiles/${TAG}_kernel_resources.txt 2>&1
ls -la profiles/${TAG}_*
