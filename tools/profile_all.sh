#!/bin/bash
# Runs on the GPU box (via gpurun): everything profiles/<tag>_* is made of, on the tree as it is.
#   tools/profile_all.sh <tag> <commit>        -> gpurun_out/<tag>_*  (tools/collect_profiles.sh copies the summaries to profiles/)
# Needs build/variants/lib_mstats.so (tools/build_variant.py mstats -DB9_MARG_STATS) of the same sources for the stats pass.
set -o pipefail
TAG=${1:-r05}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
bash tools/profile_round.sh $TAG $COMMIT > $O/${TAG}_profile_round.log 2>&1 || exit 1
echo "profile_round done"
bash tools/profile_marg.sh $TAG > $O/${TAG}_profile_marg.log 2>&1 || exit 1
echo "profile_marg done"
python3 tools/marg_stats.py $TAG > $O/${TAG}_marg_stats.log 2>&1 && cp profiles/${TAG}_marg_stats.json $O/ || echo "(marg_stats failed)"
# the bench lines LAST: they quote the counters of the passes above (profiles/<tag>_summary.json, <tag>_marg_stats.json of THESE sources)
cp $O/${TAG}_summary.json profiles/${TAG}_summary.json
python3 bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench_line.err || exit 1
echo "bench line done"
python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_line_driver_shape.json 2>> $O/${TAG}_bench_line.err || exit 1
{ echo "# Config sweep, ${TAG} (commit ${COMMIT}; tools/config_sweep.py on one MI355X)"; echo; python3 tools/config_sweep.py; } > $O/${TAG}_config_sweep.md 2>&1 || echo "(config_sweep failed)"
echo "config sweep done"
{ echo "# Walkers per GPU, 50k stars x 8 filters, ${TAG} (commit ${COMMIT}; tools/walker_scaling.py; 1-4 walkers run the tree launch)"; echo; python3 tools/walker_scaling.py; } > $O/${TAG}_walker_scaling.md 2>&1 || echo "(walker_scaling failed)"
{
  python3 tools/time_marg.py 50000 4 4 8; python3 tools/time_marg.py 50000 8 8 8; python3 tools/time_marg.py 50000 4 4 8 --filters 16
  python3 tools/time_marg.py 50000 4 4 8 --filters 4; python3 tools/time_marg.py 30000 4 4 8 --pops 2; python3 tools/time_marg.py 30000 4 4 8 --pops 2 --filters 16
  python3 tools/time_marg.py 20000 4 4 8 --wd 0.05; python3 tools/time_marg.py 20000 4 4 1; python3 tools/time_marg.py 10000 4 4 1; python3 tools/time_marg.py 10000 8 8 1; python3 tools/time_marg.py 200 4 4 1 --filters 4
  python3 tools/time_marg.py 50000 4 4 8 --sample; python3 tools/time_marg.py 30000 4 4 8 --pops 2 --sample; python3 tools/time_marg.py 50000 4 4 8 --filters 16 --sample
} > $O/${TAG}_marg_instances.txt 2>&1 || echo "(marg instances failed)"
{ python3 tools/time_step.py C0 C1 C2 C3 C4 F16 F16P2 F4; python3 tools/time_step.py C0 C1 C2 C3 C4 F16 F16P2 F4 --marg 4 4; python3 tools/time_step.py C1 C3 --marg 8 8;
  echo "# the two-launch marginalised step (b9_tuning.two_launch_steps), for comparison"; B9_TWO_LAUNCH_STEPS=1 python3 tools/time_step.py C1 C2 C3 --marg 4 4; } > $O/${TAG}_time_step.txt 2>&1 || echo "(time_step failed)"
{ python3 tools/time_logpost.py; python3 tools/time_logpost.py --marg 4 4; } > $O/${TAG}_time_logpost.txt 2>&1 || echo "(time_logpost failed)"
{ for s in C1 C2 C3 C4; do python3 tools/soak_determinism.py 50 $s --marg 4 4; done; python3 tools/soak_determinism.py 100 C2; python3 tools/soak_determinism.py 100 C3; } > $O/${TAG}_soak.txt 2>&1 || echo "(soak failed)"
if [ -f build/variants/lib_gantt.so ]; then
  { for s in C1 C2 C3 C4; do B9_HIP_LIB=build/variants/lib_gantt.so python3 tools/gantt_marg.py $s > $O/gantt_tmp.txt; head -6 $O/gantt_tmp.txt; grep -A4 "table builders" $O/gantt_tmp.txt; grep -A8 "star workgroups, dispatch" $O/gantt_tmp.txt; done; rm -f $O/gantt_tmp.txt; } > $O/${TAG}_gantt_marg.txt 2>&1 || echo "(gantt failed)"
fi
echo "all done"
