#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# usage: tools/profile_round.sh <tag>     -> gpurun_out/<tag>_*   (copy the summaries to profiles/)
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
BENCH="python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- $BENCH > $OUT/${TAG}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${TAG}_pmc_write -- $BENCH > $OUT/${TAG}_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/${TAG}_pmc_sq -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1 || exit 1
python3 tools/profile_summary.py $TAG
