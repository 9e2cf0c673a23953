#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py (never combined with
# a trace domain: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then tools/profile_summary.py.
# usage: tools/profile_round.sh <tag> <commit>     -> gpurun_out/<tag>_*   (copy the summaries to profiles/)
set -o pipefail
TAG=${1:-r05}
COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
BENCH="python3 bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-sustained"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace -- $BENCH > $O/${TAG}_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_pmc_fetch -- $BENCH > $O/${TAG}_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_pmc_write -- $BENCH > $O/${TAG}_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/${TAG}_pmc_sq -- $BENCH > $O/${TAG}_pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_lds -- $BENCH > $O/${TAG}_pmc_lds.log 2>&1 || echo "(lds pass failed: optional)"
python3 tools/profile_summary.py $TAG "$COMMIT" "$BENCH"
