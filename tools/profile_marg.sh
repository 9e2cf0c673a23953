#!/bin/bash
# Runs on the GPU box (via gpurun): the marginalised mode's kernels under rocprofv3 -- kernel-trace stats, then separate
# PMC passes (never combined with a trace domain) -- on the bench shape (50k stars x 8 filters x 8 walkers, K = Q = 4).
# usage: tools/profile_marg.sh <tag>     -> gpurun_out/<tag>_marg_*
set -o pipefail
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
CMD="python3 tools/time_marg.py 50000 4 4 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_marg_trace -- $CMD > $O/${TAG}_marg_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_marg_pmc_fetch -- $CMD > $O/${TAG}_marg_pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${TAG}_marg_pmc_write -- $CMD > $O/${TAG}_marg_pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/${TAG}_marg_pmc_sq -- $CMD > $O/${TAG}_marg_pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_marg_pmc_sca -- $CMD > $O/${TAG}_marg_pmc_sca.log 2>&1 || echo "(scalar pass failed: optional)"
python3 tools/profile_marg_summary.py $TAG
