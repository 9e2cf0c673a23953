cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $O/pmc_t1 -- python3 tools/time_step.py C2 --marg 4 4 > $O/pmc_t1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_t2 -- python3 tools/time_step.py C2 --marg 4 4 > $O/pmc_t2.log 2>&1
echo ok
