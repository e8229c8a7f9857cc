mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export DVSR_CONV_WINO=2 DVSR_CONV_WINO3=1
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d gpurun_out/dbp$i -o p -- python tools/wino_bench.py --quick > /dev/null 2>&1
  python tools/pmc_dump.py gpurun_out/dbp$i/p_results.db conv2d_wino3 2>&1 | tail -4
  rm -rf gpurun_out/dbp$i
done
