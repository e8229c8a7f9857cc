#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r02_z5; rm -rf $out; mkdir -p $out
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/db_$c -o p -- python tools/edvr_step_profile.py 44 80 6 > /dev/null 2>&1
done
python tools/pmc_mfma.py $out/db_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db $out/db_GRBM_GUI_ACTIVE/p_results.db > $out/r02_z_pmc_mfma_util_edvr_step_44x80.txt
rm -rf $out/db_*
head -30 $out/r02_z_pmc_mfma_util_edvr_step_44x80.txt
