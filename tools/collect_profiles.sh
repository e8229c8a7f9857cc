#!/bin/bash
# Runs on the GPU box (gpurun): collects the per-round profile artefacts as TEXT under gpurun_out/<tag>/
# (rocprofv3 databases are summarised in place and deleted: they exceed the 64 MiB copy-back limit).
# usage: tools/collect_profiles.sh <tag>      e.g. r02_z   -> copy what should be judged into profiles/
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf "$out"; mkdir -p "$out"
B="--no-cpu-baseline --no-inner-step --no-split --no-meta --no-validation"
# 1. the driver's bench line (all legs)
python bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench_stderr.txt
# 2. kernel trace of the headline forward
rocprofv3 --kernel-trace --stats -d $out/db1 -o r -- python bench.py --steps 10 --warmup 3 $B > /dev/null 2>&1
python tools/rocprof_summary.py $out/db1/r_results.db > $out/${tag}_kernel_trace_bench_fwd180x320.txt; rm -rf $out/db1
# 3. inner step: timings, kernel table, launch timeline of one EDVR fwd+bwd at the SLR size
rocprofv3 --kernel-trace --stats -d $out/db2 -o r -- python tools/inner_bench.py 176 320 6 2>&1 | grep -E "inner|EDVR|MFDN|full|LR|adapt_video|overlapped" > $out/${tag}_inner_step_176x320.txt
python tools/rocprof_summary.py $out/db2/r_results.db >> $out/${tag}_inner_step_176x320.txt; rm -rf $out/db2
rocprofv3 --kernel-trace --stats -d $out/db2b -o r -- python tools/edvr_step_profile.py 44 80 30 2>&1 | grep EDVR > $out/${tag}_edvr_step_44x80_timeline.txt
python tools/trace_dump.py $out/db2b/r_results.db charbonnier_partial >> $out/${tag}_edvr_step_44x80_timeline.txt; rm -rf $out/db2b
python tools/edvr_step_profile.py 44 80 40 2>&1 | grep EDVR > $out/${tag}_edvr_step_44x80.txt
# 4. per-launch forward tables (hipEvents), headline size and SLR size
python tools/op_profile.py 180 320 5 2>&1 | grep -v amdgpu > $out/${tag}_per_launch_fwd180x320.txt
python tools/op_profile.py 44 80 10 2>&1 | grep -v amdgpu > $out/${tag}_per_launch_fwd44x80.txt
# 5. PMC: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) and matrix-pipe utilisation
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/db_$c -o p -- python bench.py --steps 3 --warmup 1 $B > /dev/null 2>&1
done
python tools/pmc_traffic.py $out/db_FETCH_SIZE/p_results.db $out/db_WRITE_SIZE/p_results.db $out/${tag}_pmc_hbm_traffic.txt $out/pmc_traffic.json
python tools/pmc_mfma.py $out/db_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db $out/db_GRBM_GUI_ACTIVE/p_results.db > $out/${tag}_pmc_mfma_util.txt
rm -rf $out/db_FETCH_SIZE $out/db_WRITE_SIZE $out/db_SQ_VALU_MFMA_BUSY_CYCLES $out/db_GRBM_GUI_ACTIVE
# 6. neighbours of the path: metrics, degradation, meta step, estimator, bf16 modes, the other backbones
python tools/metrics_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_metrics_720x1280.txt
python tools/degrade_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_degradation.txt
python tools/meta_bench.py 4 1 2>&1 | grep -v amdgpu > $out/${tag}_meta_train_step.txt
python tools/estimator_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_estimator.txt
python tools/bf16_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_bf16_modes.txt
python tools/backbone_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_backbones_tof_duf.txt
# 7. EDVR-L x4 (configs[4]) forward+backward in bf16 mode: timing + kernel table
rocprofv3 --kernel-trace --stats -d $out/db7 -o r -- python tools/edvr_l_step_profile.py 1 20 2>&1 | grep EDVR-L > $out/${tag}_edvr_l_bf16_step.txt
python tools/rocprof_summary.py $out/db7/r_results.db >> $out/${tag}_edvr_l_bf16_step.txt; rm -rf $out/db7
python tools/edvr_l_step_profile.py 1 20 2>&1 | grep EDVR-L >> $out/${tag}_edvr_l_bf16_step.txt
python tools/edvr_l_step_profile.py 0 10 2>&1 | grep EDVR-L >> $out/${tag}_edvr_l_bf16_step.txt
# 7b. HBM-side traffic of the EDVR-L forward+backward step per MFMA mode (merged into pmc_traffic.json for bench.py)
for mode in 0 1 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    EDVR_L_EXACT_STEPS=1 rocprofv3 --pmc $c --kernel-trace -d $out/dbl_$c -o p -- python tools/edvr_l_step_profile.py $mode 6 > /dev/null 2>&1
  done
  echo "EDVR-L fwd+bwd bf16_mfma=$mode: $(python tools/pmc_total.py $out/dbl_FETCH_SIZE/p_results.db $out/dbl_WRITE_SIZE/p_results.db 8 --json $out/pmc_traffic.json edvr_l_fwd_bwd_mode$mode | head -1)" >> $out/${tag}_edvr_l_bf16_step.txt
  rm -rf $out/dbl_FETCH_SIZE $out/dbl_WRITE_SIZE
done
# 8. micro-measurements behind DESIGN 3.1 / 3.2: what hides behind an fp32 MFMA; cycle stamps of the DCN forward (debug build)
python tools/mfma_shadow.py 2>&1 | grep "cycles per" > $out/${tag}_mfma_shadow.txt
if [ -f dynavsr_amd/libdynavsr_hip_trace.so ]; then
  python tools/dcn_dma_trace.py 2 2>&1 | grep -v amdgpu > $out/${tag}_dcn_dma_trace.txt
  python tools/dcn_dma_trace.py 2 44 80 2>&1 | grep -v amdgpu >> $out/${tag}_dcn_dma_trace.txt
fi
# 9. r03: K frames as one batch with per-frame parameter gradients -- phases, kernel table, launch timeline; the
#    weight-gradient kernel alone on the batch shapes; DCN backward alone
python tools/inner_batch_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_inner_batch.txt
for K in 8 16; do
  python tools/inner_batch_profile.py $K 10 2>&1 | grep -v amdgpu >> $out/${tag}_inner_batch.txt
done
rocprofv3 --kernel-trace --stats -d $out/db9 -o r -- python tools/inner_batch_profile.py 8 6 > /dev/null 2>&1
python tools/rocprof_summary.py $out/db9/r_results.db > $out/${tag}_inner_batch_K8_kernels.txt
python tools/trace_dump.py $out/db9/r_results.db l1_final > $out/${tag}_inner_batch_K8_timeline.txt; rm -rf $out/db9
WGRAD_BENCH_BATCHED=1 python tools/wgrad_bench.py 20 2>&1 | grep wgrad > $out/${tag}_wgrad_kernel.txt
python tools/wgrad_bench.py 20 2>&1 | grep wgrad >> $out/${tag}_wgrad_kernel.txt
if [ -f dynavsr_amd/libdynavsr_hip_trace.so ]; then   # staging ablations of the scalar-staged weight-gradient kernel (debug build)
  for nf in 0 1 3 5 7; do
    echo "== DVSR_WGRAD_WIDE=0 DVSR_WGRAD_NOFLUSH=$nf (bit 0: no flush, bit 1: no global loads, bit 2: no LDS writes; results wrong)" >> $out/${tag}_wgrad_ablation.txt
    DVSR_WGRAD_WIDE=0 DVSR_WGRAD_NOFLUSH=$nf WGRAD_BENCH_BATCHED=1 DVSR_HIP_LIB=$PWD/dynavsr_amd/libdynavsr_hip_trace.so python tools/wgrad_bench.py 10 2>&1 | grep wgrad >> $out/${tag}_wgrad_ablation.txt
  done
fi
python tools/dcn_bwd_bench.py 20 2>&1 | grep -v amdgpu > $out/${tag}_dcn_bwd.txt
rocprofv3 --kernel-trace --stats -d $out/db9b -o r -- python tools/dcn_bwd_bench.py 20 > /dev/null 2>&1
python tools/rocprof_summary.py $out/db9b/r_results.db | head -8 >> $out/${tag}_dcn_bwd.txt; rm -rf $out/db9b
# r06: the fused kernel's phases (debug build), the fp32-MFMA form beside it, and its HBM-side traffic at 5x64x180x320
# (algorithmic: x, gout, gx 73.7 MB each + offsets / masks and their gradients 2 x 248.8 MB = 719 MB; the weight-gradient
# partials, 170 MB per call, are the kernel's own)
echo "== DVSR_DCN_BWD=fp32 (fp32-MFMA contractions)" >> $out/${tag}_dcn_bwd.txt
DVSR_DCN_BWD=fp32 python tools/dcn_bwd_bench.py 20 2>&1 | grep -v amdgpu >> $out/${tag}_dcn_bwd.txt
if [ -f dynavsr_amd/libdynavsr_hip_trace.so ]; then
  python tools/dcn_bwd_trace.py 5 180 320 2>&1 | grep -v amdgpu >> $out/${tag}_dcn_bwd.txt
  python tools/dcn_bwd_trace.py 5 44 80 2>&1 | grep -v amdgpu >> $out/${tag}_dcn_bwd.txt
fi
echo "== PMC, 5x64x180x320 (separate passes; mean per launch)" >> $out/${tag}_dcn_bwd.txt
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace -d /tmp/dbq -o p -- python $GRAFT_REPO_ROOT/tools/dcn_bwd_bench.py 4 big > /dev/null 2>&1)
  f=$(find /tmp/dbq -name "p_results.db" | head -1)
  [ -n "$f" ] && python tools/pmc_dump.py $f mdcn_ >> $out/${tag}_dcn_bwd.txt
  rm -rf /tmp/dbq
done
# PMC: matrix-pipe utilisation of the batched inner step's kernels
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/dbi_$c -o p -- python tools/inner_batch_profile.py 8 3 > /dev/null 2>&1
done
python tools/pmc_mfma.py $out/dbi_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db $out/dbi_GRBM_GUI_ACTIVE/p_results.db > $out/${tag}_pmc_mfma_util_inner_batch.txt
rm -rf $out/dbi_SQ_VALU_MFMA_BUSY_CYCLES $out/dbi_GRBM_GUI_ACTIVE
# r04: HBM-side traffic of the batched inner step (K = 16, the leg the metric is named after): 5 warm-up + 2 + 2 profiled batches
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $out/dbt_$c -o p -- python tools/inner_batch_profile.py 16 2 > /dev/null 2>&1
done
echo "batched inner step K=16 LR 176x320, per batch: $(python tools/pmc_total.py $out/dbt_FETCH_SIZE/p_results.db $out/dbt_WRITE_SIZE/p_results.db 9 --json $out/pmc_traffic.json inner_step_batched frames_per_batch=16 h=176 w=320 | head -1)" > $out/${tag}_pmc_hbm_traffic_inner_step.txt
rm -rf $out/dbt_FETCH_SIZE $out/dbt_WRITE_SIZE
# 10. the Winograd kernels per layer shape (accuracy vs fp64, time per launch incl. the weight pack): F(4x4, 3x3) on the bf16
#     pipe (round 6, conv2d_wino5.hip), F(2x2, 3x3) on the bf16 pipe (conv2d_wino4.hip), on the fp32 MFMA (conv2d_wino.hip), the
#     direct kernels
echo "== Winograd F(4x4, 3x3) on the bf16 pipe, exact 3-way split (conv2d_wino5.hip; DVSR_CONV_WINO5=2: wherever eligible)" > $out/${tag}_wino_vs_direct.txt
DVSR_CONV_WINO=2 DVSR_CONV_WINO5=2 python tools/wino_bench.py 2>&1 | grep -v amdgpu >> $out/${tag}_wino_vs_direct.txt
echo "== ... its 16 x 32-pixel workgroup tiles (DVSR_CONV_WINO5=3)" >> $out/${tag}_wino_vs_direct.txt
DVSR_CONV_WINO=2 DVSR_CONV_WINO5=3 python tools/wino_bench.py --quick 2>&1 | grep -v amdgpu >> $out/${tag}_wino_vs_direct.txt
echo "== Winograd F(2x2, 3x3) on the bf16 pipe, exact 3-way split (conv2d_wino4.hip; DVSR_CONV_WINO5=0)" >> $out/${tag}_wino_vs_direct.txt
DVSR_CONV_WINO=2 DVSR_CONV_WINO5=0 python tools/wino_bench.py 2>&1 | grep -v amdgpu >> $out/${tag}_wino_vs_direct.txt
echo "== Winograd F(2x2, 3x3) on the fp32 MFMA (conv2d_wino.hip, DVSR_CONV_WINO3=0)" >> $out/${tag}_wino_vs_direct.txt
DVSR_CONV_WINO=2 DVSR_CONV_WINO3=0 python tools/wino_bench.py 2>&1 | grep -v amdgpu >> $out/${tag}_wino_vs_direct.txt
echo "== direct kernels (DVSR_CONV_WINO=0)" >> $out/${tag}_wino_vs_direct.txt
DVSR_CONV_WINO=0 python tools/wino_bench.py 2>&1 | grep -v amdgpu >> $out/${tag}_wino_vs_direct.txt
if [ -f dynavsr_amd/libdynavsr_hip_trace.so ]; then
  # cycle-stamp timelines (debug build): the F(4x4) kernel on the 8- and the 16-chunk layer, form 4 beside it
  python tools/wino5_trace.py 5 64 0 64 180 320 2>&1 | grep -v amdgpu > $out/${tag}_wino5_trace.txt
  python tools/wino5_trace.py 5 64 64 64 180 320 2>&1 | grep -v amdgpu >> $out/${tag}_wino5_trace.txt
  DVSR_CONV_WINO5=0 python tools/wino_trace.py 2>&1 | grep -v amdgpu > $out/${tag}_wino_trace.txt
fi
# compile-time probe builds of the F(4x4) kernel (results WRONG: each bounds what one term of the chunk loop costs), same box
# (no weight-fragment probe: without its global loads the compiler re-allocates the chunk loop and the build measures 2.5x SLOWER)
if [ -x tools/wino5_variants.sh ]; then
  tools/wino5_variants.sh NOPROD NOMMA NODMA NOBR NOVW > /dev/null 2>&1
  cp dynavsr_amd/libdynavsr_hip.so /tmp/base.so
  for v in base NOPROD NOMMA NODMA NOBR NOVW base; do
    if [ $v = base ]; then cp /tmp/base.so dynavsr_amd/libdynavsr_hip.so; else cp dynavsr_amd/libdynavsr_hip_w5_$v.so dynavsr_amd/libdynavsr_hip.so; fi
    echo "== $v" >> $out/${tag}_wino5_ablation.txt
    DVSR_CONV_WINO=2 DVSR_CONV_WINO5=2 python tools/wino_bench.py --quick 2>&1 | grep -E "fe_rb|L1_offset|rc_rb" >> $out/${tag}_wino5_ablation.txt
  done
  cp /tmp/base.so dynavsr_amd/libdynavsr_hip.so; rm -f dynavsr_amd/libdynavsr_hip_w5_*.so
fi
# PMC picture of the two bf16x3 Winograd kernels on the same three layers (LDS activity, wave wait / issue split, instruction mix,
# vector-memory path)
for mode in 2 0; do
  echo "== DVSR_CONV_WINO5=$mode" >> $out/${tag}_wino5_pmc.txt
  for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
    DVSR_CONV_WINO=2 DVSR_CONV_WINO5=$mode rocprofv3 --pmc $set --kernel-trace -d $out/dbw -o p -- python tools/wino_bench.py --quick > /dev/null 2>&1
    python tools/pmc_dump.py $out/dbw/p_results.db conv2d_wino >> $out/${tag}_wino5_pmc.txt; rm -rf $out/dbw
  done
done
# r04: what the two waves of a SIMD share (micro-benchmarks behind DESIGN 3.1f)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap 2>/dev/null && /tmp/mfma_overlap > $out/${tag}_mfma_overlap.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/wave_phase.hip -o /tmp/wave_phase 2>/dev/null && /tmp/wave_phase > $out/${tag}_wave_phase.txt
python tools/graph_fwd_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_graph_vs_eager_fwd.txt
for sw in "DVSR_CONV_WINO5=0" "DVSR_CONV_WINO=0"; do
echo "== forward 180x320 with $sw" >> $out/${tag}_wino_vs_direct.txt
env $sw python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inner-step --no-split --no-meta --no-validation 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','value_one_clip_in_flight')}, d['roofline']['kernel'], d['roofline']['frac'])" >> $out/${tag}_wino_vs_direct.txt
done
# r04: clips in flight on HIP streams against one clip at a time and against one forward over a batch of clips
for cfg in "2 1" "3 1" "1 8" "2 4"; do
  python tools/fwd_concurrent.py 180 320 20 $cfg 2>&1 | grep -E "stream|diff" >> $out/${tag}_fwd_clips_in_flight.txt
done
python tools/inner_two_streams.py 16 2 2>&1 | grep "inner step" >> $out/${tag}_fwd_clips_in_flight.txt
# r05: the split's edge semantics, the bring-up orders against the stream -> hardware-queue assignment, the DCN forward on
# the bf16 pipe against the fp32-MFMA kernel (per-launch tables)
python tools/split_edge_probe.py 2>&1 | grep -v amdgpu > $out/${tag}_split_edges.txt
python tools/queue_probe_bench.py 2>&1 | grep -v amdgpu > $out/${tag}_queue_probe.txt
DVSR_DCN_FWD=dma python tools/op_profile.py 180 320 5 2>&1 | grep -E "mdcn|total" > $out/${tag}_dcn_fwd_split_vs_dma.txt
echo "== split (default)" >> $out/${tag}_dcn_fwd_split_vs_dma.txt
grep -E "mdcn|total" $out/${tag}_per_launch_fwd180x320.txt >> $out/${tag}_dcn_fwd_split_vs_dma.txt
# r05 (late): the split weight gradient -- round-4 schedule, re-scheduled, re-scheduled + row split -- per launch and in the
# batched inner step; its timeline (debug build)
for cfg in "DVSR_WGRAD_S3V=0" "DVSR_WGRAD_S3_KYS_BELOW=0 DVSR_WGRAD_S3W=0" "DVSR_WGRAD_S3_KYS_BELOW=0" "DVSR_WGRAD_S3_KYS_BELOW=4000"; do
  echo "== $cfg" >> $out/${tag}_wgrad_s3v_collect.txt
  env $cfg python tools/wgrad_bench.py 30 2>&1 | grep split >> $out/${tag}_wgrad_s3v_collect.txt
  env $cfg WGRAD_BENCH_BATCHED=1 python tools/wgrad_bench.py 30 2>&1 | grep split >> $out/${tag}_wgrad_s3v_collect.txt
  env $cfg python tools/inner_batch_bench.py 2>&1 | grep -E "batch of 8|per-frame loop" >> $out/${tag}_wgrad_s3v_collect.txt
done
if [ -f dynavsr_amd/libdynavsr_hip_trace.so ]; then
  # (the stamps are in the four-wave forms: conv2d_wgrad_split3v_kernel, row-split and not)
  DVSR_WGRAD_S3W=0 python tools/wgrad_trace.py 40 64 64 44 80 2>&1 | grep -v amdgpu > $out/${tag}_wgrad_trace_collect.txt
  DVSR_WGRAD_S3W=0 DVSR_WGRAD_S3_KYS_BELOW=0 python tools/wgrad_trace.py 40 64 64 44 80 2>&1 | grep -v amdgpu >> $out/${tag}_wgrad_trace_collect.txt
fi
du -sh gpurun_out
