mkdir -p gpurun_out
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_stderr.txt; echo rc=$?
tail -3 gpurun_out/bench_stderr.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}); print('roofline',{k:v for k,v in d['roofline'].items() if k not in('note','algorithm','method','traffic_source')})
i=d['inner_step']; print('inner', i['value'], i['ms_per_step'], {k:v for k,v in i['roofline'].items() if k!='note'}, i.get('per_rank_value'))
print('loop', i['per_frame_loop']['value'], 'step_only', i['step_only']['value'], '3steps', i['three_inner_steps']['ms_per_frame'])
p=d['per_frame_pipeline']; print('pipe', p['value'], p['ms_per_frame'], {k:v for k,v in p['roofline'].items() if k!='note'})
print('meta', {k:d['meta_step'].get(k) for k in ('value','ms_per_outer_iteration','allreduce','error')})
print('val', {k:d['validation'].get(k) for k in ('value','psnr_vector_complete_on_rank0','mean_psnr_start_db','mean_psnr_final_db','error')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('parity_vs_this_run'))
PY
echo "== torchrun 1 rank (gloo default group + RCCL subgroup)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-split > gpurun_out/bench_line_torchrun1.json 2> gpurun_out/bench_stderr_torchrun1.txt; echo rc=$?
tail -3 gpurun_out/bench_stderr_torchrun1.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_line_torchrun1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'inner', d['inner_step']['value'], d['inner_step']['per_frame_loop']['value'], 'pipe', d['per_frame_pipeline']['value'])
print('meta', {k:d['meta_step'].get(k) for k in ('value','ms_per_outer_iteration','allreduce','error')})
print('val', {k:d['validation'].get(k) for k in ('value','psnr_vector_complete_on_rank0','error')})
PY
