mkdir -p gpurun_out
B="--no-cpu-baseline --no-inner-step --no-split --no-meta --no-validation"
for w3 in 1 0 1 0; do
DVSR_CONV_WINO3=$w3 python bench.py --steps 40 --warmup 10 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WINO3=$w3', {k: round(d[k],3) for k in ('value','ms_per_step')}, {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('achieved','frac','algorithmic_tflops','mfma_flops_executed_frac','avg_launch_ms')})"
done
