WGRAD_BENCH_BATCHED=1 python tools/wgrad_bench.py 20 2>&1 | grep wgrad
python tools/wgrad_bench.py 20 2>&1 | grep wgrad
for w in 128 256 512 1024; do echo "== DVSR_WGRAD_BF_WGS=$w"; DVSR_WGRAD_BF_WGS=$w WGRAD_BENCH_BATCHED=1 python tools/wgrad_bench.py 10 2>&1 | grep "k3"; done
