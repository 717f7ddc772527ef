cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sharded.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/e18_tests.log
timeout 600 python bench.py --gpus 8 --backend gloo --single-device --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/e18_bench_8ranks_1gpu_gloo_strong.json 2> gpurun_out/e18_b8.err
timeout 600 python bench.py --gpus 2 --backend gloo --single-device --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/e18_bench_2ranks_1gpu_gloo_strong.json 2> gpurun_out/e18_b2.err
for w in c1 c2 c5; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/e18_bench_$w.json
done
