cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_wave.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/e19_tests.log
for w in c1 c5; do
timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/e19_bench_$w.json
done
timeout 600 python bench.py --workload c1 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/e19_bench_c1b.json
