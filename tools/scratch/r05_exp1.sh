cd $GRAFT_REPO_ROOT
timeout 300 python tools/scratch/r05_sub_probe.py 16 > gpurun_out/e1_sub.txt 2>&1
for s in 0 64; do
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --subchunk-kib $s > gpurun_out/e1_bench_sub$s.json 2> gpurun_out/e1_bench_sub$s.err
done
SX_MISSION_STREAMS=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e1_bench_mstreams.json 2> gpurun_out/e1_bench_mstreams.err
SX_TIMING=1 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/e1_bench_timing.json 2> gpurun_out/e1_bench_timing.err
