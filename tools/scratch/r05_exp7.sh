cd $GRAFT_REPO_ROOT
env | grep -i -E 'HSA|HIP_|ROC|GPU' > gpurun_out/e7_env.txt
export SX_BUSIEST_LAST=0
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e7_base.json 2> gpurun_out/e7_base.err
HSA_ENABLE_SDMA=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e7_sdma1.json 2> gpurun_out/e7_sdma1.err
HSA_ENABLE_SDMA=0 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e7_sdma0.json 2> gpurun_out/e7_sdma0.err
SX_REPLAY_COPY_WGS=2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e7_wgs2.json 2> gpurun_out/e7_wgs2.err
SX_REPLAY_COPY_WGS=8 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e7_wgs8.json 2> gpurun_out/e7_wgs8.err
