cd $GRAFT_REPO_ROOT
for b in 0 1 2; do
SX_BUSIEST_LAST=$b timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/e4_bl$b.json 2> gpurun_out/e4_bl$b.err
done
