cd $GRAFT_REPO_ROOT
for b in 0 2; do
SX_TIMELINE=1 SX_BUSIEST_LAST=$b timeout 200 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/e5_bl$b.json 2> gpurun_out/e5_bl$b.err
done
