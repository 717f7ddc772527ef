cd $GRAFT_REPO_ROOT
for s in 0 32 64 128; do
timeout 300 python bench.py --workload c2 --steps 30 --warmup 5 --no-cpu-baseline --subchunk-kib $s 2>/dev/null | tail -1 > gpurun_out/e23_c2_sub$s.json
done
for s in 64 128; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --subchunk-kib $s 2>/dev/null | tail -1 > gpurun_out/e23_c3_sub$s.json
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gib 8 2>/dev/null | tail -1 > gpurun_out/e23_c3_8gib.json
