#!/bin/bash
# Everything profiles/ holds for one commit, in one go on the GPU box: tools/final_profiles.sh TAG  -> gpurun_out/TAG_*
tag=$1
python bench.py > gpurun_out/${tag}_bench_c3_64gib.json 2> gpurun_out/${tag}_bench_c3.err < /dev/null
tail -c 600 gpurun_out/${tag}_bench_c3_64gib.json; echo
# the same Missions with one launch each (rounds 1-5), and the fused launch over the buffer in one piece
python bench.py --no-cpu-baseline --per-mission-launches 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c3_64gib_per_mission.json
SX_PIECE_MIB=0 python bench.py --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c3_64gib_one_piece.json
python bench.py --gib 8 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c3_8gib.json
# (a process that held 64 GiB of HBM has just ended: for about ten seconds the driver clears that memory, and device-to-host copies
# run at 43 instead of 53 GB/s meanwhile — C1 and the text workload, which are bound by exactly those copies, wait for it)
sleep 20
python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c1.json
python tools/gpu_text.py > gpurun_out/${tag}_text.txt 2>&1 < /dev/null
for w in c2 c4; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_${w}.json; done
sleep 20
python bench.py --workload c5 --warmup 2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c5.json
# kernel statistics + the launch rows (warm-up / timed / alone) of the same run
for w in c3 c5; do timeout 600 tools/kernel_stats.sh ${tag}_${w} --workload $w --steps 3 --warmup 2 > /dev/null 2>&1 < /dev/null; done
timeout 600 tools/kernel_stats.sh ${tag}_c1 --workload c1 --steps 20 --warmup 5 > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_fetch "FETCH_SIZE" --no-alone > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_write "WRITE_SIZE" --no-alone > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_sq "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" --no-alone > /dev/null 2>&1 < /dev/null
timeout 600 tools/pmc_pass.sh ${tag}_sq2 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE" --no-alone > /dev/null 2>&1 < /dev/null
# two ranks sharing the one GPU (gloo: RCCL refuses two ranks on one device): the N > 1 path of bench.py, weak and strong scaling
python bench.py --gpus 2 --backend gloo --single-device --gib 16 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_2ranks_1gpu_gloo.json
python bench.py --gpus 2 --backend gloo --single-device --scaling strong --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_2ranks_1gpu_gloo_strong.json
python bench.py --gpus 2 --backend gloo --single-device --workload c5 --gib 8 --warmup 2 --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_2ranks_1gpu_gloo_c5_8gib.json
python bench.py --gpus 8 --backend gloo --single-device --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_8ranks_1gpu_gloo_strong.json
tools/sort_reach.sh ${tag} > /dev/null 2>&1 < /dev/null
# SQ counters of the wave kernels
timeout 1500 tools/wave_pmc.sh ${tag} > /dev/null 2>&1 < /dev/null
ls -la gpurun_out/ | grep ${tag}
