#!/bin/bash
# SQ counter passes of the wave kernels (string-dense stage B): tools/wave_pmc.sh TAG  -> gpurun_out/pmc_TAG_{c1,c5}_{sq,sq2}.csv
# (separate --pmc runs, never together with tracing; C5 at 16 GiB with two warm-up buffers so that the timed one is scanned in pieces
# by the wave kernels without stage A)
tag=$1
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
timeout 900 tools/pmc_pass.sh ${tag}_c1_sq "$A" --workload c1 --gib 4 --warmup 1 > /dev/null 2>&1 < /dev/null
timeout 900 tools/pmc_pass.sh ${tag}_c1_sq2 "$B" --workload c1 --gib 4 --warmup 1 > /dev/null 2>&1 < /dev/null
timeout 1200 tools/pmc_pass.sh ${tag}_c5_sq "$A" --workload c5 --gib 16 --warmup 2 > /dev/null 2>&1 < /dev/null
timeout 1200 tools/pmc_pass.sh ${tag}_c5_sq2 "$B" --workload c5 --gib 16 --warmup 2 > /dev/null 2>&1 < /dev/null
grep -h "wave_\|scan_kernel" gpurun_out/pmc_${tag}_c?_sq*.csv | head -120
