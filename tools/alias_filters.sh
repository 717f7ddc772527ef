#!/bin/bash
# What an alias filter costs on the headline image (C3's three Missions with another -u): tools/alias_filters.sh TAG [FILTERS...]
# -> gpurun_out/TAG_bench_c3_ubf_<filter>{,_generic}.json, TAG_ubf_cjk_kernel_stats.csv, pmc_TAG_ubf_cjk_sq.csv
# (not Asian: two thirds of all UTF-16 units pass it, random bytes are then one string from end to end)
tag=$1; shift
filters=${@:-Cjk Hangul Kana All}
for f in $filters; do
  python bench.py --ubf $f --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c3_ubf_${f}.json
  python bench.py --ubf $f --no-cpu-baseline --generic-kernels 2>/dev/null < /dev/null | tail -1 > gpurun_out/${tag}_bench_c3_ubf_${f}_generic.json
  python - gpurun_out/${tag}_bench_c3_ubf_${f}.json gpurun_out/${tag}_bench_c3_ubf_${f}_generic.json <<'PY'
import json, sys
for p in sys.argv[1:]:
    try:
        j = json.loads(open(p).read()); r = j["roofline"]
        print(p.split("/")[-1], "GiB/s", j["value"], "ms/step", j["ms_per_step"], "frac", r["frac"], "per_kernel_ms", r["per_kernel_ms"], "alone", r["per_kernel_ms_alone"])
    except Exception as e:
        print(p, "unreadable:", e)
PY
done
case " $filters " in *" Cjk "*) ;; *) exit 0;; esac
timeout 600 tools/kernel_stats.sh ${tag}_ubf_cjk --ubf Cjk --steps 3 --warmup 1 > /dev/null 2>&1 < /dev/null
head -5 gpurun_out/${tag}_ubf_cjk_kernel_stats.csv | cut -c1-200
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout 600 tools/pmc_pass.sh ${tag}_ubf_cjk_sq "$A" --ubf Cjk --gib 16 > /dev/null 2>&1 < /dev/null
grep scan_kernel gpurun_out/pmc_${tag}_ubf_cjk_sq.csv
