SX_TIMELINE=1 python bench.py --no-cpu-baseline --steps 1 --warmup 2 > gpurun_out/r06e_tl.json 2> gpurun_out/r06e_tl.err
python - <<'PY'
lines=[l.rstrip() for l in open('gpurun_out/r06e_tl.err') if l.startswith('[tl')]
# the third scan_common (timed step)
idx=[i for i,l in enumerate(lines) if 'scan_common:' in l]
s=idx[2]; e=idx[3] if len(idx)>3 else len(lines)
print("\n".join(lines[s:e][:120]))
PY
