cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r6_bench_default_c.json 2> gpurun_out/r6_bench_default_c.err
tail -c 600 gpurun_out/r6_bench_default_c.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6_bench_default_c.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('end_to_end',{}).items():
    if isinstance(v,dict): print(k, {x:v[x] for x in ('msamples_s','parse_s','kernel_s','total_s','h2d_bytes','packed_streams') if x in v})
PY
