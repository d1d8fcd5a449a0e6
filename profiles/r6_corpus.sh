cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r6j
for rep in 1 2; do
  for t in r5 r6; do
    if [ $t = r5 ]; then d=.r5tree; else d=.; fi
    ( cd $d && timeout 600 python3 bench.py --workload corpus --no-extra --no-cpu --steps 50 --warmup 10 2>/dev/null | tail -1 ) > gpurun_out/r6j/${t}_$rep.json
    python3 -c "
import json
j=json.load(open('gpurun_out/r6j/${t}_$rep.json'))
print('$t rep $rep: ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'verified', j['verified_streams'])"
  done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6j/trace -- python3 bench.py --workload corpus --no-extra --no-cpu --steps 50 --warmup 10 > gpurun_out/r6j/corpus.json 2> gpurun_out/r6j/corpus.err
python3 - <<PY
import json,glob,csv
for f in glob.glob("gpurun_out/r6j/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %6s avg_us %10.1f pct %s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
