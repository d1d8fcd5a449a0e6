# The headline launch slows over the first half minute of back-to-back processes on a box (r6_level9_notes.txt section 19): what does the SMU say meanwhile?
# twelve short bench processes, a rocm-smi snapshot (temperatures, clocks, power) right after each.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
OUT=gpurun_out/r6_drift.txt
: > $OUT
snap() { rocm-smi --showtemp --showclocks --showpower 2>/dev/null | grep -E "Temperature|sclk|mclk|fclk|socclk|Power" | sed 's/^=*//' | tr -s ' ' | tr '\n' ';' >> $OUT; echo >> $OUT; }
echo "idle:" >> $OUT; snap
for rep in $(seq 1 12); do
  python3 bench.py --no-extra --no-cpu --steps 50 --warmup 10 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=j['roofline']
print('process $rep: launch_ms %.4f frac %.4f' % (r['launch_ms'], r['frac']))" >> $OUT
  snap
done
cat $OUT
