#!/bin/bash
# configs[2]-like corpus through the CLI: reference tool one file at a time (all host cores via xargs -P) vs `acmtool -B`.
# usage: profiles/cli_batch_probe.sh [files]     (GPU box; files land in /dev/shm so that no disk is measured)
N=${1:-400}
D=/dev/shm/acm_cli_probe
rm -rf $D; mkdir -p $D/in $D/ref $D/ours
python3 - $N $D/in <<'PY'
import sys; sys.path.insert(0, '.')
from concurrent.futures import ThreadPoolExecutor
from libacm_amd import synth, workload
n, out = int(sys.argv[1]), sys.argv[2]
shapes = workload.corpus_shapes(n)
def one(a):
    i, s = a
    kw = dict(seed=synth.BASE_SEED + 31000 + i, level=s["level"], rows=s["rows"], nblocks=s["nblocks"], channels=s["channels"], total_values=s["total_values"])
    open("%s/f%05d.acm" % (out, i), "wb").write(synth.generate(**kw))
    return s["total_values"]
with ThreadPoolExecutor(16) as ex:
    tot = sum(ex.map(one, enumerate(shapes)))
open(out + "/../total", "w").write(str(tot))
print("corpus: %d files, %.1f Msamples" % (n, tot / 1e6))
PY
TOT=$(cat $D/total)
NCPU=$(python3 -c "import sys; sys.path.insert(0, '.'); from libacm_amd import workload; print(workload.usable_cpus())")     # affinity mask capped by the cgroup quota
if [ -x oracle/_ref/acmtool_ref ]; then
  cp $D/in/*.acm $D/ref/
  s=$(date +%s.%N); ls $D/ref/*.acm | xargs -P $NCPU -n 8 oracle/_ref/acmtool_ref -d -q -r >/dev/null 2>&1; e=$(date +%s.%N)
  python3 -c "print('reference acmtool, xargs -P %d (all usable CPUs): %.2f s  %.1f Msamples/s' % ($NCPU, $e-$s, $TOT/($e-$s)/1e6))"
fi
cp $D/in/*.acm $D/ours/
for rep in 1 2; do
  rm -f $D/ours/*.raw
  s=$(date +%s.%N); libacm_amd/bin/acmtool -d -q -B -r $D/ours/*.acm > /dev/null 2> $D/trace.$rep; e=$(date +%s.%N)
  [ -n "${ACMTOOL_BATCH_TRACE:-}" ] && cat $D/trace.$rep
  python3 -c "print('acmtool -B (run $rep): %.2f s  %.1f Msamples/s' % ($e-$s, $TOT/($e-$s)/1e6))"
done
if [ -d $D/ref ]; then
  bad=0; for f in $D/ref/*.raw; do cmp -s $f $D/ours/$(basename $f) || bad=$((bad+1)); done; echo "files that differ from the reference tool's: $bad"
fi
rm -rf $D
