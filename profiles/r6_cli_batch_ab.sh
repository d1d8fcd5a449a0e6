# acmtool -B of the round-5 tree (.r5tree) against this tree's on the same files, same box
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-1000}
D=/dev/shm/acm_cli_probe
rm -rf $D; mkdir -p $D/in $D/ours
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
with ThreadPoolExecutor(16) as ex:
    list(ex.map(one, enumerate(shapes)))
PY
cp $D/in/*.acm $D/ours/
for rep in 1 2 3; do
  for t in r5 r6; do
    if [ $t = r5 ]; then tool=.r5tree/libacm_amd/bin/acmtool; else tool=libacm_amd/bin/acmtool; fi
    rm -f $D/ours/*.raw
    s=$(date +%s.%N); ACMTOOL_BATCH_TRACE=${TRACE:-} $tool -d -q -B -r $D/ours/*.acm > /dev/null 2> $D/trace.$t.$rep; e=$(date +%s.%N)
    python3 -c "print('$t acmtool -B $N files (run $rep): %.3f s' % ($e-$s))"
    [ -n "${TRACE:-}" ] && [ $rep = 3 ] && tail -25 $D/trace.$t.$rep
  done
done
rm -rf $D
