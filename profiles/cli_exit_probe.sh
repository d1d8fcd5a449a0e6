# Where the wall clock of `acmtool -B` goes outside the batch itself (start-up before the first trace line, exit behind the last
# one) by the number of groups parsed ahead; ACMTOOL_DETACH=1 shows the detached exit.  usage: bash profiles/cli_exit_probe.sh   (GPU box)
N=4000
D=/dev/shm/acm_cli_probe2
rm -rf $D; mkdir -p $D/in
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
for ahead in 3 4 6 10 3 4 6 10; do
  rm -f $D/in/*.raw
  s=$(date +%s.%N); ACMTOOL_GROUPS_AHEAD=$ahead ACMTOOL_BATCH_TRACE=1 libacm_amd/bin/acmtool -d -q -B -r $D/in/*.acm > /dev/null 2> $D/trace; e=$(date +%s.%N)
  x=$(grep "wall clock at exit" $D/trace | sed 's/.*exit \([0-9.]*\), the batch started \([0-9.]*\).*/\1 \2/')
  python3 -c "
x='$x'.split(); ex=float(x[0]); dur=float(x[1])
print('ahead $ahead: wall %.3f s = %.3f before the batch starts + %.3f batch + %.3f after its last line' % ($e-$s, ex-dur-$s, dur, $e-ex))"
done
rm -rf $D
