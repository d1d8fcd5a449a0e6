#!/bin/bash
# BASELINE configs[0]: one mono 22 050 Hz level-7 file through `acmtool -d -n` - reference tool (CPU) vs ours (GPU windows)
set -e
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from libacm_amd import synth
open('/tmp/c1_1k.acm', 'wb').write(synth.generate(seed=synth.BASE_SEED, level=7, rows=16, nblocks=1000))
open('/tmp/c1_20k.acm', 'wb').write(synth.generate(seed=synth.BASE_SEED, level=7, rows=16, nblocks=20000))
open('/tmp/c1_l9.acm', 'wb').write(synth.generate(seed=synth.BASE_SEED, level=9, rows=16, nblocks=5000))
PY
for f in c1_1k c1_20k c1_l9; do
  for tool in oracle/_ref/acmtool_ref libacm_amd/bin/acmtool; do
    [ -x $tool ] || continue
    $tool -d -n -q /tmp/$f.acm >/dev/null 2>&1   # warm
    s=$(date +%s.%N); $tool -d -n -q /tmp/$f.acm; e=$(date +%s.%N)
    python3 -c "import os;n={'c1_1k':2.048e6,'c1_20k':40.96e6,'c1_l9':40.96e6}['$f'];dt=$e-$s;print('%-28s %-8s %.3f s  %.1f Msamples/s' % ('$tool','$f',dt,n/dt/1e6))"
  done
done
$PWD/libacm_amd/bin/acmtool -d -r -o /tmp/ours.raw /tmp/c1_1k.acm > /dev/null; oracle/_ref/acmtool_ref -d -r -o /tmp/ref.raw /tmp/c1_1k.acm > /dev/null; cmp /tmp/ours.raw /tmp/ref.raw && echo "PCM identical (config 1 file)"
