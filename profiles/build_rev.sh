#!/bin/bash
# The WHOLE library of an earlier commit as libacm_amd/lib/exp/<name>.so (for same-box A/B against the working tree:
# profiles/ab_kernels.py --own-form ...): profiles/build_rev.sh <git-rev> <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
REV=$1; NAME=$2; shift 2
T=$(mktemp -d)
git archive "$REV" libacm_amd/csrc include | tar -x -C "$T"
mkdir -p libacm_amd/lib/exp
OBJS=""
for f in acm_kernels.hip acm_parse.hip acm_hip_api.cpp acm_fill.cpp acm_pack.cpp acm_stream.cpp acm_batch.cpp acm_host_synth.cpp; do
  [ -f "$T/libacm_amd/csrc/$f" ] || continue
  X=""; case $f in *.cpp) X="-x hip";; esac
  hipcc -O3 -g1 -std=c++17 -fPIC --offload-arch=gfx950 -I "$T/include" -I "$T/libacm_amd/csrc" "$@" $X -c "$T/libacm_amd/csrc/$f" -o "$T/$f.o" &
  OBJS="$OBJS $T/$f.o"
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o libacm_amd/lib/exp/$NAME.so $OBJS -lpthread
rm -rf "$T"
ls -la libacm_amd/lib/exp/$NAME.so
