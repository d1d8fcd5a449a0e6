#!/bin/bash
# Experimental build of the kernels into libacm_amd/lib/exp/<name>.so (travels to the GPU box; *.so is git-ignored):
#   profiles/build_variant.sh <name> [-DFLAG ...]          (the other objects are the regular build's)
# A/B on the box: python3 profiles/ab_kernels.py libacm_amd/lib/libacm_hip.so libacm_amd/lib/exp/<name>.so ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=${ACM_KERNEL_SRC:-libacm_amd/csrc/acm_kernels.hip}
python3 -c "from libacm_amd import _build; _build.build_hip()"
mkdir -p libacm_amd/lib/exp
hipcc -O3 -g1 -std=c++17 -fPIC -Wall -Wextra --offload-arch=gfx950 -I include -I libacm_amd/csrc "$@" -c $SRC -o libacm_amd/lib/exp/$NAME.kernels.o
hipcc -shared -fPIC --offload-arch=gfx950 -o libacm_amd/lib/exp/$NAME.so libacm_amd/lib/exp/$NAME.kernels.o \
  libacm_amd/lib/acm_parse.hip.o libacm_amd/lib/acm_hip_api.cpp.o libacm_amd/lib/acm_fill.cpp.o libacm_amd/lib/acm_pack.cpp.o libacm_amd/lib/acm_stream.cpp.o libacm_amd/lib/acm_batch.cpp.o libacm_amd/lib/acm_host_synth.cpp.o -lpthread
rm -f libacm_amd/lib/exp/$NAME.kernels.o
ls -la libacm_amd/lib/exp/$NAME.so
