#!/bin/bash
# Start-up cost of a one-file decode next to a do-nothing HIP program (GPU box): runtime init and exit are ~0.3 s of every process.
python3 - <<'PY'
import sys; sys.path.insert(0, '.')
from libacm_amd import synth
open('/tmp/c1_20k.acm', 'wb').write(synth.generate(seed=synth.BASE_SEED, level=7, rows=16, nblocks=20000))
open('/tmp/c1_50.acm', 'wb').write(synth.generate(seed=synth.BASE_SEED, level=7, rows=16, nblocks=50))
PY
python3 profiles/init_probe.py 2>&1 | tail -3
for f in c1_50 c1_20k; do
  for i in 1 2 3; do s=$(date +%s.%N); libacm_amd/bin/acmtool -d -n -q /tmp/$f.acm; e=$(date +%s.%N); python3 -c "print('$f %.3f' % ($e-$s))"; done
done
# how long does a trivial HIP program take (init + exit)?
cat > /tmp/hipnull.cpp <<'CPP'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
__global__ void k(int *p){ if (p) *p = 1; }
int main(){ double t0=now(); int n=0; hipGetDeviceCount(&n); double t1=now(); hipSetDevice(0); hipStream_t s; hipStreamCreate(&s); double t2=now(); int *d; hipMalloc(&d,4); double t3=now(); hipLaunchKernelGGL(k,dim3(1),dim3(64),0,s,d); hipStreamSynchronize(s); double t4=now();
 printf("count %.3f stream %.3f malloc %.3f launch %.3f\n", t1-t0,t2-t1,t3-t2,t4-t3); return 0; }
CPP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/hipnull /tmp/hipnull.cpp 2>/dev/null
for i in 1 2; do s=$(date +%s.%N); /tmp/hipnull; e=$(date +%s.%N); python3 -c "print('hipnull total %.3f' % ($e-$s))"; done
