#!/bin/bash
# Power and clocks the chip reads while the level-9 kernel / its timing-only ablations run (GPU box; needs
# libacm_amd/lib/exp/abl.so = profiles/build_variant.sh abl -DACM_ABLATION=1).  energy per launch = W x ms.
# usage: profiles/power_probe.sh [lib[@ENV=VAL] ...]
E=libacm_amd/lib/exp
LIBS=${@:-libacm_amd/lib/libacm_hip.so $E/abl.so@ACM_K2_ABL=17 $E/abl.so@ACM_K2_ABL=6 $E/abl.so@ACM_K2_ABL=8 $E/abl.so@ACM_K2_ABL=1 $E/abl.so@ACM_K2_ABL=16}
rocm-smi --showmaxpower 2>/dev/null | grep -i "power"
for lib in $LIBS; do
  tag=$(basename $lib | tr '@=' '__')
  python3 profiles/ab_kernels.py --rounds 20 --steps 600 --allow-wrong abl,x4,x4nowarm,ablnowarm $lib > gpurun_out/power_$tag.log 2>&1 &
  pid=$!
  sleep 11
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys,json
d=json.load(sys.stdin); c=list(d.values())[0]
s=[v for k,v in c.items() if 'sclk clock speed' in k][0]
p=[v for k,v in c.items() if 'ower' in k][0]
mhz=int(s.strip('()Mhz'))
if mhz>900: print('$tag sclk %d W %s'%(mhz,p))"
    sleep 0.7
  done
  tail -1 gpurun_out/power_$tag.log
done
