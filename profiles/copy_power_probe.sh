#!/bin/bash
# Socket power and shader clock while a pure copy / read / write kernel streams (GPU box): profiles/copy_power_probe.sh
for pat in 0 1 2 3 4; do
  ./profiles/ubench/copy_bw.bin loop $pat 7 > gpurun_out/copy_power_$pat.log 2>&1 &
  pid=$!
  sleep 2.5
  for i in 1 2 3 4 5; do
    rocm-smi --showpower --showclocks --json 2>/dev/null | python3 -c "
import sys,json
c=list(json.load(sys.stdin).values())[0]
s=[v for k,v in c.items() if 'sclk clock speed' in k][0]; p=[v for k,v in c.items() if 'ower' in k][0]
print('pattern $pat sclk', s.strip('()Mhz'), 'W', p)"
    sleep 0.6
  done
  wait $pid
  cat gpurun_out/copy_power_$pat.log
done
