#!/bin/bash
# The device-parse batch (profiles/e2e_ranges_probe.py, 16 ranges) with the process confined to the CPUs of each NUMA node in
# turn, and unconfined: does the read-back rate depend on where the pool threads and the pinned arenas land?  (GPU box)
export E2E_RANGES=16
for node in none 0 1; do
  if [ "$node" = none ]; then pre=""; else pre="taskset -c $(cat /sys/devices/system/node/node$node/cpulist)"; fi
  echo "== cpus of node: $node"
  $pre python3 profiles/e2e_ranges_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
done
