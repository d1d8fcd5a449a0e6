for cfg in "--level 6 --blocks 2000" "--level 7 --blocks 1000" "--level 8 --blocks 500" "--level 9 --blocks 250" "--level 10 --blocks 125" "--level 11 --rows 64 --blocks 16" "--level 12 --rows 16 --blocks 32" "--workload corpus"; do
  for so in base prio base prio; do
    v=$(ACM_HIP_LIB=libacm_amd/lib/exp/$so.so python3 bench.py $cfg --steps 100 --warmup 20 --no-extra --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['frac'])")
    echo "$cfg $so $v"
  done
done
