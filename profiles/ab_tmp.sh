for cfg in "--level 10 --blocks 125" "--level 11 --rows 64 --blocks 16" "--level 12 --rows 16 --blocks 32" "--level 13 --rows 16 --blocks 16"; do
  for so in base sub2 base sub2; do
    v=$(ACM_HIP_LIB=libacm_amd/lib/exp/$so.so python3 bench.py $cfg --steps 60 --warmup 10 --no-extra --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['frac'], d['verified_streams'])")
    echo "$cfg $so $v"
  done
done
