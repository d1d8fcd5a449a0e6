for cfg in "K2=0 --level 9 --blocks 250" "K2=0 --level 7 --blocks 1000" "K2=1 --level 5 --blocks 4000" "K2=1 --level 12 --rows 16 --blocks 32" "K2=1 --workload corpus" "K2=1 --level 9 --blocks 250"; do
  k2=${cfg%% *}; args=${cfg#* }
  for so in base prio2 base prio2; do
    v=$(ACM_K2=${k2#K2=} ACM_HIP_LIB=libacm_amd/lib/exp/$so.so python3 bench.py $args --steps 100 --warmup 20 --no-extra --no-cpu --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['frac'])")
    echo "$cfg $so $v"
  done
done
