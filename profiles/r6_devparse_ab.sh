# same-box A/B of the device parser's kernels: libraries named on the command line (default: the library before the column kernel learnt
# the 12-bit / whole-range classes and odd heights, libacm_amd/lib/exp/devp_old.so = profiles/build_rev.sh d40edb2 devp_old, against the
# working tree); kernel times from rocprofv3 --kernel-trace --stats over four device-parsed batches of the headline workload
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
LIBS=${*:-"libacm_amd/lib/exp/devp_old.so libacm_amd/lib/libacm_hip.so"}
for rep in 1 2; do
for lib in $LIBS; do
  which=$(basename $lib .so)
  D=gpurun_out/devp_${which}_$rep
  rm -rf $D
  export ACM_HIP_LIB=$PWD/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python3 profiles/dev_parse_trace.py 1 > gpurun_out/devp_${which}_$rep.log 2>&1
  echo "== $which rep $rep"; grep byteplane gpurun_out/devp_${which}_$rep.log | tail -2
  python3 - $D <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name']
        if any(k in n for k in ('acm_parse','acm_chunk','scatter')):
            print("   %-60s calls %5s avg_us %10.1f total_ms %9.2f" % (n[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
done
done
