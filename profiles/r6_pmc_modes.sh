# instruction counts of acm_chunk by path: the plain kernel (profiles/ubench/k3_plain.bin) on synthetic chunk tables
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r6g${LEVEL:-}; mkdir -p $O
for cfg in ${CFGS:-"16 0 0" "16 100 0" "16 0 100" "16 56 0" "16 22 33" "4096 0 0" "4096 100 0" "4096 0 100" "2 100 0" "2 22 33"}; do
  set -- $cfg; tag=r$1_w$2_n$3
  ./profiles/ubench/k3_plain.bin ${LEVEL:-9} $1 $2 $3 > $O/$tag.txt 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/pmc_$tag -- ./profiles/ubench/k3_plain.bin ${LEVEL:-9} $1 $2 $3 > /dev/null 2> $O/pmc_$tag.err
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob("$O/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "acm_chunk" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
line=open("$O/$tag.txt").read().strip()
print(line)
print("   per chunk: " + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_",""), acc[k]/max(1,len(n[k]))/1048576) for k in sorted(acc)))
PY
done 2>&1 | tee $O/summary.txt
