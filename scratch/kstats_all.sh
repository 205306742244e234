# usage: kstats_all.sh TAG [bench args...] -> all kernel averages from rocprofv3 --stats
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && rm -rf $R/gpurun_out/ka_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ka_$TAG -o k -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --img-dtype bf16 "$@" > $R/gpurun_out/ka_$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ka_$TAG/**/k_kernel_stats.csv",recursive=True)[0]
print("== $TAG")
for r in csv.DictReader(open(f)):
    n=r["Name"].replace("ptx::","").replace("void ","")
    n=n[:n.find("(")] if "(" in n else n
    if n.startswith("k_"): print(f'   {n:28s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us')
PY
tail -1 $R/gpurun_out/ka_$TAG.log | cut -c1-200
