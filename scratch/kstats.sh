# usage: kstats.sh TAG [ENV=VAL ...]  -> prints image-kernel averages from rocprofv3 --stats
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && rm -rf $R/gpurun_out/ks_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$TAG -o k -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --img-dtype bf16 > $R/gpurun_out/ks_$TAG.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ks_$TAG/**/k_kernel_stats.csv",recursive=True)[0]
print("== $TAG $@")
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "img" in n and ("16" in n or "_bf" in n or "pool" in n):
        print(f'   {n[:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:7.2f} us')
PY
