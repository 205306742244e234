#!/bin/bash
# copy what profiles/ keeps out of gpurun_out/ (run here, after scratch/r05_final.sh on the GPU box)
cd /root/repo; O=gpurun_out/r05; F=gpurun_out/r05_final; P=profiles
cp $O/bench.json $P/r05_final_bench.json; cp $O/bench_f32.json $P/r05_final_bench_f32.json
cp $O/r05_kernel_stats_bf16.csv $P/r05_final_kernel_stats_bf16.csv; cp $O/r05_kernel_stats_f32.csv $P/r05_final_kernel_stats_f32.csv
cp $O/r05_pmc_traffic.json $P/r05_pmc_traffic.json
cp $O/r05_timeline_bf16.txt $P/r05_final_timeline_bf16.txt; cp $O/r05_timeline_f32.txt $P/r05_final_timeline_f32.txt
for n in b1 b32 cfg1 cfg4 cfg4_b6 cfg4_room_b6 cfg5 cfg5_b16 driver_args; do cp $F/bench_$n.json $P/r05_final_bench_$n.json; done
cp $F/r05_timeline_cfg4_b6.txt $P/r05_final_timeline_cfg4_b6.txt
python - "$(find $F/stats_cfg4_b6 -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open("profiles/r05_final_kernel_stats_cfg4_b6.csv", "w") as out:
    out.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
    for r in rows[:24]:
        n = r["Name"].replace("ptx::", "").split("(")[0]
        out.write('"%s",%s,%.1f,%.2f,%.2f,%.2f,%s\n' % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                                                        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
grep -v amdgpu.ids $F/vox_time.txt > $P/r05_final_vox_time.txt; grep -v amdgpu.ids $F/ingest_time.txt > $P/r05_final_ingest_time.txt
ls -la $P | grep r05_final | wc -l
