cd /root/repo; export TMPDIR=/tmp; OUT=/root/repo/gpurun_out
cd /tmp
for k in c3 c2; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r06_prof_dprank_$k -o p -- python3 /root/repo/scripts/dp_rank_profile.py $k > $OUT/r06_prof_dprank_$k.log 2>&1
cp $(find $OUT/r06_prof_dprank_$k -name "*kernel_stats.csv" | head -1) $OUT/r06_dprank_${k}_kernel_stats.csv
tail -1 $OUT/r06_prof_dprank_$k.log
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/r06_dprank_${k}_kernel_stats.csv")))[:10]: print(r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"])
PY
done
