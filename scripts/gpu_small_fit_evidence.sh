# usage (GPU box): bash scripts/gpu_small_fit_evidence.sh -- the resident-fit evidence under profiles/: step times, soak, fuzz, API-level fits + their kernel stats
cd /root/repo
python scripts/resident_time.py > gpurun_out/r03_resident_time_new.txt 2>&1
python scripts/resident_soak.py > gpurun_out/r03_resident_soak_new.txt 2>&1
python scripts/resident_fuzz.py > gpurun_out/r03_resident_fuzz_new.txt 2>&1
python scripts/api_small_fit.py > gpurun_out/r03_api_small_fit_new.txt 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03_prof_api -o p -- python3 /root/repo/scripts/api_small_fit.py > /root/repo/gpurun_out/r03_prof_api.log 2>&1
cp $(find /root/repo/gpurun_out/r03_prof_api -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03_api_small_fit_kernel_stats_new.csv
cd /root/repo; for f in soak fuzz; do tail -n 3 gpurun_out/r03_resident_${f}_new.txt; done; tail -n 3 gpurun_out/r03_api_small_fit_new.txt
