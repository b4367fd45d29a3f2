# usage (GPU box): bash scripts/gpu_freeze.sh <tag>   -- everything profiles/ and DESIGN.md quote, from the sources as they are:
# GPU test suite, kernel stats for C1-C5 + the lmm shapes, traffic PMC, the C2 training kernel's PMC set.  The bench line
# itself (which reads the traffic file) is a second gpurun call AFTER the results were copied into profiles/.
TAG=${1:-r02}
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt; cat gpurun_out/${TAG}_gpu_tests.txt
bash scripts/gpu_profiles.sh $TAG > gpurun_out/${TAG}_profiles.log 2>&1; tail -25 gpurun_out/${TAG}_profiles.log
NT=65536 N=1048576 bash scripts/gpu_pmc.sh ${TAG}train c2 train > gpurun_out/${TAG}_train_pmc.log 2>&1
python scripts/make_train_pmc.py $TAG
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_lmm -o p -- python3 /root/repo/scripts/lmm_profile.py 128,128 > /root/repo/gpurun_out/${TAG}_prof_lmm.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_lmm -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_lmm_h128x128_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_cvae -o p -- python3 /root/repo/scripts/cvae_c5.py > /root/repo/gpurun_out/${TAG}_prof_cvae.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_cvae -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_cvae_c5_kernel_stats.csv
cd /root/repo; ls -la gpurun_out/${TAG}_*kernel_stats.csv gpurun_out/${TAG}_*pmc*.json
