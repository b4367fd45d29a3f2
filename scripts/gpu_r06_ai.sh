# freeze + bench lines in ONE call: the traffic files the bench line reads are copied into profiles/ on the box between the two
cd /root/repo
bash scripts/gpu_freeze.sh r06 2>&1 | tail -60
cp gpurun_out/r06_traffic_pmc.json gpurun_out/r06_traffic_pmc_c3c4.json profiles/
bash scripts/gpu_r06_final.sh
