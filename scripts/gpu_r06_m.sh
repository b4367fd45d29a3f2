# round 6: more machine-scheduler knobs on the whole library; C2 (the headline kernels) and C3
cd /root/repo; O=gpurun_out/r06m; mkdir -p $O
{ echo "scripts/bench_kernels.py, ms; [] product; [_nur] -amdgpu-disable-unclustered-high-rp-reschedule; [_ncl] -amdgpu-disable-clustered-low-occupancy-reschedule; [_td] -misched-prera-direction=topdown; [_bu] =bottomup; [_npm] -enable-post-misched=false; [_b100] -amdgpu-schedule-metric-bias=100"
  NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2 c3" ITERS=10 bash scripts/gpu_ab.sh "" _nur _ncl _td _bu _npm _b100; } > $O/sched_ab3.txt 2>&1; cat $O/sched_ab3.txt
