# round 6: compiler scheduling strategies on the whole library (same sources): training call, forward, inverse; C2 and C3
cd /root/repo; O=gpurun_out/r06i; mkdir -p $O
{ echo "scripts/bench_kernels.py, ms; [] product (-O3 default scheduler), [_smi] -amdgpu-sched-strategy=max-ilp, [_sii] iterative-ilp, [_trk] -amdgpu-use-amdgpu-trackers, [_b0] -amdgpu-schedule-metric-bias=0"
  NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2 c3" ITERS=10 bash scripts/gpu_ab.sh "" _smi _sii _trk _b0; } > $O/sched_ab.txt 2>&1; cat $O/sched_ab.txt
