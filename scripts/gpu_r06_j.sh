# round 6: more scheduling strategies; C4 for iterative-ilp
cd /root/repo; O=gpurun_out/r06j; mkdir -p $O
{ echo "scripts/bench_kernels.py, ms; [] product, [_sii] -amdgpu-sched-strategy=iterative-ilp, [_smr] iterative-minreg, [_smo] iterative-maxocc, [_smc] max-memory-clause"
  NT=65536 N=1048576 OPS=train,fwd,inv CFGS="c2 c3 c4" ITERS=10 bash scripts/gpu_ab.sh "" _sii _smr _smo _smc; } > $O/sched_ab2.txt 2>&1; cat $O/sched_ab2.txt
