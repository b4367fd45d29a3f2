# round 5: issue priority of the net-split wave pairs (RNVP_NS_PRIO 0..3), A/B inside one box; stamps of the base and the balanced form;
# the strong-batch per-rank sizes (8192 rows) as they stand
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40
CFGS=c2 OPS=train bash scripts/gpu_ab.sh "" _p1 _p2 _p3 > $O/prio_ab.txt 2>&1
for nt in 8192 16384 32768; do echo "NT=$nt"; NT=$nt OPS=train python scripts/bench_kernels.py c2 c3 2>&1 | grep -E "^\{" | cut -c1-220; done > $O/rank_sizes.txt 2>&1
for v in _st _st3; do echo "variant $v"; ITERS=1 OPS=train RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/bench_kernels.py c2 2>&1 | grep STAMP | tail -12 | sort -k5n; done > $O/stamp_prio.txt 2>&1
cat $O/prio_ab.txt $O/rank_sizes.txt
