# round 5: the per-rank batch of the 8-GPU configuration (8192 rows of C3) on the tile-split kernel with two row tiles per
# workgroup, against the row-parallel launch it replaces; parity of the new dispatch
cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export ITERS=40 OPS=train
for v in "" _old; do for nt in 4096 6000 8192 12288 16384; do echo "variant [$v] NT=$nt"; NT=$nt RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so python scripts/bench_kernels.py c3 2>&1 | grep -E "^\{|rror" | cut -c1-200; done; done > $O/c3_rank_ts.txt 2>&1
python -m pytest tests/test_bench_sizes_gpu.py -x -q -k "c3-8192 or c3-5000 or 2120 or c2-8192" -s 2>&1 | tail -15 >> $O/c3_rank_ts.txt
cat $O/c3_rank_ts.txt
