# round 6: does -amdgpu-mfma-vgpr-form (or -O2) move the C2 training call?  six repetitions each, alternating
cd /root/repo; O=gpurun_out/r06r; mkdir -p $O
{ echo "rnvp_loss_grad C2, 65536 rows, ms (ITERS=30); [] product, [_vf] whole library -mllvm -amdgpu-mfma-vgpr-form, [_o2] -O2"
for rep in 1 2 3 4 5 6; do for v in "" _vf _o2; do printf "[%-3s] " "$v"; RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip$v.so NT=65536 OPS=train ITERS=30 python scripts/bench_kernels.py c2 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin: print('%.4f' % json.loads(l)['train_ms'])"; done; done; } > $O/c2_vf.txt 2>&1; cat $O/c2_vf.txt
