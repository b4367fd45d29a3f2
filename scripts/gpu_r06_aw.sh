# round 6: wave number in an SGPR (readfirstlane), translation unit by translation unit, against the product ([] first, variant second)
cd /root/repo
L=/root/repo/probaforms_amd/csrc/librnvp_hip
run() { RNVP_HIP_LIB=$L$1.so "${@:2}" 2>/dev/null; }
for rep in 1 2; do
echo "== rnvp_lmm rep $rep"; for v in "" _rlmm; do echo "[$v]"; run "$v" python scripts/lmm64_time.py 128,128 | grep -v "family=auto" | cut -c1-150; run "$v" python scripts/lmm64_time.py 10,20,15 65536 8 2 0 | grep lmm16 | cut -c1-150; PREC= OPS=fwd,inv N=262144 run "$v" python scripts/bench_kernels.py 8,16,4,128 >/dev/null; done
echo "== rnvp_mfma (f32 flows) rep $rep"; for v in "" _rmfma; do echo -n "[$v] "; RNVP_HIP_LIB=$L$v.so PREC=f32 OPS=fwd,inv ITERS=30 WARM_S=0.1 python scripts/bench_kernels.py c2 8,16,4,64 2>/dev/null | grep -E "^\{" | python3 -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['config'], ' '.join('%s=%.4f' % (k, v) for k, v in j.items() if k.endswith('_ms')), end=' | ')
print()"; done
echo "== rnvp_resident rep $rep"; for v in "" _rres; do echo "[$v]"; run "$v" python scripts/resident_time.py | tail -6 | cut -c1-160; done
echo "== cvae_resident rep $rep"; for v in "" _rcres; do echo "[$v]"; run "$v" python scripts/cvae_resident_time.py | tail -4 | cut -c1-160; done
echo "== cvae_mfma rep $rep"; for v in "" _rcmf; do run "$v" python scripts/cvae_kernel_time.py | tail -1; done
done
