# developer build: librnvp_hip_stamp64.so = the product objects with rnvp_lmm64.hip compiled under -DRNVP_STAMP=1 (cycle shares per
# kind of work, printed by workgroup 0):  RNVP_HIP_LIB=.../librnvp_hip_stamp64.so python scripts/lmm64_one.py
cd /root/repo/probaforms_amd/csrc && make -j8 > /dev/null && \
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -DRNVP_STAMP=1 -c rnvp_lmm64.hip -o /tmp/rnvp_lmm64_stamp.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librnvp_hip_stamp64.so $(ls *.o | grep -v "_bxv\|lmm64") /tmp/rnvp_lmm64_stamp.o -ldl && ls -la librnvp_hip_stamp64.so
