# usage (GPU box): bash scripts/gpu_freeze.sh <tag>   -- everything profiles/ and DESIGN.md quote, from the sources as they are:
# GPU test suite, kernel stats for C1-C5 + the lmm shapes, traffic PMC, the C2 training kernel's PMC set (product kernel and the
# one-wave split-GEMM1 variant librnvp_hip_bxv.so when it has been built), flow-kernel PMC for C2 / C4, the microbenchmarks, the
# single-rank data-parallel A/B.  The bench line itself (which reads the traffic file) is a second gpurun call AFTER the results
# were copied into profiles/.
TAG=${1:-r06}
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/${TAG}_gpu_tests.txt; cat gpurun_out/${TAG}_gpu_tests.txt
bash scripts/gpu_micro.sh $TAG > gpurun_out/${TAG}_micro.log 2>&1 || echo "gpu_micro.sh FAILED (see gpurun_out/${TAG}_micro.log)"; tail -3 gpurun_out/${TAG}_micro_overlap.txt
python bench.py --no-cpu-baseline --no-api-level > gpurun_out/${TAG}_bench_plain.json 2> gpurun_out/${TAG}_bench_plain.err
bash scripts/gpu_profiles.sh $TAG > gpurun_out/${TAG}_profiles.log 2>&1; tail -25 gpurun_out/${TAG}_profiles.log
NT=65536 N=1048576 bash scripts/gpu_pmc.sh ${TAG}train c2 train > gpurun_out/${TAG}_train_pmc.log 2>&1
python scripts/make_train_pmc.py $TAG
if [ -f probaforms_amd/csrc/librnvp_hip_bxv.so ]; then
  RNVP_HIP_LIB=/root/repo/probaforms_amd/csrc/librnvp_hip_bxv.so NT=65536 N=1048576 bash scripts/gpu_pmc.sh ${TAG}trainbx c2 train > gpurun_out/${TAG}_trainbx_pmc.log 2>&1
  python scripts/make_train_pmc.py $TAG ${TAG}trainbx "k_mfma_train_bx<2, 1, 4, 0" ${TAG}_train_bx_pmc.json "C2, 65536 rows, split-GEMM1 forward + backward, one wave per SIMD, VGPR-form MFMA (librnvp_hip_bxv.so: -DRNVP_TRAIN_BX=1 -DRNVP_TRAIN_BX_NS_FIRST=0 -mllvm -amdgpu-mfma-vgpr-form), 256 workgroups x 4 waves"
  { echo "whole rnvp_loss_grad call, 65536 rows (scripts/bench_kernels.py), ms; [] = product library, [_bxv] = one-wave split-GEMM1 variant"; OPS=train bash scripts/gpu_ab.sh "" _bxv; } > gpurun_out/${TAG}_train_bx_ab.txt 2>&1
fi
N=1048576 bash scripts/gpu_pmc.sh ${TAG}flowc2 c2 fwd,inv > gpurun_out/${TAG}_flow_pmc_c2.log 2>&1
python scripts/make_train_pmc.py $TAG ${TAG}flowc2 "k_flow_bx3<2, 1" ${TAG}_flow_pmc_c2.json "C2 forward / inverse, 1M rows, barrier-free split-bf16 kernels (what precision auto runs for d <= 16 with hidden > 96 since round 6)"
N=1048576 bash scripts/gpu_pmc.sh ${TAG}flowc4 c4 fwd,inv > gpurun_out/${TAG}_flow_pmc_c4.log 2>&1
python scripts/make_train_pmc.py $TAG ${TAG}flowc4 "k_flow_bx3<8, 4" ${TAG}_flow_pmc_c4.json "C4 forward / inverse, 1M rows, bx3 kernels (LDS-staged weights)"
# single-rank data-parallel A/B: the same bench step through rnvp_fit_epoch (fused) and through rnvp_fit_epoch_dp on a one-rank RCCL communicator
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-level > gpurun_out/${TAG}_dp_a.json 2>/dev/null
BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-level > gpurun_out/${TAG}_dp_b.json 2>/dev/null
python - <<PY
import json
a = json.load(open("gpurun_out/${TAG}_dp_a.json")); b = json.load(open("gpurun_out/${TAG}_dp_b.json"))
out = {"command": "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-level, without and with BENCH_FORCE_DIST=1 (one rank, RCCL initialised: rnvp_fit_epoch_dp = loss+grad, ncclAllReduce on the same stream, Adam per batch)",
       "fused_single_gpu_ms_per_step": a["ms_per_step"], "data_parallel_one_rank_ms_per_step": b["ms_per_step"],
       "ratio": b["ms_per_step"] / a["ms_per_step"], "fused_roofline_frac": a["roofline"]["frac"], "dp_roofline_frac": b["roofline"]["frac"]}
json.dump(out, open("gpurun_out/${TAG}_dp_single_rank.json", "w"), indent=1); print(out)
PY
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_lmm -o p -- python3 /root/repo/scripts/lmm_profile.py 128,128 > /root/repo/gpurun_out/${TAG}_prof_lmm.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_lmm -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_lmm_h128x128_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/${TAG}_prof_cvae -o p -- python3 /root/repo/scripts/cvae_c5.py > /root/repo/gpurun_out/${TAG}_prof_cvae.log 2>&1
cp $(find /root/repo/gpurun_out/${TAG}_prof_cvae -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/${TAG}_cvae_c5_kernel_stats.csv
# round 4: the fit-epoch step per kernel (training kernel + ONE finish launch), the event-vs-rocprof reconciliation of the bench line,
# the reference-exact prior drawn on the device
cd /root/repo
bash scripts/gpu_step_profile.sh $TAG > gpurun_out/${TAG}_step_profile.log 2>&1; tail -12 gpurun_out/${TAG}_step_profile.log
python scripts/event_vs_rocprof.py $TAG
python scripts/api_sample_prior.py > gpurun_out/${TAG}_api_sample_prior.txt 2>&1; tail -8 gpurun_out/${TAG}_api_sample_prior.txt
# the any-shape training kernel on 64-row blocks: timings of the two forms on three nets, its counters, the shape fuzz
{ python scripts/lmm64_time.py 128,128; python scripts/lmm64_time.py 64,64; python scripts/lmm64_time.py 10,20,15 65536 8 2 0; python scripts/lmm64_time.py 128 65536 8 64 16; } 2>/dev/null | grep -v "family=auto" > gpurun_out/${TAG}_lmm64_time.txt; cat gpurun_out/${TAG}_lmm64_time.txt
bash scripts/gpu_pmc_any.sh ${TAG}lmm64 k_lmm_train64 /root/repo/scripts/lmm64_one.py > gpurun_out/${TAG}_lmm64_pmc.txt 2>&1; tail -28 gpurun_out/${TAG}_lmm64_pmc.txt
python scripts/multi_hidden_timing.py 2>/dev/null | tail -4 > gpurun_out/${TAG}_multi_hidden_timing.txt
cd /root/repo; ls -la gpurun_out/${TAG}_*kernel_stats.csv gpurun_out/${TAG}_*pmc*.json
