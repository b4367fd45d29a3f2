# usage (GPU box, via gpurun): bash scripts/gpu_micro.sh <tag>  -- the microbenchmarks DESIGN.md leans on, output under gpurun_out/
# Each scripts/micro/*.hip is built here (hipcc, gfx950) before it runs; a failed build or run fails the script loudly.
TAG=${1:-r06}
cd /root/repo; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_micro_overlap.txt
: > $OUT
rc=0
for b in bf16_overlap mfma_valu_overlap mfma4x4 unit_mix finish_read; do
  if ! /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value -o scripts/micro/$b scripts/micro/$b.hip >> $OUT 2>&1; then
    echo "BUILD FAILED: scripts/micro/$b.hip" | tee -a $OUT; rc=1; continue
  fi
  echo "== scripts/micro/$b (MI355X) ==" >> $OUT
  if ! timeout 300 scripts/micro/$b >> $OUT 2>&1; then echo "RUN FAILED: scripts/micro/$b" | tee -a $OUT; rc=1; fi
  echo >> $OUT
done
cat $OUT
exit $rc
