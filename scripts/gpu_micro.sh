# usage (GPU box, via gpurun): bash scripts/gpu_micro.sh <tag>  -- the microbenchmarks DESIGN.md leans on, output under gpurun_out/
TAG=${1:-r03}
cd /root/repo; mkdir -p gpurun_out
{
  echo "== scripts/micro/bf16_overlap (MI355X) =="; timeout 300 scripts/micro/bf16_overlap
  echo; echo "== scripts/micro/mfma_valu_overlap =="; timeout 300 scripts/micro/mfma_valu_overlap
  echo; echo "== scripts/micro/mfma4x4 =="; timeout 300 scripts/micro/mfma4x4
  echo; echo "== scripts/micro/unit_mix =="; timeout 300 scripts/micro/unit_mix
} > gpurun_out/${TAG}_micro_overlap.txt 2>&1
cat gpurun_out/${TAG}_micro_overlap.txt
