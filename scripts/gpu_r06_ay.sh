# round 6, final library: soaks (bit-identical repeats) of the barrier-free flush, the 64-row any-shape kernel and the resident epochs; box probe
cd /root/repo
bash scripts/box_probe.sh
python scripts/tflush_soak.py 1000 2>&1 | tail -4
python scripts/lmm64_soak.py 2>&1 | tail -2
python scripts/resident_soak.py 2>&1 | tail -2
FUZZ_WIDE=1 python scripts/autograd_fuzz.py 2>&1 | tail -1
