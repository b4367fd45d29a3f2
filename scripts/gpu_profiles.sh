# usage (GPU box, via gpurun): bash scripts/gpu_profiles.sh <tag>
# rocprofv3 evidence for every config, written under gpurun_out/<tag>_*; copy what is judged into profiles/.
#   1. --kernel-trace --stats: bench.py (C2 step), scripts/bench_kernels.py c2 c3 c4 (1M-row forward / inverse,
#      65536-row training call), scripts/readme_example.py (C1)
#   2. HBM traffic, --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md "HBM":
#      FETCH_SIZE x2 on gfx950), for bench.py and for the c3 / c4 kernels
# The program itself follows `--` (python3 ...): no shell or env hop under the profiler.
TAG=${1:-r06}
cd /root/repo; mkdir -p gpurun_out; OUT=/root/repo/gpurun_out
export TMPDIR=/tmp
cd /tmp
stats() {  # name, then the command
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_$name -o p -- "$@" > $OUT/${TAG}_prof_$name.log 2>&1
  cp $(find $OUT/${TAG}_prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${name}_kernel_stats.csv
  head -8 $OUT/${TAG}_${name}_kernel_stats.csv | cut -c1-200
}
# the driver's own command (default steps / warmup), minus the CPU legs that launch no kernel; its JSON line (HIP events taken UNDER
# the profiler) lands in ${TAG}_prof_bench.log and is compared with the CSV by scripts/event_vs_rocprof.py
stats bench python3 /root/repo/bench.py --no-cpu-baseline --no-api-level
export N=1048576 NT=65536
stats kernels_c2 python3 /root/repo/scripts/bench_kernels.py c2
stats kernels_c3 python3 /root/repo/scripts/bench_kernels.py c3
stats kernels_c4 python3 /root/repo/scripts/bench_kernels.py c4
EPOCHS=5 stats readme_c1 python3 /root/repo/scripts/readme_example.py
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_traffic/bench_$CNT -o p -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api-level > $OUT/${TAG}_traffic.bench_$CNT.log 2>&1
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/${TAG}_traffic/kern_$CNT -o p -- python3 /root/repo/scripts/bench_kernels.py c3 c4 > $OUT/${TAG}_traffic.kern_$CNT.log 2>&1
done
cd /root/repo
python3 - <<PY
import csv, glob, json, collections, re, sys
sys.path.insert(0, '/root/repo')
import bench
def collect(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for row in csv.DictReader(open(f)):
            m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', row['Kernel_Name'])
            agg[(m.group(0) if m else row['Kernel_Name'][:40])][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {}
    for k, d in agg.items():
        if not k.startswith('k_'): continue
        f_ = d.get('FETCH_SIZE', []); w_ = d.get('WRITE_SIZE', [])
        fetch_kb = sum(f_) / max(1, len(f_)); write_kb = sum(w_) / max(1, len(w_))
        out[k] = {"launches": len(f_), "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
                  "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0}
    return out
note = "per launch, averaged over the launches of the run; FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is; separate --pmc passes (scripts/gpu_profiles.sh)"
b = {"csrc_hash": bench.csrc_hash(), "command": "python3 bench.py --steps 2 --warmup 1 (C2 step: 16 training launches of which one ragged + 1 sampling launch per step)", "note": note,
     "kernels": collect('$OUT/${TAG}_traffic/bench_*/*counter_collection.csv')}
json.dump(b, open('$OUT/${TAG}_traffic_pmc.json', 'w'), indent=1)
k = {"csrc_hash": bench.csrc_hash(), "command": "N=1048576 NT=65536 python3 scripts/bench_kernels.py c3 c4 (forward / inverse on 1M rows, training call on 65536 rows; both configs share kernel template names where their tile geometry coincides -- none does here: c3 = <4,2,..>, c4 = <8,4,..>)", "note": note,
     "kernels": collect('$OUT/${TAG}_traffic/kern_*/*counter_collection.csv')}
json.dump(k, open('$OUT/${TAG}_traffic_pmc_c3c4.json', 'w'), indent=1)
for name, d in (("bench", b), ("c3c4", k)):
    for kk, v in d["kernels"].items(): print(name, kk[-46:], v["launches"], "%.1f MB/launch" % (v["hbm_bytes_per_launch"] / 1e6))
PY
