# HBM traffic of the dominant kernel from PMC counters, one counter per pass
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE in KB; on gfx950 FETCH_SIZE reports 1/2 of a
# wide coalesced read stream -> doubled below; WRITE_SIZE exact for 16-B streaming stores).
set -x
TAG=${1:-r01}
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --output-format csv -d /root/repo/gpurun_out/traffic_$TAG/$CNT -o p -- python3 /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/traffic_$TAG.$CNT.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/root/repo/gpurun_out/traffic_$TAG/*/*counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', row['Kernel_Name']); agg[(m.group(0) if m else row['Kernel_Name'][:40])][row['Counter_Name']].append(float(row['Counter_Value']))
out = {}
for k, d in agg.items():
    if not k.startswith('k_'): continue
    fetch_kb = sum(d.get('FETCH_SIZE', [0])) / max(1, len(d.get('FETCH_SIZE', [0])))
    write_kb = sum(d.get('WRITE_SIZE', [0])) / max(1, len(d.get('WRITE_SIZE', [0])))
    out[k] = {"launches": len(d.get('FETCH_SIZE', [])), "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb,
              "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
              "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is; separate --pmc passes"}
json.dump(out, open('/root/repo/gpurun_out/traffic_$TAG.json', 'w'), indent=1)
for k, v in out.items(): print(k[-50:], v["launches"], "fetch KB %.0f write KB %.0f -> %.1f MB/launch" % (v["FETCH_SIZE_KB_raw"], v["WRITE_SIZE_KB_raw"], v["hbm_bytes_per_launch"] / 1e6))
PY
