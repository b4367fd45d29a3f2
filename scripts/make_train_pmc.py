"""profiles/<tag>_train_pmc.json from the counter CSVs that `NT=65536 N=1048576 bash scripts/gpu_pmc.sh <tag>train c2 train`
left under gpurun_out/pmc_<tag>train/ (five separate --pmc passes): per-launch averages for the C2 training kernel and
the derived per-SIMD / per-wave figures DESIGN.md section 5 quotes.   usage: python scripts/make_train_pmc.py <tag>"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# optional: directory tag of the counter CSVs, kernel-name substring, output name, free-text description
dir_tag = sys.argv[2] if len(sys.argv) > 2 else tag + "train"
kmatch = sys.argv[3] if len(sys.argv) > 3 else "k_mfma_train<2, 1, 4, 1, 0"
out_name = sys.argv[4] if len(sys.argv) > 4 else "%s_train_pmc.json" % tag
descr = sys.argv[5] if len(sys.argv) > 5 else "C2, 65536 rows, net-split, 256 workgroups x 8 waves"
vals = collections.defaultdict(list)
durs = []
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s" % dir_tag, "s*", "*counter_collection.csv")):
    seen = set()
    for row in csv.DictReader(open(f)):
        if kmatch not in row["Kernel_Name"]:
            continue
        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = row.get("Dispatch_Id")
        if key not in seen and row.get("End_Timestamp") and row.get("Start_Timestamp"):
            seen.add(key)
            durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
assert vals, "no counter rows for %s under gpurun_out/pmc_%s" % (kmatch, dir_tag)
per = {k: sum(v) / len(v) for k, v in vals.items()}
waves = per["SQ_WAVES"]
simds = 1024.0
wave_cycles = per["SQ_WAVE_CYCLES"] * 4.0                    # the counter ticks once per 4 cycles
cycles_per_wave = wave_cycles / waves
dur_us = sum(durs) / len(durs) if durs else None
out = {
    "kernel": "%s...> (%s)" % (kmatch, descr),
    "csrc_hash": bench.csrc_hash(),
    "command": "NT=65536 N=1048576 bash scripts/gpu_pmc.sh %s <config> <ops>  (rocprofv3 --pmc, 5 separate passes over scripts/bench_kernels.py); scripts/make_train_pmc.py %s" % (dir_tag, " ".join(sys.argv[1:])),
    "per_launch": per,
    "avg_duration_us_under_pmc": dur_us,
    "derived": {
        "cycles_per_wave": cycles_per_wave,
        "wave_lifetime_over_duration_GHz (a lower bound of the clock; meaningless when waves run in several rounds)": (cycles_per_wave / (dur_us * 1e3)) if dur_us else None,
        "mfma_busy_cycles_per_simd": per["SQ_VALU_MFMA_BUSY_CYCLES"] / simds,
        "mfma_busy_frac_of_wave_lifetime": per["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cycles_per_wave,
        "valu_active_cycles_per_simd (quad-cycles x4, includes MFMA issue)": per["SQ_ACTIVE_INST_VALU"] * 4.0 / simds,
        "useful_mfma_cycles_per_simd (245760 flop/row x 65536 rows / 1024 SIMDs / 64 flop/clk)": 245760.0 * 65536 / simds / 64.0,
        "mfma_instructions_per_wave": per["SQ_INSTS_MFMA"] / waves,
        "non_mfma_valu_instructions_per_wave": (per["SQ_INSTS_VALU"] - per["SQ_INSTS_MFMA"]) / waves,
        "kernel_cycles (GRBM_GUI_ACTIVE / 8 XCDs)": per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0,
        "clock_GHz (kernel_cycles / duration)": (per["GRBM_GUI_ACTIVE"] / 8.0 / (dur_us * 1e3)) if (dur_us and "GRBM_GUI_ACTIVE" in per) else None,
        "mfma_busy_frac_of_kernel_cycles": (per["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / (per["GRBM_GUI_ACTIVE"] / 8.0)) if "GRBM_GUI_ACTIVE" in per else None,
        "valu_active_frac_of_kernel_cycles (includes MFMA issue)": (per["SQ_ACTIVE_INST_VALU"] * 4.0 / simds / (per["GRBM_GUI_ACTIVE"] / 8.0)) if "GRBM_GUI_ACTIVE" in per else None,
        "share_of_wave_cycles": {k: per[k] / per["SQ_WAVE_CYCLES"] for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                                                            "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS") if k in per},
    },
}
dst = os.path.join(ROOT, "gpurun_out", out_name)
json.dump(out, open(dst, "w"), indent=1)
print(dst, json.dumps(out["derived"]))
