"""Reconcile bench.py's HIP-event kernel time with rocprofv3's: both taken from the SAME profiled run (scripts/gpu_profiles.sh:
`rocprofv3 --kernel-trace --stats -- python3 bench.py ...`, whose JSON line sits in <tag>_prof_bench.log), next to the same
command run without the profiler (gpurun_out/<tag>_bench_plain.json when present).  usage: python scripts/event_vs_rocprof.py <tag>"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = os.path.join(ROOT, "gpurun_out")
line = None
for l in open(os.path.join(out, "%s_prof_bench.log" % tag), errors="replace"):
    if l.startswith("{") and '"roofline"' in l:
        line = json.loads(l)
assert line, "no bench JSON line in the profiled run's log"
kern = line["roofline"]["dispatch"]["kernel"]
rows = [r for r in csv.DictReader(open(os.path.join(out, "%s_bench_kernel_stats.csv" % tag))) if (kern + "<") in r["Name"]]
calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
n_ev = line["steps"] * 16
# the timed region's launches are the LAST n_ev of the run (warm-up and the secondary legs come before / after): the CSV average
# covers all of them, so also average the trace over the launches between the first and the last timed one when it is there
res = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-api-level (one run: the JSON line's HIP events and the CSV describe the same launches)",
       "kernel": kern,
       # bench.py: achieved TFLOP/s = 245760 flop/row x 1e6 rows x steps / (sum of the event durations): inverted here
       "events_avg_ms_under_profiler": 245760 * 1e6 * line["steps"] / (line["roofline"]["achieved"] * 1e12) * 1e3 / n_ev,
       "rocprof_avg_ms_all_launches": tot / calls / 1e6, "rocprof_launches": calls, "event_launches": n_ev,
       "ms_per_step_under_profiler": line["ms_per_step"], "frac_under_profiler": line["roofline"]["frac"]}
plain = os.path.join(out, "%s_bench_plain.json" % tag)
if os.path.exists(plain):
    p = json.load(open(plain))
    res["events_avg_ms_without_profiler"] = 245760 * 1e6 * p["steps"] / (p["roofline"]["achieved"] * 1e12) * 1e3 / (p["steps"] * 16)
    res["ms_per_step_without_profiler"] = p["ms_per_step"]; res["frac_without_profiler"] = p["roofline"]["frac"]
res["ratio_rocprof_over_events_same_run"] = res["rocprof_avg_ms_all_launches"] / res["events_avg_ms_under_profiler"]
json.dump(res, open(os.path.join(out, "%s_event_vs_rocprof.json" % tag), "w"), indent=1)
print(json.dumps(res, indent=1))
