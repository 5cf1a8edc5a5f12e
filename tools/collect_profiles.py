#!/usr/bin/env python3
"""Copy the judged evidence of a measurement pass (tools/r04_measure.sh <tag>) from gpurun_out/ (scratch) into profiles/ (tracked):
bench lines, rocprofv3 summaries, the bench lines with the run's own PMC traffic, SQ counter summaries; merge the per-workload traffic tables
into profiles/traffic.json.   usage: tools/collect_profiles.py <tag>"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(G, tag, "bench_*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    open(os.path.join(P, "%s_%s" % (tag, os.path.basename(f))), "w").write(lines[-1]); n += 1
tab_path = os.path.join(P, "traffic.json")
tab = json.load(open(tab_path))
for d in sorted(glob.glob(os.path.join(G, "prof_%s_*" % tag))):
    t = os.path.basename(d)[len("prof_%s_" % tag):]
    s = os.path.join(d, "summary.txt")
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, "%s_%s_rocprof_summary.txt" % (tag, t))); n += 1
    b = os.path.join(d, "bench_measured.json")
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            open(os.path.join(P, "%s_%s_bench_measured_traffic.json" % (tag, t)), "w").write(lines[-1]); n += 1
    tj = os.path.join(d, "traffic.json")
    if os.path.exists(tj):
        tab.update(json.load(open(tj)))
for f in sorted(glob.glob(os.path.join(G, tag, "sq_*.txt"))):
    shutil.copy(f, os.path.join(P, "%s_%s_sq_pmc.txt" % (tag, os.path.basename(f)[:-4]))); n += 1
json.dump(tab, open(tab_path, "w"), indent=1, sort_keys=True)
print("collected %d files for %s; traffic.json keys: %s" % (n, tag, ", ".join(sorted(tab))))
