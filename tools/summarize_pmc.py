"""Summarise the rocprofv3 passes collected by tools/profile_bench.sh: per-kernel duration stats and PMC sums."""
import csv, glob, os, sys, collections

out = sys.argv[1]
print("# kernel stats (rocprofv3 --kernel-trace --stats), bench.py --steps 1 --warmup 0 --no-cpu")
for f in glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv")):
    for r in list(csv.DictReader(open(f)))[:10]:
        print("%-44s calls=%4s total_ms=%9.2f avg_us=%9.1f pct=%s" % (r["Name"].split("(")[0][:44], r["Calls"],
              float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("# PMC sums per kernel (one counter set per pass; FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE reads 1/2 on gfx950)")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][:44], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print("%-44s %-14s launches=%4d sum=%.5g avg=%.5g" % (k, c, n, s, s / n))
