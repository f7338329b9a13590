"""Per-kernel begin -> end durations of a rocprofv3 --kernel-trace CSV: n, mean, median, p10, p90, max [us].
usage: kernel_trace_summary.py <..._kernel_trace.csv> ["header line"]"""
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
if len(sys.argv) > 2:
    print(sys.argv[2])
for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if name.startswith("void at::") or "at::native" in name:
        continue
    v.sort()
    q = lambda f: v[min(len(v) - 1, int(f * len(v)))]
    print("%-86s n %5d  mean %8.2f  median %8.2f  p10 %8.2f  p90 %8.2f  max %8.2f"
          % (name[:86], len(v), sum(v) / len(v), q(0.5), q(0.1), q(0.9), v[-1]))
