"""Where the GPU sits idle between the kernels of a run: reads a rocprofv3 --kernel-trace CSV, starts at
the first ksh:: kernel, and adds up the gaps between consecutive kernels by the pair of kernels around
them (gaps of 50 ms and more are between builds and not counted).  usage: gpu_gaps.py kernel_trace.csv"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
first = next(i for i, r in enumerate(rows) if "ksh::" in r[2])
rows = rows[first:]


def short(n):
    return n.split("ksh::")[-1].split("(")[0][:40]


busy = idle = 0
after, cnt, hist = collections.Counter(), collections.Counter(), collections.Counter()
end = rows[0][0]
prev = None
for s, e, n in rows:
    if s > end and prev is not None:
        g = s - end
        if g < 50e6:
            idle += g
            key = short(prev) + " -> " + short(n)
            after[key] += g
            cnt[key] += 1
            hist[min(int(g / 1000).bit_length(), 12)] += g
    busy += max(0, e - max(s, end))
    end = max(end, e)
    prev = n
print("busy %.1f ms, idle (gaps < 50 ms) %.1f ms" % (busy / 1e6, idle / 1e6))
for k, v in after.most_common(25):
    print("%9.2f ms  %5d gaps  avg %7.1f us  %s" % (v / 1e6, cnt[k], v / cnt[k] / 1e3, k))
print("idle by gap length (upper bound in us):", {(1 << b): round(v / 1e6, 1) for b, v in sorted(hist.items())})
