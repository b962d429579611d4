"""How the kernels of a multi-lane build share the GPU: reads a rocprofv3 --kernel-trace CSV, takes the LAST
build's span (from the last k_str_bases-led burst of decode kernels to the end), and reports
  - wall, busy (union of all kernel spans), idle, and the time during which 1 / 2 / 3+ kernels were running;
  - per kernel name: launches, summed duration (stream time), and the part of it spent alone on the GPU.
usage: lane_overlap.py kernel_trace.csv [BUILD]   (BUILD: index of the build in the trace, default -1)"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    m = re.search(r"ksh::(k_\w+)", n)
    return m.group(1) if m else n.split("(")[0][:32]


# the builds of the trace: every build starts with the Size() of its inputs (k_sum_lens, one per input, within its
# first tenth of a second); a build ends where the next one starts.  argv[2] = which build (default -1: the last
# one; bench.py's last build is its one-stream roofline leg, the one before it the last TIMED build: -2)
ksh = [r for r in rows if "ksh::" in r[2]]
starts = []
prev = None
for i, r in enumerate(ksh):
    if "k_sum_lens" in r[2]:
        if prev is None or r[0] - prev > 100e6:
            starts.append(i)
        prev = r[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
if starts:
    b0 = starts[which]
    later = [x for x in starts if x > b0]
    build = ksh[b0:(later[0] if later else len(ksh))]
else:
    build = ksh
t0, t1 = build[0][0], max(x[1] for x in build)
ev = []
for s, e, n in build:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort()
depth_time = collections.Counter()
alone = collections.Counter()
running = collections.Counter()
depth, last = 0, t0
for t, d, n in ev:
    if t > last:
        depth_time[min(depth, 4)] += t - last
        if depth == 1:
            (only,) = [k for k, v in running.items() if v > 0]
            alone[only] += t - last
    last = t
    depth += d
    running[short(n)] += d
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n in build:
    tot[short(n)] += e - s
    cnt[short(n)] += 1
wall = t1 - t0
print("build wall %.1f ms; idle %.1f; 1 kernel %.1f; 2 kernels %.1f; 3 %.1f; 4+ %.1f; stream time %.1f ms"
      % (wall / 1e6, depth_time[0] / 1e6, depth_time[1] / 1e6, depth_time[2] / 1e6, depth_time[3] / 1e6,
         depth_time[4] / 1e6, sum(tot.values()) / 1e6))
print("%-34s %7s %10s %10s" % ("kernel", "n", "sum ms", "alone ms"))
for k, v in tot.most_common(30):
    print("%-34s %7d %10.2f %10.2f" % (k, cnt[k], v / 1e6, alone[k] / 1e6))
