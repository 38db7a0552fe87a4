"""rocprofv3 --kernel-trace CSV -> per (kernel, grid) table of one training step: calls, average and total time.
    python scripts/trace_summary.py KERNEL_TRACE.csv STEPS [TOP]   (STEPS = number of steps the trace covers)"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
agg = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:70]
    key = (name, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += d
tot = sum(v[1] for v in agg.values())
print("kernel time %.2f ms / step over %g steps, %d launches / step" % (tot / 1e6 / steps, steps, len(rows) / steps))
# how much of the traced span the card ran at least one kernel (launches on two streams overlap), and the gaps between launches
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
busy, gaps_big, cur_s, cur_e = 0, 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if s - cur_e > 1000000:      # > 1 ms: between the traced steps' timing brackets / warm-up, not inside a step
            gaps_big += s - cur_e
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0] - gaps_big
print("card busy %.2f ms / step, idle between launches %.2f ms / step (gaps > 1 ms excluded), overlap of the two streams %.2f ms / step"
      % (busy / 1e6 / steps, (span - busy) / 1e6 / steps, (tot - busy) / 1e6 / steps))
print("%9s %7s %9s  %s" % ("us/launch", "n/step", "ms/step", "kernel  (workgroups x, y, z)"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%9.1f %7.1f %9.3f  %s  %s" % (v[1] / v[0] / 1e3, v[0] / steps, v[1] / 1e6 / steps, k[0], k[1:]))
