"""Turn a rocprofv3 `*_kernel_stats.csv` into the markdown table kept under profiles/."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    t = float(r["TotalDurationNs"])
    print("| `%s` | %s | %.2f | %.1f | %.1f |" % (r["Name"][:110], r["Calls"], t / 1e6, float(r["AverageNs"]) / 1e3, 100 * t / tot))
print("\nSum of kernel time %.1f ms over %g steps = %.1f ms/step" % (tot / 1e6, steps, tot / 1e6 / steps))
