"""Print the head of a rocprofv3 kernel_stats.csv: python tools/show_stats.py <csv> [n] [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/1e6:.3f} ms" + (f"  per step {tot/1e6/steps:.3f} ms" if steps else ""))
for r in rows[:n]:
    extra = f"  /step {float(r['TotalDurationNs'])/1e3/steps:8.1f} us" if steps else ""
    print(f'{float(r["TotalDurationNs"])/tot*100:5.1f}%  calls {int(r["Calls"]):6d}  avg {float(r["AverageNs"])/1e3:9.1f} us{extra}  {r["Name"][:100]}')
