"""Print the top rows of a rocprofv3 kernel_stats.csv (diagnostic helper)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[:n]:
    print(f"{r['Name'][:95]:95s} {r['Calls']:>5} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
