"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: python tools/prof_summary.py <dir> [n]"""
import csv, glob, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'{"kernel":72s} {"calls":>7s} {"total ms":>10s} {"avg us":>9s} {"%":>6s}')
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.1f}")
print(f'total GPU kernel time {tot/1e6:.1f} ms')
