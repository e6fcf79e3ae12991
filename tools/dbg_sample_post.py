import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_sample')]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# each phase: 1 prefill sample + 100 decode samples
names = ['greedy(0)', 'ras(1)', 'exit after log-softmax(2)', 'exit after per-wave top-25(3)', 'exit after merge(4)', 'ras(1) again']
for i, n in enumerate(names):
    seg = rows[i * 101 + 1:(i + 1) * 101]
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in seg)
    print(f'{n:34s} median {d[len(d)//2]:6.1f} us  min {d[0]:6.1f}  max {d[-1]:6.1f}  n={len(d)}')
