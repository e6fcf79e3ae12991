"""Sum rocprofv3 --pmc counter CSVs per decode step: python tools/pmc_decode.py <dir> <counter> <steps_profiled>"""
import csv, glob, sys
d, counter = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
dec = [r for r in rows if any(k in r['Kernel_Name'] for k in ('k_qkv<1', 'k_attn<', 'k_store<1', 'k_gateup<1', 'k_sample'))]
tot = sum(float(r['Counter_Value']) for r in dec)
nsample = sum(1 for r in dec if 'k_sample' in r['Kernel_Name'])
per = {}
for r in dec:
    k = r['Kernel_Name'].split('(')[0]
    per[k] = per.get(k, 0.0) + float(r['Counter_Value'])
print(f'{counter}: {len(dec)} decode dispatches over {nsample} steps (incl. prefill sample), total {tot:.4g}, per step {tot / max(nsample, 1):.4g}')
for k, v in sorted(per.items(), key=lambda x: -x[1]):
    print(f'   {k:40s} {v / max(nsample, 1):12.4g} per step')
