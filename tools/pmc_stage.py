"""Per-kernel sums of rocprofv3 --pmc counters and kernel-trace durations of one stage -> JSON.
    python tools/pmc_stage.py <stage> <reps> <out.json> <trace_dir> <pmc_dir> [<pmc_dir> ...]
Counters are collected in separate passes (no trace domains beside --pmc); FETCH_SIZE is doubled (gfx950: 128-byte requests are
tallied at 64 B, MI355X_MICROARCH.md section HBM); FETCH_SIZE / WRITE_SIZE are in KiB.  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES
/ (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import csv, glob, json, sys, collections

stage, reps, out_path, trace_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
pmc_dirs = sys.argv[5:]


def short(name):
    return name.split('(')[0].replace('void ', '').strip()


dur = collections.defaultdict(lambda: [0.0, 0])
f = glob.glob(trace_dir + '/**/*kernel_trace.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    k = short(r['Kernel_Name'])
    dur[k][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    dur[k][1] += 1
cnt = collections.defaultdict(dict)
for d in pmc_dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            c = r['Counter_Name']
            cnt[k][c] = cnt[k].get(c, 0.0) + float(r['Counter_Value'])
rows = []
for k, (us, n) in sorted(dur.items(), key=lambda x: -x[1][0]):
    c = cnt.get(k, {})
    row = {'kernel': k, 'launches_per_rep': n / reps, 'us_per_rep': round(us / reps, 1), 'avg_us': round(us / n, 2)}
    if 'FETCH_SIZE' in c or 'WRITE_SIZE' in c:
        rd = 2.0 * c.get('FETCH_SIZE', 0.0) * 1024 / reps
        wr = c.get('WRITE_SIZE', 0.0) * 1024 / reps
        row.update({'hbm_read_MB_per_rep': round(rd / 1e6, 2), 'hbm_write_MB_per_rep': round(wr / 1e6, 2),
                    'hbm_GBs': round((rd + wr) / (us / reps * 1e-6) / 1e9, 1) if us else None})
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c.get('GRBM_GUI_ACTIVE'):
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        row['mfma_util'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * cyc), 4)
        row['clock_GHz'] = round(cyc / (us * 1e3) , 3) if us else None
        # GRBM_GUI_ACTIVE also counts dispatch / drain time around a short kernel (apparent clocks of 4-5 GHz above): the same busy
        # cycles over the kernel-trace duration at the 2.4 GHz peak engine clock is the figure to hold against the MFMA roof
        row['mfma_util_wall'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * 256 * us * 2400.0), 4) if us else None
    for extra in ('SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_BUSY_CYCLES', 'SQ_WAVES'):
        if extra in c:
            row[extra + '_per_rep'] = c[extra] / reps
    rows.append(row)
tot_us = sum(r['us_per_rep'] for r in rows)
summary = {'stage': stage, 'reps': reps, 'kernel_us_per_rep': round(tot_us, 1),
           'hbm_read_MB_per_rep': round(sum(r.get('hbm_read_MB_per_rep', 0) for r in rows), 1),
           'hbm_write_MB_per_rep': round(sum(r.get('hbm_write_MB_per_rep', 0) for r in rows), 1),
           'note': 'rocprofv3 --pmc, separate passes (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE) + one --kernel-trace pass; '
                   'FETCH_SIZE x 2 (gfx950 correction); per rep = one utterance of the configs[1] shape',
           'kernels': rows[:24]}
w = [r for r in rows if 'mfma_util' in r]
if w:
    summary['mfma_util_time_weighted'] = round(sum(r['mfma_util'] * r['us_per_rep'] for r in w) / sum(r['us_per_rep'] for r in w), 4)
    summary['mfma_util_wall_time_weighted'] = round(sum(r['mfma_util_wall'] * r['us_per_rep'] for r in w) / sum(r['us_per_rep'] for r in w), 4)
tb = summary['hbm_read_MB_per_rep'] + summary['hbm_write_MB_per_rep']
summary['hbm_GBs_over_kernel_time'] = round(tb * 1e6 / (tot_us * 1e-6) / 1e9, 1) if tot_us else None
json.dump(summary, open(out_path, 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'kernels'}))
for r in rows[:10]:
    print(r)
