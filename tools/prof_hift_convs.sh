# per-launch durations of the HiFT convolutions of one utterance (500 frames), grouped by grid and LDS size
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_hc
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hc -- python3 $GRAFT_REPO_ROOT/tools/prof_hift_run.py > $GRAFT_REPO_ROOT/gpurun_out/prof_hc.log 2>&1
cd $GRAFT_REPO_ROOT && python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/prof_hc/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_conv')]
n = len(rows) // 3                       # three calls: keep the last
rows = rows[-n:]
g = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'][:8], r['Grid_Size_X'], r['Grid_Size_Y'], r['LDS_Block_Size'] if 'LDS_Block_Size' in r else r.get('Dynamic_LDS_Size', ''))
    g.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = 0
for k, v in g.items():
    tot += sum(v)
    print(k, 'n=%d avg %.1f us  sum %.0f us' % (len(v), sum(v) / len(v), sum(v)))
print('total conv us', tot)
P
find gpurun_out/prof_hc -name '*_kernel_trace.csv' -delete
