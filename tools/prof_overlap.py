"""How much of a streaming run's flow, HiFT and decode kernel time overlaps: python tools/prof_overlap.py <kernel-trace dir>
Reads rocprofv3's *_kernel_trace.csv; busy time = union of the kernels' [start, end) intervals per class."""
import csv, glob, sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
cls = {'decode': ('k_qkv', 'k_attn<', 'k_store', 'k_gateup', 'k_prep', 'k_sample', 'k_step', 'k_attn_combine', 'k_rms_split', 'k_rope_cache', 'k_attn_prefill', 'k_swiglu', 'k_gather_rows'),
       'hift': ('k_conv', 'k_phase', 'k_source', 'k_stft', 'k_istft', 'k_f0_head', 'k_set_seed', 'k_fade'),
       'flow': ('k_gemm', 'k_attn_est', 'k_tail', 'k_kv_append', 'k_conv_tail', 'k_layernorm', 'k_euler', 'k_embed', 'k_repeat2', 'k_spk', 'k_mel_out', 'k_relsoftmax')}
iv = {k: [] for k in cls}
iv['other'] = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('void ', '')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    for k, pre in cls.items():
        if n.startswith(pre):
            iv[k].append((s, e))
            break
    else:
        iv['other'].append((s, e))


def union(a):
    a = sorted(a)
    out = []
    for s, e in a:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(u):
    return sum(e - s for s, e in u)


def inter(u, v):
    i = j = 0
    t = 0
    while i < len(u) and j < len(v):
        s, e = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if s < e:
            t += e - s
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return t


U = {k: union(v) for k, v in iv.items()}
allu = union([x for v in iv.values() for x in v])
print('busy ms: ' + '  '.join(f'{k} {total(u) / 1e6:.1f}' for k, u in U.items()) + f'   any kernel {total(allu) / 1e6:.1f}')
for a, b in (('flow', 'hift'), ('flow', 'decode'), ('hift', 'decode')):
    print(f'overlap {a} & {b}: {inter(U[a], U[b]) / 1e6:.1f} ms')
