"""Round 6: WHO is the slowest block of each role of the one-row decode step, and why (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_chain_tails.py [steps]
profiles/r5_chain_timeline.txt: in every role the last block publishes ~1 us after the median (Q 2.96 vs 1.44, O 8.56 vs 7.62, gate/up 11.68 vs
10.80, down 15.12 vs 14.05 us) = 4-5 of a layer's 14 us.  Over `steps` single decode steps (layer CV2_DBG_LAYER, default 12) this records for
each role's LAST publisher: its index in the role, the XCD / CU it ran on (XCC_ID / HW_ID registers), and which phase made it late -- its
operand arriving late (the wait for its own producers), operand -> staged (the gather / fold of the operand granules), staged -> result
(weights + MFMAs + reduction) -- each relative to the role's median block; and whether the first consumer of the next role shares its XCD."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import numpy as np
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sd = synth.make_llm(layers=24)
MAXPOS = 2048
NT = (MAXPOS + 127) // 128
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=MAXPOS, max_out=MAXPOS)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
eng.add_request(0, x, MAXPOS - 400, MAXPOS - 400, mode=1, seed=7, force_len=True)
roles = (('Q', 36), ('A', NT * 2), ('O', 56), ('gate/up', 304), ('down', 112))
per = sum(n for _, n in roles)
eng.step(1, 16)
torch.cuda.synchronize()
acc = {name: collections.defaultdict(list) for name, _ in roles}
last_idx = {name: collections.Counter() for name, _ in roles}
last_xcd = {name: collections.Counter() for name, _ in roles}
last_cu = {name: collections.Counter() for name, _ in roles}
same_xcd = {name: [] for name, _ in roles}
for it in range(STEPS):
    eng.step(1, 8)                                # (the graph replays 8 steps: the stamps are the last step's)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8))()
    L.check(L.lib().cv2_debug_chain(buf))
    t = np.array(buf, dtype=np.uint64).reshape(1024, 8)
    where = t[:, 7]
    t = t.astype(np.float64)
    base, prev = 0, None
    for name, n in roles:
        r, w = t[base:base + n], where[base:base + n]
        m = r[:, 2] > 0
        idx = np.nonzero(m)[0]
        base += n
        if not m.any():
            prev = None
            continue
        r, w = r[m], w[m]
        us = r / 100.0
        pub = us[:, 2]
        j = int(np.argmax(pub))
        med = lambda a: float(np.median(a))
        a = acc[name]
        a['tail_us'].append(pub[j] - med(pub))
        a['late_start'].append(us[j, 0] - med(us[:, 0]))
        if (r[:, 3] > 0).all():
            a['late_operand'].append(us[j, 3] - med(us[:, 3]))
            if (r[:, 4] > 0).all() and name != 'A':
                a['operand_to_staged'].append((us[j, 4] - us[j, 3]) - med(us[:, 4] - us[:, 3]))
                a['staged_to_result'].append((us[j, 1] - us[j, 4]) - med(us[:, 1] - us[:, 4]))
            else:
                a['operand_to_published'].append((us[j, 2] - us[j, 3]) - med(us[:, 2] - us[:, 3]))
        a['result_to_published'].append((us[j, 2] - us[j, 1]) - med(us[:, 2] - us[:, 1]) if (r[:, 1] > 0).all() else 0.0)
        last_idx[name][int(idx[j])] += 1
        xcd = int((int(w[j]) >> 32) & 15)
        last_xcd[name][xcd] += 1
        last_cu[name][(xcd, (int(w[j]) >> 13) & 7, (int(w[j]) >> 12) & 1, (int(w[j]) >> 8) & 15)] += 1
        if prev is not None:                       # the previous role's last publisher against THIS role's earliest-served consumer
            k = int(np.argmin(us[:, 3])) if (r[:, 3] > 0).all() else 0
            same_xcd[name].append(int(((int(w[k]) >> 32) & 15) == prev))
        prev = xcd
print(f'one-row decode step, layer {os.environ.get("CV2_DBG_LAYER", "12")}, {STEPS} samples (every 8th step): the LAST publisher of each role against the role\'s median block (us)')
for name, n in roles:
    a = acc[name]
    if not a['tail_us']:
        continue
    f = lambda k: f'{np.mean(a[k]):+5.2f}' if a[k] else '  n/a'
    print(f'  {name:8s} tail {np.mean(a["tail_us"]):5.2f} (p90 {np.percentile(a["tail_us"], 90):5.2f}) = started {f("late_start")}, operand arrived {f("late_operand")}, '
          f'operand->staged {f("operand_to_staged")}, staged->result {f("staged_to_result")}, operand->published (A) {f("operand_to_published")}, result->published {f("result_to_published")}')
    top = ', '.join(f'{i}: {c}' for i, c in last_idx[name].most_common(6))
    print(f'           last publisher, index in the role (of {n}): {top}   [{len(last_idx[name])} different blocks]')
    print(f'           its XCD: {dict(sorted(last_xcd[name].items()))};  most frequent (XCD, SE, SH, CU): {last_cu[name].most_common(3)}')
    if same_xcd[name]:
        print(f'           first-served consumer of this role on the XCD of the previous role\'s last publisher: {100.0 * np.mean(same_xcd[name]):.0f} % (12.5 % = chance)')
