"""Phase timing inside the flow GEMMs (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_stamps_flow.py
Block (0,0,0) of every k_gemm launch stamps s_memtime (shader cycles) at its phase boundaries into a 64-entry ring; this prints
the last 64 launches of one estimator pass grouped by (tile, K, N, blocks)."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.flow import FlowEngine

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # utterances per batch (32: the configs[2] tile shapes)
flow = FlowEngine(synth.make_flow(), 'cuda:0', max_utts=NU, max_len=2 * (320 + 512))
utts = []
for i in range(NU):
    inp = synth.synthetic_inputs(seed=1986 + i, text_len=50, prompt_len=255 - (37 * i) % 100, prompt_text_len=20)
    utts.append(dict(token=torch.randint(0, 6561, (1, 250 - (53 * i) % 100), dtype=torch.int32), prompt_token=inp['prompt_token'].to('cuda:0'),
                     prompt_feat=inp['prompt_feat'].to('cuda:0'), embedding=inp['embedding'].to('cuda:0')))
for rep in range(2):
    flow.inference_batch(utts, streaming=False, finalize=True)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (64 * 8))()
L.check(L.lib().cv2_debug_stamps_flow(buf))
ph = ['issue', 'first stage', 'K loop', 'C stage', 'row epilogue']
t = [buf[60 * 8 + i] for i in range(8)]
print(f'k_attn_est block 0: prologue={t[1]-t[0]}  staging+wait(sum)={t[2]}  compute(sum)={t[3]}  loop total={t[4]-t[1]}  store={t[5]-t[4]}  tiles={t[6]} blocks={t[7]}')
t = [buf[61 * 8 + i] for i in range(8)]
if t[0]:
    print(f'k_attn_est_dma block 0 (s_memtime ticks): prologue={t[1]-t[0]}  loop total={t[4]-t[1]} = wait for the tile (sum) {t[2]} + barrier (sum) {t[3]} + DMA issue (sum) {t[6]} '
          f'+ tile body (sum) {t[7]}  store={t[5]-t[4]}  tiles={buf[62 * 8]} blocks={buf[62 * 8 + 1]}')
groups = collections.defaultdict(list)
for k in range(56):
    t = [buf[k * 8 + i] for i in range(8)]
    if not t[0]: continue
    key = (t[7] >> 48, (t[7] >> 32) & 0xffff, t[6] >> 32, t[6] & 0xffffffff, t[7] & 0xffffffff)
    groups[key].append([t[i + 1] - t[i] for i in range(5)])
for key, rows in sorted(groups.items()):
    n = len(rows)
    avg = [sum(r[i] for r in rows) / n for i in range(5)]
    print(f'tile {key[0]}x{key[1]} K={key[2]} N={key[3]} blocks={key[4]} launches={n}: ' +
          '  '.join(f'{p}={v:.0f}' for p, v in zip(ph, avg)) + f'   total={sum(avg):.0f} cycles')
