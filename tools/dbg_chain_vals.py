"""One decode step of a 1-layer model through k_step and through the launches: compare every intermediate vector."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

plen = int(sys.argv[1]) if len(sys.argv) > 1 else 12
inp = synth.synthetic_inputs(text_len=6, prompt_len=plen, prompt_text_len=4)
sd = synth.make_llm(layers=1)
res = {}
for chain in ('0', '1'):
    os.environ['CV2_LLM_CHAIN'] = chain
    eng = LLMEngine(sd, 'cuda:0', max_seqs=4, max_pos=512, max_out=64)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    eng.add_request(0, x, 50, 50, force_len=True)
    torch.cuda.synchronize()
    pos = int(eng.state[0, 0])
    eng.step(1, 1)
    torch.cuda.synchronize()
    ptr = (C.c_uint64 * 16)()
    lib = L.lib(); lib.cv2_llm_debug_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.check(lib.cv2_llm_debug_ptrs(eng.handle, ptr))
    ws0 = eng.workspace.data_ptr()
    def f32(addr, n):
        off = addr - ws0
        return eng.workspace[off:off + 4 * n].view(torch.float32).cpu().clone()
    def gran(idx, n):
        off = ptr[8] - ws0 + idx * 8
        g = eng.workspace[off:off + 8 * n].view(torch.int32).cpu().clone().view(n, 2)
        return g[:, 0].contiguous().view(torch.float32), g[:, 1]
    H, NQ, I = 896, 896, 4864
    d = {'logits': eng.logits[0, :6564].cpu().clone(), 'pos': pos}
    if chain == '0':
        d['q'] = f32(ptr[0], NQ)
        d['o'] = f32(ptr[3], H)
        d['h'] = f32(ptr[4], I)
        d['parts'] = f32(ptr[5], 4 * 32 * H).view(4, 32, H)[:, 0]
        d['xa'] = f32(ptr[6], H); d['xb'] = f32(ptr[7], H)
        d['att'] = f32(ptr[1], 16 * 32 * NQ).view(16, 32, NQ)[:, 0]
        d['ml'] = f32(ptr[2], 16 * 32 * 14 * 2).view(16, 32, 14, 2)[:, 0]
    else:
        gl, odg, oqg, okv, oag, ohg = [int(ptr[i]) for i in range(9, 15)]
        d['xmid'], d['xmid_tag'] = gran(0, H)
        d['parts'] = gran(odg, 4 * H)[0].view(4, H)
        d['q'], d['q_tag'] = gran(oqg, NQ)
        d['kv'] = gran(okv, 256)[0]
        d['att'], d['att_tag'] = gran(oag, 8 * 2 * 464)
        d['h'] = gran(ohg, I)[0]
        d['epoch'] = int(eng.workspace[(ptr[8] - ws0 - 256):(ptr[8] - ws0 - 252)].view(torch.int32)[0]) if False else None
    res[chain] = d
    del eng
a, b = res['0'], res['1']
print('pos', a['pos'], b['pos'])
cmp = lambda n, x, y: print(f'{n:8s} max|diff| {float((x - y).abs().max()):.3e}  max|ref| {float(x.abs().max()):.3e}')
cmp('q', a['q'], b['q'])
print('q tags', b['q_tag'].unique().tolist(), 'xmid tags', b['xmid_tag'].unique().tolist())
# attention partials of tile 0: launches layout [split][NQ] with 64-key splits at max_pos 512
att0 = b['att'].view(8, 2, 464)
for s in range(2):
    for g in range(2):
        cmp(f'att s{s} g{g}', a['att'][s, g * 448:(g + 1) * 448], att0[s, g, :448])
        cmp(f'ml  s{s} g{g}', a['ml'][s, g * 7:(g + 1) * 7].reshape(-1), att0[s, g, 448:462])
xm_ref = a['xa'] if (a['xa'] - b['xmid']).abs().max() < (a['xb'] - b['xmid']).abs().max() else a['xb']
cmp('x_mid', xm_ref, b['xmid'])
cmp('h', a['h'], b['h'])
cmp('parts', a['parts'], b['parts'])
cmp('logits', a['logits'], b['logits'])
