"""Phase timing inside the skinny decode kernels (diagnostic build only):
    ./build.sh -DCV2_STAMPS -o cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so
    CV2_AMD_LIB=$PWD/cosyvoice2-eu_amd/cv2amd/libcv2amd_dbg.so python tools/dbg_stamps.py
Prints, for layer 1's four weight-streaming kernels of the last decode step, s_memtime deltas (shader cycles) of
block 0 / wave 0 between the phase boundaries marked SK_STAMP in csrc/skinny.h."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth, lib as L
from cv2amd.llm import LLMEngine

sd = synth.make_llm(layers=24)
eng = LLMEngine(sd, 'cuda:0', max_seqs=1, max_pos=2048, max_out=2048)
inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
eng.add_request(0, x, 5000, 5000, mode=1, seed=7, force_len=True)      # RAS sampler, forced length
names = ['qkv', 'attn', 'o-proj', 'gate/up', 'down', 'sample']
phases = ['issue loads', 'x arrive+rms', 'stage B', 'weights+mfma', 'reduce']
for rep in range(3):
    eng.step(1, 64)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 8))()
    L.check(L.lib().cv2_debug_stamps(buf))
    print('rep', rep)
    for k in range(6):
        t = [buf[k * 8 + i] for i in range(7)]
        if names[k] == 'attn':      # k_attn's own boundaries (csrc/llm.hip): block 0 = kv group 0, key split 0
            ph = ['issue+pos', 'q arrive', 'K+scores', 'softmax', 'V+PV', 'store']
            d = [t[i + 1] - t[i] for i in range(6)]
            print(f'  {names[k]:8s} ' + '  '.join(f'{p}={v}' for p, v in zip(ph, d)) + f'   total={t[6]-t[0]} cycles')
            continue
        if names[k] == 'sample':    # k_sample (RAS mode)
            ph = ['logits', 'log-softmax', 'wave top-25', 'merge', 'draw', 'embed+state']
            d = [t[i + 1] - t[i] for i in range(6)]
            print(f'  {names[k]:8s} ' + '  '.join(f'{p}={v}' for p, v in zip(ph, d)) + f'   total={t[6]-t[0]} cycles')
            continue
        d = [(t[i + 1] - t[i]) if t[i + 1] and t[i] else float('nan') for i in range(5)]
        print(f'  {names[k]:8s} ' + '  '.join(f'{p}={v}' for p, v in zip(phases, d)) + f'   total={t[5]-t[0]} cycles')
