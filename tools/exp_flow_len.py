"""Flow stage of ONE utterance against its length: the estimator attention's block forms switch on the packed row count (flow.hip
est_tblock), and a form whose grid is just above one block per CU runs two rounds.  ms per call for generated-token counts N (T = 2 (255 + N)
frames), under the default dispatch and with CV2_ATT_KSP=2 (two key groups of a 32-row block).  The variable is read once per process: the
script runs itself.  python tools/exp_flow_len.py [N ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if 'LEN_CHILD' not in os.environ:
    for mode in os.environ.get('LEN_MODES', 'default,2').split(','):
        env = dict(os.environ, LEN_CHILD='1')
        if mode != 'default':
            env['CV2_ATT_KSP'] = mode
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.flow import FlowEngine
dev = 'cuda:0'
Ns = [int(a) for a in sys.argv[1:]] or [250, 280, 300, 400, 550, 700]
flow = FlowEngine(synth.make_flow(), dev, max_utts=1, max_len=2 * (320 + 800))
inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
for N in Ns:
    utt = dict(token=torch.randint(0, 6561, (1, N), dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
               prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev))
    for _ in range(2):
        flow.inference_batch([utt], streaming=False, finalize=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        flow.inference_batch([utt], streaming=False, finalize=True)
    e1.record(); torch.cuda.synchronize()
    T = 2 * (255 + N)
    print(f'CV2_ATT_KSP={os.environ.get("CV2_ATT_KSP", "default")}: N = {N:4d} (T = {T:5d} frames, {2 * ((T + 8 + 127) // 128 * 128)} rows): {e0.elapsed_time(e1) / 5:7.2f} ms', flush=True)
