"""One stage alone for the profilers: python tools/prof_stage_run.py {flow|hift|decode} [reps]
flow: one utterance, T = 1010 (configs[1] shape), encoder + 10 Euler steps with CFG; hift: 500 frames; decode: 40 one-row steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = 'cuda:0'
if what == 'flow':
    from cv2amd.flow import FlowEngine
    flow = FlowEngine(synth.make_flow(), dev, max_utts=1, max_len=2 * (320 + 512))
    inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
    utt = dict(token=torch.randint(0, 6561, (1, 250), dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
               prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(reps):                          # exactly `reps` calls (the counter tools divide by it); the last one is timed
        if i == reps - 1:
            e0.record()
        flow.inference_batch([utt], streaming=False, finalize=True)
    e1.record()
    torch.cuda.synchronize()
    print(f'flow, one utterance: {e0.elapsed_time(e1):.2f} ms (last of {reps} calls)')
elif what == 'hift':
    from cv2amd.hift import HiftEngine
    eng = HiftEngine(synth.make_hift(), dev, max_frames=512)
    mel = torch.randn(1, 80, 500, device=dev) * 0.5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(reps):                          # exactly `reps` calls; the last one is timed
        if i == reps - 1:
            e0.record()
        eng.inference(mel, None, seed=1)
    e1.record()
    torch.cuda.synchronize()
    print(f'hift, 500 frames: {e0.elapsed_time(e1):.3f} ms (last of {reps} calls)')
else:
    from cv2amd.llm import LLMEngine
    eng = LLMEngine(synth.make_llm(layers=24), dev, max_seqs=1, max_pos=2048, max_out=2048)
    inp = synth.synthetic_inputs(seed=0, text_len=50, prompt_len=255)
    x = eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token'])
    eng.add_request(0, x, 2000, 2000, mode=1, seed=7, force_len=True)
    eng.step(1, 40)
torch.cuda.synchronize()
print('REPS', reps)
