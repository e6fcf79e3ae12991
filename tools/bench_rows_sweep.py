"""Decode step time against the number of live rows (cv2_llm_decode_rows over slots 1..n, so that one row also takes the launches):
python tools/bench_rows_sweep.py [prompt_len]      (CV2_SWEEP_ROWS=32,24,17: those row counts only)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine

P = int(sys.argv[1]) if len(sys.argv) > 1 else 255
eng = LLMEngine(synth.make_llm(layers=24), 'cuda:0', max_seqs=32, max_pos=2048, max_out=2048)
xs = []
for b in range(32):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=P)
    xs.append(eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token']))


def reset(slots):
    """fresh requests in `slots`: every measurement runs at the same context (prompt + 16 .. 80 generated positions)"""
    eng.park()
    eng.add_requests(slots, [xs[b] for b in slots], [(2000, 2000)] * len(slots), 1, 0, True)


def timed(fn):
    fn(16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(64); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 64 * 1e3


print(f'decode step against the number of live rows, P = {P} prompt tokens + 50 text tokens, positions {P + 52 + 16} .. {P + 52 + 80} (us per step)')
ROWS = tuple(int(v) for v in os.environ['CV2_SWEEP_ROWS'].split(',')) if os.environ.get('CV2_SWEEP_ROWS') else (32, 28, 24, 22, 20, 18, 17, 16, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1)
for n in ROWS:
    slots = list(range(32 - n, 32))
    res = []
    for shared in ((False, True) if n <= 24 else (False,)):         # <= 24 rows: the one-launch step (k_step2 pairs, k_step<true> at 3 rows, k_step4 from 9 rows), and the launches beside it
        reset(slots)
        res.append(timed(lambda k: eng.step_rows(slots, k, shared=shared)))
    print(f'rows {n:2d}: ' + (f'{res[0]:7.1f} (one launch)   {res[1]:7.1f} (launches)' if len(res) == 2 else f'{res[0]:7.1f} (launches)'), flush=True)
reset([0])
print(f'rows  1 (slot 0, one-row kernel k_step<false>): {timed(lambda k: eng.step(1, k)):7.1f}', flush=True)
st = eng.state.cpu()
assert not int(st[:, 10].any()), 'a slot reports an error'
