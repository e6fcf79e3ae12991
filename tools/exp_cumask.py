"""Round 6: does a many-row decode burst overlap with a flow batch when the two streams own DISJOINT sets of CUs (hipExtStreamCreateWithCUMask)?
Round 3 measured the overlap on ordinary streams (tools/exp_overlap.py): both sides stretch, 10 % saved, the batch pipelining lost.  With CU
masks the latency-bound decode chain keeps CUs no flow block can occupy.   python tools/exp_cumask.py [rows] [utts] [steps]
Prints, per split (decode CUs / flow CUs): decode alone, flow alone (each on its masked stream), both at once."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.llm import LLMEngine
from cv2amd.flow import FlowEngine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
utts_n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 120
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)
hip = C.CDLL('libamdhip64.so')
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
NCU = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(bits):
    """bits: iterable of CU-mask bit indices that are ON.  None -> an ordinary stream."""
    if bits is None:
        return torch.cuda.Stream(dev)
    words = (NCU + 31) // 32
    m = (C.c_uint32 * words)()
    for b in bits:
        m[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, m)
    assert rc == 0, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(s.value, device=dev)


eng = LLMEngine(synth.make_llm(layers=24), dev, max_seqs=32, max_pos=2048, max_out=2048)
for b in range(rows):
    inp = synth.synthetic_inputs(seed=b, text_len=50, prompt_len=255)
    eng.add_request(b, eng.build_lm_input(inp['text'], inp['prompt_text'], inp['prompt_token']), 2000, 2000, force_len=True)
flow = FlowEngine(synth.make_flow(), dev, max_utts=utts_n, max_len=2 * (320 + 512))
inp = synth.synthetic_inputs(seed=1986, text_len=50, prompt_len=255, prompt_text_len=20)
utts = [dict(token=torch.randint(0, 6561, (1, 250), dtype=torch.int32), prompt_token=inp['prompt_token'].to(dev),
             prompt_feat=inp['prompt_feat'].to(dev), embedding=inp['embedding'].to(dev)) for _ in range(utts_n)]


def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3


print(f'{NCU} CUs; decode: {rows} rows x {steps} steps (launches form, as beside other work); flow: {utts_n} utterances of 10 s', flush=True)
splits = [('ordinary streams (round 3)', None, None)]
for nd in (32, 64, 96, 128):
    splits.append((f'mask bits [0, {nd}) decode / [{nd}, {NCU}) flow', range(nd), range(nd, NCU)))
splits.append((f'decode unmasked / flow bits [64, {NCU})', None, range(64, NCU)))
splits.append(('decode every 4th bit (64 CUs) / flow the rest', range(0, NCU, 4), [b for b in range(NCU) if b % 4]))
for name, bd, bf in splits:
    try:
        sd, sf = masked_stream(bd), masked_stream(bf)
    except AssertionError as e:
        print(name, '->', e, flush=True)
        continue

    def dec():
        with torch.cuda.stream(sd):
            eng.step(rows, steps, shared=True)

    def fl():
        with torch.cuda.stream(sf):
            flow.inference_batch(utts, streaming=False, finalize=True)
    for _ in range(2):
        dec(); fl(); torch.cuda.synchronize()
    a, b = min(t(dec), t(dec)), min(t(fl), t(fl))
    c = min(t(lambda: (dec(), fl())), t(lambda: (dec(), fl())))
    print(f'{name:52s}: decode alone {a:7.1f} ms ({a / steps * 1e3:6.0f} us per step), flow alone {b:6.1f} ms, sum {a + b:7.1f}, both at once {c:7.1f} ms '
          f'({100 * (1 - c / (a + b)):.0f} % saved vs this split\'s sum)', flush=True)
