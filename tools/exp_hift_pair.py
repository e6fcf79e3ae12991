"""Vocoder with every ResBlock (dilated conv, conv) pair of the 64- / 128-channel stages as ONE launch (k_respair, the default) against
the two k_conv6 launches per pair (CV2_HIFT_PAIR=0): ms per call and a hash of the waveform (the two forms are bit-identical).
The variable is read once per process: this script runs itself.  python tools/exp_hift_pair.py [frames ...]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if 'PAIR_CHILD' not in os.environ:
    for mode in os.environ.get('PAIR_MODES', '0,1,0,1').split(','):           # 'on:min_blocks' sets CV2_HIFT_PAIR_MIN too
        env = dict(os.environ, PAIR_CHILD='1', CV2_HIFT_PAIR=mode.split(':')[0])
        if ':' in mode:
            env['CV2_HIFT_PAIR_MIN'] = mode.split(':')[1]
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, check=False)
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
from cv2amd import synth
from cv2amd.hift import HiftEngine
frames = [int(a) for a in sys.argv[1:]] or [500, 90, 37]
eng = HiftEngine(synth.make_hift(), 'cuda:0', max_frames=max(512, max(frames)))
for T in frames:
    g = torch.Generator().manual_seed(T)
    mel = (torch.randn(1, 80, T, generator=g) * 2 - 4).clamp(-11.5, 2).to('cuda:0')
    for _ in range(3):
        wav, src = eng.inference(mel, None, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        wav, src = eng.inference(mel, None, seed=1)
    e1.record()
    torch.cuda.synchronize()
    h = hashlib.sha256(wav.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f'CV2_HIFT_PAIR={os.environ["CV2_HIFT_PAIR"]} min {os.environ.get("CV2_HIFT_PAIR_MIN", "-")}: {T:4d} frames {e0.elapsed_time(e1) / 20:7.3f} ms per call; waveform sha {h} finite {bool(torch.isfinite(wav).all())} '
          f'absmax {float(wav.abs().max()):.4f}', flush=True)
