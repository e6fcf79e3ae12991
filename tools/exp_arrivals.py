"""Round 6 (ADVICE): the first-round hold under SUSTAINED arrivals.  The hold (cosyvoice/cli/model.py: a streaming call's first chunk waits for
newcomers that are about to submit theirs, so that they share one chunk round) was tuned on bursts of 8 calls within 40 ms; a server sees calls
arriving all the time.  Streaming calls arrive as a Poisson process (mean gap `gap_ms`, default 80 ms = ~8 concurrent streams) for `n` calls;
time from each call's own start to its first chunk, per hold setting: off, the round-5 form (no window: every newcomer younger than 80 ms
counts, cap 120 ms) and the round-6 default (only newcomers expected within 40 ms).   python tools/exp_arrivals.py [n] [gap_ms]"""
import os, random, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
import torch
import bench as B
args = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(args[0]) if len(args) > 0 else 60
gap = float(args[1]) if len(args) > 1 else 80.0
dev = torch.device('cuda:0')
model = B.build_model(dev, 32)
sreq = B.request(1986, B.P_TOK, 12, dev)
B.run_calls(model, [sreq] * 8, [None] * 8, stream=True)          # graphs, prompt cache
B.run_calls(model, [sreq] * 8, [None] * 8, stream=True)
if '--trace' in sys.argv:       # the scheduler's log between one steady-state newcomer's call and its first chunk (default hold), two newcomers
    rng = random.Random(7)
    offs, t = [], 0.0
    for _ in range(n):
        t += rng.expovariate(1.0 / gap) * 1e-3
        offs.append(t)
    calls, firsts_t = [None] * n, [None] * n
    model.newcomer_beside = model.first_chunk_lane = '--beside' in sys.argv
    model._sched_log = []
    t_start = time.perf_counter()

    def twork(i):
        time.sleep(max(0.0, t_start + offs[i] - time.perf_counter()))
        calls[i] = time.perf_counter()
        for o in model.tts(**sreq, stream=True):
            if firsts_t[i] is None:
                firsts_t[i] = time.perf_counter()
    ths = [threading.Thread(target=twork, args=(i,)) for i in range(n)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    log, model._sched_log = sorted(model._sched_log, key=lambda e: e[0]), None
    for i in (n // 2, n // 2 + 3):
        print(f'--- call {i}: first chunk after {(firsts_t[i] - calls[i]) * 1e3:.1f} ms; running streams at its start: '
              f'{sum(1 for j in range(n) if calls[j] is not None and calls[j] < calls[i] and firsts_t[j] is not None)} started before it')
        for tt, kind, info in log:
            if calls[i] - 0.002 <= tt <= firsts_t[i] + 0.002:
                print(f'{(tt - calls[i]) * 1e3:8.2f} ms  {kind:7s} {info}')
    sys.exit(0)
SETTINGS = (('hold off', 0.0, 40.0, 3), ('hold 80 ms, no window (round 5)', 80.0, 1e6, 3), ('hold 80 ms, window 40 ms (default)', 80.0, 40.0, 3),
            ('hold off', 0.0, 40.0, 3), ('hold 80 ms, window 40 ms (default)', 80.0, 40.0, 3))
if '--beside' in sys.argv:      # CosyVoice2Model.newcomer_beside (a newcomer's prefill and first tokens beside the chunk round in progress) off / on
    SETTINGS = (('both switches off (rounds 1-5)', 80.0, 40.0, 0), ('newcomer_beside', 80.0, 40.0, 1), ('newcomer_beside + first-chunk lane', 80.0, 40.0, 3),
                ('both switches off (rounds 1-5)', 80.0, 40.0, 0), ('newcomer_beside', 80.0, 40.0, 1), ('newcomer_beside + first-chunk lane', 80.0, 40.0, 3))
if '--giveway' in sys.argv:     # (d) later chunks give way to a newcomer about to submit: off / 15 / 30 ms, on top of the defaults
    SETTINGS = (('defaults', 80.0, 40.0, 3), ('+ later chunks give way 15 ms', 80.0, 40.0, 3 + 4 * 15), ('+ later chunks give way 30 ms', 80.0, 40.0, 3 + 4 * 30),
                ('defaults', 80.0, 40.0, 3), ('+ later chunks give way 15 ms', 80.0, 40.0, 3 + 4 * 15), ('+ later chunks give way 30 ms', 80.0, 40.0, 3 + 4 * 30))
for name, hold, window, beside in SETTINGS:
    model.first_round_hold_ms, model.first_round_hold_window_ms = hold, window
    model.newcomer_beside, model.first_chunk_lane, model.later_chunk_wait_ms = bool(int(beside) & 1), bool(int(beside) & 2), float(int(beside) >> 2)
    rng = random.Random(1986)
    offs, t = [], 0.0
    for _ in range(n):
        t += rng.expovariate(1.0 / gap) * 1e-3
        offs.append(t)
    firsts, audio, errs = [None] * n, [0.0] * n, []
    t_start = time.perf_counter()

    def work(i):
        try:
            time.sleep(max(0.0, t_start + offs[i] - time.perf_counter()))
            t_call = time.perf_counter()
            for o in model.tts(**sreq, stream=True):
                if firsts[i] is None:
                    firsts[i] = time.perf_counter() - t_call
                audio[i] += o['tts_speech'].shape[1] / 24000.0
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    wall = time.perf_counter() - t_start
    assert not errs, errs[0]
    f = sorted(x * 1e3 for x in firsts)
    print(f'{name:38s}: {n} calls, mean gap {gap:.0f} ms: first chunk p50 {f[n // 2]:6.1f}  p90 {f[int(0.9 * n)]:6.1f}  max {f[-1]:6.1f} ms; '
          f'{sum(audio) / wall:6.1f} audio-s/s over {wall:.1f} s', flush=True)
