#!/usr/bin/env python3
"""Headline benchmark: CosyVoice2-0.5B-EU zero-shot synthesis throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N = 1 (default): a "step" = ONE utterance of BASELINE configs[1] (zero-shot FR, batch 1, bf16 weights, non-streaming) through the
  PRODUCT call `CosyVoice2Model.tts()` — scheduler, LLM prefill + autoregressive decode with the RAS sampler, flow encoder + 10
  Euler steps with CFG, HiFT vocoder, and the device->host copy of the waveform.  Inputs (text ids, prompt speech tokens, prompt
  mel, speaker embedding) are resident in HBM before the timed region.  Weights are synthetic at the real shapes (no checkpoints
  offline), decode length forced to 250 tokens = 10 s of audio.  The same run also measures, as `extra`:
  configs[2] (32 concurrent calls coalesced into one batch, FR + DE, ragged lengths), configs[4] (8 concurrent streams: p50
  first-chunk latency) and the CPU fp32 oracle on one whole utterance (`cpu_baseline`).
N > 1: configs[3] — 32 N utterances sharded over N ranks (one process per GPU): rank 0 broadcasts the shared prompt, scatters the
  text ids, every rank synthesises its 32 utterances as the evaluation harness does (a thread pool calling tts(), coalesced into
  one batch by the scheduler), rank 0 gathers the waveforms; RCCL broadcast / scatter / gather only, nothing per step.  The driver
  launches the ranks with torch.distributed.run; started WITHOUT a launcher (`python bench.py --gpus N`) this script spawns that
  launcher as a child process before anything touches the GPU.  Weak scaling: per-GPU work (32 utterances) is fixed; the N = 1
  reference for it is the N = 1 line's `extra.batch32`.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel group, the LLM decode step (HBM-bound weight stream):
algorithmic bytes per step / average step duration from HIP events recorded on the LLM's launch stream around every
decode burst of the timed region.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))

N_TOK = 250          # generated speech tokens per utterance (10 s)
P_TOK = 255          # FR prompt: 10.22 s -> 255 speech tokens
P_TOK_DE = 310       # DE prompt: 12.42 s
TEXT_LEN = 50
PROMPT_TEXT_LEN = 20
PER_GPU = 32         # utterances per GPU in the sharded configuration (configs[3]: 256 over 8 GPUs)
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK = 2500.0   # TFLOP/s dense
FP32_MFMA_PEAK = 157.3


def llm_step_bytes(weight_bytes, kv_positions_sum, layers=24, n_kv=2, kv_elem_bytes=4):
    # bf16 weights once per step + the KV read: layers x {k, v} x kv heads x 64 x element size per cached position.
    # kv_elem_bytes = 4: what this implementation reads (fp32 cache, 24 576 B per position); 2: SURVEY.md §8(d)'s figure (bf16 cache, 12 288 B)
    return weight_bytes + layers * 2 * n_kv * 64 * kv_elem_bytes * kv_positions_sum


def hift_products():
    """bf16 MFMA products the vocoder's convolutions issue per fp32-equivalent term: 3 (two planes per operand, the default), 6 (CV2_HIFT_PLANES=3);
    None with CV2_HIFT_FP32=1 (fp32 matrix cores)."""
    if os.environ.get('CV2_HIFT_FP32', '0')[:1] == '1':
        return None
    return 6 if os.environ.get('CV2_HIFT_PLANES', '2')[:1] == '3' else 3


def hift_mfma_fields(alg_tflops):
    """The conv stack's matrix-core rate: issued FLOP/s (algorithmic x products per term) against the roof of the instructions it issues."""
    n = hift_products()
    if n is None:
        return {'achieved': round(alg_tflops, 2), 'peak': FP32_MFMA_PEAK, 'unit': 'TFLOP/s', 'frac': round(alg_tflops / FP32_MFMA_PEAK, 4),
                'note': 'CV2_HIFT_FP32=1: fp32 matrix cores, algorithmic FLOP/s against the fp32 MFMA roof'}
    return {'achieved': round(n * alg_tflops, 1), 'peak': MFMA_BF16_PEAK, 'unit': 'TFLOP/s', 'frac': round(n * alg_tflops / MFMA_BF16_PEAK, 4),
            'algorithmic_tflops': round(alg_tflops, 2), 'products_per_term': n,
            'note': f'issued bf16 MFMA FLOP/s of the conv stack ({n} plane products per fp32-equivalent term of the 30.6 GFLOP per audio-second) against '
                    'the bf16 roof its instructions run on; the f0 predictor (3.3 of the 30.6) runs on the fp32 matrix cores'}


def rows_breakdown(rec):
    """Decode bursts by live-row count: {group: {'steps', 'avg_step_us'}} for the row-count families of the decode step (<= 8 rows: k_step /
    k_step2, 9-24: k_step4, 25-32: the launches); rec = StepTimer.rec after a device synchronize."""
    groups = (('rows_25_32', 25, 32), ('rows_9_24', 9, 24), ('rows_1_8', 1, 8))
    out = {}
    for name, lo, hi in groups:
        sel = [(e0.elapsed_time(e1), k) for e0, e1, n, k in rec if lo <= n <= hi]
        steps = sum(k for _, k in sel)
        out[name] = {'steps': steps, 'avg_step_us': round(sum(ms for ms, _ in sel) / steps * 1e3, 1) if steps else None}
    return out


def flow_flops(T):
    # estimator call (CFG pair) = 0.2644 GFLOP*T + 229376*T^2 ; 10 Euler steps ; + encoder ~80 GFLOP at T=1010
    return (0.2644e9 * T + 229376.0 * T * T) * 10 + 80e9 * (T / 1010.0)


def host_cores():
    """CPU share of this process: cgroup quota if any, else the affinity mask (a GPU box exposes 256 logical CPUs but grants ~16 to
    one GPU's container; oversubscribing torch threads makes the baseline meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(n_threads):
    """The CPU fp32 oracle (oracle/, a port of the reference's algorithm) on ONE WHOLE configs[1] utterance: same inputs as the GPU
    run, 250 forced greedy tokens through 24 layers, flow encoder + 10 Euler steps with CFG at T = 1010, HiFT on 500 frames."""
    import torch
    from cv2amd import synth
    from oracle import llm as OL, flow as OF, hift as OH
    torch.set_num_threads(n_threads)
    inp = synth.synthetic_inputs(text_len=TEXT_LEN, prompt_len=P_TOK, prompt_text_len=PROMPT_TEXT_LEN)
    with torch.inference_mode():
        sd = synth.make_llm()
        t0 = time.time()
        toks = OL.inference(sd, inp['text'], inp['prompt_text'], inp['prompt_token'], force_len=N_TOK)
        t_llm = time.time() - t0
        del sd
        fsd = synth.make_flow()
        t0 = time.time()
        mel = OF.inference(fsd, torch.tensor([toks], dtype=torch.int32), inp['prompt_token'], inp['prompt_feat'], inp['embedding'], False, True)
        t_flow = time.time() - t0
        del fsd
        hsd = synth.make_hift()
        g = torch.Generator().manual_seed(1)
        T = mel.shape[2]
        t0 = time.time()
        wav, _ = OH.inference(hsd, mel, torch.zeros(1, 1, 0), torch.rand(1, 9, generator=g), torch.randn(1, 480 * T, 9, generator=g))
        t_hift = time.time() - t0
    total = t_llm + t_flow + t_hift
    audio_s = wav.shape[1] / 24000.0
    return {'value': round(audio_s / total, 4), 'unit': 'audio-s/s', 'cores': n_threads, 'kind': 'port',
            'sample': f'oracle/ (torch CPU fp32) on one whole configs[1] utterance ({audio_s:.1f} s of audio): LLM prefill + {N_TOK} decode steps '
                      f'{t_llm:.1f}s, flow (encoder + 10 Euler steps, T={2 * (P_TOK + N_TOK)}) {t_flow:.1f}s, HiFT {T} frames {t_hift:.1f}s; '
                      f'total {total:.1f}s (RTF {total / audio_s:.2f}); nothing extrapolated'}


# ------------------------------------------------------------------------------------------------ helpers (GPU ranks)
class StepTimer:
    """HIP events on the LLM's launch stream around every decode burst (`LLMEngine.step` runs under `torch.cuda.stream(llm_stream)`
    in the scheduler, or on the current stream in a coalesced batch: the events are recorded on whichever stream is current there,
    i.e. the stream the graph replays are launched on)."""

    def __init__(self, llm):
        import torch
        self.torch, self.llm, self.rec, self.on = torch, llm, [], False
        self._orig, self._orig_rows = llm.step, llm.step_rows
        llm.step, llm.step_rows = self._step, self._step_rows

    def _step(self, n_seqs, n_steps=1, **kw):
        if not self.on:
            return self._orig(n_seqs, n_steps, **kw)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        self._orig(n_seqs, n_steps, **kw)
        e1.record()
        self.rec.append((e0, e1, n_seqs, n_steps))

    def _step_rows(self, slots, n_steps=1, **kw):                # decode over the live slots only (a batch past its first finished request)
        if not self.on:
            return self._orig_rows(slots, n_steps, **kw)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        self._orig_rows(slots, n_steps, **kw)
        e1.record()
        self.rec.append((e0, e1, len(slots), n_steps))

    def result(self):
        """(total ms, total steps) of the recorded bursts; call after a device synchronize."""
        ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in self.rec)
        return ms, sum(k for _, _, _, k in self.rec)


class StageTimer:
    """Events on the current stream around a bound method (flow / HiFT stage split of the B = 1 run; diagnostics only)."""

    def __init__(self, obj, name):
        import torch
        self.torch, self.rec, self.on = torch, [], False
        self._orig = getattr(obj, name)
        setattr(obj, name, self._call)

    def _call(self, *a, **k):
        if not self.on:
            return self._orig(*a, **k)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = self._orig(*a, **k)
        e1.record()
        self.rec.append((e0, e1))
        return r

    def ms(self):
        return sum(e0.elapsed_time(e1) for e0, e1 in self.rec)


def request(seed, prompt_len, text_len, dev, prompt_text_len=PROMPT_TEXT_LEN):
    """model_input of frontend_zero_shot (cli/frontend.py:504-512) on synthetic data, resident on the device."""
    from cv2amd import synth
    inp = synth.synthetic_inputs(seed=seed, text_len=text_len, prompt_len=prompt_len, prompt_text_len=prompt_text_len)
    d = dev
    return dict(text=inp['text'].to(d), prompt_text=inp['prompt_text'].to(d), llm_prompt_speech_token=inp['prompt_token'].to(d),
                flow_prompt_speech_token=inp['prompt_token'].to(d), prompt_speech_feat=inp['prompt_feat'].to(d),
                flow_embedding=inp['embedding'].to(d), llm_embedding=inp['embedding'].to(d))


class CallerPool:
    """The callers of concurrent tts() calls: persistent worker threads, as in the reference's harness and servers (a ThreadPoolExecutor in
    evaluation/cosyvoice_synthesizer.py:260 and runtime/python/grpc/server.py).  A thread's FIRST device copy pays the per-thread HIP
    initialisation (20-40 ms on this image, during which it holds the GIL): with fresh threads per round that lands inside every first
    chunk (8 streams: p50 146 ms against 108 ms on the same box).  fresh=True (--fresh-threads) starts new threads per round instead."""
    fresh = False
    _inst = None

    def __init__(self):
        import queue
        self._queue, self.q, self.done = queue, [], queue.Queue()

    @classmethod
    def run(cls, fn, n):
        if cls.fresh:
            ths = [threading.Thread(target=fn, args=(i,)) for i in range(n)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            return
        if cls._inst is None:
            cls._inst = cls()
        self = cls._inst
        while len(self.q) < n:
            q = self._queue.Queue()
            self.q.append(q)
            threading.Thread(target=self._loop, args=(q, len(self.q) - 1), daemon=True).start()
        for i in range(n):
            self.q[i].put(fn)
        errs = [e for e in (self.done.get() for _ in range(n)) if e is not None]
        if errs:
            raise errs[0]

    def _loop(self, q, i):
        while True:                                   # a worker survives a failing call: the round's first error is re-raised by run()
            fn = q.get()
            try:
                fn(i)
                self.done.put(None)
            except BaseException as e:      # noqa: BLE001
                self.done.put(e)


def run_calls(model, reqs, forces, stream=False, chunk_times=None):
    """The evaluation harness pattern (evaluation/cosyvoice_synthesizer.py:219,260): one thread per utterance calling tts() on ONE
    model.  Returns (list of waveforms [1, n] CPU, list of first-yield times relative to the common start); chunk_times (a list of n
    lists) receives the yield time of every chunk."""
    n = len(reqs)
    if n == 1:
        t0 = time.perf_counter()
        outs, first = [], None
        for o in model.tts(**reqs[0], stream=stream, force_len=forces[0]):
            first = first if first is not None else time.perf_counter() - t0
            if chunk_times is not None:
                chunk_times[0].append(time.perf_counter() - t0)
            outs.append(o['tts_speech'])
        import torch
        return [outs[0] if len(outs) == 1 else torch.cat(outs, 1)], [first]
    import torch
    wavs, first, errs = [None] * n, [None] * n, []
    t0 = time.perf_counter()

    def work(i):
        try:
            outs = []
            for o in model.tts(**reqs[i], stream=stream, force_len=forces[i]):
                if first[i] is None:
                    first[i] = time.perf_counter() - t0
                if chunk_times is not None:
                    chunk_times[i].append(time.perf_counter() - t0)
                outs.append(o['tts_speech'])
            wavs[i] = outs[0] if len(outs) == 1 else torch.cat(outs, 1)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    CallerPool.run(work, n)
    if errs:
        raise errs[0]
    return wavs, first


def pmc_stage_file(stage):
    """The newest committed profiles/r<N>_pmc_<stage>.json (tools/pmc_stages.sh: rocprofv3 --pmc passes of one stage alone) or None.
    The dict carries its own file name ('_file'): counters cannot be read from inside the bench, so every field derived from them is
    stamped with where it came from and 'measured_in_this_run': False."""
    for rnd in (6, 5, 4, 3):
        p = os.path.join(ROOT, 'profiles', f'r{rnd}_pmc_{stage}.json')
        if os.path.exists(p):
            d = json.load(open(p))
            d['_file'] = f'profiles/r{rnd}_pmc_{stage}.json'
            return d
    return None


_PMC_DEADLINE = [None]               # wall-clock end of the live counter passes' common budget (set by live_counters)


def _pmc_pass(stage, reps, counters, timeout_s=150):
    """One `rocprofv3 --pmc <counters>` run of tools/prof_stage_run.py <stage> <reps> as a child process (no trace domain beside --pmc; the
    program directly behind `--`).  Returns {kernel: {'n': dispatches, 'ns': summed duration, counter: summed value}} or None."""
    import collections, csv, glob, shutil, tempfile
    tmp = tempfile.mkdtemp(prefix='cv2_pmc_', dir='/tmp')
    try:
        cmd = ['rocprofv3', '--pmc'] + list(counters) + ['--output-format', 'csv', '-d', tmp, '--', sys.executable,
                                                         os.path.join(ROOT, 'tools', 'prof_stage_run.py'), stage, str(reps)]
        # the profiler and the program behind `--` run as their own process group: on a time-out the WHOLE group is killed and reaped, so no
        # orphaned prof_stage_run.py keeps the GPU busy beside the timed region; one overall budget for all passes (_PMC_DEADLINE)
        left = _PMC_DEADLINE[0] - time.time() if _PMC_DEADLINE[0] is not None else timeout_s
        if left < 5:
            return None
        pr = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            rc = pr.wait(timeout=min(timeout_s, left))
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            pr.wait()
            _PMC_DEADLINE[0] = 0.0                                            # a hung profiler: no further passes (committed counter files instead)
            return None
        if rc != 0:
            return None
        acc, seen = collections.defaultdict(lambda: collections.defaultdict(float)), set()
        for f in glob.glob(tmp + '/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                k = row['Kernel_Name'].split('(')[0].replace('void ', '').strip()
                acc[k][row['Counter_Name']] += float(row['Counter_Value'])
                if (f, row['Dispatch_Id']) not in seen:
                    seen.add((f, row['Dispatch_Id']))
                    acc[k]['n'] += 1
                    acc[k]['ns'] += float(row['End_Timestamp']) - float(row['Start_Timestamp'])
        return {k: dict(v) for k, v in acc.items()} or None
    except Exception:                   # noqa: BLE001 -- a profiler problem must not fail the benchmark
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_counters():
    """The counter-based fields of the line measured NOW: each stage alone (tools/prof_stage_run.py) under `rocprofv3 --pmc`, one pass per
    counter group, as child processes that finish before this process touches the GPU.  FETCH_SIZE is doubled (gfx950 tallies 128-byte
    requests at 64 B, MI355X_MICROARCH.md); FETCH_SIZE / WRITE_SIZE are in KiB.  Returns {} when rocprofv3 is missing or the bench itself
    runs under a profiler; a stage whose pass fails is left out (the committed counter files of profiles/ are used for it)."""
    import shutil
    out = {}
    if shutil.which('rocprofv3') is None or os.environ.get('CV2_BENCH_LIVE_PMC', '1') == '0':
        return out
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):          # already inside a profiler run (tools/prof_b1.sh)
        return out
    how = 'measured in this run: rocprofv3 --pmc, one pass per counter, child processes before the timed region; FETCH_SIZE x 2 (gfx950 correction)'
    # all passes together get CV2_BENCH_PMC_BUDGET_S (default 300 s: ~60 s of passes + a fresh box's first `import torch`); a pass that
    # times out ends the live measurement (its process group is killed) and the committed counter files stand in for what is missing
    _PMC_DEADLINE[0] = time.time() + float(os.environ.get('CV2_BENCH_PMC_BUDGET_S', '300'))
    f, w = _pmc_pass('decode', 1, ['FETCH_SIZE'], timeout_s=300), None       # (the first child also pays a fresh box's first `import torch`)
    if f is None:
        return out                                                           # the profiler does not work here: do not try the other passes
    w = _pmc_pass('decode', 1, ['WRITE_SIZE'])
    ks = [k for k in f if k.startswith('k_step')]
    if w is not None and ks and ks[0] in w:
        k = ks[0]
        per = (2.0 * f[k]['FETCH_SIZE'] / f[k]['n'] + w[k]['WRITE_SIZE'] / w[k]['n']) * 1024.0
        out['decode'] = (int(per), f'{how}; {int(f[k]["n"])} {k} launches of the decode stage alone')
    f, w = _pmc_pass('hift', 3, ['FETCH_SIZE']), _pmc_pass('hift', 3, ['WRITE_SIZE'])
    if f is not None and w is not None:
        rd = sum(2.0 * v.get('FETCH_SIZE', 0.0) for v in f.values()) * 1024.0 / 3
        wr = sum(v.get('WRITE_SIZE', 0.0) for v in w.values()) * 1024.0 / 3
        ns = sum(v['ns'] for v in f.values()) / 3
        convs = {k: v for k, v in f.items() if k.startswith(('k_conv', 'k_respair'))}
        top = max(convs, key=lambda k: convs[k]['ns'], default=None)
        topd = None
        if top is not None:
            tb = (2.0 * f[top].get('FETCH_SIZE', 0.0) + w.get(top, {}).get('WRITE_SIZE', 0.0)) * 1024.0
            topd = {'kernel': top, 'hbm_gbs': round(tb / f[top]['ns'], 1), 'avg_us': round(f[top]['ns'] / f[top]['n'] / 1e3, 2)}
        out['hift'] = {'hbm_gbs': round((rd + wr) / ns, 1), 'hbm_GB_per_10s_audio': round((rd + wr) / 1e9, 3), 'survey_8d_GB_per_10s_audio': 0.28,
                       'conv_launches_per_call': round(sum(v['n'] for v in convs.values()) / 3), 'top_conv_kernel': topd, 'measured_in_this_run': True,
                       'source': how + '; 500 mel frames = 10 s of audio per call, 3 calls; bytes over the summed kernel time'}
    m = _pmc_pass('flow', 3, ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'])
    if m is not None:
        busy = sum(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) for v in m.values())
        ns = sum(v['ns'] for v in m.values())
        gui = sum(v.get('GRBM_GUI_ACTIVE', 0.0) for v in m.values()) / 8.0
        out['flow'] = {'util_counter': round(busy / (4 * 256 * ns * 2.4), 4), 'util_counter_gui_active': round(busy / (4 * 256 * gui), 4) if gui else None,
                       'measured_in_this_run': True,
                       'source': how + '; one utterance (T = 1010), 3 calls: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x cycles) over all kernels of the stage; '
                                 'cycles = kernel duration x 2.4 GHz (util_counter) or GRBM_GUI_ACTIVE / 8 (util_counter_gui_active)'}
    return out


def pmc_decode_traffic(desc):
    """HBM bytes per decode-step launch from the committed counter files: the one-launch step (k_step) of round 3, else the older
    per-step totals."""
    note = '(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch correction)'
    pj = pmc_stage_file('decode')
    if pj is not None and 'k_step' in desc:
        for r in pj['kernels']:
            if r['kernel'].startswith('k_step') and r.get('hbm_read_MB_per_rep') is not None:
                per = (r['hbm_read_MB_per_rep'] + r['hbm_write_MB_per_rep']) * 1e6 / r['launches_per_rep']
                return int(per), f'{pj["_file"]}, k_step {note}; NOT measured in this run'
    for name in ('r2_pmc_decode.json', 'r1_pmc_decode.json'):
        pmc = os.path.join(ROOT, 'profiles', name)
        if 'k_step' not in desc and os.path.exists(pmc):
            return json.load(open(pmc))['hbm_bytes_per_step'], f'profiles/{name} {note}'
    return None, None


def counter_fields(pmc_flow, pmc_hift, hift_tf):
    """Counter-based companions of the computed stage figures (north_star: rocprof-reported MFMA utilisation for the GEMM stage, HBM
    GB/s for the conv stacks).  From the committed PMC files; null when they are absent."""
    flow = hift = None
    if pmc_flow:
        flow = {'util_counter': pmc_flow.get('mfma_util_wall_time_weighted'), 'util_counter_gui_active': pmc_flow.get('mfma_util_time_weighted'),
                'hbm_read_MB_per_utt': pmc_flow['hbm_read_MB_per_rep'], 'hbm_write_MB_per_utt': pmc_flow['hbm_write_MB_per_rep'],
                'measured_in_this_run': False, 'source': pmc_flow['_file'] + ': SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x cycles), time-weighted over the stage\'s kernels; '
                          'cycles = kernel-trace duration x 2.4 GHz (util_counter) or GRBM_GUI_ACTIVE / 8 (util_counter_gui_active)'}
    if pmc_hift:
        tot = (pmc_hift['hbm_read_MB_per_rep'] + pmc_hift['hbm_write_MB_per_rep']) / 1e3
        convs = [r for r in pmc_hift['kernels'] if r['kernel'].startswith(('k_conv6', 'k_respair'))]
        k6 = max(convs, key=lambda r: r.get('us_per_rep') or 0, default={})     # the convolution kernel with the largest share of the stage
        hift = {'hbm_gbs': pmc_hift.get('hbm_GBs_over_kernel_time'), 'hbm_GB_per_10s_audio': round(tot, 3), 'survey_8d_GB_per_10s_audio': 0.28,
                'conv_launches_per_call': round(sum(r['launches_per_rep'] for r in pmc_hift['kernels'] if r['kernel'].startswith(('k_conv', 'k_respair')))),
                'top_conv_kernel': {'kernel': k6.get('kernel'), 'hbm_gbs': k6.get('hbm_GBs'), 'mfma_util_counter': k6.get('mfma_util_wall'), 'avg_us': k6.get('avg_us')},
                'measured_in_this_run': False, 'source': pmc_hift['_file'] + ' (500 mel frames = 10 s of audio per rep; FETCH_SIZE x 2 + WRITE_SIZE over the stage\'s kernel time); '
                          'SURVEY 8(d)\'s 0.28 GB assumes whole ResBlocks fused (1.4 GB unfused); here the (dilated conv, conv) pairs of the 64- and 128-channel '
                          'stages are one launch each (k_respair), the 256-channel stage is one launch per convolution: input + halo, the residual / MRF '
                          'accumulator read by the epilogue, and the weight planes once per L2 that serves the layer\'s channel tiles'}
    return flow, hift


def kv_positions_mean(L0s, forces):
    """mean over the decode steps of (sum over live sequences of cached positions): step i of a sequence attends L0 + i keys."""
    n_steps = max(forces) - 1
    tot = 0
    for l0, f in zip(L0s, forces):
        tot += sum(l0 + i for i in range(1, f))
    return tot / n_steps, n_steps


def lm_rows(r):
    return 1 + r['prompt_text'].numel() + r['text'].numel() + 1 + r['llm_prompt_speech_token'].numel()


def build_model(dev, max_batch):
    from cv2amd import synth
    from cosyvoice.cli.model import CosyVoice2Model
    m = CosyVoice2Model(synth.make_llm(), synth.make_flow(), synth.make_hift(), device=dev, max_batch=max_batch, max_text=128,
                        max_prompt_tokens=320, max_new_tokens=512, coalesce_ms=0.0)
    return m


# ------------------------------------------------------------------------------------------------ N = 1
def run_single(args):
    # the live counter passes run first: child processes, done before this process initialises the GPU
    live = live_counters() if (args.batch == 1 and not args.no_extra and 'RANK' not in os.environ) else {}
    live_traffic = live.get('decode')
    import torch
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the hot path has no CPU fallback)'
    dev_index = int(os.environ.get('CV2_BENCH_DEVICE', os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    use_dist = 'RANK' in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', device_id=dev)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    B = args.batch
    model = build_model(dev, max(B, 1) if args.no_extra else 32)
    st = StepTimer(model.llm)
    flow_t, hift_t = StageTimer(model.flow, 'inference_batch'), StageTimer(model.hift_pool, 'inference_many')
    if model.max_batch == 1:                  # serial scheduler path: HiFT runs inside _mel2wav (slice + vocoder), not through the pool
        hift_t = StageTimer(model, '_mel2wav')
    if B == 1:
        reqs, forces = [request(1986, P_TOK, TEXT_LEN, dev)], [N_TOK]
    else:
        g = torch.Generator().manual_seed(1986)
        reqs = [request(1986 + b, P_TOK if b % 2 == 0 else P_TOK_DE, TEXT_LEN, dev) for b in range(B)]
        forces = [int(torch.randint(150, 501, (1,), generator=g)) for _ in range(B)]
    model.coalesce_ms = 0.0 if B == 1 else 20.0

    for _ in range(args.warmup):
        wavs, _ = run_calls(model, reqs, forces)
    barrier()
    st.on = flow_t.on = hift_t.on = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wavs, _ = run_calls(model, reqs, forces)          # yields CPU tensors: the device->host copy is inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    st.on = flow_t.on = hift_t.on = False
    assert all(w.device.type == 'cpu' and torch.isfinite(w).all() for w in wavs)
    assert [w.shape[1] for w in wavs] == [960 * f for f in forces], 'forced-length run must produce 2 frames x 480 samples per token'
    audio_per_step = sum(w.shape[1] for w in wavs) / 24000.0
    value = audio_per_step * args.steps / dt

    dec_ms, dec_steps = st.result()
    kv_mean, n_steps = kv_positions_mean([lm_rows(r) for r in reqs], forces)
    assert dec_steps == n_steps * args.steps, (dec_steps, n_steps, args.steps)
    step_us = dec_ms / dec_steps * 1e3
    step_bytes = llm_step_bytes(model.llm.weight_bytes, kv_mean)
    step_bytes_8d = llm_step_bytes(model.llm.weight_bytes, kv_mean, kv_elem_bytes=2)
    achieved = step_bytes / (step_us * 1e-6) / 1e9
    Ts = [2 * (r['flow_prompt_speech_token'].numel() + f) for r, f in zip(reqs, forces)]
    flow_tf = sum(flow_flops(T) for T in Ts) * args.steps / (flow_t.ms() * 1e-3) / 1e12
    hift_tf = 30.6e9 * audio_per_step * args.steps / (hift_t.ms() * 1e-3) / 1e12
    traffic, traffic_src = None, None
    if B == 1:          # PMC counters cannot be read from inside the bench: committed rocprofv3 --pmc measurements of the same kernels
        traffic, traffic_src = live_traffic if live_traffic is not None else pmc_decode_traffic(model.llm.decode_kernel_desc(B))
    pmc_flow, pmc_hift = pmc_stage_file('flow'), pmc_stage_file('hift')
    out = {
        'metric': 'audio-sec/sec, CosyVoice2-0.5B-EU zero-shot FR', 'value': round(value, 3), 'unit': 'audio-s/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'rtf': round(dt / (audio_per_step * args.steps), 5),
        'config': {'workload': f'configs[{1 if B == 1 else 2}]: zero-shot FR{"+DE" if B > 1 else ""}, batch={B}, non-streaming, through CosyVoice2Model.tts() '
                               f'(scheduler + D2H of the waveform inside the timed region), P=255{"/310" if B > 1 else ""} prompt tokens, '
                               f'{TEXT_LEN} text tokens, {"250" if B == 1 else "U{150..500}"} generated tokens (forced), 10 Euler steps + CFG, RAS sampler; '
                               f'LLM bf16 weights / fp32 KV, flow bf16 MFMA, HiFT fp32 data with two-plane bf16 MFMA products (three per term, ~2^-15; CV2_HIFT_PLANES=3: six, fp32-equivalent) and fp32 MFMA for the f0 predictor', 'batch_per_gpu': B,
                   'audio_s_per_step_per_gpu': round(audio_per_step, 3)},
        'roofline': {'bound': 'hbm', 'kernel': model.llm.decode_kernel_desc(B),
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                     'traffic': traffic, 'traffic_source': traffic_src, 'bytes_per_launch': int(step_bytes), 'avg_launch_us': round(step_us, 2),
                     # both byte counts: `achieved` / `frac` use what THIS implementation must read (fp32 KV cache: 24 576 B per cached
                     # position); SURVEY.md 8(d) prices a bf16 cache (12 288 B per position)
                     'bytes_per_launch_survey_8d': int(step_bytes_8d),
                     'frac_survey_8d_bytes': round(step_bytes_8d / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
        'stages': {'ms_per_step': {'llm_decode': round(dec_ms / args.steps, 3), 'flow': round(flow_t.ms() / args.steps, 3),
                                   'hift': round(hift_t.ms() / args.steps, 3),
                                   'rest (prefill, scheduler, D2H)': round((dt * 1e3 - dec_ms - flow_t.ms() - hift_t.ms()) / args.steps, 3)},
                   'flow_mfma': {'achieved': round(flow_tf, 1), 'peak': MFMA_BF16_PEAK, 'unit': 'TFLOP/s', 'frac': round(flow_tf / MFMA_BF16_PEAK, 4)},
                   'hift_mfma': hift_mfma_fields(hift_tf)},
    }
    cf, ch = counter_fields(pmc_flow, pmc_hift, hift_tf)
    if B == 1 and live.get('flow'):               # the live passes profile the configs[1] shapes: they belong to the B=1 line
        cf = live['flow']
    if B == 1 and live.get('hift'):
        ch = live['hift']
    if cf:
        out['stages']['flow_mfma'].update(cf)
    if ch:
        out['stages']['hift'] = ch
    if not args.no_extra and B == 1:
        out['extra'] = extras(model, st, flow_t, hift_t, dev)
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(host_cores())
    print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def extras(model, st, flow_t, hift_t, dev):
    """configs[2] and configs[4] measured in the same run, on the same model object."""
    import torch
    ex = {}
    # ---- configs[2]: 32 concurrent non-streaming calls (16 FR + 16 DE), ragged forced lengths, coalesced into one batch
    g = torch.Generator().manual_seed(1986)
    reqs = [request(1986 + b, P_TOK if b % 2 == 0 else P_TOK_DE, TEXT_LEN, dev) for b in range(32)]
    forces = [int(torch.randint(150, 501, (1,), generator=g)) for _ in range(32)]
    model.coalesce_ms = 50.0
    run_calls(model, reqs, forces)
    torch.cuda.synchronize()
    st.rec, flow_t.rec, hift_t.rec = [], [], []
    st.on = flow_t.on = hift_t.on = True
    n0 = len(model.batch_sizes)
    K = 2
    t0 = time.perf_counter()
    for _ in range(K):
        wavs, _ = run_calls(model, reqs, forces)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st.on = flow_t.on = hift_t.on = False
    audio = sum(w.shape[1] for w in wavs) / 24000.0
    dec_ms, dec_steps = st.result()
    kv_mean, _ = kv_positions_mean([lm_rows(r) for r in reqs], forces)
    step_us = dec_ms / dec_steps * 1e3
    sb = llm_step_bytes(model.llm.weight_bytes, kv_mean)
    sb8d = llm_step_bytes(model.llm.weight_bytes, kv_mean, kv_elem_bytes=2)
    Ts = [2 * (r['flow_prompt_speech_token'].numel() + f) for r, f in zip(reqs, forces)]
    ex['batch32'] = {'workload': 'configs[2]: 32 concurrent tts() calls on one model (16 FR P=255 + 16 DE P=310, U{150..500} forced tokens), coalesced',
                     'value': round(audio * K / dt, 2), 'unit': 'audio-s/s', 'rtf': round(dt / (audio * K), 5), 'ms_per_step': round(dt / K * 1e3, 1),
                     'batch_sizes': model.batch_sizes[n0:],
                     'roofline': {'bound': 'hbm', 'kernel': model.llm.decode_kernel_desc(32), 'achieved': round(sb / (step_us * 1e-6) / 1e9, 1),
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(sb / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                  'bytes_per_launch': int(sb), 'avg_launch_us': round(step_us, 1),
                                  'bytes_per_launch_survey_8d': int(sb8d), 'frac_survey_8d_bytes': round(sb8d / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                  # the batch's steps by live-row count (finished requests leave the decode rows): each family is its own kernel form
                                  'by_rows': rows_breakdown(st.rec)},
                     'stages_ms': {'llm_decode': round(dec_ms / K, 1), 'flow': round(flow_t.ms() / K, 1), 'hift': round(hift_t.ms() / K, 1)},
                     'flow_mfma_frac': round(sum(flow_flops(T) for T in Ts) * K / (flow_t.ms() * 1e-3) / 1e12 / MFMA_BF16_PEAK, 4),
                     'hift_mfma_frac': hift_mfma_fields(30.6e9 * audio * K / (hift_t.ms() * 1e-3) / 1e12)['frac']}
    # ---- configs[4]: streaming, 8 concurrent calls on one model; 12 text tokens -> at most 240 speech tokens per stream (EOS is live)
    sreq = request(1986, P_TOK, 12, dev)
    out = {}
    for n in (1, 8):
        run_calls(model, [sreq] * n, [None] * n, stream=True)            # warm-up: graphs for 1..n slots
        firsts, audio, dts = [], 0.0, 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            wavs, first = run_calls(model, [sreq] * n, [None] * n, stream=True)
            dts += time.perf_counter() - t0
            audio += sum(w.shape[1] for w in wavs) / 24000.0
            firsts += first
        firsts.sort()
        out[n] = {'first_chunk_ms_p50': round(firsts[len(firsts) // 2] * 1e3, 1), 'first_chunk_ms_max': round(firsts[-1] * 1e3, 1),
                  'audio_s_per_s': round(audio / dts, 1)}
        if n == 8 and not CallerPool.fresh:           # the same with new caller threads per round (what rounds 1-3 of this bench measured)
            CallerPool.fresh = True
            try:
                ff = []
                for _ in range(3):
                    ff += run_calls(model, [sreq] * n, [None] * n, stream=True)[1]
                ff.sort()
                out[n]['first_chunk_ms_p50_fresh_threads'] = round(ff[len(ff) // 2] * 1e3, 1)
            finally:
                CallerPool.fresh = False
    # calls that do NOT start together (a server's arrivals): every call of a round waits a random 0 .. 40 ms first; time from the call's OWN
    # start to its first chunk, 10 rounds of 8
    try:
        import random
        rng = random.Random(1986)
        sf = []
        for _ in range(10):
            offs = [rng.random() * 0.040 for _ in range(8)]
            firsts8 = [None] * 8

            def swork(i):
                time.sleep(offs[i])
                t_call = time.perf_counter()
                for o in model.tts(**sreq, stream=True):
                    if firsts8[i] is None:
                        firsts8[i] = time.perf_counter() - t_call
            CallerPool.run(swork, 8)
            sf += firsts8
        sf.sort()
        out[8]['first_chunk_ms_p50_staggered'] = round(sf[len(sf) // 2] * 1e3, 1)
        out[8]['first_chunk_ms_max_staggered'] = round(sf[-1] * 1e3, 1)
        out[8]['staggered'] = ('8 calls per round, each after a random 0-40 ms offset, 10 rounds; from the call\'s own start to its first chunk; '
                               f'first_round_hold_ms = {getattr(model, "first_round_hold_ms", None)}')
    except Exception as e:      # noqa: BLE001
        out[8]['staggered'] = repr(e)
    # first chunk when the prompt has NOT been seen before (no prompt flow cache to start from): the number a new voice gets
    if hasattr(model, '_prompt_caches'):
        keep_max = model.prompt_cache_max
        try:
            firsts = []
            for _ in range(3):
                model._prompt_caches.clear()                      # the model has never seen this prompt
                _, first = run_calls(model, [sreq] * 8, [None] * 8, stream=True)
                firsts += first
            firsts.sort()
            out[8]['first_chunk_ms_p50_new_prompt'] = round(firsts[len(firsts) // 2] * 1e3, 1)
        finally:
            model.prompt_cache_max = keep_max
    # the same with 250 forced tokens (10 s of audio, 9 chunks of 25 tokens after the first + the final call): time between chunks
    for n in (1, 8):
        run_calls(model, [sreq] * n, [250] * n, stream=True)
        gaps_early, gaps_late, audio, dts = [], [], 0.0, 0.0
        for _ in range(2):
            ct = [[] for _ in range(n)]
            t0 = time.perf_counter()
            wavs, _ = run_calls(model, [sreq] * n, [250] * n, stream=True, chunk_times=ct)
            dts += time.perf_counter() - t0
            audio += sum(w.shape[1] for w in wavs) / 24000.0
            for c in ct:                                   # chunk k arrives c[k]; non-final chunks are c[0..8], the final call c[9]
                g = [b - a for a, b in zip(c[:-2], c[1:-1])]
                gaps_early += g[:3]
                gaps_late += g[-3:]
        gaps_early.sort(); gaps_late.sort()
        out[n].update({'forced250_audio_s_per_s': round(audio / dts, 1),
                       'chunk_gap_ms_p50_chunks_2_4': round(gaps_early[len(gaps_early) // 2] * 1e3, 1),
                       'chunk_gap_ms_p50_chunks_8_10': round(gaps_late[len(gaps_late) // 2] * 1e3, 1)})
    # ---- configs[4] as BASELINE words it: "bistream LLM + chunk-CFM, batch=8" -- the text of every call is a Python GENERATOR (llm_job's
    # bistream branch, cli/model.py:120-128): 120 text tokens arriving in pieces of 5, interleaved 5 text : 15 speech tokens on the device.
    # Random weights never draw the fill / EOS ids on their own: for this leg the decoder bias of the three special ids is raised (the idea
    # of tests/test_fullsize_gpu.py:_bistream_sd; EOS +24, fill +6: the fill id appears once on its own and is forced every
    # 16 entries from then on, llm.py:783-804; in the final decode EOS always beats it) and the sampler is the greedy harness
    # (with RAS an EOS that dominates while text is still expected exhausts the 100 re-draws of llm.py:242-250); both are put back afterwards.
    bis = None
    keep_mode = model.sampling_mode
    try:
        from cv2amd.llm import MODE_GREEDY
        bd = model.llm.bdec
        keep = bd[6561:6564].clone()
        bd[6563] += 6.0; bd[6561] += 24.0; bd[6562] = -30.0
        model.sampling_mode = MODE_GREEDY
        breq = request(1986, P_TOK, 120, dev)          # the first 17 blocks of 5 text tokens carry the 255 prompt speech tokens (5 : 15)
        pieces = [breq['text'][:, i:i + 5] for i in range(0, 120, 5)]

        def bcall(n):
            firsts, audio = [None] * n, [0.0] * n
            errs = []
            t0 = time.perf_counter()

            def work(i):
                try:
                    kw = dict(breq)
                    kw['text'] = (p for p in pieces)
                    for o in model.tts(**kw, stream=True):
                        if firsts[i] is None:
                            firsts[i] = time.perf_counter() - t0
                        audio[i] += o['tts_speech'].shape[1] / 24000.0
                except Exception as e:      # noqa: BLE001
                    errs.append(e)
            CallerPool.run(work, n)
            if errs:
                raise errs[0]
            return firsts, sum(audio), time.perf_counter() - t0
        bcall(8)
        fs, au, dts = [], 0.0, 0.0
        for _ in range(3):
            f, a, d = bcall(8)
            fs += f; au += a; dts += d
        fs.sort()
        bis = {'workload': '8 concurrent tts(text=<generator>, stream=True) calls (llm_job bistream branch): 120 text tokens in pieces of 5, P=255, greedy harness, EOS live',
               'first_chunk_ms_p50': round(fs[len(fs) // 2] * 1e3, 1), 'first_chunk_ms_max': round(fs[-1] * 1e3, 1),
               'audio_s_per_s': round(au / dts, 1), 'audio_s_per_call': round(au / 24, 2)}
    except Exception as e:      # noqa: BLE001
        bis = {'error': repr(e)}
    finally:
        model.llm.bdec[6561:6564] = keep
        model.sampling_mode = keep_mode
        torch.cuda.synchronize()
    ex['streaming'] = {'flow_cache': bool(getattr(model, 'flow_cache', False)), 'bistream_8': bis,
                       'prompt_cache': 'the streams share a prompt the model has served before: its whole chunks come from the prompt flow cache '
                                       '(first_chunk_ms_p50_new_prompt: without it)' if getattr(model, 'prompt_cache_max', 0) > 0 else 'off',
                       'workload': 'configs[4]: streaming (hop 25, look-ahead 3), P=255: the first chunk needs prefill + 48 tokens, chunk-masked flow at '
                                   'T=600, HiFT on 90 frames; time from the tts() call to its first yielded chunk, 3 rounds',
                       'callers': ('new threads per round' if CallerPool.fresh else 'persistent worker threads (CallerPool: a server\'s / the evaluation harness\'s '
                                   'thread pool); first_chunk_ms_p50_fresh_threads = the same calls from new threads per round, each paying its first '
                                   'device copy\'s per-thread HIP initialisation (20-40 ms, GIL held) inside the first chunk'),
                       'callers_streams_1': 'one call from the benchmark\'s own thread (no pool, no new thread): the same in every round of this bench',
                       'streams_1': out[1], 'streams_8': out[8]}
    return ex


# ------------------------------------------------------------------------------------------------ N > 1: configs[3]
def run_sharded(args):
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    backend = os.environ.get('CV2_BENCH_BACKEND', 'nccl')
    fake = os.environ.get('CV2_BENCH_FAKE_SYNTH') == '1'          # CPU plumbing test (tests/test_host_cpu.py): no model, gloo
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    if fake:
        dev = torch.device('cpu')
        dist.init_process_group(backend)
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (the hot path has no CPU fallback)'
        if 'CV2_BENCH_DEVICE' not in os.environ and torch.cuda.device_count() < world:
            sys.exit(f'bench.py: --gpus {world} but only {torch.cuda.device_count()} device(s) are visible (one process per GPU)')
        dev_index = int(os.environ.get('CV2_BENCH_DEVICE', os.environ.get('LOCAL_RANK', 0)))
        torch.cuda.set_device(dev_index)
        dev = torch.device('cuda', dev_index)
        dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
    from cv2amd import shard, synth

    def sync():
        if not fake:
            torch.cuda.synchronize()

    def barrier():
        sync()
        dist.barrier()
        sync()

    per = int(os.environ.get('CV2_BENCH_PER_GPU', PER_GPU))
    n_utts = per * world
    # rank 0 owns the inputs: texts of 30..100 tokens (the forced speech length is 5 x the text length: 150..500 tokens, so the
    # length-balanced dealing of shard.assign balances the work), one shared FR prompt
    texts, prompt = None, None
    if rank == 0:
        g = torch.Generator().manual_seed(1986)
        texts = [torch.randint(0, 151000, (int(torch.randint(30, 101, (1,), generator=g)),), generator=g, dtype=torch.int32) for _ in range(n_utts)]
        inp = synth.synthetic_inputs(seed=1986, text_len=1, prompt_len=P_TOK, prompt_text_len=PROMPT_TEXT_LEN)
        prompt = dict(prompt_text=inp['prompt_text'], prompt_token=inp['prompt_token'], prompt_feat=inp['prompt_feat'], embedding=inp['embedding'])
    model = None if fake else build_model(dev, per)
    if model is not None:
        model.coalesce_ms = 50.0
    st = StepTimer(model.llm) if model is not None else None

    work = {'s': 0.0, 'utts': 0, 'audio_s': 0.0}          # this rank's own synthesis time inside the timed steps (collectives excluded)

    def synth_fn(my_texts, p):
        t_in = time.perf_counter()
        try:
            return synth_inner(my_texts, p)
        finally:
            sync()
            if timing['on']:
                work['s'] += time.perf_counter() - t_in
                work['utts'] += len(my_texts)
                work['audio_s'] += sum(960 * 5 * t.numel() for t in my_texts) / 24000.0
    timing = {'on': False}

    def synth_inner(my_texts, p):
        if fake:
            return [torch.full((960 * 5 * t.numel(),), float(t.numel())) for t in my_texts]
        base = dict(prompt_text=p['prompt_text'].to(dev), llm_prompt_speech_token=p['prompt_token'].to(dev),
                    flow_prompt_speech_token=p['prompt_token'].to(dev), prompt_speech_feat=p['prompt_feat'].to(dev),
                    flow_embedding=p['embedding'].to(dev), llm_embedding=p['embedding'].to(dev))
        reqs = [dict(base, text=t.reshape(1, -1).to(dev), device_output=True) for t in my_texts]     # the waveforms stay in HBM for the gather
        wavs, _ = run_calls(model, reqs, [5 * t.numel() for t in my_texts])
        return [w.reshape(-1) for w in wavs]

    for _ in range(args.warmup):
        shard.synthesize_sharded(texts, prompt, synth_fn)
    barrier()
    if st is not None:
        st.on = True
    timing['on'] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        waves = shard.synthesize_sharded(texts, prompt, synth_fn)
    barrier()
    dt = time.perf_counter() - t0
    timing['on'] = False
    # bench bookkeeping only (the product path has no all-reduce): every rank's wall time and own work, gathered as objects
    mine = {'rank': rank, 'wall_s': round(dt, 4), 'work_s': round(work['s'], 4), 'utts_per_step': work['utts'] // max(args.steps, 1),
            'audio_s_per_step': round(work['audio_s'] / max(args.steps, 1), 2),
            'device': 'cpu' if fake else f'cuda:{dev.index} ' + torch.cuda.get_device_name(dev)}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    dt = max(r['wall_s'] for r in per_rank)
    if rank == 0:
        assert len(waves) == n_utts and all(w.numel() == 960 * 5 * x.numel() for w, x in zip(waves, texts)), 'gather returned the wrong waveforms'
        audio_per_step = sum(w.numel() for w in waves) / 24000.0
        out = {'metric': 'audio-sec/sec, CosyVoice2-0.5B-EU zero-shot FR', 'value': round(audio_per_step * args.steps / dt, 3), 'unit': 'audio-s/s',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'FAKE (plumbing test, no model)' if fake else 'synthetic',
               'rtf': round(dt / (audio_per_step * args.steps), 5),
               'ranks_seen': dist.get_world_size(),
               'backend': (f'nccl = RCCL {".".join(map(str, torch.cuda.nccl.version()))} over xGMI' if backend == 'nccl' and not fake else backend),
               'per_rank': per_rank,
               'imbalance': round(max(r['work_s'] for r in per_rank) / max(min(r['work_s'] for r in per_rank), 1e-9), 3),
               # the one-GPU reference measured INSIDE this run: what each rank's own shard (its utterances, no collectives) ran at; the
               # committed N = 1 line's extra.batch32.value (the same 32-utterance workload through the same path) is quoted beside it
               'n1_reference': n1_reference(per_rank, args.steps),
               'config': {'workload': f'configs[3]: {n_utts} utterances sharded over {world} ranks ({per} per GPU, length-balanced), zero-shot FR, P=255, texts of '
                                      f'30..100 tokens, 5 x text length forced speech tokens (150..500), non-streaming; broadcast prompt / scatter text ids / '
                                      f'gather waveforms over {"RCCL" if backend == "nccl" else backend} inside the timed region; every rank runs its shard as '
                                      f'one coalesced batch through CosyVoice2Model.tts()', 'batch_per_gpu': per,
                          'audio_s_per_step_total': round(audio_per_step, 2),
                          'scaling_reference': f"per-GPU work is fixed at {per} utterances: compare with N x the N=1 line's extra.batch32.value (32 per GPU there)"}}
        if st is not None:
            sync()
            dec_ms, dec_steps = st.result()
            out['roofline'] = {'bound': 'hbm', 'kernel': model.llm.decode_kernel_desc(per) + ' (rank 0)', 'avg_launch_us': round(dec_ms / max(dec_steps, 1) * 1e3, 1),
                               'achieved': None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': None, 'traffic': None,
                               'note': 'bytes per step depend on the shard; the per-GPU roofline is the N=1 line (extra.batch32.roofline)'}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def n1_reference(per_rank, steps):
    """Per-GPU throughput of the shards themselves (audio of a rank's shard / the time that rank spent synthesising it), and the
    committed one-GPU measurement of the same workload, so that the N > 1 line carries its own scaling reference."""
    rates = sorted(r['audio_s_per_step'] * steps / max(r['work_s'], 1e-9) for r in per_rank)
    ref = {'unit': 'audio-s/s per GPU', 'per_rank_shard_rate_median': round(rates[len(rates) // 2], 2), 'per_rank_shard_rate_min': round(rates[0], 2),
           'source': "each rank's own shard inside this run, collectives excluded"}
    for name in ('r6_bench_b1.json', 'r5_bench_b1.json', 'r4_bench_b1.json', 'r3_bench_b1.json'):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            with open(path) as f:
                line = json.loads(f.read().strip().splitlines()[-1])
            ref['committed_n1_batch32'] = {'value': line['extra']['batch32']['value'], 'file': 'profiles/' + name}
            break
        except Exception:      # noqa: BLE001
            continue
    return ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--batch', type=int, default=1, help='N=1 only: utterances per step (1 = configs[1], 32 = configs[2])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='N=1: skip the configs[2] / configs[4] measurements')
    ap.add_argument('--fresh-threads', action='store_true', help='concurrent calls from new threads per round instead of persistent workers (CallerPool)')
    args = ap.parse_args()
    CallerPool.fresh = bool(args.fresh_threads)
    world = int(os.environ.get('WORLD_SIZE', 1))
    # CV2_BENCH_FORCE_SHARDED=1 --gpus 1: the configs[3] path (run_sharded: RCCL process group, broadcast / scatter / gather on device tensors,
    # tts(device_output=True)) with a world of ONE rank -- the only way to execute the `nccl` branch of cv2amd/shard.py on a one-GPU box
    force_sharded = os.environ.get('CV2_BENCH_FORCE_SHARDED') == '1'
    if args.steps is None:
        args.steps = 6 if (args.gpus == 1 and not force_sharded) else 2
    if args.warmup is None:
        args.warmup = 2 if (args.gpus == 1 and not force_sharded) else 1
    if (args.gpus > 1 or force_sharded) and 'RANK' not in os.environ:
        # started without a launcher: spawn one rank per GPU as CHILD processes before anything in this process touches the GPU
        # (a process that has initialised the GPU must never exec another program on this pool)
        port = os.environ.get('MASTER_PORT', '29531')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus} '
                 f'(or without a launcher: the script spawns it)')
    from cv2amd import lib as L
    if not os.path.exists(L.LIB_PATH) and os.environ.get('CV2_BENCH_FAKE_SYNTH') != '1':
        import __graft_entry__
        __graft_entry__.build()
    if world == 1 and not force_sharded:
        run_single(args)
    else:
        run_sharded(args)


if __name__ == '__main__':
    main()
