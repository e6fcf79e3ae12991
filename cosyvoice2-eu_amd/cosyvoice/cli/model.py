"""`CosyVoice2Model` — the synthesis scheduler (replaces cosyvoice/cli/model.py:255-401 of the reference).

Same contract: `tts(**model_input, stream, speed)` is a generator of `{'tts_speech': float32 CPU tensor [1, n]}`;
per-call state lives in uuid-keyed dicts (`tts_speech_token_dict`, `llm_end_dict`, `hift_cache_dict`) created under
`self.lock` and released when the generator finishes (model.py:342-345, 395-398); `token2wav` slices / caches / cross-fades
exactly as model.py:300-334.

What is different, by design:
  * no LLM thread and no 100 ms polling loop: the LLM decode runs on its own HIP stream (the reference's `llm_context`
    side stream, model.py:278) in bursts of exactly the tokens the next chunk needs, and the next burst is enqueued BEFORE
    the current chunk's flow + HiFT are launched on the main stream, so the two overlap on the device while the host
    never sleeps; first-chunk latency is the work itself, not a poll quantum;
  * tokens stay on the device between the LLM and the flow (the reference round-trips a Python list, model.py:385);
  * the cross-fade runs on the device (the reference's fade_in_out moves both tensors to the CPU, utils/common.py:144);
  * one model object is shared by concurrent tts() calls (the evaluation harness calls it from up to 8 threads,
    evaluation/cosyvoice_synthesizer.py:219,260; BASELINE config 5 runs 8 streams at once) — SURVEY.md §8(b) "Threading", §8(f) rank 2:
      - STREAMING calls each own one LLM slot (KV cache + sampler state) and share the decode steps: whenever one of them needs
        tokens, a step advances every active slot (continuous batching of the LLM; a stream that already has its tokens simply
        runs ahead, as the reference's LLM thread does); each device operation (prefill, step, poll, token2wav of one chunk) is a
        short critical section under `self.run_lock` (FIFO), so the chunks of different streams interleave instead of queueing whole
        calls; calls that start together share one batched prefill, and every round of ready chunks runs the flow as one ragged
        batch and HiFT on the pool's HIP streams;
      - NON-STREAMING calls are COALESCED: each call queues its request, one caller becomes the batch leader, waits `coalesce_ms`
        for stragglers, and runs up to `max_batch` queued requests as one batch through the three stages (one decode step serves
        all of them, the flow runs over the packed ragged batch, HiFT on a pool of streams);
      - the two kinds exclude each other (a batch uses slots 0..n-1): streams hold the model in shared mode, a batch in exclusive mode.
"""
import os
import threading
import time
import uuid

import numpy as np
import torch

from cv2amd.flow import FlowEngine
from cv2amd.hift import HiftEngine, HiftPool
from cv2amd.llm import LLMEngine, MODE_RAS, MODE_GREEDY
from cv2amd import lib as L


class _FairLock:
    """FIFO mutex (ticket lock).  threading.Lock lets a thread that releases and immediately re-acquires win again and again: one
    stream would then run all its chunks before the others get their first.  With tickets the streams' operations go round-robin."""

    def __init__(self):
        self._cv = threading.Condition()
        self._next, self._serving = 0, 0

    def acquire(self):
        with self._cv:
            t = self._next
            self._next += 1
            while self._serving != t:
                self._cv.wait()

    def release(self):
        with self._cv:
            self._serving += 1
            self._cv.notify_all()

    __enter__ = lambda self: self.acquire()
    __exit__ = lambda self, *a: self.release()


class CosyVoice2Model:
    def __init__(self, llm_sd=None, flow_sd=None, hift_sd=None, fp16=False, device=None, max_text=512, max_prompt_tokens=750,
                 max_new_tokens=3000, sampling='ras', seed=0, max_batch=8, coalesce_ms=2.0, config=None):
        if not torch.cuda.is_available():
            raise L.Cv2Error('CosyVoice2Model (MI355X build) needs a GPU: the hot path has no CPU fallback')
        self.device = torch.device(device or 'cuda')
        if self.device.index is None:                  # set_device needs an index (threads started later switch to this device)
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.fp16 = fp16                       # accepted for signature compatibility; the HIP path fixes its own dtypes
        from cv2amd.config import Config
        self.config = config or Config()       # hyper-parameters of cosyvoice2.yaml (the reference receives built modules instead)
        self.min_token_text_ratio, self.max_token_text_ratio = 2, 20    # defaults of Qwen2LM.inference (llm.py:586-587)
        self.token_hop_len = 25                # must match the training static_chunk_size (model.py:271)
        self.mel_cache_len = 8
        self.source_cache_len = int(self.mel_cache_len * 480)
        self.speech_window = np.hamming(2 * self.source_cache_len)
        self.lock = threading.Lock()           # guards the per-uuid dicts, as in the reference
        self.run_lock = _FairLock()            # one device operation at a time (a chunk's work, or a whole coalesced batch), FIFO
        self._leader_lock = threading.Lock()   # election of the batch leader among queued non-streaming callers
        # the LLM engine has its own lock (taken INSIDE run_lock by the unistream paths, alone by the bistream hub): the hub's rounds
        # (poll / batched text-block feed / shared decode burst on the LLM stream) go on while a chunk round holds run_lock
        self.llm_lock = threading.Lock()
        self._bi_cv = threading.Condition()    # guards _bi_calls and every BiStream's text side; signalled on new text / tokens / calls
        self._bi_calls, self._bi_thread, self._bi_gen = [], None, 0
        self.bistream_coalesce_ms = 8.0        # calls that start together are fed together (one pass over the weights for all first feeds)
        self.bistream_text_ahead = 8           # text pieces a generator-text call may be pulled ahead of their consumption (the reference: the one in work)
        self.bistream_burst = 16
        # True: no decode burst beside a FIRST chunk's flow + HiFT round.  Measured with 8 generator-text calls: first chunk 139 instead of
        # 142-148 ms, but 90.5 instead of 96-98 audio-s/s (the later chunks' tokens come 15-25 ms later): off by default
        self.bistream_first_chunk_alone = os.environ.get('CV2_BI_FIRST_ALONE', '0') == '1'
        # chunks of one round: concurrent streams come back from a chunk round (or are woken by the hub's poll) a few hundred
        # microseconds apart (thread wake-ups under the GIL).  The first submitter waits until the queue has stopped growing for
        # `chunk_quiet_ms` (at most `chunk_wave_ms`, and never when it is the only stream) instead of running the flow for a batch
        # of a few and leaving the rest to a second round (measured, 8 generator-text streams: rounds of 4 + 4 chunks at 67 ms each
        # instead of one of 8)
        self.chunk_wave_ms, self.chunk_quiet_ms = 3.0, 1.0
        self._joining, self.join_burst = set(), 8      # slots taken but not prefilled yet; decode burst length while there are any
        self.open_burst, self.burst_piece = 16, 8      # burst length while slots are free; steps per completion event inside a burst
        self.prefill_wave_ms = 3.0             # a prefill waits this long for calls that hold a slot but have not queued their prompt yet
        # Calls that do NOT start together (a server's arrivals, 0-40 ms apart): the earliest stream's first chunk used to start a round of its
        # own, and a round holds the device for ~40 ms -- the prefills and first bursts of the calls that arrived meanwhile waited behind it
        # (8 calls within 40 ms: first chunk p50 151 ms, max 215-257 ms from each call's own start).  A FIRST chunk therefore waits while
        # newcomers -- calls that hold a slot, started less than first_round_hold_ms ago and have not submitted their first chunk -- are on
        # their way (their prefills and shared decode bursts run meanwhile, this stream's rows included).  0: off.
        self.first_round_hold_ms = float(os.environ.get('CV2_FIRST_ROUND_HOLD_MS', '80'))
        self.first_round_hold_cap = float(os.environ.get('CV2_FIRST_ROUND_HOLD_CAP', '1.5'))      # longest hold of one chunk, in units of first_round_hold_ms
        # ... and only for newcomers that are CLOSE to their first chunk: a newcomer is expected to submit as long after its call as the
        # holding call took (call -> first submit); the hold covers newcomers expected within this window, the others meet the next round.
        # Under sustained arrivals (there is always a newcomer) a first chunk then pays at most the window, not the cap, for lock-step
        self.first_round_hold_window_ms = float(os.environ.get('CV2_FIRST_ROUND_HOLD_WINDOW_MS', '40'))
        self._first_pending = {}               # uuid -> time of the call, until its first chunk is submitted
        # Newcomers beside a chunk round (round 6; CV2_NEWCOMER_BESIDE=0 turns it off).  ONE lock (run_lock) serialises device control and
        # a chunk round holds it from its first launch to its last device-to-host copy (40-125 ms with 8-13 chunks): a call that arrives
        # meanwhile used to wait for the round before its prefill could even be enqueued, and the LLM stream idled for most of the round
        # (profiles/r6_arrivals.txt: prefill at +95 ms, first chunk at +306 ms with 12 streams running).  Now (a) a tensor-text call is
        # prefilled at once on the LLM stream whatever holds run_lock (its own small lock, as the bistream hub does), and while a round is
        # in progress its thread keeps enqueuing the decode steps of its first chunk in short shared bursts; (b) a round's leader tops up
        # the first-chunk tokens of the newcomers already prefilled before it launches the round's flow; (c) CV2_FIRST_CHUNK_LANE: first
        # chunks go in a round of their own ahead of later chunks (which have ~1 s of audio in their listener's buffer).  Measured
        # (profiles/r6_arrivals.txt, r6_beside_bench.txt): Poisson arrivals at half load first chunk p50 193 -> 167 ms, p90 265 -> 218; at
        # capacity 350-456 -> 279 ms with the same or better throughput; 8 calls 0-40 ms apart p50 130 -> 121, max 162 -> 143 ms; the
        # lock-step legs (8 calls at once, generator text, batches) unchanged.  Tokens never depend on any of this (tests).
        self.newcomer_beside = os.environ.get('CV2_NEWCOMER_BESIDE', '1') != '0'
        self.first_chunk_lane = os.environ.get('CV2_FIRST_CHUNK_LANE', '1') != '0'
        self.later_chunk_wait_ms = float(os.environ.get('CV2_LATER_CHUNK_WAIT_MS', '0'))       # (d) a later chunk gives way to a newcomer about to submit (see _chunk_submit)
        self._prefill_lock = threading.Lock()  # one thread at a time drains _prefill_q (alone with newcomer_beside, inside run_lock without it)
        self._adv_lock = threading.Lock()      # _llm_advance's bookkeeping (it used to rely on run_lock alone)
        self._first_need = {}                  # slot -> tokens the call's first chunk needs, until that chunk is submitted
        self._chunks_active = 0                # chunk rounds (flow + HiFT) in progress: decode bursts beside them take the launches
        self._bi_incoming = 0                  # generator-text calls that have entered tts() but not yet joined the hub
        self._sched_log = None                 # diagnostics (tools/bench_bistream.py): list receiving (t, kind, info) of hub / chunk rounds
        self._mode = threading.Condition()     # shared (streams, one LLM slot each) / exclusive (a batch) use of the engines
        self._n_shared, self._excl, self._excl_waiting = 0, False, 0
        self._slot_free, self._active_slots = [], set()
        self._bursts, self._enq, self._slot_ready = [], {}, {}      # events of enqueued decode bursts; tokens per slot once they have run; prefill events
        self.tts_speech_token_dict = {}
        self.llm_end_dict = {}
        self.hift_cache_dict = {}
        self._hift_pin, self._pin_rr = {}, -1  # uuid -> index of the HiftPool engine / HIP stream serving that call's chunks
        self.llm_stream = torch.cuda.Stream(self.device)   # (torch.cuda.Stream.priority_range() is (0, -1) here: no priority below the default exists)
        self.sampling_mode = MODE_RAS if sampling == 'ras' else MODE_GREEDY
        self.seed = seed
        self._limits = (max_text, max_prompt_tokens, max_new_tokens)
        self.max_batch = max(1, int(max_batch))  # concurrent non-streaming calls coalesced into one batch (1 = serialise only)
        # True: decode bursts of the streams step the active slots only (cv2_llm_decode_rows) instead of slots 0 .. highest active.  Off:
        # 8 equal streams measured 85-88 audio-s/s in lock step against 81-82 with it (every new row count captures its graphs in the
        # middle of a round, and equal streams never drop a row); worth turning on for long-lived servers with ragged streams
        self.stream_live_rows = False
        self.coalesce_ms = coalesce_ms
        self._pending = []                     # queued non-streaming requests, guarded by self.lock
        self._chunk_q = []                     # queued chunks of streaming calls, guarded by self.lock
        self._prefill_q = []                   # queued prefills of calls that start together, guarded by self.lock
        self.batch_sizes = []                  # sizes of the coalesced batches run so far (diagnostics / tests)
        self.llm = self.flow = self.hift = None
        self._noise_hook = None                # tests: callable(T) -> [1, 480 T, 9] N(0,1) injected in place of the device Philox draws
        self._noise_hook_takes_uuid = False    # tests with concurrent calls: callable(T, uuid)
        self._on_call = None                   # tests: callable(uuid), invoked in the caller's thread when a tts() call starts
        self._trace = None                     # tests: list receiving (flow mel, token_offset, finalize, noise) per token2wav call
        self._token_log = None                 # tests: dict uuid -> the call's speech tokens, filled when a call ends
        # non-final chunks of streaming calls keep a per-call flow cache (cv2_flow_inference_chunk): a chunk costs its own 50 frames
        # instead of the whole prefix the reference re-runs (model.py:351-381).  CV2_FLOW_CACHE=0: recompute like the reference.
        self.flow_cache = os.environ.get('CV2_FLOW_CACHE', '1') != '0'
        self._flow_caches = {}                 # uuid -> cv2amd.flow.FlowCache, touched under run_lock only
        self.flow_cache_min_group, self.flow_cache_min_frames, self.flow_cache_min_first = 2, 1500, 4
        import collections
        self._prompt_caches = collections.OrderedDict()    # prompt key -> FlowCache of the prompt alone (LRU), touched under run_lock
        self._prompt_building, self._builds_due = set(), []
        self.prompt_cache_max = int(os.environ.get('CV2_PROMPT_CACHES', '4'))       # 1.15 GB per 10 s prompt; 0 disables
        self.flow_cache_headroom = 0.5         # capacity beyond the chunk at hand when the caller's estimate is smaller
        if llm_sd is not None:
            self.load_state_dicts(llm_sd, flow_sd, hift_sd)

    # ---- model.py:67-90 ---------------------------------------------------------------------------------------
    def load(self, llm_model, flow_model, hift_model):
        """model.py:67-90: strict key / shape validation of the three checkpoints (strict=False fallback for the LLM only), the
        hifigan `generator.` prefix stripped, training metadata dropped (cv2amd/checkpoint.py)."""
        from cv2amd import checkpoint
        llm_sd, flow_sd, hift_sd = checkpoint.validate(torch.load(llm_model, map_location='cpu'), torch.load(flow_model, map_location='cpu'),
                                                       torch.load(hift_model, map_location='cpu'))
        self.load_state_dicts(llm_sd, flow_sd, hift_sd)

    def load_state_dicts(self, llm_sd, flow_sd, hift_sd):
        max_text, max_prompt, max_new = self._limits
        B = self.max_batch
        cfg = self.config
        self.llm = LLMEngine(llm_sd, self.device, max_seqs=B, max_pos=max_text + max_prompt + max_new + 8, max_out=max_new,
                             sampling=cfg.sampling)
        self.flow = FlowEngine(flow_sd, self.device, max_utts=B, max_len=2 * (max_prompt + max_new), n_timesteps=cfg.n_timesteps,
                               cfg_rate=cfg.inference_cfg_rate)
        self.flow.pre_lookahead_len, self.flow.token_mel_ratio, self.flow.input_frame_rate = cfg.pre_lookahead_len, cfg.token_mel_ratio, cfg.input_frame_rate
        self.hift_pool = HiftPool(hift_sd, self.device, max_frames=2 * max_new + self.mel_cache_len, n=min(4, B), batch_lanes=0 if os.environ.get('CV2_HIFT_BATCH') == '0' else min(8, B))      # (CV2_HIFT_BATCH=0: A/B switch, one vocoder call per chunk)
        self.hift = self.hift_pool.engines[0]
        self.llm.park()                            # no slot is live: decode steps that cover a free slot leave it alone
        torch.cuda.synchronize(self.device)
        self._slot_free, self._active_slots = list(range(B)), set()
        self._window_dev = torch.from_numpy(self.speech_window).float().to(self.device)

    def load_jit(self, *a, **k):
        raise NotImplementedError('load_jit: TorchScript flow encoder is an NVIDIA-path accelerator of the reference; not used on MI355X')

    def load_trt(self, *a, **k):
        raise NotImplementedError('load_trt: the TensorRT estimator seam is replaced by cv2_flow_estimator')

    def load_vllm(self, *a, **k):
        raise NotImplementedError('load_vllm: the vLLM seam is replaced by cv2_llm_*')

    # ---- model.py:300-334 -------------------------------------------------------------------------------------
    def token2wav(self, token, prompt_token, prompt_feat, embedding, token_offset, uuid, stream=False, finalize=False, speed=1.0):
        tts_mel, _ = self.flow.inference(token=token, token_len=None, prompt_token=prompt_token, prompt_token_len=None,
                                         prompt_feat=prompt_feat, prompt_feat_len=None, embedding=embedding, streaming=stream,
                                         finalize=finalize)
        return self._mel2wav(tts_mel, token_offset, uuid, finalize, speed)

    def _chunk_post(self, uuid, cache, tts_mel, tts_speech, tts_source, hift):
        """model.py:316-324 after the vocoder of a non-final chunk: cross-fade with the cached speech tail, new caches, trim."""
        if cache is not None:
            hift.fade_in_out(tts_speech, cache['speech'], self._window_dev)
        self.hift_cache_dict[uuid] = {'mel': tts_mel[:, :, -self.mel_cache_len:].clone(),
                                      'source': tts_source[:, :, -self.source_cache_len:].clone(),
                                      'speech': tts_speech[:, -self.source_cache_len:].clone()}
        return tts_speech[:, :-self.source_cache_len]

    def _chunks_batched_hift(self, grp, mels):
        """The non-final chunks of one round through ONE vocoder call per shape (HiftPool.batch_engine: gridDim.z = chunks) on the
        current stream; returns the list of speech tensors (None where the chunk failed), or None when batching does not apply."""
        eng = self.hift_pool.batch_engine
        live = [i for i, m in enumerate(mels) if m is not None]
        if eng is None or self._noise_hook is not None or self._trace is not None or len(live) < 2:
            return None
        cur, pre, out = torch.cuda.current_stream(), {}, [None] * len(grp)
        for i in live:
            c, (mel, first) = grp[i], mels[i]
            assert c.offset * self.flow.token_mel_ratio >= first
            mel = mel[:, :, c.offset * self.flow.token_mel_ratio - first:]
            cache = self.hift_cache_dict[c.uuid]
            cs = None
            if cache is not None:
                for t in cache.values():
                    t.record_stream(cur)
                mel, cs = torch.concat([cache['mel'], mel], dim=2), cache['source']
            pre[i] = (cache, mel.contiguous(), cs)
        shapes = {}
        for i in live:
            _, mel, cs = pre[i]
            shapes.setdefault((mel.shape[2], 0 if cs is None else cs.numel()), []).append(i)
        for (T, _nc), idx in shapes.items():
            for j0 in range(0, len(idx), eng.lanes):
                part = idx[j0:j0 + eng.lanes]
                try:
                    if len(part) >= 2 and T <= eng.max_frames:
                        res = eng.inference_batch([pre[i][1] for i in part], [pre[i][2] for i in part])
                    else:
                        res = [self.hift.inference(speech_feat=pre[i][1], cache_source=pre[i][2]) for i in part]
                    for i, (wav, src) in zip(part, res):
                        out[i] = self._chunk_post(grp[i].uuid, pre[i][0], pre[i][1], wav, src, self.hift)
                except Exception as e:
                    for i in part:
                        if out[i] is None:
                            grp[i].exc = e
        return out

    def _mel2wav(self, tts_mel, token_offset, uuid, finalize, speed, hift=None, mel_first=0):
        """model.py:311-334: everything of token2wav after the flow (slice, mel / source / speech caches, HiFT, cross-fade).
        mel_first: index of tts_mel's first frame in the mel flow.inference would have returned (cached chunks hold only the tail)."""
        hift = hift or self.hift
        flow_mel = tts_mel
        if mel_first and self._trace is not None:                              # tests slice the traced mel themselves: full-length view
            flow_mel = torch.cat([tts_mel.new_zeros(1, tts_mel.shape[1], mel_first), tts_mel], dim=2)
        assert token_offset * self.flow.token_mel_ratio >= mel_first
        tts_mel = tts_mel[:, :, token_offset * self.flow.token_mel_ratio - mel_first:]
        cache = self.hift_cache_dict[uuid]
        if cache is not None:
            for t in cache.values():                                           # (the previous round may have built them on another stream of the pool)
                t.record_stream(torch.cuda.current_stream())
            tts_mel = torch.concat([cache['mel'], tts_mel], dim=2)
            hift_cache_source = cache['source']
        else:
            hift_cache_source = None
        noise = None
        if self._noise_hook is not None:
            n_frames = int(tts_mel.shape[2] / speed) if (finalize and speed != 1.0) else tts_mel.shape[2]
            noise = self._noise_hook(n_frames, uuid) if self._noise_hook_takes_uuid else self._noise_hook(n_frames)
        if self._trace is not None:
            self._trace.append((flow_mel.cpu(), token_offset, finalize, noise, uuid))
        if finalize is False:
            tts_speech, tts_source = hift.inference(speech_feat=tts_mel.contiguous(), cache_source=hift_cache_source, noise=noise)
            tts_speech = self._chunk_post(uuid, cache, tts_mel, tts_speech, tts_source, hift)
        else:
            if speed != 1.0:
                assert cache is None, 'speed change only support non-stream inference mode'
                tts_mel = hift.change_speed(tts_mel, speed)                     # F.interpolate(mode='linear') of model.py:329, on the device
            tts_speech, tts_source = hift.inference(speech_feat=tts_mel.contiguous(), cache_source=hift_cache_source, noise=noise)
            if cache is not None:
                hift.fade_in_out(tts_speech, cache['speech'], self._window_dev)
        return tts_speech

    # ---- chunks of concurrent streams: one ragged flow batch for every chunk that is ready -------------------------------
    class _Chunk:
        __slots__ = ('token', 'fpt', 'feat', 'femb', 'offset', 'uuid', 'stream', 'finalize', 'done', 'speech', 'exc', 'cap_hint', 'pkey', 'taken')

    def _flow_cache_for(self, c):
        """The call's flow cache, large enough for this chunk.  Capacity: the caller's estimate of the utterance (frames), at least
        twice what this chunk needs; a cache that turns out too small is replaced by a larger empty one, which makes this chunk a
        recompute of the whole prefix (what the reference does for every chunk)."""
        need = self.flow.token_mel_ratio * (c.fpt.shape[1] + c.token.shape[1] - self.flow.pre_lookahead_len)
        fc = self._flow_caches.get(c.uuid)
        if fc is None or fc.frames < need:
            cap = max(min(max(c.cap_hint or 0, need + int(need * self.flow_cache_headroom)), self.flow.max_len), need)
            pc = self._prompt_caches.get(c.pkey) if (fc is None and c.offset == 0) else None
            if pc is not None:                                                 # a prompt this model has run before: start from its frames
                self._prompt_caches.move_to_end(c.pkey)
                fc = self._flow_caches[c.uuid] = self.flow.clone_cache(pc, cap)
            else:
                fc = self._flow_caches[c.uuid] = self.flow.new_cache(cap)
        return fc

    def _build_prompt_cache(self, pkey, fpt, feat, femb):
        """Background: the flow cache of a prompt alone (its whole chunks), kept for later calls with the same prompt — their first
        chunk then computes ~100 frames instead of the prompt's ~500 + its own.  At most `prompt_cache_max` prompts are kept (LRU)."""
        try:
            torch.cuda.set_device(self.device)
            with self.run_lock:
                if pkey not in self._prompt_caches:
                    pc = self.flow.prompt_cache(fpt, feat, femb, hop=self.token_hop_len)
                    torch.cuda.current_stream().synchronize()
                    if pc is not None:
                        self._prompt_caches[pkey] = pc
                        while len(self._prompt_caches) > self.prompt_cache_max:
                            self._prompt_caches.popitem(last=False)
        except Exception:                                                      # an optimisation only: the calls go on without it
            pass
        finally:
            self._prompt_building.discard(pkey)

    def _flow_batch(self, grp, streaming, finalize):
        """[(mel, first frame index)] of one group of chunks.  Non-final chunks of streaming calls go through the per-call flow cache
        when that pays: never a call's first chunk (its latency is the one that is felt; the second chunk fills the cache instead),
        and only when at least `flow_cache_min_group` chunks share the round or a prefix is long — one short stream alone is
        latency-bound either way (measured: 45 ms per chunk cached, 37 ms recomputed; 8 streams: 87 ms against 127 ms per round).  A
        cache that sat out some rounds is still valid for the frames it holds: the next cached call computes everything after them."""
        utts = [dict(token=c.token, prompt_token=c.fpt, prompt_feat=c.feat, embedding=c.femb) for c in grp]
        outs, cached = [None] * len(grp), []
        if streaming and not finalize and self.flow_cache:
            # first chunks join only when their prompt's cache exists already (a voice used before)
            firsts = [i for i, c in enumerate(grp) if c.offset == 0 and c.pkey is not None and c.pkey in self._prompt_caches]
            if len(firsts) < self.flow_cache_min_first:                        # a cached call costs ~40 ms whatever it holds, a recomputed
                firsts = []                                                    # first chunk ~19 + 5.5 ms per stream: worth it from 4 on
            cached = sorted([i for i, c in enumerate(grp) if c.offset > 0] + firsts)
            frames = [self.flow.token_mel_ratio * (grp[i].fpt.shape[1] + grp[i].token.shape[1]) for i in cached]
            if len(cached) < self.flow_cache_min_group and not any(f >= self.flow_cache_min_frames for f in frames):
                cached = []
            if len(grp) >= self.flow_cache_min_group and self.prompt_cache_max > 0:
                for c in grp:                                                  # several streams at once and a prompt not seen before:
                    if c.offset == 0 and c.pkey is not None and c.pkey not in self._prompt_caches and c.pkey not in self._prompt_building:
                        self._prompt_building.add(c.pkey)                      # prepare it for the calls to come: started when this round
                        self._builds_due.append((c.pkey, c.fpt, c.feat, c.femb))   # has delivered (its waiters are in line for the device first)
        if cached:
            res = self.flow.inference_chunk_batch([utts[i] for i in cached], [self._flow_cache_for(grp[i]) for i in cached], finalize=False)
            for i, r in zip(cached, res):
                outs[i] = r
        rest = [i for i in range(len(grp)) if outs[i] is None]
        if rest:
            res = self.flow.inference_batch([utts[i] for i in rest], streaming=streaming, finalize=finalize)
            for i, m in zip(rest, res):
                outs[i] = (m, 0)
        return outs

    def _run_chunks(self, batch):
        """One round of ready chunks.  A failure is delivered only to the call it belongs to (an utterance failure must not poison its
        batch, SURVEY.md §5: evaluation/cosyvoice_synthesizer.py:265-297 records per-sample errors): the ragged flow batch is retried
        chunk by chunk when it fails as a whole, HiFT / cache / cross-fade errors stay with their chunk."""
        try:
            for key in sorted({(c.stream, c.finalize) for c in batch}):
                grp = [c for c in batch if (c.stream, c.finalize) == key]
                try:
                    mels = self._flow_batch(grp, key[0], key[1])
                except Exception:                                              # find the offender: run the chunks one by one (a failed
                    mels = []                                                  # cached call left its caches where they were)
                    for c in grp:
                        try:
                            mels.append(self._flow_batch([c], key[0], key[1])[0])
                        except Exception as e:
                            c.exc = e
                            mels.append(None)
                if key[1]:                                                     # the final chunk is a full-context recompute (model.py:374-380
                    for c in grp:                                              # passes no `stream`): the call's cache is not needed any more
                        self._flow_caches.pop(c.uuid, None)
                # HiFT (and the cache / cross-fade bookkeeping) of a chunk always runs on the pool engine + HIP stream its call was pinned
                # to at its first chunk: the per-uuid caches are then allocated, read and freed on ONE stream (a cache block freed on
                # stream A while stream B still reads it could be handed out again by the caching allocator), joined before anything is read
                pool, main, sp = self.hift_pool, torch.cuda.current_stream(), []
                batched = self._chunks_batched_hift(grp, mels) if (key[0] and not key[1]) else None
                if batched is not None:                                        # one vocoder call per chunk shape instead of one per chunk
                    for c, w in zip(grp, batched):
                        if w is not None:
                            c.speech = w.cpu()
                    continue
                used = sorted({self._hift_pin.setdefault(c.uuid, self._next_pin()) for c in grp})
                for k in used:
                    pool.streams[k].wait_stream(main)
                for c, mel in zip(grp, mels):
                    if mel is None:
                        sp.append(None)
                        continue
                    k = self._hift_pin[c.uuid]
                    try:
                        mel, first = mel
                        with torch.cuda.stream(pool.streams[k]):
                            mel.record_stream(pool.streams[k])                 # produced on the main stream, consumed on the pinned one
                            sp.append(self._mel2wav(mel, c.offset, c.uuid, c.finalize, 1.0, hift=pool.engines[k], mel_first=first))
                    except Exception as e:
                        c.exc = e
                        sp.append(None)
                for k in used:
                    main.wait_stream(pool.streams[k])
                for c, w in zip(grp, sp):
                    if w is not None:
                        w.record_stream(main)
                        c.speech = w.cpu()
        except BaseException as e:                                            # not attributable to one chunk: every waiting stream sees it
            for c in batch:
                if c.speech is None and c.exc is None:
                    c.exc = e
        finally:
            for c in batch:
                c.done = True
            due, self._builds_due = self._builds_due, []
            for args in due:
                threading.Thread(target=self._build_prompt_cache, args=args, daemon=True).start()

    def _next_pin(self):
        self._pin_rr = (self._pin_rr + 1) % len(self.hift_pool.engines)
        return self._pin_rr

    def _chunk_submit(self, token, fpt, feat, femb, offset, this_uuid, stream, finalize, cap_hint=None, pkey=None, first_slot=None):
        """token2wav for one chunk of a streaming call.  The chunk is queued; whoever gets the device next runs the flow over ALL
        queued chunks as one ragged batch (streams that share decode steps become ready together), then HiFT per chunk."""
        c = self._Chunk()
        c.token, c.fpt, c.feat, c.femb, c.offset, c.uuid, c.stream, c.finalize = token, fpt, feat, femb, offset, this_uuid, stream, finalize
        c.done, c.speech, c.exc, c.cap_hint, c.pkey, c.taken = False, None, None, cap_hint, pkey, False
        with self.lock:
            self._chunk_q.append(c)
        if self._sched_log is not None:
            self._sched_log.append((time.perf_counter(), 'submit', dict(queued=len(self._chunk_q), offset=offset)))
        if stream and not finalize and offset == 0:
            with self.lock:
                t_call = self._first_pending.pop(this_uuid, None)
                if first_slot is not None:
                    self._first_need.pop(first_slot, None)
            hold = self.first_round_hold_ms * 1e-3
            if hold > 0:
                t_sub, held = time.perf_counter(), False
                t_cap = t_sub + self.first_round_hold_cap * hold              # (arrivals that never stop must not hold a chunk for ever)
                own = t_sub - t_call if t_call is not None else hold          # what this call took from tts() to its first chunk
                window = self.first_round_hold_window_ms * 1e-3
                while not c.done and not c.taken:                             # (taken: a round has this chunk -- get in line)
                    now = time.perf_counter()
                    with self.lock:                                           # (one consistent view of the newcomers and of the queue)
                        pend = list(self._first_pending.values())
                        n_wait = sum(1 for q in self._chunk_q if q.offset == 0 and q.stream and not q.finalize)
                    # (from three calls on: two calls 30 ms apart are as well off with a round each -- measured 73 / 121 ms p50 / max either way)
                    n_new = sum(1 for t0 in pend if now - t0 < hold and t0 + own - now <= window)
                    if now >= t_cap or n_new == 0 or n_new + max(n_wait, 1) < 3:
                        break
                    held = True
                    time.sleep(0.0002)
                if held and self._sched_log is not None:
                    self._sched_log.append((time.perf_counter(), 'held', dict(ms=round((time.perf_counter() - t_sub) * 1e3, 1))))
        if stream and not finalize and offset > 0 and self.later_chunk_wait_ms > 0 and self._first_need:
            # a LATER chunk (its listener holds ~1 s of audio) gives way to a newcomer that is about to submit its FIRST chunk: when every decode
            # step a newcomer's first chunk needs is already enqueued, its chunk arrives within one burst -- waiting for it (at most
            # later_chunk_wait_ms) puts it into the round that is about to start (the first-chunk lane) instead of behind it
            t_end = time.perf_counter() + self.later_chunk_wait_ms * 1e-3
            waited = False
            while not c.done and not c.taken and time.perf_counter() < t_end:
                with self.lock:
                    near = any(self._enq.get(sl, 0) >= n for sl, n in self._first_need.items() if sl not in self._joining)
                    have_first = any(q.offset == 0 and q.stream and not q.finalize for q in self._chunk_q)
                if not near or have_first:
                    break
                waited = True
                time.sleep(0.0002)
            if waited and self._sched_log is not None:
                self._sched_log.append((time.perf_counter(), 'gaveway', dict(ms=round((time.perf_counter() - t_end) * 1e3 + self.later_chunk_wait_ms, 1))))
        n_streams = min(self._n_shared, self.max_batch)
        if n_streams > 1 and self.chunk_wave_ms > 0:
            t_last = time.perf_counter()
            t_end, last = t_last + self.chunk_wave_ms * 1e-3, len(self._chunk_q)
            while not c.done and not c.taken:                                 # (taken: a leader has this chunk in its round already -- get in line)
                n, now = len(self._chunk_q), time.perf_counter()
                if n >= n_streams or now >= t_end:
                    break
                if n != last:
                    last, t_last = n, now
                elif now - t_last > self.chunk_quiet_ms * 1e-3:
                    break
                time.sleep(0.0001)
        with self.run_lock:
            while not c.done:                                                 # (one round; two when the first-chunk lane ran first without c)
                with self.lock:
                    batch = self._chunk_q[:self.max_batch]
                    if self.first_chunk_lane and len(batch) > 1:
                        # first chunks in a round of their own AHEAD of the others: a later chunk of a stream has ~1 s of audio in its
                        # listener's buffer, a first chunk is what the caller is waiting for (profiles/r6_arrivals.txt: a newcomer's
                        # first chunk used to ride a round of 10-11 chunks, 85-125 ms)
                        lane = [b for b in batch if b.stream and not b.finalize and b.offset == 0]
                        if lane and len(lane) < len(batch):
                            batch = lane
                    for b in batch:
                        self._chunk_q.remove(b)
                        b.taken = True
                if not batch:                                                 # (cannot happen while c is queued; never spin on an empty queue)
                    break
                t0 = time.perf_counter()
                if self.newcomer_beside and self._first_need:                 # (b) newcomers already prefilled: their first chunk's tokens run
                    with self.lock:                                           # on the LLM stream beside this round instead of after it
                        short = [n - self._enq[sl] for sl, n in self._first_need.items() if sl in self._enq and sl not in self._joining]
                    if short and max(short) > 0:
                        self._llm_advance(min(max(short), 2 * self.token_hop_len), shared=True)
                self._chunks_active += 1
                try:
                    self._run_chunks(batch)
                finally:
                    self._chunks_active -= 1
                if self._sched_log is not None:
                    self._sched_log.append((t0, 'chunks', dict(n=len(batch), final=sum(1 for b in batch if b.finalize), ms=round((time.perf_counter() - t0) * 1e3, 2))))
        if c.exc is not None:
            raise c.exc
        return c.speech

    # ---- llm side: llm_job (model.py:118-139) as bursts on the LLM stream -------------------------------------------
    def _enter_shared(self, joining=True):
        """A streaming / serial call takes one LLM slot; waits while a coalesced batch runs (or waits to run) or no slot is free."""
        with self._mode:
            while self._excl or self._excl_waiting > 0 or not self._slot_free:
                self._mode.wait(0.05)
            self._n_shared += 1
            slot = self._slot_free.pop(0)
            self._active_slots.add(slot)
            if joining:                                                       # (prefilled by _llm_start; the bistream hub feeds its slots itself)
                self._joining.add(slot)
        return slot

    def _exit_shared(self, slot):
        with self._mode:
            self._active_slots.discard(slot)
            self._joining.discard(slot)
            self._enq.pop(slot, None)
            self._slot_ready.pop(slot, None)
            self._slot_free.append(slot)
            self._slot_free.sort()
            self._n_shared -= 1
            self._mode.notify_all()

    def _enter_excl(self):
        with self._mode:
            self._excl_waiting += 1
            while self._excl or self._n_shared > 0:
                self._mode.wait(0.05)
            self._excl_waiting -= 1
            self._excl = True

    def _exit_excl(self):
        with self._mode:
            self._excl = False
            self._mode.notify_all()

    class _Prefill:
        __slots__ = ('slot', 'text', 'prompt_text', 'ptok', 'done', 'exc', 'force_len')

    def _llm_start(self, slot, text, prompt_text, llm_prompt_speech_token, force_len=None):
        """llm.py:684-719 step 0 (prefill + first draw) for one call.  Calls that start together are prefilled TOGETHER: the request
        is queued, whoever gets the device next runs one batched prefill (one pass over the weights) for every queued request."""
        if self._sched_log is not None:
            self._sched_log.append((time.perf_counter(), 'start', dict(slot=slot)))
        p = self._Prefill()
        p.slot, p.text, p.prompt_text, p.ptok, p.done, p.exc, p.force_len = slot, text, prompt_text, llm_prompt_speech_token, False, None, force_len
        with self.lock:
            self._prefill_q.append(p)
        if self.newcomer_beside:
            # never behind a chunk round: the prefill goes on the LLM stream at once (run_lock is a ticket lock -- a thread that queues for it
            # an instant before a round's leader waits the whole round), then the first chunk's decode steps keep coming while a round lasts
            self._prefill_drain(p)
            if p.exc is None:
                self._first_tokens_beside(slot)
        else:
            with self.run_lock:
                self._prefill_drain(p)
        if p.exc is not None:
            raise p.exc

    def _prefill_drain(self, p):
        """One batched prefill for every queued request (p among them), unless another thread has served p meanwhile."""
        with self._prefill_lock:
            if not p.done:
                # calls that have taken a slot but not queued their prompt yet (they are copying their inputs) join this prefill: one that
                # misses it by a millisecond waits for the whole prefill + the bursts behind it (trace of 8 staggered calls: 26 ms)
                t_end = time.perf_counter() + self.prefill_wave_ms * 1e-3
                while len(self._prefill_q) < len(self._joining) and time.perf_counter() < t_end:
                    time.sleep(0.0002)
                with self.lock:
                    batch = self._prefill_q[:]
                    self._prefill_q.clear()
                try:
                    with self.llm_lock, torch.cuda.stream(self.llm_stream):
                        xs = [self.llm.build_lm_input(b.text, b.prompt_text, b.ptok) for b in batch]
                        mms = [(int(b.text.shape[1] * self.min_token_text_ratio), int(b.text.shape[1] * self.max_token_text_ratio))
                               if b.force_len is None else (b.force_len, b.force_len) for b in batch]     # llm.py:643-644 (target text only)
                        self.seed += 1
                        groups = [[b for b in batch if b.force_len is None], [b for b in batch if b.force_len is not None]]
                        for grp, forced in zip(groups, (False, True)):
                            if not grp:
                                continue
                            gx, gm = [xs[batch.index(b)] for b in grp], [mms[batch.index(b)] for b in grp]
                            if sum(x.shape[0] for x in gx) <= self.llm.dims.max_prefill_rows:
                                self.llm.add_requests([b.slot for b in grp], gx, gm, self.sampling_mode, self.seed, forced)
                            else:
                                for b, x, mm in zip(grp, gx, gm):
                                    self.llm.add_requests([b.slot], [x], [mm], self.sampling_mode, self.seed, forced)
                        ev = torch.cuda.Event()
                        ev.record(self.llm_stream)
                    for b in batch:
                        self._slot_ready[b.slot], self._enq[b.slot] = ev, 1          # the prefill draws token 0
                        self._joining.discard(b.slot)
                    if self._sched_log is not None:
                        self._sched_log.append((time.perf_counter(), 'prefill', dict(n=len(batch), rows=sum(x.shape[0] for x in xs),
                                                                                       beside=self._chunks_active > 0)))
                except BaseException as e:
                    for b in batch:
                        b.exc = e
                finally:
                    for b in batch:
                        b.done = True

    def _first_tokens_beside(self, slot):
        """newcomer_beside (a): while a chunk round holds run_lock, enqueue the decode steps this slot's first chunk still needs, in bursts of
        at most `open_burst` shared steps, waiting for each burst before the next so that the LLM stream stays short for the next newcomer's
        prefill.  Every active slot advances with it (the running streams' next chunks need those tokens anyway).  Stops when the round is
        over: the caller's ordinary loop (under run_lock) takes it from there."""
        need = self._first_need.get(slot)
        while need is not None and self._chunks_active > 0:
            left = need - self._enq.get(slot, 0)
            if left <= 0:
                break
            ev = self._llm_advance(min(left, self.open_burst), shared=True)
            if ev is None:
                break
            ev.synchronize()

    # the helpers below run under self.run_lock (newcomer_beside: _llm_advance also beside a chunk round, under its own lock)
    def _llm_advance(self, n_steps, shared=False):
        """n_steps decode steps for EVERY active slot (slots 0..highest active, parked slots in between idle; or the active slots only,
        see stream_live_rows), enqueued on the LLM stream; `_enq[slot]` = the tokens a live slot holds at most once everything enqueued so far has run.  shared: the burst runs
        beside a chunk's flow + HiFT on the other streams."""
        if n_steps <= 0:
            return None
        with self._adv_lock:
            return self._llm_advance_locked(n_steps, shared or self._chunks_active > 0)

    def _llm_advance_locked(self, n_steps, shared):
        with self._mode:
            act = sorted(self._active_slots)
            # a call that has taken its slot but is not prefilled yet (it started a moment after the others): its prefill queues on the
            # LLM stream behind whatever is enqueued now, so keep that short -- the caller's loop asks again (measured with 8 calls
            # starting within 5 ms of each other: three of them missed the first prefill by < 1 ms and waited 41 ms behind a 47-step burst)
            if self._joining:
                n_steps = min(n_steps, self.join_burst)
            elif self._slot_free and self.max_batch > 1:
                # slots are free: a call may arrive any moment, and its prefill waits for the device lock, which the stream that polls for
                # this burst's tokens holds until the burst has run (trace of 8 staggered calls: two newcomers waited 17 and 24 ms behind a
                # 39-step burst of the first two streams).  The caller's loop asks again.
                n_steps = min(n_steps, self.open_burst)
        evs = []
        with self.llm_lock, torch.cuda.stream(self.llm_stream):
            # an event every `burst_piece` steps: a stream that needs 8 more tokens waits for 8 steps, not for the 25 another stream asked for
            # in the same burst (the same trace: four newcomers 8 tokens short of their first chunk sat out a 25-step burst, 15 ms)
            left = n_steps
            while left > 0:
                k = min(left, self.burst_piece) if left > self.burst_piece + 2 else left
                if self.stream_live_rows:
                    self.llm.step_rows(act, k, shared=shared)
                else:
                    self.llm.step(act[-1] + 1, k, shared=shared)     # slots 0 .. highest active, parked ones idle
                ev = torch.cuda.Event()
                ev.record(self.llm_stream)
                evs.append(ev)
                left -= k
        while self._bursts and self._bursts[0].query():                       # finished bursts nobody had to wait for
            self._bursts.pop(0)
        self._bursts.extend(evs)
        for sl in act:
            self._enq[sl] = self._enq.get(sl, 0) + n_steps
        if self._sched_log is not None:
            self._sched_log.append((time.perf_counter(), 'burst', dict(rows=len(act), steps=n_steps, shared=bool(shared))))
        return evs[-1] if evs else None

    def _llm_poll(self, this_uuid, slot, need=None):
        """Publish the tokens of the slot so far (the reference's thread appends to the same list).  need = None: wait for all the
        enqueued LLM work.  Otherwise wait only until the slot holds `need` tokens: the burst that another stream's next chunk asked
        for keeps running on the LLM stream BESIDE this chunk's flow and HiFT (a full synchronize here would put every decode burst in
        front of the chunk round: 24 + 58 ms per round of 8 streams instead of max(24, 58))."""
        ready = self._slot_ready.pop(slot, None)
        if ready is not None:
            ready.synchronize()                                               # the slot's prefill: before it the state row is the previous call's
        if need is None:
            self.llm_stream.synchronize()
            del self._bursts[:]
        while True:
            st, toks = self.llm.read_slot(slot)
            if need is None or len(toks) >= need or bool(st[L.ST_DONE]) or not self._bursts:
                break
            try:
                ev = self._bursts.pop(0)
            except IndexError:                                                # (newcomer_beside: a pump emptied the list meanwhile)
                continue
            ev.synchronize()
        if not self._bursts:
            self._enq[slot] = len(toks)                                       # nothing in flight: the count is exact (ids above EOS are steps without a token)
        self.tts_speech_token_dict[this_uuid] = toks
        self.llm_end_dict[this_uuid] = bool(st[L.ST_DONE])
        return toks

    # ---- coalesced non-streaming calls ---------------------------------------------------------------------------
    class _Pending:
        __slots__ = ('text', 'prompt_text', 'llm_ptok', 'fpt', 'feat', 'femb', 'speed', 'uuid', 'done', 'speech', 'exc', 'force_len', 'dev_out')

    def _run_batch(self, batch):
        """llm_job + token2wav(finalize=True) of model.py:118-139,300-334 for several queued calls at once.  Only the caller whose
        request failed gets the exception (the sampler's RuntimeError of llm.py:249 is per request): the others' audio is delivered."""
        self.batch_sizes.append(len(batch))
        try:
            self.seed += 1
            fl = [p.force_len for p in batch]
            if any(f is None for f in fl):
                assert all(f is None for f in fl), 'forced-length (synthetic-weights) calls cannot share a batch with free-running ones'
                fl = None
            toks, errs = self.llm.generate([(p.text, p.prompt_text, p.llm_ptok) for p in batch], mode=self.sampling_mode, seed=self.seed,
                                           return_errors=True, min_ratio=self.min_token_text_ratio, max_ratio=self.max_token_text_ratio,
                                           force_len=fl)
            with self.lock:
                for p, t in zip(batch, toks):
                    self.tts_speech_token_dict[p.uuid], self.llm_end_dict[p.uuid] = t, True
            for p, err in zip(batch, errs):
                if err is not None:
                    p.exc = err
            live = [(p, t) for p, t, err in zip(batch, toks, errs) if err is None]
            utts = [dict(token=torch.tensor(t, dtype=torch.int32).unsqueeze(0), prompt_token=p.fpt, prompt_feat=p.feat, embedding=p.femb)
                    for p, t in live]
            try:
                mels = self.flow.inference_batch(utts, streaming=False, finalize=True) if utts else []
            except Exception:                                                  # e.g. one utterance beyond the flow's length limit
                mels = []
                for (p, _), u in zip(live, utts):
                    try:
                        mels.append(self.flow.inference_batch([u], streaming=False, finalize=True)[0])
                    except Exception as e:
                        p.exc = e
                        mels.append(None)
            good = [(p, m) for (p, _), m in zip(live, mels) if m is not None]
            gm = []
            for p, m in good:
                if p.speed != 1.0:                                            # model.py:328-330
                    m = self.hift.change_speed(m, p.speed)
                gm.append(m.contiguous())
            outs = self.hift_pool.inference_many(gm) if gm else []
            torch.cuda.synchronize(self.device)
            for (p, _), (wav, _s) in zip(good, outs):
                p.speech = wav.clone() if p.dev_out else wav.cpu()      # (device_output: the waveform stays in HBM for a device-side consumer)
        except BaseException as e:                                            # not attributable to one request
            for p in batch:
                if p.speech is None and p.exc is None:
                    p.exc = e
        finally:
            for p in batch:
                p.done.set()

    def _tts_coalesced(self, p):
        with self.lock:
            self._pending.append(p)
        while not p.done.is_set():
            # another caller is the leader: block on its lock (it is released right after the batch's results are set).  Polling
            # here (31 waiting callers waking every 0.5 ms) fought the leader's Python for the GIL: 20-70 ms per batch of 32
            if not self._leader_lock.acquire(timeout=0.05):
                continue
            try:
                if p.done.is_set():
                    break
                self._enter_excl()                                             # streams drain first; new ones wait for the batch
                try:
                    if self.coalesce_ms > 0:
                        t_end = time.perf_counter() + self.coalesce_ms * 1e-3  # give concurrent callers a moment to queue up
                        while time.perf_counter() < t_end and len(self._pending) < self.max_batch:
                            time.sleep(0.0002)
                    with self.lock:
                        batch = self._pending[:self.max_batch]
                        del self._pending[:len(batch)]
                    if batch:
                        with self.run_lock:
                            self._run_batch(batch)
                finally:
                    self._exit_excl()
            finally:
                self._leader_lock.release()
        if p.exc is not None:
            raise p.exc
        return p.speech

    # ---- llm_job's bistream branch (model.py:120-128) for ALL generator-text calls: one round loop -------------------------------
    class _BiCall:
        __slots__ = ('bs', 'uuid', 'toks', 'want_total', 'closed', 'exc', 'ended', 'first_done')

    def _bi_register(self, bs, this_uuid, text):
        """A generator-text call joins the hub: its text generator is drained by a small pump thread (host only: the reference's
        llm_job thread iterates it the same way, model.py:118-128), the device side is driven by the hub thread for all calls."""
        c = self._BiCall()
        c.bs, c.uuid, c.toks, c.want_total, c.closed, c.exc, c.ended = bs, this_uuid, self.tts_speech_token_dict[this_uuid], 0, False, None, False
        c.first_done = False                  # the call's first chunk has been delivered (until then its chunk round has the device to itself, see _bi_loop)

        def pump():
            # The reference pulls the next piece only when the previous one has been worked off (llm.py `for this_text in text`: a piece that
            # leads to a decode holds the loop until the fill id).  Here: at most `bistream_text_ahead` pieces wait un-consumed, so the caller's
            # generator sees the same back-pressure (within that margin) and is never drained ahead of the synthesis.
            failed = False
            try:
                it = iter(text)
                while True:
                    with self._bi_cv:
                        while not c.closed and len(bs.pieces) >= self.bistream_text_ahead:
                            self._bi_cv.wait(0.05)
                        if c.closed:
                            return
                    try:
                        piece = next(it)
                    except StopIteration:
                        break
                    with self._bi_cv:
                        if c.closed:
                            return
                        bs.push(piece)
                        self._bi_gen += 1
                        self._bi_cv.notify_all()
            except BaseException as e:      # noqa: BLE001 -- the caller's generator failed: the call fails with it
                failed = True
                with self._bi_cv:           # no final feed, no further decode for a call that will raise: the stream is over as it stands
                    c.exc = e
                    c.ended = True
                    bs.finished = True
                    self._bi_gen += 1
                    self._bi_cv.notify_all()
            finally:
                if not failed:
                    with self._bi_cv:
                        bs.close()
                        self._bi_gen += 1
                        self._bi_cv.notify_all()
        with self._bi_cv:
            self._bi_calls.append(c)
            self._bi_incoming -= 1
            self._bi_gen += 1
            if self._bi_thread is None or not self._bi_thread.is_alive():
                self._bi_thread = threading.Thread(target=self._bi_loop, daemon=True, name='cv2-bistream-hub')
                self._bi_thread.start()
            self._bi_cv.notify_all()
        threading.Thread(target=pump, daemon=True, name='cv2-bistream-text').start()
        return c

    def _bi_next(self, bs):
        """BiStream.next_feed() under _bi_cv; the text pumps are told when it consumed pieces (they pull the next ones)"""
        n0 = len(bs.pieces)
        r = bs.next_feed()
        if len(bs.pieces) != n0:
            self._bi_cv.notify_all()
        return r

    def _bi_unregister(self, c):
        with self._bi_cv:
            c.closed = True
            if c in self._bi_calls:
                self._bi_calls.remove(c)
            self._bi_cv.notify_all()
        with self.llm_lock:                                                   # (a round in flight has finished with the slot once this is held)
            with torch.cuda.stream(self.llm_stream):
                self.llm.park(c.bs.slot)
            self.llm_stream.synchronize()

    def _bi_loop(self):
        """Scheduler-owned rounds over every generator-text call (Qwen2LM.inference_bistream x the concurrent calls, llm.py:721-834):
          1. poll -- one read of the running slots' state records and new ids; tokens are published to the calls' lists;
          2. feed -- every idle slot whose next input is there (text pieces are consumed on the host, no device work) goes into ONE
             cv2_llm_extend_batch: the first [sos, 5 text : 15 prompt-speech blocks] of calls that start together, the 5-token text
             blocks after a fill id, the final [pending, remaining text, task id];
          3. burst -- one shared decode burst over the running slots only (cv2_llm_decode_rows: a slot stopped on the fill id is not
             a row), as long as the earliest known stop / the earliest chunk somebody waits for.
        The LLM work runs on the LLM stream under llm_lock only, so it goes on beside the chunk rounds (flow + HiFT under run_lock)."""
        torch.cuda.set_device(self.device)
        idle_since = None
        while True:
            with self._bi_cv:
                if not self._bi_calls:
                    if idle_since is None:
                        idle_since = time.perf_counter()
                    if time.perf_counter() - idle_since > 5.0:                # nothing to do for a while: the next call starts a new thread
                        self._bi_thread = None
                        return
                    self._bi_cv.wait(0.5)
                    continue
                idle_since = None
                calls = [c for c in self._bi_calls if not c.closed]
                # calls that start together are fed together: wait (briefly) until the newcomers' first input is complete
                if any(not c.bs.started for c in calls) and not any(c.bs.running for c in calls):
                    t_end = time.perf_counter() + self.bistream_coalesce_ms * 1e-3
                    while time.perf_counter() < t_end:
                        gen = self._bi_gen
                        self._bi_cv.wait(0.0005)
                        if self._bi_gen == gen and self._bi_incoming <= 0 and all(c.bs.started or self._bi_next(c.bs) is not None
                                                                                   for c in self._bi_calls if not c.closed):
                            break
                    calls = [c for c in self._bi_calls if not c.closed]
            if not calls:
                continue
            try:
                ev = None
                with self.llm_lock, torch.cuda.stream(self.llm_stream):
                    eng = self.llm
                    streams = [c.bs for c in calls if not c.closed]
                    hold = False
                    t_r0 = time.perf_counter()
                    eng.bi_poll(streams)
                    t_r1 = time.perf_counter()
                    self._bi_publish(calls)
                    with self._bi_cv:                                         # text pieces are pushed under this lock: consume them under it,
                        for c in calls:                                       # (tokens the consumers still wait for, after this poll's ids)
                            c.bs.want = max(0, c.want_total - len(c.toks)) if c.want_total else None
                        for b in streams:                                     # feed outside it (the consumers' pull() takes it too)
                            if not b.running and not b.finished:
                                self._bi_next(b)
                    fed = eng.bi_feed(streams, prepared=True)
                    t_r2 = time.perf_counter()
                    n = eng.bi_burst_len(streams, self.bistream_burst)
                    # a FIRST chunk is in its flow + HiFT round: no decode burst beside it (the latency that is felt; the bursts resume with
                    # the round's end -- every later chunk's tokens are ready long before its round)
                    if n and self.bistream_first_chunk_alone and self._chunks_active > 0 and any(
                            not c.first_done and c.want_total and len(c.toks) >= c.want_total for c in calls):
                        n, hold = 0, True
                    if n:                                                     # beside a chunk round: the launches; alone: the one-launch step
                        eng.bi_burst(streams, n, shared=bool(self._chunk_q) or self._chunks_active > 0)
                    if fed or n:
                        ev = torch.cuda.Event()
                        ev.record(self.llm_stream)
                    if self._sched_log is not None:
                        self._sched_log.append((t_r0, 'hub', dict(fed=len(fed), rows=sum(len(b._last_feed) for b in fed), burst=n,
                                                                  running=sum(1 for b in streams if b.running), poll_ms=round((t_r1 - t_r0) * 1e3, 2),
                                                                  feed_ms=round((t_r2 - t_r1) * 1e3, 2), burst_ms=round((time.perf_counter() - t_r2) * 1e3, 2))))
                if ev is not None:
                    ev.synchronize()                                          # the round's device work, outside the lock
                elif hold:
                    time.sleep(0.001)
                else:
                    with self._bi_cv:                                         # every slot waits for text (or has ended): sleep until something arrives
                        if all(not c.bs.running and (c.bs.finished or self._bi_next(c.bs) is None) for c in self._bi_calls):
                            self._bi_cv.wait(0.05)
            except BaseException as e:      # noqa: BLE001 -- not attributable to one call: all of this round's calls see it
                with self._bi_cv:
                    for c in calls:
                        if c.exc is None:
                            c.exc = e
                        c.ended = True
                        c.bs.finished, c.bs.running = True, False
                    self._bi_cv.notify_all()

    def _bi_publish(self, calls):
        with self._bi_cv:
            for c in calls:
                new = c.bs.take()
                if new:
                    c.toks.extend(new)
                if c.bs.err is not None and c.exc is None:
                    c.exc = c.bs.err
                if c.bs.finished and not c.ended:
                    c.ended = True
                    self.llm_end_dict[c.uuid] = True
            self._bi_cv.notify_all()

    def _tts_pulled(self, text, prompt_text, llm_ptok, source_speech_token, fpt, feat, femb, this_uuid, stream, speed, vc):
        """tts() for the two token sources that are not the batched unistream LLM: voice conversion (vc_job, model.py:141-143: the
        speech tokens of the source utterance ARE the tokens) and generator text (inference_bistream, llm.py:721-834: the call owns
        one LLM slot, the hub above drives it together with every other generator-text call).  The chunk arithmetic is model.py:351-394."""
        hop, la = self.token_hop_len, self.flow.pre_lookahead_len
        incoming = not vc                                                      # counted in _bi_incoming until the call has joined the hub
        slot = None
        toks = self.tts_speech_token_dict[this_uuid]
        call = None
        try:
            slot = None if vc else self._enter_shared(joining=False)
            if vc:
                toks.extend(source_speech_token.flatten().tolist())
                self.llm_end_dict[this_uuid] = True
            else:
                self.seed += 1
                call = self._bi_register(self.llm.new_bistream(slot, prompt_text, llm_ptok, mode=self.sampling_mode, seed=self.seed), this_uuid, text)
                incoming = False

            def pull(n_total):
                """wait until the call holds n_total tokens or its LLM has ended (the reference polls the shared list, model.py:353-366)"""
                if call is None:
                    return
                with self._bi_cv:
                    call.want_total = n_total
                    while len(toks) < n_total and not call.ended and call.exc is None:
                        self._bi_cv.wait(0.5)
                    if call.exc is not None:
                        raise call.exc
            if stream is True:
                token_offset = 0
                prompt_token_pad = int(np.ceil(fpt.shape[1] / hop) * hop - fpt.shape[1])
                while True:
                    this_hop = hop + prompt_token_pad if token_offset == 0 else hop
                    pull(token_offset + this_hop + la)
                    if len(toks) - token_offset < this_hop + la:
                        break
                    this_tok = torch.tensor(toks[:token_offset + this_hop + la], dtype=torch.int32).unsqueeze(0)
                    speech = self._chunk_submit(this_tok, fpt, feat, femb, token_offset, this_uuid, True, False)
                    if call is not None:
                        call.first_done = True
                    token_offset += this_hop
                    yield {'tts_speech': speech}
                pull(1 << 30)
                this_tok = torch.tensor(list(toks), dtype=torch.int32).unsqueeze(0)
                yield {'tts_speech': self._chunk_submit(this_tok, fpt, feat, femb, token_offset, this_uuid, False, True)}
            else:
                pull(1 << 30)
                with self.run_lock:
                    this_tok = torch.tensor(list(toks), dtype=torch.int32).unsqueeze(0)
                    speech = self.token2wav(this_tok, fpt, feat, femb, 0, this_uuid, finalize=True, speed=speed).cpu()
                yield {'tts_speech': speech}
        finally:
            if incoming:
                with self._bi_cv:
                    self._bi_incoming -= 1
            if call is not None:
                self._bi_unregister(call)
            if slot is not None:
                self._exit_shared(slot)
            with self.lock:
                if self._token_log is not None:
                    self._token_log[this_uuid] = list(self.tts_speech_token_dict.get(this_uuid, []))
                self.tts_speech_token_dict.pop(this_uuid, None)
                self.llm_end_dict.pop(this_uuid, None)
                self.hift_cache_dict.pop(this_uuid, None)
                self._hift_pin.pop(this_uuid, None)
                self._flow_caches.pop(this_uuid, None)

    # ---- model.py:336-401 -------------------------------------------------------------------------------------
    def tts(self, text=torch.zeros(1, 0, dtype=torch.int32), flow_embedding=torch.zeros(0, 192), llm_embedding=torch.zeros(0, 192),
            prompt_text=torch.zeros(1, 0, dtype=torch.int32),
            llm_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            flow_prompt_speech_token=torch.zeros(1, 0, dtype=torch.int32),
            prompt_speech_feat=torch.zeros(1, 0, 80), source_speech_token=torch.zeros(1, 0, dtype=torch.int32), stream=False, speed=1.0,
            force_len=None, device_output=False, **kwargs):
        """force_len (an extension; the reference swallows unknown keywords in **kwargs): synthetic-weights mode of SURVEY.md §8(d) —
        exactly that many speech tokens, EOS and fill ids never drawn — so that benchmark work is deterministic without a checkpoint.
        device_output (an extension, non-streaming coalesced calls only): 'tts_speech' stays on the model's device instead of the reference's
        CPU tensor -- for a caller that hands the waveform to a device-side consumer (cv2amd/shard.py: the RCCL gather of configs[3])."""
        from collections.abc import Generator
        vc = source_speech_token.shape[1] != 0                   # vc_job (model.py:141-143): the tokens are given, no LLM
        bistream = isinstance(text, Generator)                    # llm_job's bistream branch (model.py:120-128)
        if not vc and not bistream and not isinstance(text, torch.Tensor):
            raise TypeError('text must be a tensor of token ids or a generator of such tensors')
        this_uuid = str(uuid.uuid1())
        # the current device is per thread and a new thread starts on device 0: callers run tts() from pool threads (the evaluation
        # harness, evaluation/cosyvoice_synthesizer.py:260) while the model may live on another GPU of the node (one rank per GPU)
        if self._sched_log is not None:
            self._sched_log.append((time.perf_counter(), 'call', {}))
        torch.cuda.set_device(self.device)
        if self._sched_log is not None:
            self._sched_log.append((time.perf_counter(), 'device', {}))
        if self._on_call is not None:
            self._on_call(this_uuid)
        with self.lock:
            self.tts_speech_token_dict[this_uuid], self.llm_end_dict[this_uuid] = [], False
            self.hift_cache_dict[this_uuid] = None
        if bistream and not vc:
            with self._bi_cv:                                                 # the hub waits (briefly) for calls on their way in before its first feed
                self._bi_incoming += 1
        dev = self.device
        try:
            fpt = flow_prompt_speech_token.to(dev)
            feat = prompt_speech_feat.to(dev)
            femb = flow_embedding.to(dev)
        except BaseException:
            if bistream and not vc:
                with self._bi_cv:
                    self._bi_incoming -= 1
            raise
        if vc or bistream:
            yield from self._tts_pulled(text, prompt_text, llm_prompt_speech_token, source_speech_token, fpt, feat, femb, this_uuid,
                                        stream, speed, vc)
            return
        if stream is not True and self.max_batch > 1 and self._noise_hook is None and self._trace is None:
            p = self._Pending()
            p.text, p.prompt_text, p.llm_ptok = text.to(dev), prompt_text.to(dev), llm_prompt_speech_token.to(dev)
            p.fpt, p.feat, p.femb, p.speed, p.uuid, p.force_len = fpt, feat, femb, speed, this_uuid, force_len
            p.dev_out = bool(device_output)
            p.done, p.speech, p.exc = threading.Event(), None, None
            try:
                yield {'tts_speech': self._tts_coalesced(p)}
            finally:
                with self.lock:
                    self.tts_speech_token_dict.pop(this_uuid, None)
                    self.llm_end_dict.pop(this_uuid, None)
                    self.hift_cache_dict.pop(this_uuid, None)
                self._hift_pin.pop(this_uuid, None)
                self._flow_caches.pop(this_uuid, None)
            return
        t_call = time.perf_counter()
        slot = self._enter_shared()
        hop, la = self.token_hop_len, self.flow.pre_lookahead_len
        try:
            if self._sched_log is not None:
                self._sched_log.append((time.perf_counter(), 'slot', dict(slot=slot)))
            text_d, ptext_d, lptok_d = text.to(dev), prompt_text.to(dev), llm_prompt_speech_token.to(dev)
            if stream is True:
                token_offset = 0
                prompt_token_pad = int(np.ceil(fpt.shape[1] / hop) * hop - fpt.shape[1])
                # flow-cache capacity in frames: the forced length, else a typical 8 speech tokens per text token (grown when exceeded)
                n_guess = force_len if force_len is not None else min(self._limits[2], 8 * int(text.shape[1]) + 64)
                cap_hint = self.flow.token_mel_ratio * (fpt.shape[1] + n_guess)
                pkey = None
                if self.flow_cache and self.prompt_cache_max > 0:              # identity of the prompt: tokens + checksums of mel and embedding
                    pkey = (tuple(flow_prompt_speech_token.flatten().tolist()), float(prompt_speech_feat.double().sum()),
                            float(flow_embedding.double().sum()))
                with self.lock:
                    self._first_pending[this_uuid] = t_call
                    self._first_need[slot] = hop + prompt_token_pad + la        # (model.py:353-357: what the first chunk waits for)
                self._llm_start(slot, text_d, ptext_d, lptok_d, force_len)     # prefill draws token 0; the first pass of the loop below
                while True:                                                    # requests the rest of the first chunk's tokens
                    this_tok, finished = None, False
                    with self.run_lock:
                        this_token_hop_len = hop + prompt_token_pad if token_offset == 0 else hop
                        need = token_offset + this_token_hop_len + la
                        toks = self._llm_poll(this_uuid, slot, need)
                        ended = self.llm_end_dict[this_uuid]
                        if len(toks) >= need:
                            if not ended:                                      # the next chunk's tokens beyond what is already enqueued (by this
                                self._llm_advance(need + hop - max(len(toks), self._enq.get(slot, 0)), shared=True)    # or another stream): the burst
                            this_tok = torch.tensor(toks[:need], dtype=torch.int32).unsqueeze(0)          # overlaps this chunk's flow + HiFT
                        elif ended:
                            finished = True
                        else:
                            self._llm_advance(need - len(toks))                # nothing in flight any more (ids above EOS are steps without a token)
                    if this_tok is not None:
                        speech = self._chunk_submit(this_tok, fpt, feat, femb, token_offset, this_uuid, True, False, cap_hint, pkey,
                                                    first_slot=slot if token_offset == 0 else None)
                        token_offset += this_token_hop_len
                        yield {'tts_speech': speech}
                    if finished:
                        break
                this_tok = torch.tensor(self.tts_speech_token_dict[this_uuid], dtype=torch.int32).unsqueeze(0)
                speech = self._chunk_submit(this_tok, fpt, feat, femb, token_offset, this_uuid, False, True)
                yield {'tts_speech': speech}
            else:
                self._llm_start(slot, text_d, ptext_d, lptok_d, force_len)
                n_min = force_len if force_len is not None else int(text.shape[1] * self.min_token_text_ratio)
                n_max = force_len if force_len is not None else int(text.shape[1] * self.max_token_text_ratio)
                done_steps = 1                                                 # the prefill drew step 0
                while True:
                    with self.run_lock:
                        # the call cannot end before min_len steps (llm.py:242-250): one burst up to there, then 32 at a time
                        burst = max(1, min(max(32, n_min - done_steps), n_max - done_steps))
                        self._llm_advance(burst)
                        done_steps += burst
                        toks = self._llm_poll(this_uuid, slot)
                    if self.llm_end_dict[this_uuid]:
                        break
                with self.run_lock:
                    this_tok = torch.tensor(toks, dtype=torch.int32).unsqueeze(0)
                    speech = self.token2wav(this_tok, fpt, feat, femb, 0, this_uuid, finalize=True, speed=speed).cpu()
                yield {'tts_speech': speech}
        finally:
            with self.run_lock, self.llm_lock:                                # an abandoned generator leaves a live slot behind: park it
                with torch.cuda.stream(self.llm_stream):
                    self.llm.park(slot)
                self.llm_stream.synchronize()
            with self.lock:
                self._first_pending.pop(this_uuid, None)
                self._first_need.pop(slot, None)
            self._exit_shared(slot)
            with self.lock:
                self.tts_speech_token_dict.pop(this_uuid, None)
                self.llm_end_dict.pop(this_uuid, None)
                self.hift_cache_dict.pop(this_uuid, None)
                self._hift_pin.pop(this_uuid, None)
                self._flow_caches.pop(this_uuid, None)
