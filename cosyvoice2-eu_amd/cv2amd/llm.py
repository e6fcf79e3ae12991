"""Host side of stage 1: owns the device buffers and drives cv2_llm_* (csrc/llm.hip).

Mirrors the engine seam the reference already has for vLLM (cosyvoice/llm/llm.py:651-680): requests are added
with their prompt embeddings (`lm_input`, llm.py:641), `step()` advances every active request, ids are read back.
"""
import ctypes as C

import torch

from . import lib as L
from . import weights as W

EOS = 6561
MODE_GREEDY, MODE_RAS = 0, 1


class LLMEngine:
    def __init__(self, sd, device='cuda:0', max_seqs=32, max_pos=2048, max_out=2048, max_prefill_rows=None, sampling=None):
        """sampling: dict(top_p, top_k, win_size, tau_r) of ras_sampling (conf/cosyvoice2.yaml:33-37); None = those defaults."""
        self.device = torch.device(device)
        self.lib = L.lib()
        dev = self.device
        hidden = sd['llm.model.model.norm.weight'].numel()
        layers = 0
        while f'llm.model.model.layers.{layers}.input_layernorm.weight' in sd:
            layers += 1
        pfx = 'llm.model.model.layers.0.'
        n_q = sd[pfx + 'self_attn.q_proj.weight'].shape[0] // 64
        n_kv = sd[pfx + 'self_attn.k_proj.weight'].shape[0] // 64
        inter = sd[pfx + 'mlp.gate_proj.weight'].shape[0]
        vocab = sd['llm_decoder.weight'].shape[0]
        vocab_pad = (vocab + 15) // 16 * 16
        self.dims = L.LlmDims(hidden=hidden, inter=inter, layers=layers, n_q=n_q, n_kv=n_kv, vocab=vocab,
                              vocab_pad=vocab_pad, eos=EOS, max_seqs=max_seqs, max_pos=max_pos, max_out=max_out,
                              rms_eps=1e-6,
                              max_prefill_rows=(max_prefill_rows if max_prefill_rows is not None else min(max_seqs * 512, max_seqs * max_pos)),
                              **{k: (sampling or {}).get(k, v) for k, v in (('top_p', 0.8), ('top_k', 25), ('win_size', 10), ('tau_r', 0.1))})
        self.hidden, self.max_seqs, self.max_out, self.vocab, self.vocab_pad = hidden, max_seqs, max_out, vocab, vocab_pad
        keep = []

        def dv(t, dtype=torch.float32):
            t = t.to(device=dev, dtype=dtype).contiguous()
            keep.append(t)
            return t

        self._layers = (L.LlmLayer * layers)()
        for i in range(layers):
            p = f'llm.model.model.layers.{i}.'
            wqkv = torch.cat([sd[p + 'self_attn.q_proj.weight'], sd[p + 'self_attn.k_proj.weight'],
                              sd[p + 'self_attn.v_proj.weight']], dim=0).to(dev)
            bqkv = torch.cat([sd[p + 'self_attn.q_proj.bias'], sd[p + 'self_attn.k_proj.bias'], sd[p + 'self_attn.v_proj.bias']])
            wgu = W.interleave_tiles(sd[p + 'mlp.gate_proj.weight'].to(dev), sd[p + 'mlp.up_proj.weight'].to(dev))
            ly = self._layers[i]
            for name, t in (('wqkv', W.pack_bf16(wqkv)), ('wo', W.pack_bf16(sd[p + 'self_attn.o_proj.weight'].to(dev))),
                            ('wgu', W.pack_bf16(wgu)), ('wdown', W.pack_bf16(sd[p + 'mlp.down_proj.weight'].to(dev)))):
                keep.append(t)
                setattr(ly, name, t.data_ptr())
            ly.bqkv = dv(bqkv).data_ptr()
            ly.ln1 = dv(sd[p + 'input_layernorm.weight']).data_ptr()
            ly.ln2 = dv(sd[p + 'post_attention_layernorm.weight']).data_ptr()
        wdec = W.pack_bf16(sd['llm_decoder.weight'].to(dev))
        keep.append(wdec)
        bdec = torch.zeros(vocab_pad)
        bdec[:vocab] = sd['llm_decoder.bias']
        cos, sin = W.rope_tables(max_pos)
        # one table [text ++ speech ++ (sos, task)] with the three embeddings as views: the input rows of a feed (any mixture of the
        # three kinds, llm.py:766-786) are ONE gather over host-built indices
        n_text = sd['llm.model.model.embed_tokens.weight'].shape[0]
        self.emb_all = dv(torch.cat([sd['llm.model.model.embed_tokens.weight'].float(), sd['speech_embedding.weight'].float(),
                                     sd['llm_embedding.weight'].float()], dim=0))
        self.text_emb, self.speech_emb, self.llm_emb = self.emb_all[:n_text], self.emb_all[n_text:n_text + vocab], self.emb_all[n_text + vocab:]
        self.IDX_SPEECH, self.IDX_SOS, self.IDX_TASK = n_text, n_text + vocab, n_text + vocab + 1
        self.bdec = dv(bdec)                           # llm_decoder.bias on the device (padded to vocab_pad)
        self._w = L.LlmWeights(layers=C.cast(self._layers, C.POINTER(L.LlmLayer)),
                               final_norm=dv(sd['llm.model.model.norm.weight']).data_ptr(), wdec=wdec.data_ptr(),
                               bdec=self.bdec.data_ptr(), speech_emb=self.speech_emb.data_ptr(),
                               rope_cos=dv(cos).data_ptr(), rope_sin=dv(sin).data_ptr())
        self.state = torch.zeros(max_seqs, L.STATE_STRIDE, dtype=torch.int32, device=dev)
        self.out_tokens = torch.zeros(max_seqs, max_out, dtype=torch.int32, device=dev)
        self.logits = torch.zeros(32, vocab_pad, dtype=torch.float32, device=dev)
        self._io = L.LlmIO(state=self.state.data_ptr(), out_tokens=self.out_tokens.data_ptr(), logits=self.logits.data_ptr())
        nbytes = self.lib.cv2_llm_workspace_bytes(C.byref(self.dims))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._keep = keep
        h = C.c_void_p()
        L.check(self.lib.cv2_llm_create(C.byref(self.dims), C.byref(self._w), C.byref(self._io), self.workspace.data_ptr(),
                                        nbytes, C.byref(h)))
        self.handle = h
        # a one-launch step whose in-launch hand-off timed out (CV2_ST_ERR = 3) commits nothing; the host clears the flag, repeats the
        # steps and keeps this engine on the launches for a while (_recover_handoff): 30 s after the first time-out, twice as long after
        # every further one (up to an hour) -- contention that broke the step's forward progress once is given time to go away
        self._chain_off_until, self._chain_backoff, self.handoff_recoveries = 0.0, 30.0, 0
        self.weight_bytes = sum(t.numel() * t.element_size() for t in keep if t.dtype == torch.int16)

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.cv2_llm_destroy(self.handle)
        except Exception:
            pass

    # ---- llm.py:625-641 -------------------------------------------------------------------------
    def build_lm_input(self, text, prompt_text, prompt_speech_token):
        """[sos, embed_tokens(prompt_text ++ text), task_id, speech_embedding(prompt_speech)] -> [L0, hidden] fp32."""
        dev = self.device
        ids = torch.cat([prompt_text.reshape(-1), text.reshape(-1)]).to(dev).long()
        ps = prompt_speech_token.reshape(-1).to(dev).long()
        return torch.cat([self.llm_emb[0:1], self.text_emb[ids], self.llm_emb[1:2], self.speech_emb[ps]], dim=0).contiguous()

    def add_request(self, slot, lm_input, min_len, max_len, mode=MODE_GREEDY, seed=0, force_len=False):
        """Step 0 of inference_wrapper (llm.py:684-719): prefill + first draw for `slot`."""
        self._init_state(slot, min_len, max_len, mode, seed, force_len)
        assert lm_input.dtype == torch.float32 and lm_input.is_contiguous() and lm_input.shape[1] == self.hidden
        L.check(self.lib.cv2_llm_prefill(self.handle, slot, L.ptr(lm_input), lm_input.shape[0], L.stream_ptr()))

    def _init_state(self, slot, min_len, max_len, mode, seed, force_len):
        # every step emits at most one token into out_tokens[slot, :max_out]: a request may not run longer than the buffer
        # (the reference has no such buffer: `for i in range(max_len)`, llm.py:684; max_out is sized for 20 x the text limit)
        if max_len > self.max_out:
            import logging
            logging.warning('LLM request max_len %d exceeds the output buffer (%d tokens): decode is capped there', max_len, self.max_out)
            max_len = self.max_out
        min_len = min(min_len, max_len)
        st = torch.zeros(L.STATE_STRIDE, dtype=torch.int32)
        st[L.ST_MINLEN], st[L.ST_MAXLEN], st[L.ST_MODE], st[L.ST_FORCE] = min_len, max_len, mode, int(force_len)
        st[L.ST_SEED_LO] = (seed & 0x7FFFFFFF) - (seed & 0x80000000)
        st[L.ST_SEED_HI] = ((seed >> 32) & 0x7FFFFFFF) - ((seed >> 32) & 0x80000000)
        self.state[slot].copy_(st)

    def add_requests(self, slots, lm_inputs, lens_minmax, mode=MODE_GREEDY, seed=0, force_len=False):
        """Step 0 for several slots in one pass over the weights (cv2_llm_prefill_batch): lm_inputs = list of [L0, hidden] fp32."""
        n = len(slots)
        for slot, (mn, mx) in zip(slots, lens_minmax):
            self._init_state(slot, mn, mx, mode, seed, force_len)
        emb = torch.cat([x.reshape(-1, self.hidden) for x in lm_inputs], 0).to(torch.float32).contiguous()
        sl = (C.c_int32 * n)(*slots)
        ln = (C.c_int32 * n)(*[x.shape[0] for x in lm_inputs])
        L.check(self.lib.cv2_llm_prefill_batch(self.handle, n, sl, ln, L.ptr(emb), L.stream_ptr()))

    def decode_kernel_desc(self, n_seqs):
        """What one decode step launches (for bench.py's roofline record)."""
        L24 = self.dims.layers
        if n_seqs <= 24 and self.lib.cv2_llm_one_launch_step(self.handle) and not self.chain_broken:
            rows = ('' if n_seqs == 1 else f'; {n_seqs} rows = {n_seqs} interleaved chains sharing every weight tile through one L2' if n_seqs == 3 else
                    f'; {n_seqs} rows = {(n_seqs + 1) // 2} interleaved chains of row PAIRS (k_step2: two MFMA columns per block)' if n_seqs <= 8 else
                    f'; {n_seqs} rows = {(n_seqs + 3) // 4} interleaved chains of FOUR rows (k_step4: four MFMA columns per block)')
            return (f'LLM decode step = ONE launch k_step ({L24} x {{Q, attention, O, gate/up, down}} roles + head as blocks of one grid, granule '
                    f'hand-offs{rows}) + k_sample, {n_seqs} row(s)')
        if n_seqs <= 16:
            return f'LLM decode step = one hipGraph replay ({L24} x {{k_qkv, k_attn, k_store(o), k_gateup, k_store(down)}} + head + k_sample), {n_seqs} row(s)'
        return (f'LLM decode step = one hipGraph replay ({L24} x {{k_prep, k_qkv, k_attn, k_prep, k_store(o)+norm2, k_gateup, k_store(down)}} + head + '
                f'k_sample), {n_seqs} rows')

    @property
    def chain_broken(self):
        """True while the engine keeps off the one-launch step after a hand-off time-out"""
        import time
        return time.monotonic() < self._chain_off_until

    def step(self, n_seqs, n_steps=1, shared=False):
        """shared: kernels of other streams run beside these steps (CV2_DECODE_SHARED: the launches instead of the one-launch step)."""
        L.check(self.lib.cv2_llm_decode_ex(self.handle, n_seqs, n_steps, 1 if (shared or self.chain_broken) else 0, L.stream_ptr()))

    def step_rows(self, slots, n_steps=1, shared=False):
        """n_steps over the live slots only (cv2_llm_decode_rows): row r of every step serves slot slots[r]."""
        arr = (C.c_int32 * len(slots))(*slots)
        L.check(self.lib.cv2_llm_decode_rows(self.handle, arr, len(slots), n_steps, 1 if (shared or self.chain_broken) else 0, L.stream_ptr()))

    def _recover_handoff(self, slots):
        """CV2_ST_ERR = 3: a hand-off inside the one-launch step (k_step, csrc/chain.h) did not arrive within its bound -- the device
        was contended in a way that broke the step's forward progress.  k_sample commits nothing for such a step (nor for the steps
        enqueued behind it): tokens, positions and the pending input are those before it and the KV rows it wrote are rewritten by the
        repeat.  The flag is cleared, the caller's loop issues the missing steps again, and for the next 30 s (doubling with every
        further time-out, up to an hour) this engine runs every step on the launches (the structure that needs no co-residency)."""
        import logging
        import time
        for s in slots:
            self.state[s, L.ST_ERR] = 0
        self._chain_off_until = time.monotonic() + self._chain_backoff
        self.handoff_recoveries += 1
        logging.warning('LLM decode: an in-launch hand-off of the one-launch step timed out (slot %s); the step is repeated on the launches and '
                        'this engine stays on them for %.0f s', list(slots), self._chain_backoff)
        self._chain_backoff = min(2 * self._chain_backoff, 3600.0)

    ERR_MSG = 'sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!'      # llm.py:249
    ERR_HANDOFF = 'LLM decode: an in-launch hand-off (k_step) timed out; the device was too contended for the step to finish'

    def _err(self, code):
        return RuntimeError(self.ERR_HANDOFF if int(code) == 3 else self.ERR_MSG)

    def read(self, n_seqs, raise_on_error=True):
        """(state [n, 16] int32 cpu, list of emitted-token lists). One device->host sync.  A slot whose sampler gave up (ST_ERR,
        llm.py:242-250) is finished; with raise_on_error=False the caller inspects st[:, ST_ERR] and fails only that request."""
        st = self.state[:n_seqs].cpu()
        bad = [b for b in range(n_seqs) if int(st[b, L.ST_ERR]) == 3]
        if bad:
            self._recover_handoff(bad)
            st[bad, L.ST_ERR] = 0
        toks = self.out_tokens[:n_seqs].cpu()
        if raise_on_error:
            for b in range(n_seqs):
                if int(st[b, L.ST_ERR]):
                    raise self._err(st[b, L.ST_ERR])
        return st, [toks[b, :min(int(st[b, L.ST_NOUT]), self.max_out)].tolist() for b in range(n_seqs)]

    def read_slot(self, slot):
        """(state row [16] int32 cpu, emitted tokens of one slot).  One device->host sync."""
        st = self.state[slot].cpu()
        if int(st[L.ST_ERR]) == 3:
            self._recover_handoff([slot])
            st[L.ST_ERR] = 0
        if int(st[L.ST_ERR]):
            raise self._err(st[L.ST_ERR])
        n = min(int(st[L.ST_NOUT]), self.max_out)
        return st, self.out_tokens[slot, :n].cpu().tolist()

    def park(self, slot=None):
        """Mark a slot (or every slot) finished: a parked slot idles on its last position when a decode step covers it."""
        if slot is None:
            self.state[:, L.ST_DONE] = 1
        else:
            self.state[slot, L.ST_DONE] = 1

    def generate(self, requests, mode=MODE_GREEDY, seed=0, force_len=None, sync_every=16, min_ratio=2, max_ratio=20, batch_prefill=True,
                 return_errors=False, compact=True):
        """requests: list of (text, prompt_text, prompt_speech_token) int tensors.  Returns list of token lists; with
        return_errors=True also a list holding None or the RuntimeError of each request (a failing request finishes its own slot and
        never stops the others), otherwise the first error is raised."""
        n = len(requests)
        assert n <= self.max_seqs
        xs, mm = [], []
        forced = force_len is not None
        fl = list(force_len) if isinstance(force_len, (list, tuple)) else [force_len] * n     # one forced length for all, or one per request
        for b, (text, ptxt, ptok) in enumerate(requests):
            mn, mx = int(text.numel() * min_ratio), int(text.numel() * max_ratio)
            if forced:
                mn, mx = fl[b], fl[b]
            xs.append(self.build_lm_input(text, ptxt, ptok))
            mm.append((mn, mx))
        if batch_prefill and sum(x.shape[0] for x in xs) <= self.dims.max_prefill_rows:
            self.add_requests(list(range(n)), xs, mm, mode, seed, forced)
        else:
            for b in range(n):
                self.add_request(b, xs[b], mm[b][0], mm[b][1], mode, seed, forced)
        while True:
            st = self.state[:n].cpu()                 # a poll reads the state records only (2 KB); the tokens are fetched once, at the end
            S = st.tolist()
            bad = [b for b in range(n) if S[b][L.ST_ERR] == 3]
            if bad:                                   # nothing of the failed steps was committed: the loop below issues them again
                self._recover_handoff(bad)
                for b in bad:
                    S[b][L.ST_ERR] = 0
            if all(r[L.ST_DONE] for r in S):
                st, toks = self.read(n, raise_on_error=not return_errors)
                if return_errors:
                    return toks, [self._err(st[b, L.ST_ERR]) if int(st[b, L.ST_ERR]) else None for b in range(n)]
                return toks
            if not return_errors:
                for r in S:
                    if r[L.ST_ERR]:
                        raise self._err(r[L.ST_ERR])
            # no live request can finish before its min_len (EOS is re-drawn until then, llm.py:242-250): poll again only after the
            # earliest possible finish, then every sync_every steps (a finished slot idles inside a burst)
            live = [b for b in range(n) if not S[b][L.ST_DONE]]
            to_min = min(S[b][L.ST_MINLEN] - S[b][L.ST_STEP] for b in live)
            to_max = max(S[b][L.ST_MAXLEN] - S[b][L.ST_STEP] for b in live)
            k = max(1, min(max(sync_every, to_min), to_max))
            if compact and len(live) < n:
                self.step_rows(live, k)               # the finished requests' rows are not computed any more
            else:
                self.step(n, k)

    # ---- llm.py:721-834 ------------------------------------------------------------------------------------------
    def new_bistream(self, slot, prompt_text, prompt_speech_token, mode=MODE_GREEDY, seed=0, mix_ratio=(5, 15)):
        """Host bookkeeping of one Qwen2LM.inference_bistream call on slot `slot` (no device work)."""
        return BiStream(self, slot, prompt_text, prompt_speech_token, mode, seed, mix_ratio)

    def bi_poll(self, streams):
        """Read the slots of the RUNNING streams (one copy of the state records, one of the new out_tokens entries): new ids, fill /
        EOS stops (llm.py:805-811, 826-832), errors.  Blocks until the current stream's work has run."""
        run = [b for b in streams if b.running]
        if not run:
            return
        st = self.state.cpu().tolist()
        bad = [b.slot for b in run if st[b.slot][L.ST_ERR] == 3]
        if bad:
            self._recover_handoff(bad)
            for s_ in bad:
                st[s_][L.ST_ERR] = 0
        lo = min(b.n_read for b in run)
        hi = max(min(st[b.slot][L.ST_NOUT], self.max_out) for b in run)
        toks = self.out_tokens[:, lo:hi].cpu().tolist() if hi > lo else None
        for b in run:
            b._polled(st[b.slot], toks[b.slot][b.n_read - lo:min(st[b.slot][L.ST_NOUT], self.max_out) - lo] if toks is not None else [])

    def bi_feed(self, streams, prepared=False):
        """Every idle stream that has its next input (the text block after a fill, the first [sos, text / prompt-speech blocks], or the
        final [pending input, remaining text, task id]; llm.py:766-786, 817) is fed in ONE pass over the weights
        (cv2_llm_extend_batch) followed by one draw per slot.  Returns the streams fed.  prepared: the caller has already called
        next_feed() on the idle streams (under the lock its text producers push under); only those results are used."""
        feeds = []
        for b in streams:
            if not b.running and not b.finished:
                f = b._pending if prepared else b.next_feed()
                if f is not None:
                    feeds.append((b, f))
        if not feeds:
            return []
        dev = self.device
        rows = torch.tensor([b._state_row(final) for b, (_, final) in feeds], dtype=torch.int32)
        self.state.index_copy_(0, torch.tensor([b.slot for b, _ in feeds], dtype=torch.long).to(dev), rows.to(dev))
        emb = self.emb_all[torch.tensor([i for _, (idx, _) in feeds for i in idx], dtype=torch.long).to(dev)]
        cap = (self.dims.max_prefill_rows // 128) * 128
        r0, grp = 0, []

        def flush():
            nonlocal grp
            if grp:
                n = len(grp)
                a0 = grp[0][2]
                arr = lambda v: (C.c_int32 * n)(*v)     # noqa: E731
                part = emb[a0:a0 + sum(g[1] for g in grp)]
                L.check(self.lib.cv2_llm_extend_batch(self.handle, n, arr([g[0].slot for g in grp]), arr([g[1] for g in grp]),
                                                      arr([g[0].pos for g in grp]), L.ptr(part), L.stream_ptr()))
                grp = []
        for b, (idx, final) in feeds:
            n = len(idx)
            if n > cap:                                   # one feed beyond the GEMM path's capacity: 32 rows at a time
                flush()
                part = emb[r0:r0 + n]
                L.check(self.lib.cv2_llm_extend(self.handle, b.slot, L.ptr(part), n, b.pos, L.stream_ptr()))
            else:
                if sum(g[1] for g in grp) + n > cap:
                    flush()
                grp.append((b, n, r0))
            r0 += n
        flush()
        self._keep_rows = emb                            # alive until the stream has consumed it
        for b, (idx, final) in feeds:
            b._last_feed = idx
            b._fed(len(idx), final)
        return [b for b, _ in feeds]

    def bi_burst(self, streams, n_steps, shared=False):
        """n_steps decode steps over the running streams' slots (a slot that stops on a fill id inside the burst idles from there)."""
        run = sorted(b.slot for b in streams if b.running)
        if run and n_steps > 0:
            self.step_rows(run, n_steps, shared=shared)
            for b in streams:
                if b.running and b.eta is not None:
                    b.eta = max(0, b.eta - n_steps)

    @staticmethod
    def bi_burst_len(streams, burst=16):
        """Steps worth enqueueing before the next poll: up to the earliest known stop (the forced fill, llm.py:796-798) or the earliest
        point a consumer waits for; 0 when every running stream is expected to have stopped already."""
        run = [b for b in streams if b.running]
        if not run or all(b.eta == 0 for b in run):
            return 0
        c = [b.eta for b in run if b.eta] + [b.want for b in run if b.want is not None and b.want > 0]
        # a stream that has not shown its first fill id yet stops at an unknown step (a trained model: when the prompt's last 15-token
        # block is complete, see _fed): half bursts until then, so that few steps run behind the stop
        if any(b.eta is None and not b.final_fed and not b.seen_fill for b in run):
            c.append(max(1, burst // 2))
        return max(1, min(c + [burst]))

    def bistream(self, slot, text, prompt_text, prompt_speech_token, mode=MODE_GREEDY, seed=0, mix_ratio=(5, 15), burst=16,
                 n_seqs=None, on_device=None):
        """Qwen2LM.inference_bistream on slot `slot`: `text` is an iterable of int tensors [1, n] arriving over time; yields speech
        token ids as the reference generator does.  The LM input interleaves mix_ratio[0] text tokens with mix_ratio[1] speech
        tokens; the device stops the slot on the fill id (k_sample, CV2_ST_WAIT) and the next text block is fed exactly where the
        reference breaks out of its inner `while True` (llm.py:807-808).  The bookkeeping lives in BiStream (shared with the
        scheduler's rounds over several calls, cosyvoice/cli/model.py); `on_device(fn)` runs fn() under the caller's device lock /
        stream; the next piece of `text` is only pulled when the slot has nothing left to feed, as the reference's `for` does."""
        run = on_device or (lambda fn: fn())
        bs = self.new_bistream(slot, prompt_text, prompt_speech_token, mode, seed, mix_ratio)
        it = iter(text)
        while not bs.finished:
            if bs.running:
                run(lambda: self.bi_poll([bs]))
                yield from bs.take()
                if bs.err is not None:
                    raise bs.err
                if bs.running:
                    run(lambda: self.bi_burst([bs], self.bi_burst_len([bs], burst)))
                continue
            if run(lambda: self.bi_feed([bs])):
                n = self.bi_burst_len([bs], burst)
                if n:
                    run(lambda: self.bi_burst([bs], n))
                continue
            try:
                bs.push(next(it))
            except StopIteration:
                bs.close()

    def generate_fixed(self, requests, n_tokens, mode=MODE_RAS, seed=0):
        """Synthetic-weights mode: exactly n_tokens per request, EOS never drawn, no host sync inside the
        decode loop.  Returns (list of token lists, (start, end) torch events bracketing the n_tokens-1 decode steps)."""
        n = len(requests)
        assert n <= self.max_seqs and n_tokens <= self.max_out
        xs = [self.build_lm_input(text, ptxt, ptok) for (text, ptxt, ptok) in requests]
        self.add_requests(list(range(n)), xs, [(n_tokens, n_tokens)] * n, mode, seed, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.step(n, n_tokens - 1)
        e1.record()
        st, toks = self.read(n)
        assert bool(st[:, L.ST_DONE].all()) and all(len(t) == n_tokens for t in toks)
        return toks, (e0, e1)


class BiStream:
    """Host half of one Qwen2LM.inference_bistream call (llm.py:721-834): the text cache, the prompt speech tokens still to interleave,
    the pending LM input and the reference's `out_tokens` list, all as plain id lists (indices into LLMEngine.emb_all); the device
    half is the slot's state machine in k_sample.  No method touches the device: LLMEngine.bi_poll / bi_feed / bi_burst do, for one
    stream or for all the streams of a scheduler round at once."""

    def __init__(self, eng, slot, prompt_text, prompt_speech_token, mode, seed, mix_ratio=(5, 15)):
        import collections
        self.eng, self.slot, self.mode, self.seed, self.mix = eng, slot, mode, seed, tuple(mix_ratio)
        self.text_cache = [int(t) for t in prompt_text.reshape(-1).tolist()]                       # llm.py:750-757
        self.sp_left = [eng.IDX_SPEECH + int(t) for t in prompt_speech_token.reshape(-1).tolist()]
        self.n_prompt_speech, self.seen_fill = len(self.sp_left), False
        self.lm_input = [eng.IDX_SOS]                                                              # llm.py:743
        self.outs = []                        # the reference's out_tokens (fill / EOS entries included)
        self.pieces, self.text_done = collections.deque(), False
        self.n_read, self.pos = 0, 0          # entries of the device's out_tokens already seen / KV rows fed so far
        self.started = self.running = self.finished = self.final_fed = False
        self.eta = None                       # expected decode steps until the slot stops (None: unknown)
        self.want = None                      # tokens a consumer is waiting for (scheduler hint for the burst length)
        self.err = None
        self._row = None                      # the slot's state record at the last poll
        self._new = []                        # emitted ids not yet taken
        self._pending = None
        self._last_feed = ()

    # ---- text side (any thread; the scheduler serialises these with its own lock) ----
    def push(self, piece):
        self.pieces.append([int(t) for t in piece.reshape(-1).tolist()])

    def close(self):
        self.text_done = True

    def take(self):
        """the ids emitted since the last take (what the reference generator has yielded in the meantime)"""
        new, self._new = self._new, []
        return new

    # ---- llm.py:759-786, 817 for an idle slot ----
    def next_feed(self):
        """(indices of the next input rows, final) or None while the call waits for text.  Consumes text pieces exactly as the
        reference's `for this_text in text` does: one piece at a time until one of them leads to a decode."""
        if self._pending is not None:
            return self._pending
        n_text, n_speech = self.mix
        fill = EOS + 2
        while self.pieces:
            self.text_cache += self.pieces.popleft()
            while self.sp_left:                                                                    # llm.py:766-774
                if len(self.text_cache) < n_text:
                    break
                self.lm_input = self.lm_input + self.text_cache[:n_text] + self.sp_left[:n_speech]
                self.text_cache, self.sp_left = self.text_cache[n_text:], self.sp_left[n_speech:]
            if self.sp_left:
                continue
            last_fill = bool(self.outs) and self.outs[-1] == fill                                  # llm.py:776-786
            if last_fill or (not self.outs and len(self.lm_input) == 1):
                if len(self.text_cache) < n_text:
                    continue
                self.lm_input = self.text_cache[:n_text] if last_fill else self.lm_input + self.text_cache[:n_text]
                self.text_cache = self.text_cache[n_text:]
            self._pending = (list(self.lm_input), False)
            return self._pending
        if self.text_done:                                                                         # llm.py:817 (the stale lm_input goes in again)
            self._pending = (self.lm_input + self.text_cache + [self.eng.IDX_TASK], True)
            return self._pending
        return None

    def _state_row(self, final):
        """the slot's state record for the feed: a fresh one for the first feed, the polled one with the stop flags cleared afterwards"""
        if not self.started:
            eng, seed = self.eng, self.seed
            row = [0] * L.STATE_STRIDE
            row[L.ST_MINLEN], row[L.ST_MAXLEN], row[L.ST_MODE] = 0, eng.max_out, self.mode
            row[L.ST_SEED_LO] = (seed & 0x7FFFFFFF) - (seed & 0x80000000)
            row[L.ST_SEED_HI] = ((seed >> 32) & 0x7FFFFFFF) - ((seed >> 32) & 0x80000000)
            row[L.ST_NEXTFILL] = -1
        else:
            row = list(self._row)
        row[L.ST_DONE], row[L.ST_WAIT], row[L.ST_BIMODE] = 0, 0, (2 if final else 1)
        return row

    def _fed(self, n_rows, final):
        last_fill = bool(self.outs) and self.outs[-1] == EOS + 2
        self.started = self.running = True
        self.final_fed = final
        self.pos += n_rows
        self._pending = None
        # after a fill the next one is forced mix[1] + 1 entries later (llm.py:796-804): the feed draws the first of them
        # the FIRST feed ends inside the prompt's last block: a model trained on the 5 : 15 interleave completes that block (the prompt's
        # speech tokens modulo 15 are in it) and then asks for text; if it does not stop there the poll says so and eta becomes unknown
        if final:
            self.eta = None
        elif last_fill:
            self.eta = self.mix[1]
        else:
            self.eta = (self.mix[1] - self.n_prompt_speech % self.mix[1]) % self.mix[1]

    def _polled(self, row, new):
        self._row = row
        self.n_read += len(new)
        self.pos = row[L.ST_POS]
        for t in new:
            self.outs.append(t)
            if t < EOS:
                self._new.append(t)
            elif t == EOS + 2:
                self.seen_fill = True
        real = [t for t in new if t < EOS]
        if real:                                          # lm_input = speech_embedding[last emitted id] (llm.py:811)
            self.lm_input = [self.eng.IDX_SPEECH + real[-1]]
        err = row[L.ST_ERR]
        if err in (1, 3):
            self.err = self.eng._err(err)
        elif err == 2:
            self.err = ValueError('should not get token {}'.format(self.outs[-1]))
        if row[L.ST_DONE] or self.err is not None:        # stopped: on the fill id (waits for text), on EOS, on an error or out of room
            self.running = False
            self.eta = None
            if self.final_fed or self.err is not None or not row[L.ST_WAIT]:
                self.finished = True
        elif self.eta == 0:
            self.eta = None                               # did not stop where expected: poll at the default cadence
