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
        self.speech_emb = dv(sd['speech_embedding.weight'])
        self.bdec = dv(bdec)                           # llm_decoder.bias on the device (padded to vocab_pad)
        self.llm_emb = dv(sd['llm_embedding.weight'])
        self.text_emb = dv(sd['llm.model.model.embed_tokens.weight'])
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
        if n_seqs == 1 and self.lib.cv2_llm_one_launch_step(self.handle):
            return (f'LLM decode step = ONE launch k_step ({L24} x {{Q, attention, O, gate/up, down}} roles + head as blocks of one grid, granule '
                    f'hand-offs) + k_sample, 1 row')
        if n_seqs <= 16:
            return f'LLM decode step = one hipGraph replay ({L24} x {{k_qkv, k_attn, k_store(o), k_gateup, k_store(down)}} + head + k_sample), {n_seqs} row(s)'
        return (f'LLM decode step = one hipGraph replay ({L24} x {{k_prep, k_qkv, k_attn, k_prep, k_store(o)+norm2, k_gateup, k_store(down)}} + head + '
                f'k_sample), {n_seqs} rows')

    def step(self, n_seqs, n_steps=1, shared=False):
        """shared: kernels of other streams run beside these steps (CV2_DECODE_SHARED: the launches instead of the one-launch step)."""
        L.check(self.lib.cv2_llm_decode_ex(self.handle, n_seqs, n_steps, 1 if shared else 0, L.stream_ptr()))

    def step_rows(self, slots, n_steps=1, shared=False):
        """n_steps over the live slots only (cv2_llm_decode_rows): row r of every step serves slot slots[r]."""
        arr = (C.c_int32 * len(slots))(*slots)
        L.check(self.lib.cv2_llm_decode_rows(self.handle, arr, len(slots), n_steps, 1 if shared else 0, L.stream_ptr()))

    ERR_MSG = 'sampling reaches max_trials 100 and still get eos when ignore_eos is True, check your input!'      # llm.py:249
    ERR_HANDOFF = 'LLM decode: an in-launch hand-off (k_chain) timed out; the device was too contended for the step to finish'

    def _err(self, code):
        return RuntimeError(self.ERR_HANDOFF if int(code) == 3 else self.ERR_MSG)

    def read(self, n_seqs, raise_on_error=True):
        """(state [n, 16] int32 cpu, list of emitted-token lists). One device->host sync.  A slot whose sampler gave up (ST_ERR,
        llm.py:242-250) is finished; with raise_on_error=False the caller inspects st[:, ST_ERR] and fails only that request."""
        st = self.state[:n_seqs].cpu()
        toks = self.out_tokens[:n_seqs].cpu()
        if raise_on_error:
            for b in range(n_seqs):
                if int(st[b, L.ST_ERR]):
                    raise self._err(st[b, L.ST_ERR])
        return st, [toks[b, :min(int(st[b, L.ST_NOUT]), self.max_out)].tolist() for b in range(n_seqs)]

    def read_slot(self, slot):
        """(state row [16] int32 cpu, emitted tokens of one slot).  One device->host sync."""
        st = self.state[slot].cpu()
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
    def bistream(self, slot, text, prompt_text, prompt_speech_token, mode=MODE_GREEDY, seed=0, mix_ratio=(5, 15), burst=16,
                 n_seqs=None, on_device=None):
        """Qwen2LM.inference_bistream on slot `slot`: `text` is an iterable of int tensors [1, n] arriving over time; yields speech
        token ids as the reference generator does.  The LM input interleaves mix_ratio[0] text tokens with mix_ratio[1] speech
        tokens; the device stops the slot on the fill id (k_sample, CV2_ST_WAIT) and this loop feeds the next text block with
        cv2_llm_extend, exactly where the reference breaks out of its inner `while True` (llm.py:807-808).

        Same bookkeeping as the reference, including its quirks: the stale `lm_input` is replaced by the text block after a fill
        but fed again in front of the remaining text + task id at the end (llm.py:817).  `on_device(fn)` runs fn() under the
        caller's device lock / stream (the scheduler passes one); `n_seqs` = slots covered by a decode step (default slot + 1)."""
        n_text, n_speech = mix_ratio
        dev = self.device
        run = on_device or (lambda fn: fn())          # every device operation below goes through run()
        S = {'pos': 0, 'n_read': 0, 'started': False}     # KV rows fed so far / entries of the device's out_tokens already seen
        outs = []                                     # the reference's out_tokens (fill / EOS included)

        def emb(ids):
            return self.text_emb[ids.reshape(-1).to(dev).long()]

        def feed(rows, final=False):
            rows = rows.contiguous()
            if not S['started']:
                self.state[slot, L.ST_BIMODE], self.state[slot, L.ST_NEXTFILL] = 1, -1
                S['started'] = True
            self.state[slot, L.ST_DONE], self.state[slot, L.ST_WAIT], self.state[slot, L.ST_BIMODE] = 0, 0, (2 if final else 1)
            L.check(self.lib.cv2_llm_extend(self.handle, slot, L.ptr(rows), rows.shape[0], S['pos'], L.stream_ptr()))
            self._keep_rows = rows                    # keep the rows alive until the stream has consumed them

        def poll():
            st = self.state[slot].cpu()
            n = min(int(st[L.ST_NOUT]), self.max_out)
            new = self.out_tokens[slot, S['n_read']:n].cpu().tolist()
            S['n_read'], S['pos'] = n, int(st[L.ST_POS])
            return new, bool(st[L.ST_DONE]), int(st[L.ST_ERR])

        def init():
            self._init_state(slot, 0, self.max_out, mode, seed, False)
            return (self.speech_emb[prompt_speech_token.reshape(-1).to(dev).long()], self.llm_emb[0:1], emb(prompt_text))
        sp_left, lm_input, text_cache = run(init)

        def decode_until_stop():
            """the inner `while True` of llm.py:787-811 / 819-832: yields emitted ids until fill (mid) or EOS (final)."""
            nonlocal lm_input
            while True:
                new, done, err = run(poll)
                for t in new:
                    outs.append(t)
                    if t < EOS:
                        yield t
                if err in (1, 3):
                    raise self._err(err)
                if err == 2:
                    raise ValueError('should not get token {}'.format(outs[-1]))
                real = [t for t in new if t < EOS]
                if real:                              # lm_input = speech_embedding[last emitted id] (llm.py:811)
                    last = real[-1]
                    lm_input = run(lambda: self.speech_emb[last:last + 1])
                if done:
                    return
                run(lambda: self.step((n_seqs() if callable(n_seqs) else n_seqs) or slot + 1, burst))

        for this_text in text:
            def block():
                nonlocal sp_left, lm_input, text_cache
                text_cache = torch.cat([text_cache, emb(this_text)], dim=0)
                while sp_left.shape[0] != 0:                               # llm.py:766-774
                    if text_cache.shape[0] >= n_text:
                        lm_input = torch.cat([lm_input, text_cache[:n_text], sp_left[:n_speech]], dim=0)
                        text_cache, sp_left = text_cache[n_text:], sp_left[n_speech:]
                    else:
                        break
                if sp_left.shape[0] != 0:
                    return False
                last_fill = len(outs) != 0 and outs[-1] == EOS + 2         # llm.py:776-786
                if last_fill or (len(outs) == 0 and lm_input.shape[0] == 1):
                    if text_cache.shape[0] < n_text:
                        return False
                    lm_input = text_cache[:n_text] if last_fill else torch.cat([lm_input, text_cache[:n_text]], dim=0)
                    text_cache = text_cache[n_text:]
                feed(lm_input)
                return True
            if run(block):
                yield from decode_until_stop()

        def final():
            feed(torch.cat([lm_input, text_cache, self.llm_emb[1:2]], dim=0), final=True)     # llm.py:817
        run(final)
        yield from decode_until_stop()

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
