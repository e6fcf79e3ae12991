"""Host side of stage 2 (token -> mel): packs a `flow.pt` state dict into the HBM layout of include/cv2_amd.h and
drives cv2_flow_* (csrc/flow.hip).

Mirrors `CausalMaskedDiffWithXvec.inference` (cosyvoice/flow/flow.py:235-283): same arguments, same return value
`(mel [1, 80, T2] fp32, None)`; attributes the scheduler reads (`token_mel_ratio`, `pre_lookahead_len`,
`input_frame_rate`, cli/model.py:311,357,44) are kept.  There is no CPU fallback: without libcv2amd.so this raises.
"""
import ctypes as C

import torch

from . import lib as L
from . import weights as W


class Lin(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p)]


class Ln(C.Structure):
    _fields_ = [('g', C.c_void_p), ('b', C.c_void_p)]


class Conformer(C.Structure):
    _fields_ = [('norm_mha', Ln), ('norm_ff', Ln), ('qkv', Lin), ('pos', Lin), ('out', Lin), ('w1', Lin), ('w2', Lin)]


class TBlock(C.Structure):
    _fields_ = [('norm1', Ln), ('norm3', Ln), ('qkv', Lin), ('out', Lin), ('ff1', Lin), ('ff2', Lin)]


class Resnet(C.Structure):
    _fields_ = [('conv1', Lin), ('ln1', Ln), ('mlp', Lin), ('conv2', Lin), ('ln2', Ln), ('res', Lin)]


class UnetBlock(C.Structure):
    _fields_ = [('rn', Resnet), ('tb', TBlock * 4), ('tail', Lin)]


class FlowWeights(C.Structure):
    _fields_ = [('input_embedding', C.c_void_p), ('spk_w', C.c_void_p), ('spk_b', C.c_void_p),
                ('embed', Lin), ('embed_ln', Ln), ('pre1', Lin), ('pre2', Lin), ('enc', Conformer * 6),
                ('up_conv', Lin), ('up_embed', Lin), ('up_embed_ln', Ln), ('up', Conformer * 4), ('after_norm', Ln),
                ('enc_proj', Lin), ('time1', Lin), ('time2', Lin),
                ('down', UnetBlock), ('mid', UnetBlock * 12), ('up_blk', UnetBlock),
                ('final_conv', Lin), ('final_ln', Ln), ('final_proj', Lin), ('rand_noise', C.c_void_p)]


class FlowDims(C.Structure):
    _fields_ = [('max_rows', C.c_int32), ('max_seqs', C.c_int32), ('max_len', C.c_int32), ('n_timesteps', C.c_int32),
                ('cfg_rate', C.c_float)]


class FlowUtt(C.Structure):
    _fields_ = [('tokens', C.c_void_p), ('n_tok', C.c_int32), ('prompt_feat', C.c_void_p), ('n_prompt_feat', C.c_int32),
                ('embedding', C.c_void_p), ('mel_out', C.c_void_p)]


class FlowCacheRef(C.Structure):
    _fields_ = [('cache', C.c_void_p), ('cache_frames', C.c_int32), ('n_cached', C.c_int32), ('gen', C.c_int32)]


class FlowCache:
    """Per-stream state of cv2_flow_inference_chunk: the device buffer (zero-initialised) plus what the next call needs to know."""
    __slots__ = ('buf', 'frames', 'n_cached', 'gen')

    def __init__(self, engine, max_frames):
        self.frames = (int(max_frames) + 63) // 64 * 64
        nbytes = engine.lib.cv2_flow_cache_bytes(engine.handle, self.frames)
        if nbytes == 0:
            raise ValueError(f'flow cache: bad capacity {self.frames}')
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=engine.device)
        self.n_cached, self.gen = 0, 0


def _bind(lib):
    if getattr(lib, '_flow_bound', False):
        return
    lib.cv2_flow_workspace_bytes.restype = C.c_size_t
    lib.cv2_flow_workspace_bytes.argtypes = [C.POINTER(FlowDims)]
    lib.cv2_flow_create.argtypes = [C.POINTER(FlowDims), C.POINTER(FlowWeights), C.c_void_p, C.c_size_t, C.c_void_p,
                                    C.POINTER(C.c_void_p)]
    lib.cv2_flow_destroy.argtypes = [C.c_void_p]
    lib.cv2_flow_inference.argtypes = [C.c_void_p, C.POINTER(FlowUtt), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.cv2_flow_cache_bytes.restype = C.c_size_t
    lib.cv2_flow_cache_bytes.argtypes = [C.c_void_p, C.c_int32]
    lib.cv2_flow_cache_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.cv2_flow_inference_chunk.argtypes = [C.c_void_p, C.POINTER(FlowUtt), C.POINTER(FlowCacheRef), C.c_int32, C.c_int32, C.c_void_p]
    lib.cv2_flow_estimator.argtypes = [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p]
    lib.cv2_flow_encoder.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.cv2_gemm_bf16.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p]
    lib._flow_bound = True


def rand_noise():
    """flow_matching.py:197-198: set_all_random_seed(0); torch.randn([1, 80, 50 * 300]) on the CPU generator."""
    g = torch.Generator(device='cpu')
    g.manual_seed(0)
    return torch.randn([1, 80, 50 * 300], generator=g)


class FlowEngine:
    token_mel_ratio = 2
    pre_lookahead_len = 3
    input_frame_rate = 25

    def __init__(self, sd, device='cuda:0', max_utts=1, max_len=2048, n_timesteps=10, cfg_rate=0.7):
        """sd: state dict in the reference's flow.pt schema."""
        self.device = dev = torch.device(device)
        self.lib = L.lib()
        _bind(self.lib)
        self._keep = keep = []

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        def packed(w, pad_to=16):
            n = w.shape[0]
            if n % pad_to:
                w = torch.cat([w, w.new_zeros(pad_to - n % pad_to, w.shape[1])], 0)
            t = W.pack_bf16(w.to(dev))
            keep.append(t)
            return t.data_ptr()

        def lin(name, bias=True, pad_to=16):
            w = sd[name + '.weight']
            if w.dim() == 3:                                  # Conv1d [Co][Ci][k] -> [Co][k*Ci], column = tap*Ci + ci
                w = w.permute(0, 2, 1).reshape(w.shape[0], -1)
            b = None
            if bias:
                b = sd[name + '.bias']
                if b.numel() % pad_to:
                    b = torch.cat([b, b.new_zeros(pad_to - b.numel() % pad_to)])
            return Lin(packed(w, pad_to), f32(b) if b is not None else None)

        def ln(name):
            return Ln(f32(sd[name + '.weight']), f32(sd[name + '.bias']))

        def conformer(p):
            a = p + '.self_attn.'
            wq, bq = sd[a + 'linear_q.weight'], sd[a + 'linear_q.bias']
            wqkv = torch.cat([wq, wq, sd[a + 'linear_k.weight'], sd[a + 'linear_v.weight']], 0)
            bqkv = torch.cat([bq + sd[a + 'pos_bias_u'].reshape(-1), bq + sd[a + 'pos_bias_v'].reshape(-1),
                              sd[a + 'linear_k.bias'], sd[a + 'linear_v.bias']])
            return Conformer(ln(p + '.norm_mha'), ln(p + '.norm_ff'), Lin(packed(wqkv), f32(bqkv)),
                             lin(a + 'linear_pos', bias=False), lin(a + 'linear_out'), lin(p + '.feed_forward.w_1'),
                             lin(p + '.feed_forward.w_2'))

        def tblock(p):
            wqkv = torch.cat([sd[p + '.attn1.to_q.weight'], sd[p + '.attn1.to_k.weight'], sd[p + '.attn1.to_v.weight']], 0)
            return TBlock(ln(p + '.norm1'), ln(p + '.norm3'), Lin(packed(wqkv), None), lin(p + '.attn1.to_out.0'),
                          lin(p + '.ff.net.0.proj'), lin(p + '.ff.net.2'))

        def resnet(p):
            return Resnet(lin(p + '.block1.block.0'), ln(p + '.block1.block.2'), lin(p + '.mlp.1'),
                          lin(p + '.block2.block.0'), ln(p + '.block2.block.2'), lin(p + '.res_conv'))

        def unet(p, tail):
            b = UnetBlock()
            b.rn = resnet(p + '.0')
            for j in range(4):
                b.tb[j] = tblock(f'{p}.1.{j}')
            b.tail = lin(p + '.2') if tail else Lin(None, None)
            return b

        w = FlowWeights()
        w.input_embedding = f32(sd['input_embedding.weight'])
        w.spk_w, w.spk_b = f32(sd['spk_embed_affine_layer.weight']), f32(sd['spk_embed_affine_layer.bias'])
        w.embed, w.embed_ln = lin('encoder.embed.out.0'), ln('encoder.embed.out.1')
        w.pre1, w.pre2 = lin('encoder.pre_lookahead_layer.conv1'), lin('encoder.pre_lookahead_layer.conv2')
        for i in range(6):
            w.enc[i] = conformer(f'encoder.encoders.{i}')
        w.up_conv = lin('encoder.up_layer.conv')
        w.up_embed, w.up_embed_ln = lin('encoder.up_embed.out.0'), ln('encoder.up_embed.out.1')
        for i in range(4):
            w.up[i] = conformer(f'encoder.up_encoders.{i}')
        w.after_norm = ln('encoder.after_norm')
        w.enc_proj = lin('encoder_proj', pad_to=128)
        P = 'decoder.estimator'
        w.time1, w.time2 = lin(P + '.time_mlp.linear_1'), lin(P + '.time_mlp.linear_2')
        w.down = unet(P + '.down_blocks.0', True)
        for i in range(12):
            w.mid[i] = unet(f'{P}.mid_blocks.{i}', False)
        w.up_blk = unet(P + '.up_blocks.0', True)
        w.final_conv, w.final_ln = lin(P + '.final_block.block.0'), ln(P + '.final_block.block.2')
        w.final_proj = lin(P + '.final_proj', pad_to=128)
        w.rand_noise = f32(rand_noise()[0].t())
        self._w = w

        per = (max_len + 8 + 127) // 128 * 128
        self.dims = FlowDims(max_rows=2 * max_utts * per, max_seqs=2 * max_utts, max_len=max_len, n_timesteps=n_timesteps,
                             cfg_rate=cfg_rate)
        self.max_utts, self.max_len = max_utts, max_len
        nbytes = self.lib.cv2_flow_workspace_bytes(C.byref(self.dims))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        h = C.c_void_p()
        L.check(self.lib.cv2_flow_create(C.byref(self.dims), C.byref(w), self.workspace.data_ptr(), nbytes, L.stream_ptr(),
                                         C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.cv2_flow_destroy(self.handle)
        except Exception:
            pass

    # ---- CausalMaskedDiffWithXvec.inference (flow.py:235-283), batched ---------------------------------------
    def inference_batch(self, utts, streaming=False, finalize=True):
        """utts: list of dicts token [1,n] int, prompt_token [1,P] int, prompt_feat [1,F,80], embedding [1,192].
        Returns a list of mel tensors [1, 80, T2] (device, fp32)."""
        assert 1 <= len(utts) <= self.max_utts
        dev = self.device
        arr = (FlowUtt * len(utts))()
        outs, keep = [], []
        la = 0 if finalize else self.pre_lookahead_len
        for i, u in enumerate(utts):
            tok = torch.cat([u['prompt_token'].reshape(-1).to(dev), u['token'].reshape(-1).to(dev)]).to(dtype=torch.int32).contiguous()
            pf = u['prompt_feat'].reshape(-1, 80).to(device=dev, dtype=torch.float32).contiguous()
            emb = u['embedding'].reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
            n2 = 2 * (tok.numel() - la) - pf.shape[0]
            assert n2 >= 0
            out = torch.empty(1, 80, n2, dtype=torch.float32, device=dev)
            keep += [tok, pf, emb]
            outs.append(out)
            arr[i] = FlowUtt(tok.data_ptr(), tok.numel(), pf.data_ptr(), pf.shape[0], emb.data_ptr(), out.data_ptr())
        L.check(self.lib.cv2_flow_inference(self.handle, arr, len(utts), int(streaming), int(finalize), L.stream_ptr()))
        self._last_keep = keep
        return outs

    # ---- the same call for streams that keep a cache: only the frames after the cached ones are computed and returned --------
    def new_cache(self, max_frames):
        return FlowCache(self, max_frames)

    def clone_cache(self, src, max_frames):
        """A new cache of capacity >= max_frames that starts with everything `src` holds (a prompt another call has already run)."""
        dst = FlowCache(self, max(max_frames, src.n_cached))
        L.check(self.lib.cv2_flow_cache_copy(self.handle, src.buf.data_ptr(), src.frames, dst.buf.data_ptr(), dst.frames, src.n_cached,
                                             L.stream_ptr()))
        dst.n_cached, dst.gen = src.n_cached, src.gen
        return dst

    def prompt_cache(self, prompt_token, prompt_feat, embedding, hop=25):
        """The cache of a prompt alone: the whole chunks of the prompt whose look-ahead lies inside the prompt (flow.py:260-263 needs
        pre_lookahead_len tokens after the last one).  None when the prompt is shorter than one chunk plus the look-ahead."""
        P = int(prompt_token.numel())
        n_full = (P - self.pre_lookahead_len) // hop * hop
        if n_full <= 0:
            return None
        c = FlowCache(self, self.token_mel_ratio * n_full)
        u = dict(token=prompt_token.reshape(1, -1)[:, :0], prompt_token=prompt_token.reshape(1, -1)[:, :n_full + self.pre_lookahead_len],
                 prompt_feat=prompt_feat.reshape(1, -1, 80)[:, :self.token_mel_ratio * n_full], embedding=embedding)
        self.inference_chunk_batch([u], [c], finalize=False)
        return c

    def inference_chunk_batch(self, utts, caches, finalize):
        """utts as for inference_batch (token = the whole prefix so far), caches[i] = the stream's FlowCache.  Returns
        [(mel [1, 80, n] fp32, first) ...]: the frames from index `first` of the mel the recompute would return (flow.py:281).
        The caches advance only when the whole call succeeded."""
        assert 1 <= len(utts) <= self.max_utts and len(caches) == len(utts)
        dev = self.device
        arr, refs = (FlowUtt * len(utts))(), (FlowCacheRef * len(utts))()
        outs, keep, ends = [], [], []
        la = 0 if finalize else self.pre_lookahead_len
        for i, (u, c) in enumerate(zip(utts, caches)):
            tok = torch.cat([u['prompt_token'].reshape(-1).to(dev), u['token'].reshape(-1).to(dev)]).to(dtype=torch.int32).contiguous()
            pf = u['prompt_feat'].reshape(-1, 80).to(device=dev, dtype=torch.float32).contiguous()
            emb = u['embedding'].reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
            t2 = 2 * (tok.numel() - la)
            first = max(c.n_cached, pf.shape[0])
            n = t2 - first
            if n < 0 or t2 <= c.n_cached:
                raise ValueError(f'flow cache: the call ends at frame {t2} but {c.n_cached} frames are cached')
            out = torch.empty(1, 80, n, dtype=torch.float32, device=dev)
            keep += [tok, pf, emb]
            outs.append((out, first - pf.shape[0]))
            ends.append(t2)
            arr[i] = FlowUtt(tok.data_ptr(), tok.numel(), pf.data_ptr(), pf.shape[0], emb.data_ptr(), out.data_ptr())
            refs[i] = FlowCacheRef(c.buf.data_ptr(), c.frames, c.n_cached, c.gen)
        L.check(self.lib.cv2_flow_inference_chunk(self.handle, arr, refs, len(utts), int(finalize), L.stream_ptr()))
        for c, t2 in zip(caches, ends):
            c.n_cached, c.gen = t2, c.gen + 1
        self._last_keep = keep
        return outs

    def inference(self, token, token_len, prompt_token, prompt_token_len, prompt_feat, prompt_feat_len, embedding, streaming,
                  finalize):
        assert token.shape[0] == 1                                            # flow.py:246
        mel = self.inference_batch([dict(token=token, prompt_token=prompt_token, prompt_feat=prompt_feat, embedding=embedding)],
                                   streaming, finalize)[0]
        return mel, None

    # ---- inner seams, exposed for parity tests -----------------------------------------------------------------
    def forward_estimator(self, x, mask, mu, t, spks, cond, streaming=False):
        """flow_matching.py:125-150 (TensorRT seam): result written in place into x (2,80,T), also returned."""
        for tns in (x, mask, mu, t, spks, cond):
            assert tns.is_cuda and tns.dtype == torch.float32 and tns.is_contiguous()
        T = x.shape[2]
        L.check(self.lib.cv2_flow_estimator(self.handle, L.ptr(x), L.ptr(mask), L.ptr(mu), L.ptr(t), L.ptr(spks), L.ptr(cond),
                                            T, int(streaming), L.stream_ptr()))
        return x

    def encoder(self, xs, context=None, streaming=False):
        """UpsampleConformerEncoder.forward on already-embedded tokens xs [1,T,512] -> [1,2T,512]."""
        xs = xs.to(self.device, torch.float32).contiguous()
        ctx = context.to(self.device, torch.float32).contiguous() if context is not None else None
        T = xs.shape[1]
        out = torch.empty(1, 2 * T, 512, dtype=torch.float32, device=self.device)
        L.check(self.lib.cv2_flow_encoder(self.handle, L.ptr(xs), T, L.ptr(ctx), int(streaming), L.ptr(out), L.stream_ptr()))
        return out
