"""Host side of stage 3 (HiFT vocoder): folds weight-norm, packs a `hift.pt` state dict into the fp32 MFMA layout of
include/cv2_amd.h and drives cv2_hift_inference (csrc/hift.hip).

Mirrors `HiFTGenerator.inference(speech_feat, cache_source)` (cosyvoice/hifigan/generator.py:570-582): returns
`(wav [1, 480 T], source [1, 1, 480 T])`.  There is no CPU fallback.
"""
import ctypes as C
import os
import threading

import torch

from . import lib as L


class Conv(C.Structure):
    _fields_ = [('w', C.c_void_p), ('b', C.c_void_p)] + [(n, C.c_int32) for n in
                                                          ('cin', 'cout', 'cin_pad', 'cout_pad', 'taps', 'dil', 'pad_left')] + [('w3', C.c_void_p)]


class ResBlock(C.Structure):
    _fields_ = [('c1', Conv * 3), ('c2', Conv * 3), ('a1', C.c_void_p * 3), ('a2', C.c_void_p * 3)]


class HiftWeights(C.Structure):
    _fields_ = [('f0_conv', Conv * 5), ('f0_w', C.c_void_p), ('f0_b', C.c_void_p), ('src_w', C.c_void_p), ('src_b', C.c_void_p),
                ('conv_pre', Conv), ('ups', Conv * 3), ('sd_w', C.c_void_p * 3), ('sd_b', C.c_void_p * 3), ('sd_conv', Conv * 3),
                ('src_rb', ResBlock * 3), ('rb', ResBlock * 9), ('conv_post', Conv)]


class HiftDims(C.Structure):
    _fields_ = [('max_frames', C.c_int32), ('lanes', C.c_int32)]


def _bind(lib):
    if getattr(lib, '_hift_bound', False):
        return
    lib.cv2_hift_workspace_bytes.restype = C.c_size_t
    lib.cv2_hift_workspace_bytes.argtypes = [C.POINTER(HiftDims)]
    lib.cv2_hift_create.argtypes = [C.POINTER(HiftDims), C.POINTER(HiftWeights), C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.cv2_hift_destroy.argtypes = [C.c_void_p]
    lib.cv2_hift_inference.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    lib.cv2_fade_in_out.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.cv2_hift_inference_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
    lib.cv2_interp_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib._hift_bound = True


def weight_norm(sd, name):
    """w = g * v / ||v|| over all dims but 0 (torch.nn.utils.parametrizations.weight_norm); plain weights pass through."""
    if name + '.weight' in sd:
        return sd[name + '.weight'].float()
    g = sd[name + '.parametrizations.weight.original0'].float()
    v = sd[name + '.parametrizations.weight.original1'].float()
    return torch._weight_norm(v, g, 0)


def _pad64(n):
    return (n + 63) // 64 * 64


def pack_conv_f32(w):
    """[C_out][C_in][k] fp32 -> [k][cin_pad/2][cout_pad/32][64]: lane = (c_in & 1) * 32 + (c_out & 31)."""
    co, ci, k = w.shape
    cip, cop = _pad64(ci), _pad64(co)
    wp = w.new_zeros(k, cip, cop)
    wp[:, :ci, :co] = w.permute(2, 1, 0)
    return wp.view(k, cip // 2, 2, cop // 32, 32).permute(0, 1, 3, 2, 4).contiguous().reshape(-1), cip, cop


def pack_conv_split3(w):
    """[C_out][C_in][k] fp32 -> three bf16 planes (w = w0 + w1 + w2, each the bf16 rounding of what is left) in the MFMA 32x32x16
    B-operand order [k][cin_pad/16][cout_pad/32][3][64 lanes][8]: lane = ((c_in % 16) // 8) * 32 + (c_out & 31), element = c_in % 8.
    Returned as an int16 tensor of bf16 bit patterns."""
    co, ci, k = w.shape
    cip, cop = _pad64(ci), _pad64(co)
    wp = w.new_zeros(k, cip, cop)
    wp[:, :ci, :co] = w.permute(2, 1, 0)
    wp = wp.view(k, cip // 16, 2, 8, cop // 32, 32).permute(0, 1, 4, 2, 5, 3).contiguous()          # [k][kb][nt][khalf][n][j]
    planes, rest = [], wp
    for _ in range(3):
        b = rest.to(torch.bfloat16)
        planes.append(b)
        rest = rest - b.float()
    out = torch.stack(planes, 3)                                                                       # [k][kb][nt][3][khalf][n][j]
    return out.contiguous().view(torch.int16).reshape(-1)


def polyphase(wt, u, pad):
    """ConvTranspose1d weight [C_in][C_out][k] -> equivalent Conv1d weight [u*C_out][C_in][3] over input taps q-1, q, q+1."""
    ci, co, k = wt.shape
    w = wt.new_zeros(u * co, ci, 3)
    for r in range(u):
        for j in range(3):
            kk = r + pad + u * (1 - j)
            if 0 <= kk < k:
                w[r * co:(r + 1) * co, :, j] = wt[:, :, kk].t()
    return w


class HiftEngine:
    def __init__(self, sd, device='cuda:0', max_frames=2048, share_weights_with=None, split_products=True, lanes=1):
        """share_weights_with: another HiftEngine whose packed weights are reused (several engines = several workspaces, so
        independent utterances can run on different HIP streams at once).  split_products=False: every convolution on the fp32
        matrix cores (k_conv) instead of the three-plane bf16 products (k_conv6); the reference path of the A/B test."""
        sd = {k[len('generator.'):] if k.startswith('generator.') else k: v for k, v in sd.items()}     # cli/model.py:88
        self.device = dev = torch.device(device)
        self.lib = L.lib()
        _bind(self.lib)
        self._keep = keep = []

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        def conv(w, bias, dil, pad_left, split=True):
            """split: also pack the three-plane bf16 form (k_conv6); False keeps the convolution on the fp32 matrix cores."""
            co, ci, k = w.shape
            wp, cip, cop = pack_conv_f32(w.float())
            b = torch.zeros(cop)
            b[:co] = bias.float()
            w3 = None
            if split and split_products:
                t3 = pack_conv_split3(w.float()).to(dev)
                keep.append(t3)
                w3 = t3.data_ptr()
            return Conv(f32(wp), f32(b), ci, co, cip, cop, k, dil, pad_left, w3)

        def resblock(p, k, dils=(1, 3, 5)):
            rb = ResBlock()
            for i, d in enumerate(dils):
                rb.c1[i] = conv(weight_norm(sd, f'{p}.convs1.{i}'), sd[f'{p}.convs1.{i}.bias'], d, (k * d - d) // 2)
                rb.c2[i] = conv(weight_norm(sd, f'{p}.convs2.{i}'), sd[f'{p}.convs2.{i}.bias'], 1, (k - 1) // 2)
                rb.a1[i] = f32(sd[f'{p}.activations1.{i}.alpha'])
                rb.a2[i] = f32(sd[f'{p}.activations2.{i}.alpha'])
            return rb

        if share_weights_with is not None:
            w = share_weights_with._w
            self._keep = share_weights_with._keep
        else:
            w = self._pack(sd, conv, resblock, f32)
        self._w = w
        self.dims = HiftDims(max_frames=max_frames, lanes=lanes)
        self.lanes = lanes
        self.max_frames = max_frames
        nbytes = self.lib.cv2_hift_workspace_bytes(C.byref(self.dims))
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        h = C.c_void_p()
        L.check(self.lib.cv2_hift_create(C.byref(self.dims), C.byref(w), self.workspace.data_ptr(), nbytes, C.byref(h)))
        self.handle = h
        self.seed = 0
        self._seed_fn = None

    @staticmethod
    def _pack(sd, conv, resblock, f32):
        w = HiftWeights()
        for n, i in enumerate((0, 2, 4, 6, 8)):
            # the f0 predictor stays on the fp32 matrix cores: its output is integrated into a phase of ~1e5 rad (k_phase)
            w.f0_conv[n] = conv(weight_norm(sd, f'f0_predictor.condnet.{i}'), sd[f'f0_predictor.condnet.{i}.bias'], 1, 1, split=False)
        w.f0_w, w.f0_b = f32(sd['f0_predictor.classifier.weight'].reshape(-1)), f32(sd['f0_predictor.classifier.bias'])
        w.src_w, w.src_b = f32(sd['m_source.l_linear.weight'].reshape(-1)), f32(sd['m_source.l_linear.bias'])
        w.conv_pre = conv(weight_norm(sd, 'conv_pre'), sd['conv_pre.bias'], 1, 3)
        for i, (u, k) in enumerate(((8, 16), (5, 11), (3, 7))):
            wt = weight_norm(sd, f'ups.{i}')
            b = sd[f'ups.{i}.bias'].float()
            w.ups[i] = conv(polyphase(wt, u, (k - u) // 2), b.repeat(u), 1, 1)
            w.sd_w[i] = f32(sd[f'source_downs.{i}.weight'].float().permute(2, 1, 0))      # [C][18][k] -> [k][18][C]
            w.sd_b[i] = f32(sd[f'source_downs.{i}.bias'])
            # the same convolution over flat windows of the [F][18] rows: [C][18][k] -> [C][18 k][1] with input index j * 18 + c
            wsd = sd[f'source_downs.{i}.weight'].float()
            w.sd_conv[i] = conv(wsd.permute(0, 2, 1).reshape(wsd.shape[0], -1, 1), sd[f'source_downs.{i}.bias'], 1, 0)
            w.src_rb[i] = resblock(f'source_resblocks.{i}', (7, 7, 11)[i])
            for j, k2 in enumerate((3, 7, 11)):
                w.rb[i * 3 + j] = resblock(f'resblocks.{i * 3 + j}', k2)
        w.conv_post = conv(weight_norm(sd, 'conv_post'), sd['conv_post.bias'], 1, 3)
        return w

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.cv2_hift_destroy(self.handle)
        except Exception:
            pass

    def inference(self, speech_feat, cache_source=None, noise=None, seed=None):
        """speech_feat [1,80,T] fp32 device; cache_source [1,1,k]; noise [1,480T,9] injects the N(0,1) draws
        (parity tests), otherwise they come from the device Philox stream (seed advances per call)."""
        assert speech_feat.shape[0] == 1 and speech_feat.shape[1] == 80
        dev = self.device
        mel = speech_feat.to(dev, torch.float32).contiguous()
        T = mel.shape[2]
        wav = torch.empty(1, 480 * T, dtype=torch.float32, device=dev)
        src = torch.empty(1, 1, 480 * T, dtype=torch.float32, device=dev)
        cs = None
        if cache_source is not None and cache_source.numel() > 0:
            cs = cache_source.to(dev, torch.float32).contiguous()
        nz = noise.to(dev, torch.float32).contiguous() if noise is not None else None
        if seed is None:
            if self._seed_fn is not None:                 # engines of a pool draw from ONE counter (no two chunks share a noise stream)
                seed = self._seed_fn()
            else:
                self.seed += 1
                seed = self.seed
        L.check(self.lib.cv2_hift_inference(self.handle, L.ptr(mel), T, L.ptr(cs), cs.numel() if cs is not None else 0,
                                            L.ptr(nz), C.c_uint64(seed), L.ptr(wav), L.ptr(src), L.stream_ptr()))
        return wav, src

    def debug_f0(self, T):
        """Test hook: the f0 track [T] (Hz) of the last inference() call of this engine (the graph path of short calls keeps its own)."""
        out = torch.empty(T, dtype=torch.float32, device=self.device)
        L.check(self.lib.cv2_hift_debug_f0(self.handle, L.ptr(out), T, L.stream_ptr()))
        return out

    def change_speed(self, mel, speed):
        """cli/model.py:328-330: F.interpolate(tts_mel, size=int(T / speed), mode='linear') on the device (cv2_interp_linear);
        mel [1, 80, T] fp32."""
        mel = mel.contiguous()
        n_in = mel.shape[2]
        n_out = int(n_in / speed)
        out = torch.empty(mel.shape[0], mel.shape[1], n_out, dtype=torch.float32, device=mel.device)
        L.check(self.lib.cv2_interp_linear(L.ptr(mel), L.ptr(out), mel.shape[0] * mel.shape[1], n_in, n_out, L.stream_ptr()))
        return out

    def inference_batch(self, mels, cache_sources, seeds=None):
        """n <= lanes chunks of one shape as ONE set of launches (cv2_hift_inference_batch): mels = list of [1,80,T] fp32 device tensors,
        cache_sources = list of [1,1,k] tensors (all the same k, or all None / empty).  Returns a list of (wav [1,480T], source [1,1,480T]);
        equal to n inference() calls with the same seeds."""
        n, dev = len(mels), self.device
        assert 1 <= n <= self.lanes
        ms = [m.to(dev, torch.float32).contiguous() for m in mels]
        T = ms[0].shape[2]
        assert all(m.shape == (1, 80, T) for m in ms)
        cs = [None if (c is None or c.numel() == 0) else c.to(dev, torch.float32).contiguous() for c in cache_sources]
        nc = 0 if cs[0] is None else cs[0].numel()
        assert all((0 if c is None else c.numel()) == nc for c in cs)
        if seeds is None:
            seeds = [self._seed_fn() if self._seed_fn is not None else self._next_own_seed() for _ in range(n)]
        wav = [torch.empty(1, 480 * T, dtype=torch.float32, device=dev) for _ in range(n)]
        src = [torch.empty(1, 1, 480 * T, dtype=torch.float32, device=dev) for _ in range(n)]
        P = C.c_void_p * n
        L.check(self.lib.cv2_hift_inference_batch(self.handle, n, P(*[m.data_ptr() for m in ms]), T, P(*[0 if c is None else c.data_ptr() for c in cs]), nc,
                                                  (C.c_uint64 * n)(*seeds), P(*[w.data_ptr() for w in wav]), P(*[s_.data_ptr() for s_ in src]),
                                                  L.stream_ptr()))
        return list(zip(wav, src))

    def _next_own_seed(self):
        self.seed += 1
        return self.seed

    def fade_in_out(self, fade_in, fade_out_tail, window):
        """utils/common.py:142-150, on the device and in place: fade_in [1,n], fade_out_tail [1,w], window [2w]."""
        w = window.numel() // 2
        L.check(self.lib.cv2_fade_in_out(L.ptr(fade_in), L.ptr(fade_out_tail), L.ptr(window), w, L.stream_ptr()))
        return fade_in


class HiftPool:
    """Several HiftEngines (shared weights, own workspaces) on their own HIP streams: independent utterances of a batch run
    concurrently, which fills the chip where a single utterance's early conv stages launch fewer blocks than there are CUs."""

    def __init__(self, sd, device='cuda:0', max_frames=2048, n=4, batch_lanes=0):
        self.engines = [HiftEngine(sd, device, max_frames)]
        for _ in range(1, n):
            self.engines.append(HiftEngine(sd, device, max_frames, share_weights_with=self.engines[0]))
        self.streams = [torch.cuda.Stream(device) for _ in self.engines]
        # the chunks of a streaming round (one per stream, one shape) run as ONE set of launches on an engine with a workspace lane per chunk
        self.batch_engine = HiftEngine(sd, device, min(max_frames, 160), share_weights_with=self.engines[0], lanes=batch_lanes) if batch_lanes > 1 else None
        self._seed = 0
        self._seed_lock = threading.Lock()
        # one Philox seed counter for the whole pool, salted with the rank: concurrent streams served by different engines, other
        # model instances' pools on other ranks, never replay each other's noise (the reference draws from the device RNG)
        self._salt = (int(os.environ.get('RANK', '0')) + 1) << 40
        for e in self.engines + ([self.batch_engine] if self.batch_engine is not None else []):
            e._seed_fn = self.next_seed

    def next_seed(self):
        with self._seed_lock:
            self._seed += 1
            return self._salt ^ self._seed

    def inference_many(self, mels):
        """mels: list of [1,80,T] device tensors.  Returns list of (wav, source)."""
        main = torch.cuda.current_stream()
        outs = [None] * len(mels)
        for st in self.streams:
            st.wait_stream(main)
        for i, mel in enumerate(mels):
            k = i % len(self.engines)
            with torch.cuda.stream(self.streams[k]):
                outs[i] = self.engines[k].inference(mel, None, seed=self.next_seed())
            for t in outs[i]:
                t.record_stream(main)
        for st in self.streams:
            main.wait_stream(st)
        return outs
