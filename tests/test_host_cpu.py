"""CPU-side tests (run without a GPU): the C-ABI library loads and exports every symbol include/cv2_amd.h declares, the host
packing logic, the Philox reference, the scheduler arithmetic, the API surface and the multi-process sharding over gloo."""
import inspect
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'cosyvoice2-eu_amd')


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'cv2_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cv2_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    """No compute calls here (no GPU): dlopen + symbol lookup only."""
    import ctypes
    import __graft_entry__
    __graft_entry__.build()
    from cv2amd import lib as L
    cdll = ctypes.CDLL(L.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(cdll, name), f'{name} is declared in include/cv2_amd.h but not exported by libcv2amd.so'
    assert set(L.EXPORTS) <= set(declared)
    cdll.cv2_last_error.restype = ctypes.c_char_p
    import re
    hdr = open(os.path.join(ROOT, 'include', 'cv2_amd.h')).read()
    abi = int(re.search(r'#define\s+CV2_ABI_VERSION\s+(\d+)', hdr).group(1))
    assert cdll.cv2_version() == abi == L.ABI_VERSION          # header, library and ctypes mirrors agree


def test_stale_library_is_refused(monkeypatch):
    """A library of another ABI revision (struct layouts differ from the ctypes mirrors) must not be used."""
    from cv2amd import lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'ABI_VERSION', L.ABI_VERSION + 1)
    with pytest.raises(L.Cv2Error, match='ABI revision'):
        L.lib()


def test_product_path_refuses_to_run_without_the_library(monkeypatch):
    from cv2amd import lib as L
    monkeypatch.setattr(L, '_lib', None)
    monkeypatch.setattr(L, 'LIB_PATH', '/nonexistent/libcv2amd.so')
    with pytest.raises(L.Cv2Error):
        L.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'cosyvoice2-eu_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f'{f} imports oracle/'


def test_philox_known_answer():
    """Random123 Philox4x32-10 known-answer vectors (kat_vectors): the sampler's uniforms are built on these."""
    from cv2amd import philox
    assert philox.philox4x32((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert philox.philox4x32((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert philox.philox4x32((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    u = philox.uniforms(1, 2, 3, 0x1234ABCD5)
    assert all(0.0 <= x < 1.0 for x in u)


def test_pack_bf16_layout():
    from cv2amd import weights as W
    w = torch.arange(32 * 64, dtype=torch.float32).reshape(32, 64) / 64.0
    p = W.pack_bf16(w).view(torch.bfloat16).reshape(2, 2, 64, 8)            # [nt][ks][lane][8]
    for nt in range(2):
        for ks in range(2):
            for lane in (0, 5, 17, 63):
                row, col = 16 * nt + (lane & 15), 32 * ks + 8 * (lane >> 4)
                assert torch.equal(p[nt, ks, lane].float(), w[row, col:col + 8].to(torch.bfloat16).float())


def test_conv_packing_and_polyphase_transpose():
    from cv2amd import hift as H
    g = torch.Generator().manual_seed(0)
    w = torch.randn(70, 18, 7, generator=g)
    wp, cip, cop = H.pack_conv_f32(w)
    assert (cip, cop) == (64, 128)
    wp = wp.view(7, cip // 2, cop // 32, 64)
    for tap, ci, co in ((0, 0, 0), (3, 17, 69), (6, 5, 33)):
        assert wp[tap, ci // 2, co // 32, (ci & 1) * 32 + (co & 31)] == w[co, ci, tap]
    assert wp[0, 9, 0, 0] == 0                                                # padded input channel
    # ConvTranspose1d == polyphase Conv1d over taps q-1, q, q+1 with [frame][phase][channel] output
    for (u, k), (ci, co) in zip(((8, 16), (5, 11), (3, 7)), ((6, 4), (5, 3), (4, 2))):
        wt = torch.randn(ci, co, k, generator=g)
        x = torch.randn(1, ci, 9, generator=g)
        ref = torch.nn.functional.conv_transpose1d(x, wt, stride=u, padding=(k - u) // 2)
        wpoly = H.polyphase(wt, u, (k - u) // 2)
        y = torch.nn.functional.conv1d(x, wpoly, padding=1)                   # [1, u*co, 9]
        y = y.view(1, u, co, 9).permute(0, 2, 3, 1).reshape(1, co, 9 * u)
        assert torch.allclose(y, ref, atol=1e-5)


def test_shard_assignment_is_balanced_and_complete():
    from cv2amd import shard
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(5, 120, (37,), generator=g).tolist()
    shards = shard.assign(lens, 8)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(37))
    assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    load = [sum(lens[i] for i in s) for s in shards]
    assert max(load) - min(load) <= max(lens)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))
    from cv2amd import shard
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    texts = [torch.randint(0, 1000, (int(n),), generator=g, dtype=torch.int32) for n in (5, 17, 9, 30, 2, 11, 23)] if rank == 0 else None
    prompt = dict(tok=torch.arange(20, dtype=torch.int32), feat=torch.ones(40, 80), emb=torch.full((192,), 0.5)) if rank == 0 else None

    def fake_synth(my_texts, p):            # "waveform" encodes the text so the gather can be verified
        assert p['feat'].shape == (40, 80) and float(p['emb'][0]) == 0.5 and int(p['tok'][19]) == 19
        return [torch.cat([t.float(), torch.full((3,), float(t.numel()))]) for t in my_texts]
    out = shard.synthesize_sharded(texts, prompt, fake_synth)
    if rank == 0:
        ok = all(torch.equal(o, torch.cat([t.float(), torch.full((3,), float(t.numel()))])) for o, t in zip(out, texts))
        q.put(ok and len(out) == len(texts))
    dist.destroy_process_group()


def test_sharded_synthesis_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_sharded_synthesis_gloo_world4_uneven_shards():
    """Four ranks, seven utterances: the shards hold 2 / 2 / 2 / 1 texts of very different lengths; padded scatter rows and the
    padded gather must still return every waveform in input order on rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_api_surface_matches_reference_signatures():
    """Names, order and defaults of the public entry points (standalone_infer/src/cosyvoice2_eu/__init__.py:43-128,
    cosy_repo/cosyvoice/cli/cosyvoice.py:92-115,144-151, cli/model.py:300,336)."""
    from cosyvoice.cli.cosyvoice import CosyVoice2
    from cosyvoice.cli.model import CosyVoice2Model
    import cosyvoice2_eu

    def params(f):
        return [(n, p.default) for n, p in inspect.signature(f).parameters.items() if n != 'self']
    assert params(CosyVoice2.__init__)[:12] == [('model_dir', inspect._empty), ('load_jit', False), ('load_trt', False), ('load_vllm', False),
                                                ('fp16', False), ('trt_concurrent', 1), ('setting', 'original'), ('llm_run_id', None),
                                                ('flow_run_id', None), ('hifigan_run_id', None), ('final', False), ('backbone', None)]
    assert params(CosyVoice2.inference_zero_shot) == [('tts_text', inspect._empty), ('prompt_text', inspect._empty), ('prompt_speech_16k', inspect._empty),
                                                     ('zero_shot_spk_id', ''), ('stream', False), ('speed', 1.0), ('text_frontend', True)]
    assert params(CosyVoice2.inference_sft) == [('tts_text', inspect._empty), ('spk_id', inspect._empty), ('stream', False), ('speed', 1.0),
                                                ('text_frontend', True)]                                  # cli/cosyvoice.py:81
    assert params(CosyVoice2.inference_cross_lingual) == [('tts_text', inspect._empty), ('prompt_speech_16k', inspect._empty), ('zero_shot_spk_id', ''),
                                                         ('stream', False), ('speed', 1.0), ('text_frontend', True)]
    assert params(CosyVoice2.inference_instruct2) == [('tts_text', inspect._empty), ('instruct_text', inspect._empty), ('prompt_speech_16k', inspect._empty),
                                                     ('zero_shot_spk_id', ''), ('stream', False), ('speed', 1.0), ('text_frontend', True)]
    assert [n for n, _ in params(CosyVoice2Model.token2wav)] == ['token', 'prompt_token', 'prompt_feat', 'embedding', 'token_offset', 'uuid', 'stream',
                                                                 'finalize', 'speed']
    assert [n for n, _ in params(CosyVoice2Model.tts)][:10] == ['text', 'flow_embedding', 'llm_embedding', 'prompt_text', 'llm_prompt_speech_token',
                                                                'flow_prompt_speech_token', 'prompt_speech_feat', 'source_speech_token', 'stream', 'speed']
    assert [n for n, _ in params(cosyvoice2_eu.load)] == ['model_dir', 'repo_id', 'download', 'setting', 'llm_run_id', 'flow_run_id', 'hifigan_run_id',
                                                          'final', 'backbone']
    assert dict(params(cosyvoice2_eu.load))['setting'] == 'llm_flow_hifigan'
    assert [n for n, _ in params(cosyvoice2_eu.Cosy2EU.tts)] == ['text', 'prompt', 'speed', 'text_frontend']


def test_frontend_modes_build_the_reference_model_input():
    """frontend_zero_shot / cross_lingual / instruct2 (cli/frontend.py:491-537) for a registered speaker: which keys reach tts()."""
    from cosyvoice.cli.frontend import PrecomputedFrontEnd
    spk = {'prompt_text': torch.tensor([[1, 2, 3]], dtype=torch.int32), 'prompt_text_len': torch.tensor([3]),
           'llm_prompt_speech_token': torch.zeros(1, 5, dtype=torch.int32), 'llm_prompt_speech_token_len': torch.tensor([5]),
           'flow_prompt_speech_token': torch.zeros(1, 5, dtype=torch.int32), 'flow_prompt_speech_token_len': torch.tensor([5]),
           'prompt_speech_feat': torch.zeros(1, 10, 80), 'prompt_speech_feat_len': torch.tensor([10]),
           'llm_embedding': torch.zeros(1, 192), 'flow_embedding': torch.zeros(1, 192)}
    fe = PrecomputedFrontEnd(lambda t: [ord(c) for c in t], {'a': spk})
    z = fe.frontend_zero_shot('hi', 'x', None, 24000, 'a')
    assert set(z) == set(spk) | {'text', 'text_len'} and z['text'].tolist() == [[104, 105]] and z['text'].dtype == torch.int32
    # frontend_sft (frontend.py:485-489): text + the stored embedding; a CosyVoice2 model dir has no spk2info -> the reference's KeyError
    fe.spk2info['sft'] = {'embedding': torch.ones(1, 192)}
    sft = fe.frontend_sft('hi', 'sft')
    assert set(sft) == {'text', 'text_len', 'llm_embedding', 'flow_embedding'} and torch.equal(sft['llm_embedding'], torch.ones(1, 192))
    with pytest.raises(KeyError):
        fe.frontend_sft('hi', 'nobody')
    c = fe.frontend_cross_lingual('hi', None, 24000, 'a')
    assert set(c) == set(z) - {'prompt_text', 'prompt_text_len', 'llm_prompt_speech_token', 'llm_prompt_speech_token_len'}
    i2 = fe.frontend_instruct2('hi', 'speak fast', None, 24000, 'a')
    assert set(i2) == set(z) - {'llm_prompt_speech_token', 'llm_prompt_speech_token_len'}
    assert set(spk) == set(fe.spk2info['a'])                  # the registered speaker record is not mutated by the pops


def test_load_wav_mono_and_resample(tmp_path):
    import wave
    from cosyvoice.utils.file_utils import load_wav
    sr = 48000
    t = np.arange(sr) / sr
    x = np.stack([np.sin(2 * np.pi * 440 * t), np.sin(2 * np.pi * 440 * t)], 1)
    pcm = (x * 20000).astype('<i2')
    p = str(tmp_path / 'a.wav')
    with wave.open(p, 'wb') as w:
        w.setnchannels(2), w.setsampwidth(2), w.setframerate(sr)
        w.writeframes(pcm.tobytes())
    y = load_wav(p, 16000)
    assert y.shape == (1, 16000) and y.dtype == torch.float32
    ref = torch.from_numpy(np.sin(2 * np.pi * 440 * np.arange(16000) / 16000).astype(np.float32)) * (20000 / 32768)
    assert (y[0, 200:-200] - ref[200:-200]).abs().max() < 2e-3


def test_prompt_feature_oracle_conventions():
    """oracle/frontend.py, the CPU restatement of the prompt feature path (matcha/utils/audio.py:45-82 behind cosyvoice2.yaml:152-160,
    torchaudio's Resample(16000, 24000) of cli/frontend.py:497).  The STFT half is pinned by torch.stft itself (the reference's call)
    against the DFT by definition in float64; the Slaney filterbank and the sinc kernel are third-party algorithms (librosa,
    torchaudio: absent here) checked through their structural properties only."""
    from oracle import frontend as OF
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 24000 + 137, generator=g) * 0.1
    a, b = OF.stft_mag(x), OF.stft_mag_direct(x)
    assert a.shape == b.shape == (1, 961, 50)
    assert (a - b).abs().max().item() < 2e-4 * a.abs().max().item()
    m = OF.mel_spectrogram(x)
    assert m.shape == (1, 80, 50) and torch.isfinite(m).all() and m.min() >= np.log(1e-5) - 1e-6
    fb = OF.mel_filterbank()
    assert fb.shape == (80, 961) and (fb >= 0).all()
    area = fb.sum(1) * 12.5                               # bin spacing 24000 / 1920 Hz; Slaney norm: unit-area triangles
    assert np.all(np.abs(area[2:] - 1.0) < 0.08), area
    peaks = fb.argmax(1)
    assert np.all(np.diff(peaks) > 0) and peaks[-1] * 12.5 < 8000.0
    k, width, orig, new = OF.resample_kernel()
    assert (orig, new, width) == (2, 3, 7) and tuple(k.shape) == (3, 16)
    assert np.allclose(k.sum(1).numpy(), 1.0, atol=2e-2)                      # every output phase passes DC
    t16 = torch.arange(16000, dtype=torch.float64) / 16000
    y = OF.resample(torch.sin(2 * np.pi * 440 * t16).float()[None])
    assert y.shape == (1, 24000)
    t24 = torch.arange(24000, dtype=torch.float64) / 24000
    assert (y[0, 50:-50] - torch.sin(2 * np.pi * 440 * t24).float()[50:-50]).abs().max().item() < 2e-3


def test_streaming_chunk_arithmetic():
    """cli/model.py:351-381: first chunk needs hop + pad + look-ahead tokens, pad aligns prompt + chunk to the 25-token grid."""
    hop, la = 25, 3
    for P, first in ((87, 25 + 13 + 3), (255, 25 + 20 + 3), (310, 25 + 15 + 3), (250, 25 + 0 + 3)):
        pad = int(np.ceil(P / hop) * hop - P)
        assert hop + pad + la == first
        assert (P + hop + pad) % hop == 0


def test_bench_gpus2_spawns_ranks_and_gathers(tmp_path):
    """`python bench.py --gpus 2` without a launcher must spawn 2 ranks itself (torch.distributed.run as a child), run the sharded
    configuration through shard.synthesize_sharded and print ONE JSON line with n_gpus == 2.  CPU plumbing run: gloo + a fake
    synthesiser (CV2_BENCH_FAKE_SYNTH) whose waveform lengths follow the forced-length rule, so the gather is checked too."""
    import json
    import subprocess
    env = dict(os.environ, CV2_BENCH_BACKEND='gloo', CV2_BENCH_FAKE_SYNTH='1', CV2_BENCH_PER_GPU='5', MASTER_PORT=str(_free_port()))
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['warmup'] == 1 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['config']['batch_per_gpu'] == 5 and 'configs[3]' in out['config']['workload'] and out['data'].startswith('FAKE')
    # what a driver run needs to judge the line: the world the process group saw, its backend, every rank's own time and share
    assert out['ranks_seen'] == 2 and out['backend'] == 'gloo' and [r['rank'] for r in out['per_rank']] == [0, 1]
    assert all(r['utts_per_step'] == 5 and r['work_s'] >= 0 and r['wall_s'] > 0 for r in out['per_rank']) and out['imbalance'] >= 1.0
    # the N > 1 line carries its own one-GPU reference: what every rank's shard ran at, and the committed N = 1 measurement of the same workload
    ref = out['n1_reference']
    assert ref['per_rank_shard_rate_median'] > 0 and ref['per_rank_shard_rate_min'] > 0 and ref['committed_n1_batch32']['value'] > 0
    # a launcher / flag mismatch fails loudly instead of reporting a 1-GPU number as N = 2's line
    env2 = dict(env, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env2, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_serving_wire_formats_match_protobuf_and_pcm():
    """runtime/python/wire.py against google.protobuf on the schema of runtime/python/grpc/cosyvoice.proto:8-42, the int16 PCM rule of
    the servers (fastapi/server.py:40-43), and one gRPC round trip through the generic handler with a fake model."""
    sys.path.insert(0, os.path.join(PKG, 'runtime', 'python'))
    import wire
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='cosyvoice_test.proto', package='cosyvoice', syntax='proto3')

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for i, (fname, ftype) in enumerate(fields, 1):
            m.field.add(name=fname, number=i, type=ftype, label=1)
        return m
    S, B = 9, 12
    msg('sftRequest', [('spk_id', S), ('tts_text', S)])
    msg('zeroshotRequest', [('tts_text', S), ('prompt_text', S), ('prompt_audio', B)])
    msg('crosslingualRequest', [('tts_text', S), ('prompt_audio', B)])
    msg('instructRequest', [('tts_text', S), ('spk_id', S), ('instruct_text', S)])
    req = fd.message_type.add(name='Request')
    req.oneof_decl.add(name='RequestPayload')
    for i, (n, t) in enumerate((('sft_request', 'sftRequest'), ('zero_shot_request', 'zeroshotRequest'),
                                ('cross_lingual_request', 'crosslingualRequest'), ('instruct_request', 'instructRequest')), 1):
        req.field.add(name=n, number=i, type=11, label=1, type_name='.cosyvoice.' + t, oneof_index=0)
    msg('Response', [('tts_audio', B)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Request = message_factory.GetMessageClass(pool.FindMessageTypeByName('cosyvoice.Request'))
    Response = message_factory.GetMessageClass(pool.FindMessageTypeByName('cosyvoice.Response'))
    audio = np.arange(-300, 300, dtype=np.int16).tobytes()
    r = Request()
    r.zero_shot_request.tts_text = 'Bonjour à tous'
    r.zero_shot_request.prompt_text = 'salut'
    r.zero_shot_request.prompt_audio = audio
    assert wire.decode_request(r.SerializeToString()) == ('zero_shot_request', {'tts_text': 'Bonjour à tous', 'prompt_text': 'salut', 'prompt_audio': audio})
    assert wire.encode_request('zero_shot_request', tts_text='Bonjour à tous', prompt_text='salut', prompt_audio=audio) == r.SerializeToString()
    r = Request()
    r.instruct_request.tts_text, r.instruct_request.spk_id = 'x', 'fr'
    assert wire.decode_request(r.SerializeToString()) == ('instruct_request', {'tts_text': 'x', 'spk_id': 'fr', 'instruct_text': ''})
    assert wire.encode_request('cross_lingual_request', tts_text='hé', prompt_audio=audio) == \
        Request(cross_lingual_request=dict(tts_text='hé', prompt_audio=audio)).SerializeToString()
    assert wire.encode_response(audio) == Response(tts_audio=audio).SerializeToString() and wire.decode_response(Response(tts_audio=audio).SerializeToString()) == audio
    w = torch.tensor([[0.0, 0.5, -0.5, 0.999]])
    assert np.frombuffer(wire.pcm16(w), dtype=np.int16).tolist() == [0, 16384, -16384, 32735]
    assert np.allclose(wire.pcm16_to_float(wire.pcm16(w)), w.numpy(), atol=2 ** -15)

    # one streamed rpc through the generic handler, fake model yielding two chunks
    import grpc
    import importlib.util
    spec = importlib.util.spec_from_file_location('cv2_grpc_server', os.path.join(PKG, 'runtime', 'python', 'grpc', 'server.py'))
    srv_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(srv_mod)

    class Fake:
        def inference_cross_lingual(self, text, prompt):
            assert text == 'hé' and prompt.shape == (1, 600)
            yield {'tts_speech': torch.full((1, 4), 0.25)}
            yield {'tts_speech': torch.full((1, 2), -0.25)}
    server, port = srv_mod.make_server(Fake(), 0)
    server.start()
    try:
        ch = grpc.insecure_channel('127.0.0.1:{}'.format(port))
        call = ch.unary_stream('/cosyvoice.CosyVoice/Inference', request_serializer=lambda kw: wire.encode_request(**kw),
                               response_deserializer=wire.decode_response)
        chunks = list(call(dict(kind='cross_lingual_request', tts_text='hé', prompt_audio=audio), timeout=30))
        assert [np.frombuffer(c, dtype=np.int16).tolist() for c in chunks] == [[8192] * 4, [-8192] * 2]
    finally:
        server.stop(0)


class _SimBistreamEngine:
    """The device half of cv2amd.llm's bistream rounds simulated on the CPU: k_sample's state machine (csrc/llm.hip: fill id stops the
    slot, forced fill every mix[1] + 1 entries, final decode ends on EOS) over the oracle's backbone, greedy harness.  Only the methods
    LLMEngine.bistream() calls on `self` exist; BiStream and the generator loop are the product's own code."""

    def __init__(self, sd, max_out=512):
        from cv2amd.llm import LLMEngine
        from oracle import llm as OL
        self.OL, self.sd, self.d, self.max_out = OL, sd, OL.LLMDims(sd), max_out
        n_text, vocab = sd['llm.model.model.embed_tokens.weight'].shape[0], sd['speech_embedding.weight'].shape[0]
        self.emb_all = torch.cat([sd['llm.model.model.embed_tokens.weight'], sd['speech_embedding.weight'], sd['llm_embedding.weight']], 0)
        self.IDX_SPEECH, self.IDX_SOS, self.IDX_TASK = n_text, n_text + vocab, n_text + vocab + 1
        self.slots = {}
        self.feeds = []                                 # (slot, n rows, final) of every feed, in order
        for name in ('new_bistream', 'bistream', 'bi_burst_len'):
            setattr(self, name, getattr(LLMEngine, name).__get__(self) if name != 'bi_burst_len' else LLMEngine.bi_burst_len)

    def _err(self, code):
        return RuntimeError(f'err {code}')

    def _draw(self, s, x):
        from cv2amd import lib as L
        OL, st = self.OL, s['st']
        y = OL.qwen2_step(self.sd, self.d, x, s['cache'])
        logp = torch.nn.functional.linear(y[-1], self.sd['llm_decoder.weight'], self.sd['llm_decoder.bias']).log_softmax(dim=-1)
        nout, fill = st[L.ST_NOUT], OL.FILL_TOKEN
        if st[L.ST_BIMODE] == 1 and st[L.ST_NEXTFILL] != -1 and nout == st[L.ST_NEXTFILL]:
            top = fill
        else:
            top = OL.greedy_ids(logp, st[L.ST_BIMODE] == 1)
        if top == fill:
            st[L.ST_NEXTFILL] = nout + 16
        s['out'].append(top)
        st[L.ST_NOUT], st[L.ST_LAST] = nout + 1, top
        if top >= OL.SPEECH_TOKEN_SIZE:
            st[L.ST_DONE] = 1
            if st[L.ST_BIMODE] == 1 and top == fill:
                st[L.ST_WAIT] = 1
            elif not (st[L.ST_BIMODE] == 2 and top == OL.SPEECH_TOKEN_SIZE):
                st[L.ST_ERR] = 2

    def bi_feed(self, streams, prepared=False):
        from cv2amd import lib as L
        fed = []
        for b in streams:
            if b.running or b.finished:
                continue
            f = b.next_feed()
            if f is None:
                continue
            idx, final = f
            s = self.slots.setdefault(b.slot, {'cache': [None] * self.d.layers, 'out': [], 'st': None})
            s['st'] = b._state_row(final)
            assert (s['st'][L.ST_POS] if b.started else 0) == b.pos
            self._draw(s, self.emb_all[torch.tensor(idx)])
            s['st'][L.ST_POS] = b.pos + len(idx)
            self.feeds.append((b.slot, len(idx), final))
            b._fed(len(idx), final)
            fed.append(b)
        return fed

    def bi_burst(self, streams, n_steps, shared=False):
        from cv2amd import lib as L
        for b in streams:
            if not b.running:
                continue
            s = self.slots[b.slot]
            for _ in range(n_steps):
                if s['st'][L.ST_DONE]:
                    break
                self._draw(s, self.sd['speech_embedding.weight'][s['st'][L.ST_LAST]][None])
                if not s['st'][L.ST_DONE] or s['st'][L.ST_WAIT]:
                    s['st'][L.ST_POS] += 1
            if b.eta is not None:
                b.eta = max(0, b.eta - n_steps)

    def bi_poll(self, streams):
        for b in streams:
            if b.running:
                s = self.slots[b.slot]
                b._polled(list(s['st']), s['out'][b.n_read:])


def test_bistream_host_bookkeeping_matches_the_oracle():
    """cv2amd.llm.BiStream + LLMEngine.bistream (the host half of inference_bistream, llm.py:721-834: text cache, 5 : 15 interleave with the
    prompt speech tokens, the feed after a fill id, the stale lm_input in front of the final feed) drive a SIMULATED device (the oracle's
    backbone under k_sample's state machine): emitted ids and the out_tokens list equal oracle.llm.inference_bistream, for text arriving in
    uneven pieces, with and without prompt speech tokens, and for every burst length (a burst may overshoot a stop)."""
    from cv2amd import synth
    from oracle import llm as OL
    sd = synth.make_llm(layers=1)
    b = sd['llm_decoder.bias'].clone()
    b[6563] += 6.0; b[6561] += 14.0; b[6562] = -30.0
    sd['llm_decoder.bias'] = b
    for seed, P, cuts, burst in ((1, 31, (0, 3, 10, 15, 23), 16), (2, 0, (0, 5, 6, 17), 16), (3, 45, (0, 12, 13, 14, 30), 3), (4, 30, (0, 2, 4, 9), 1)):
        inp = synth.synthetic_inputs(seed=seed, text_len=cuts[-1], prompt_len=max(P, 1), prompt_text_len=6)
        ptok = inp['prompt_token'][:, :P]
        pieces = [inp['text'][:, a:b2] for a, b2 in zip(cuts[:-1], cuts[1:])]
        want, want_out = OL.inference_bistream(sd, pieces, inp['prompt_text'], ptok)
        eng = _SimBistreamEngine(sd)
        pulled = []

        def text():
            for p in pieces:
                pulled.append(len(eng.feeds))          # feeds issued before this piece was asked for
                yield p
        got = list(eng.bistream(2, text(), inp['prompt_text'], ptok, burst=burst))
        assert got == want and eng.slots[2]['out'] == want_out, (seed, got[:8], want[:8])
        assert eng.feeds[-1][2] and sum(1 for f in eng.feeds if f[2]) == 1             # exactly one final feed, the last one
        assert pulled == sorted(pulled) and pulled[0] == 0                              # pieces are pulled lazily, never ahead of a feed they enable


def test_bench_caller_pool_runs_every_call_once_in_both_modes():
    """bench.CallerPool: n calls on persistent worker threads (the same threads round after round) or on new threads per round; every index
    runs exactly once per round and an exception inside a call does not lose the round."""
    import threading
    import bench
    seen = []
    lock = threading.Lock()

    def fn(i):
        with lock:
            seen.append((i, threading.get_ident()))
        if i == 1:
            raise RuntimeError('a failing call')
    keep = bench.CallerPool.fresh
    try:
        bench.CallerPool.fresh = False
        for _ in range(3):
            with pytest.raises(RuntimeError, match='a failing call'):     # re-raised once the whole round has run; the worker survives
                bench.CallerPool.run(fn, 4)
        assert sorted(i for i, _ in seen) == sorted(list(range(4)) * 3)
        per_index = {i: {t for j, t in seen if j == i} for i in range(4)}
        assert all(len(v) == 1 for v in per_index.values())            # index i always lands on worker i
        seen.clear()
        bench.CallerPool.fresh = True
        bench.CallerPool.run(lambda i: fn(i) if i != 1 else None, 3)
        assert sorted(i for i, _ in seen) == [0, 2]
    finally:
        bench.CallerPool.fresh = keep


def test_xcd_split_block_mapping_is_a_bijection():
    """csrc/common.h xcd_tile_split / hift.hip conv_launch: the convolution grids are dealt to the 8 XCDs `nch` ways over the output-channel
    tiles and 8 / nch ways over the frame tiles; consecutive block ids go round the XCDs, the launcher pads x to a multiple of 8 / nch and
    blocks past x_real leave.  Restated here: every real (x, y) tile is visited exactly once, padding blocks only land on x >= x_real, and
    all tiles of one XCD lie in its (channel part, frame part)."""
    def mapping(gx, gy, nch, x_real):
        nfr, hy, qn = 8 // nch, gy // nch, gx // (8 // nch)
        out = {}
        for L in range(gx * gy):
            x, n = L & 7, L >> 3
            c, f = x % nch, x // nch
            dx = n // hy
            out[L] = (f * qn + dx, c * hy + (n - dx * hy), x)
        return out
    for gy in (2, 4, 8, 32):
        for nch in (2, 4, 8):
            if gy % nch:
                continue
            nfr = 8 // nch
            for x_real in (1, 3, 7, 8, 63, 75, 314):
                gx = (x_real + nfr - 1) // nfr * nfr
                m = mapping(gx, gy, nch, x_real)
                tiles = [(bx, by) for bx, by, _ in m.values()]
                assert len(set(tiles)) == gx * gy and all(0 <= bx < gx and 0 <= by < gy for bx, by in tiles)        # a bijection of the padded grid
                real = [(bx, by) for bx, by in tiles if bx < x_real]
                assert sorted(real) == [(x, y) for x in range(x_real) for y in range(gy)]                          # every real tile exactly once
                hy, qn = gy // nch, gx // nfr
                for bx, by, x in m.values():                                                                      # an XCD's tiles: one channel part, one frame part
                    assert by // hy == x % nch and bx // qn == x // nch


def test_bench_live_counter_fields(monkeypatch):
    """bench.live_counters: the arithmetic of the counter passes (FETCH_SIZE doubled, KiB units, per-launch / per-call division, MFMA busy over
    4 SIMDs x 256 CUs x duration x 2.4 GHz) on made-up pass results; a failing first pass gives {} (the committed files are the fallback)."""
    import shutil
    import bench
    monkeypatch.setattr(shutil, 'which', lambda name: '/opt/rocm/bin/rocprofv3')
    for k in list(os.environ):
        if k.startswith(('ROCPROF', 'ROCP_')):
            monkeypatch.delenv(k)
    monkeypatch.delenv('CV2_BENCH_LIVE_PMC', raising=False)

    def fake(stage, reps, counters, timeout_s=150):
        c = counters[0]
        if stage == 'decode':
            return {'k_step<false, true>': {'n': 40.0, 'ns': 40 * 330e3, c: 40 * (380000.0 if c == 'FETCH_SIZE' else 2000.0)}, 'k_sample': {'n': 40.0, 'ns': 4e5, c: 40.0}}
        if stage == 'hift':
            return {'k_conv6<1>': {'n': 81.0, 'ns': 3 * 1.2e6, c: 3 * 300000.0}, 'k_respair<1>': {'n': 36.0, 'ns': 3 * 0.6e6, c: 3 * 100000.0}, 'k_phase': {'n': 3.0, 'ns': 5e4, c: 3.0}}
        return {'k_gemm<64>': {'n': 100.0, 'ns': 1e6, 'SQ_VALU_MFMA_BUSY_CYCLES': 4 * 256 * 1e6 * 2.4 * 0.1, 'GRBM_GUI_ACTIVE': 8 * 4e6}}
    monkeypatch.setattr(bench, '_pmc_pass', fake)
    out = bench.live_counters()
    per, src = out['decode']
    assert per == int((2 * 380000.0 + 2000.0) * 1024) and 'measured in this run' in src and '40 k_step' in src
    h = out['hift']
    assert h['measured_in_this_run'] and h['conv_launches_per_call'] == 39
    assert abs(h['hbm_GB_per_10s_audio'] - (2 * 400003.0 + 400003.0) * 1024 / 1e9) < 1e-3
    assert h['top_conv_kernel']['kernel'] == 'k_conv6<1>'
    f = out['flow']
    assert abs(f['util_counter'] - 0.1) < 1e-4 and abs(f['util_counter_gui_active'] - 4 * 256 * 1e6 * 2.4 * 0.1 / (4 * 256 * 4e6)) < 1e-4
    monkeypatch.setattr(bench, '_pmc_pass', lambda *a, **k: None)
    assert bench.live_counters() == {}
