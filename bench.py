#!/usr/bin/env python3
"""Headline benchmark: CosyVoice2-0.5B-EU zero-shot synthesis throughput on MI355X (BASELINE.json metric).

A "step" = `--batch` utterances (default 1 = BASELINE configs[1]: zero-shot FR, batch 1, bf16 weights, non-streaming)
through the whole hot path on this rank's GPU: LLM prefill + autoregressive decode with the RAS sampler, flow encoder +
10 Euler steps with CFG, HiFT vocoder.  Inputs (text ids, prompt speech tokens, prompt mel, speaker embedding) are
resident in HBM before the timed region.  Weights are synthetic at the real shapes (no checkpoints offline), decode
length forced to 250 tokens = 10 s of audio per utterance.

  python bench.py --gpus N --steps K --warmup W [--batch B]
N > 1: launched by torch.distributed.run, one rank per GPU, every rank synthesises its own utterances (weak scaling, no
data-path collective); time = max over ranks between barriers.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel group, the LLM decode step (HBM-bound weight
stream): algorithmic bytes per step / average step duration from HIP events on the launch stream.
`cpu_baseline` is the CPU fp32 oracle (oracle/, a port of the reference's algorithm) timed on this box's host cores on a
bounded sample and extrapolated linearly to the full utterance.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'cosyvoice2-eu_amd'))

N_TOK = 250          # generated speech tokens per utterance (10 s)
P_TOK = 255          # FR prompt: 10.22 s -> 255 speech tokens
TEXT_LEN = 50
PROMPT_TEXT_LEN = 20
HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK = 2500.0   # TFLOP/s dense
FP32_MFMA_PEAK = 157.3


def llm_step_bytes(weight_bytes, kv_positions_sum):
    # bf16 weights once per step + fp32 KV read: 24 layers x 2 (k, v) x 2 kv heads x 64 x 4 B per cached position
    return weight_bytes + 24 * 2 * 2 * 64 * 4 * kv_positions_sum


def flow_flops(T):
    # estimator call (CFG pair) = 0.2644 GFLOP*T + 229376*T^2 ; 10 Euler steps ; + encoder ~80 GFLOP at T=1010
    return (0.2644e9 * T + 229376.0 * T * T) * 10 + 80e9 * (T / 1010.0)


def host_cores():
    """CPU share of this process: cgroup quota if any, else the affinity mask (a GPU box exposes 256 logical CPUs but
    grants ~16 to one GPU's container; oversubscribing torch threads makes the baseline meaningless)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(n_threads):
    """CPU fp32 oracle on a bounded sample, extrapolated linearly to the full C2 utterance."""
    import torch
    from cv2amd import synth
    from oracle import llm as OL, flow as OF, hift as OH
    torch.set_num_threads(n_threads)
    inp = synth.synthetic_inputs(text_len=TEXT_LEN, prompt_len=P_TOK, prompt_text_len=PROMPT_TEXT_LEN)
    with torch.inference_mode():
        sd = synth.make_llm()
        d = OL.LLMDims(sd)
        x = OL.build_lm_input(sd, inp['text'], inp['prompt_text'], inp['prompt_token'])
        cache = [None] * d.layers
        t0 = time.time()
        y = OL.qwen2_step(sd, d, x, cache)
        t_prefill = time.time() - t0
        n_dec = 6
        t0 = time.time()
        for i in range(n_dec):
            logp = torch.nn.functional.linear(y[-1], sd['llm_decoder.weight'], sd['llm_decoder.bias']).log_softmax(-1)
            top = int(logp[:6561].argmax())
            y = OL.qwen2_step(sd, d, sd['speech_embedding.weight'][top:top + 1], cache)
        t_step = (time.time() - t0) / n_dec
        del sd, cache
        fsd = synth.make_flow()
        T = 2 * (P_TOK + N_TOK)
        g = torch.Generator().manual_seed(0)
        tok = torch.randint(0, 6561, (1, P_TOK + N_TOK), generator=g)
        t0 = time.time()
        h = OF.encoder(fsd, fsd['input_embedding.weight'][tok], None, False)
        t_enc = time.time() - t0
        mu = OF.lin(fsd, 'encoder_proj', h).transpose(1, 2).contiguous()
        z = OF.rand_noise()[:, :, :T]
        t0 = time.time()
        OF.solve_euler(fsd, z, mu, torch.ones(1, 1, T), torch.zeros(1, 80), torch.zeros(1, 80, T), n_timesteps=1)
        t_est = time.time() - t0
        del fsd
        hsd = synth.make_hift()
        Th = 100
        mel = (torch.randn(1, 80, Th, generator=g) * 2 - 4).clamp(-11.5, 2)
        t0 = time.time()
        OH.inference(hsd, mel, torch.zeros(1, 1, 0), torch.rand(1, 9), torch.randn(1, 480 * Th, 9))
        t_hift = time.time() - t0
    total = t_prefill + N_TOK * t_step + t_enc + 10 * t_est + (2 * N_TOK / Th) * t_hift
    audio_s = N_TOK / 25.0
    return {'value': round(audio_s / total, 4), 'unit': 'audio-s/s', 'cores': n_threads, 'kind': 'port',
            'sample': f'oracle/ (torch CPU fp32): LLM prefill {x.shape[0]} rows ({t_prefill:.2f}s) + {n_dec} decode steps ({t_step * 1e3:.0f} ms/step), '
                      f'flow encoder ({t_enc:.2f}s) + 1 of 10 Euler steps at T={T} ({t_est:.2f}s), HiFT on {Th} of {2 * N_TOK} frames ({t_hift:.2f}s); '
                      f'extrapolated linearly to one full 10 s utterance = {total:.1f}s (RTF {total / audio_s:.2f})'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the hot path has no CPU fallback)'
    # test hooks (one-GPU boxes): CV2_BENCH_DEVICE pins every rank to one GPU, CV2_BENCH_BACKEND=gloo replaces RCCL for the barriers
    dev_index = int(os.environ.get('CV2_BENCH_DEVICE', local_rank))
    backend = os.environ.get('CV2_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = f'cuda:{dev_index}'
    if world > 1 or 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)
    use_dist = dist.is_initialized()

    from cv2amd import lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    from cv2amd import synth
    from cv2amd.llm import MODE_RAS
    from cv2amd.pipeline import Synthesizer, synthetic_request

    B = args.batch
    syn = Synthesizer(synth.make_llm(), synth.make_flow(), synth.make_hift(), dev, max_batch=B, max_text=128,
                      max_prompt_tokens=320, max_new_tokens=512)
    # C2: FR prompt; C3 (batch 32): half FR (P=255), half DE (P=310), N ~ U{150..500} seeded
    g = torch.Generator().manual_seed(1986)
    reqs, ntoks = [], []
    for b in range(B):
        P = P_TOK if (B == 1 or b % 2 == 0) else 310
        reqs.append(synthetic_request(seed=1986 + 100 * rank + b, text_len=TEXT_LEN, prompt_len=P, device=dev, prompt_text_len=PROMPT_TEXT_LEN))
        ntoks.append(N_TOK if B == 1 else int(torch.randint(150, 501, (1,), generator=g)))
    n_fixed = N_TOK if B == 1 else max(ntoks)
    llm_reqs = [(r['text'], r['prompt_text'], r['llm_prompt_speech_token']) for r in reqs]
    L0 = [1 + r['prompt_text'].numel() + r['text'].numel() + 1 + r['llm_prompt_speech_token'].numel() for r in reqs]

    ev = lambda: torch.cuda.Event(enable_timing=True)
    stage_ms = {'llm_prefill': 0.0, 'llm_decode': 0.0, 'flow': 0.0, 'hift': 0.0}
    dec_steps = 0

    def step(timed, seed):
        nonlocal dec_steps
        e = [ev() for _ in range(4)]
        e[0].record()
        toks, (d0, d1) = syn.llm.generate_fixed(llm_reqs, n_fixed, mode=MODE_RAS, seed=seed)
        toks = [t[:n] for t, n in zip(toks, ntoks)]          # batch 32: ragged lengths (the LLM ran max(N) steps for all)
        e[1].record()
        utts = [dict(token=torch.tensor(t, dtype=torch.int32, device=dev).unsqueeze(0), prompt_token=r['flow_prompt_speech_token'],
                     prompt_feat=r['prompt_speech_feat'], embedding=r['flow_embedding']) for r, t in zip(reqs, toks)]
        mels = syn.flow.inference_batch(utts, streaming=False, finalize=True)
        e[2].record()
        wavs = [w for w, _ in syn.hift_pool.inference_many(mels)]
        e[3].record()
        if timed:
            torch.cuda.synchronize()
            stage_ms['llm_decode'] += d0.elapsed_time(d1)
            stage_ms['llm_prefill'] += e[0].elapsed_time(e[1]) - d0.elapsed_time(d1)
            stage_ms['flow'] += e[1].elapsed_time(e[2])
            stage_ms['hift'] += e[2].elapsed_time(e[3])
            dec_steps += n_fixed - 1
        return wavs

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        wavs = step(False, i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wavs = step(True, 1000 + i)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(w).all() for w in wavs)
    audio_per_step = sum(w.shape[1] for w in wavs) / 24000.0
    value = audio_per_step * world * args.steps / dt

    if rank == 0:
        # roofline of the dominant kernel group: LLM decode step
        kv_sum = sum(sum(l0 + i for i in range(1, n_fixed)) for l0 in L0) / (n_fixed - 1)     # mean cached positions per step, all sequences
        step_ms = stage_ms['llm_decode'] / dec_steps
        step_bytes = llm_step_bytes(syn.llm.weight_bytes, kv_sum)
        achieved = step_bytes / (step_ms * 1e-3) / 1e9
        Ts = [2 * (r['flow_prompt_speech_token'].numel() + n) for r, n in zip(reqs, ntoks)]
        flow_tf = sum(flow_flops(T) for T in Ts) * args.steps / (stage_ms['flow'] * 1e-3) / 1e12
        hift_tf = 30.6e9 * audio_per_step * args.steps / (stage_ms['hift'] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, 'profiles', 'r1_pmc_decode.json')
        if B == 1 and os.path.exists(pmc):          # PMC counters cannot be read from inside the bench: committed rocprofv3 --pmc measurement
            pj = json.load(open(pmc))
            traffic, traffic_src = pj['hbm_bytes_per_step'], 'profiles/r1_pmc_decode.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 fetch correction)'
        out = {
            'metric': 'audio-sec/sec, CosyVoice2-0.5B-EU zero-shot FR', 'value': round(value, 3), 'unit': 'audio-s/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'rtf': round(dt / (audio_per_step * args.steps), 5),
            'config': {'workload': f'configs[{1 if B == 1 else 2}]: zero-shot FR{"+DE" if B > 1 else ""}, batch={B} per GPU, non-streaming, P=255{"/310" if B > 1 else ""} prompt tokens, '
                                   f'{TEXT_LEN} text tokens, {"250" if B == 1 else "U{150..500}"} generated tokens (forced), 10 Euler steps + CFG, RAS sampler; '
                                   f'LLM bf16 weights / fp32 KV, flow bf16 MFMA, HiFT fp32 MFMA', 'batch_per_gpu': B,
                       'audio_s_per_step_per_gpu': round(audio_per_step, 3)},
            'roofline': {'bound': 'hbm', 'kernel': 'LLM decode step = one hipGraph replay (24 x {k_qkv, k_attn, k_store(o), k_gateup, k_store(down)} + head + k_sample)',
                         'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'traffic_source': traffic_src, 'bytes_per_launch': int(step_bytes), 'avg_launch_us': round(step_ms * 1e3, 2)},
            'stages': {'ms_per_step': {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
                       'flow_mfma': {'achieved': round(flow_tf, 1), 'peak': MFMA_BF16_PEAK, 'unit': 'TFLOP/s', 'frac': round(flow_tf / MFMA_BF16_PEAK, 4)},
                       'hift_fp32_mfma': {'achieved': round(hift_tf, 2), 'peak': FP32_MFMA_PEAK, 'unit': 'TFLOP/s', 'frac': round(hift_tf / FP32_MFMA_PEAK, 4)}},
        }
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is a rank-0, one-GPU-run item
            out['cpu_baseline'] = cpu_baseline(host_cores())
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
