"""Utterance sharding over the GPUs of one node: one process per GPU, weights replicated, utterances are
independent units, so the only communication is
  broadcast  the shared prompt tensors (speech tokens, prompt mel, speaker embedding; ~165 KB)   once
  scatter    the padded text-id matrix rows of every rank's shard                                  once
  gather     lengths, then padded waveforms, to rank 0                                             once, at the end
over torch.distributed ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  No all-reduce, nothing per step.
"""
import torch
import torch.distributed as dist


def assign(lengths, world):
    """Length-balanced assignment: sort by expected length (text tokens; AR length is 2-20x that, llm.py:643-644) descending
    and deal round-robin with alternating direction (snake), so every rank gets the same count (+-1) and similar work.
    Returns list of index lists, one per rank."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards = [[] for _ in range(world)]
    for j, idx in enumerate(order):
        r = j % world
        if (j // world) % 2 == 1:
            r = world - 1 - r
        shards[r].append(idx)
    return shards


def _dev():
    return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')


def broadcast_prompt(prompt, src=0):
    """prompt: dict of tensors on rank `src` (None elsewhere).  Returns the dict on every rank (same keys, shapes, dtypes)."""
    meta = [{k: (tuple(v.shape), str(v.dtype)) for k, v in prompt.items()}] if dist.get_rank() == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for k, (shape, dtype) in meta[0].items():
        t = prompt[k].to(_dev()).contiguous() if dist.get_rank() == src else torch.empty(shape, dtype=getattr(torch, dtype.split('.')[1]), device=_dev())
        dist.broadcast(t, src=src)
        out[k] = t
    return out


def scatter_texts(texts, src=0):
    """texts: list of 1-D int32 tensors on rank `src`.  Returns (this rank's list of texts, their global indices)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == src:
        shards = assign([t.numel() for t in texts], world)
        maxlen = max(t.numel() for t in texts)
        per = max(len(s) for s in shards)
        mats = []
        for s in shards:                                   # rows: [index, length, ids...], padded to `per` rows
            m = torch.full((per, 2 + maxlen), -1, dtype=torch.int32)
            for j, idx in enumerate(s):
                m[j, 0], m[j, 1] = idx, texts[idx].numel()
                m[j, 2:2 + texts[idx].numel()] = texts[idx].to(torch.int32)
            mats.append(m.to(_dev()))
        shape = [list(mats[0].shape)]
    else:
        mats, shape = None, [None]
    dist.broadcast_object_list(shape, src=src)
    mine = torch.empty(shape[0], dtype=torch.int32, device=_dev())
    dist.scatter(mine, mats, src=src)
    mine = mine.cpu()
    rows = [r for r in mine if int(r[0]) >= 0]
    return [r[2:2 + int(r[1])].clone() for r in rows], [int(r[0]) for r in rows]


def gather_waves(waves, indices, n_total, dst=0):
    """waves: this rank's list of 1-D float32 waveforms with their global indices.  Rank `dst` gets the full list in the
    original order (CPU tensors); the other ranks get None.  Waveforms that are still on the rank's device (tts(device_output=True))
    go into the gather buffer device to device, and rank `dst` makes ONE device-to-host copy of what it gathered (CPU waveforms used to be
    copied back to the device for the collective: ~250 MB bounced twice at configs[3])."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = _dev()
    lens = [int(w.numel()) for w in waves]
    meta = [None] * world
    dist.all_gather_object(meta, (indices, lens))
    per = max(len(m[0]) for m in meta)
    maxlen = max([l for m in meta for l in m[1]] + [1])
    buf = torch.zeros(per, maxlen, dtype=torch.float32, device=dev)
    for j, w in enumerate(waves):
        buf[j, :lens[j]].copy_(w.reshape(-1), non_blocking=True)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    host = torch.stack(out).cpu()                          # [world][per][maxlen]
    res = [None] * n_total
    for r, (idxs, ls) in enumerate(meta):
        for j, (idx, l) in enumerate(zip(idxs, ls)):
            res[idx] = host[r, j, :l].clone()
    return res


def synthesize_sharded(texts, prompt, synth_fn, src=0):
    """texts / prompt are only needed on rank `src`.  synth_fn(list_of_text_tensors, prompt_dict) -> list of 1-D waveforms.
    Returns the waveforms in input order on rank `src`, None elsewhere."""
    n = [len(texts) if dist.get_rank() == src else None]
    dist.broadcast_object_list(n, src=src)
    p = broadcast_prompt(prompt, src)
    mine, idx = scatter_texts(texts, src)
    waves = synth_fn(mine, p) if mine else []
    return gather_waves(waves, idx, n[0], src)
