"""TEST INFRASTRUCTURE: the calling pattern of the reference's evaluation harness, restated (not imported: the harness needs
torchaudio / jiwer / whisper and does not exist on the GPU box) so that a test can drive the drop-in `CosyVoice2` exactly the way
`evaluation/run_evaluation_pipeline.py` does.  Follows `/root/reference/evaluation/cosyvoice_synthesizer.py`:

  _ensure_prompt_cached   :97-110   add_zero_shot_spk(prompt_text, prompt_speech, id) once per id, failures only logged
  synthesize_single       :122-181  method -> inference_cross_lingual / inference_zero_shot / inference_instruct2 with stream=False,
                                    speed, text_frontend, zero_shot_spk_id; the yielded 'tts_speech' chunks concatenated on the CPU
  synthesize_batch        :183-302  warm-up call "warmup." (:203-209), ThreadPoolExecutor(workers) with workers = inference.workers or
                                    batch_size (:216-218, eval_config.yaml:28), language hint prefix (:223-225), as_completed +
                                    fut.result(timeout=timeout_s) (:261-266), per-sample try / except -> error rows, result keys
                                    utterance_id / audio_tensor / audio_path / sample_rate / synthesis_time [/ error] (:242-258, :272-300)
and the per-utterance real-time factor of `run_evaluation_pipeline.py:265-274` (synthesis_time / (samples / sample_rate))."""
import logging
import math
import time
from concurrent.futures import ThreadPoolExecutor, TimeoutError as FuturesTimeout, as_completed

import torch

log = logging.getLogger('harness_replay')


class SynthesizerReplay:
    def __init__(self, model, prompt_speech):
        self.model = model                     # a CosyVoice2 (the reference constructs it in load_model, :55-82)
        self.prompt_speech = prompt_speech     # load_wav(prompt, 16000) in the reference (:84-95)
        self.cached_spk_id = None

    def _ensure_prompt_cached(self, cfg):
        try:
            text, spk = cfg.get('prompt_text'), cfg.get('zero_shot_spk_id')
            if text and spk and self.cached_spk_id != spk:
                if self.model.add_zero_shot_spk(text, self.prompt_speech, spk):
                    self.cached_spk_id = spk
        except Exception as e:                 # noqa: BLE001 -- the harness only warns (:109-110)
            log.warning('Failed to cache zero-shot speaker: %s', e)

    def synthesize_single(self, text, cfg):
        if self.prompt_speech is None:
            raise ValueError('Prompt speech not loaded')
        method = cfg.get('method', 'cross_lingual')
        kw = dict(stream=False, speed=cfg.get('speed', 1.0), text_frontend=cfg.get('text_frontend', False))
        spk = cfg.get('zero_shot_spk_id', '') or ''
        self._ensure_prompt_cached(cfg)
        if method == 'cross_lingual':
            gen = self.model.inference_cross_lingual(text, self.prompt_speech, zero_shot_spk_id=spk, **kw)
        elif method == 'zero_shot':
            gen = self.model.inference_zero_shot(text, cfg.get('prompt_text', ''), self.prompt_speech, zero_shot_spk_id=spk, **kw)
        elif method == 'instruct2':
            gen = self.model.inference_instruct2(text, cfg.get('instruct_text', ''), self.prompt_speech, **kw)
        else:
            raise ValueError(f'Unknown inference method: {method}')
        chunks = [out['tts_speech'].cpu() for out in gen]
        if not chunks:
            raise RuntimeError('No output from model')
        return chunks[0] if len(chunks) == 1 else torch.cat(chunks, dim=1)

    @staticmethod
    def _row(sample, audio=None, sr=None, t=0.0, error=None):
        row = {'utterance_id': sample['utterance_id'], 'audio_tensor': audio, 'audio_path': None, 'sample_rate': sr, 'synthesis_time': t}
        if error is not None:
            row['error'] = error
        return row

    def synthesize_batch(self, samples, cfg):
        self._ensure_prompt_cached(cfg)
        if cfg.get('warmup', False):
            try:
                self.synthesize_single('warmup.', cfg)
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception as e:             # noqa: BLE001
                log.warning('Warm-up failed: %s', e)
        workers = int(cfg.get('workers', cfg.get('batch_size', 1)) or 1)
        timeout_s = float(cfg.get('timeout_s', 30))
        results = [None] * len(samples)

        def work(idx, sample):
            t0 = time.time()
            text = sample['text']
            if cfg.get('add_language_hint', False) and cfg.get('language') in ('fr', 'de'):
                text = ('<|fr|><|endofprompt|> ' if cfg.get('language') == 'fr' else '<|de|><|endofprompt|> ') + text
            audio = self.synthesize_single(text, cfg)
            return idx, audio, time.time() - t0

        def finish(idx, get):
            sample = samples[idx]
            try:
                i, audio, elapsed = get()
                results[i] = self._row(sample, audio, self.model.sample_rate, elapsed)
            except FuturesTimeout:
                results[idx] = self._row(sample, t=timeout_s, error=f'timeout {timeout_s}s')
            except Exception as e:             # noqa: BLE001 -- one failing sample is one error row (:290-300)
                results[idx] = self._row(sample, error=str(e))

        if workers <= 1:
            for idx, sample in enumerate(samples):
                finish(idx, lambda: work(idx, sample))
        else:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                futs = {ex.submit(work, idx, sample): idx for idx, sample in enumerate(samples)}
                for fut in as_completed(futs):
                    finish(futs[fut], lambda: fut.result(timeout=timeout_s))
        return results


def rtf(row):
    """run_evaluation_pipeline.py:265-274"""
    try:
        if row.get('sample_rate') and row['audio_tensor'] is not None:
            dur = float(row['audio_tensor'].shape[1]) / float(row['sample_rate'])
            if dur > 0:
                return float(row['synthesis_time']) / dur
    except Exception:                          # noqa: BLE001
        pass
    return math.nan
