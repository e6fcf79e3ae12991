"""`cosyvoice2_eu` — the standalone package surface (standalone_infer/src/cosyvoice2_eu/__init__.py:33-128) on the
MI355X build: `load(...) -> Cosy2EU`, `Cosy2EU.tts(text, prompt) -> (wav [1, T], sample_rate)`, `.stream(...)`."""
import os
from typing import Iterator, Optional, Tuple

import torch

from cosyvoice.cli.cosyvoice import CosyVoice2
from cosyvoice.utils.file_utils import load_wav

__all__ = ['__version__', 'Cosy2EU', 'load']
__version__ = '0.2.8+mi355x'


class Cosy2EU:
    """Lightweight wrapper around CosyVoice2 for interactive inference."""

    def __init__(self, model: CosyVoice2):
        self._model = model

    @property
    def sample_rate(self) -> int:
        return getattr(self._model, 'sample_rate', 24000)

    def tts(self, text: str, prompt: str, *, speed: float = 1.0, text_frontend: bool = False) -> Tuple[torch.Tensor, int]:
        prompt_16k = load_wav(prompt, 16000)
        segments = [out['tts_speech'] for out in self._model.inference_cross_lingual(text, prompt_16k, stream=False, speed=speed,
                                                                                       text_frontend=text_frontend)]
        wav = segments[0] if len(segments) == 1 else torch.cat(segments, dim=1)
        return wav, self.sample_rate

    def stream(self, text: str, prompt: str, *, speed: float = 1.0, text_frontend: bool = False) -> Iterator[torch.Tensor]:
        prompt_16k = load_wav(prompt, 16000)
        for out in self._model.inference_cross_lingual(text, prompt_16k, stream=True, speed=speed, text_frontend=text_frontend):
            yield out['tts_speech']


def load(*, model_dir: Optional[str] = None, repo_id: str = 'hi-paris/CosyVoice2-0.5B-EU', download: bool = True,
         setting: str = 'llm_flow_hifigan', llm_run_id: str = 'latest', flow_run_id: str = 'latest', hifigan_run_id: str = 'latest',
         final: Optional[bool] = None, backbone: str = 'blanken') -> Cosy2EU:
    """Load CosyVoice2-EU once and reuse for multiple in-memory calls."""
    model_dir = model_dir or os.path.expanduser('~/.cache/cosyvoice2-eu')
    if download:
        from huggingface_hub import snapshot_download
        snapshot_download(repo_id=repo_id, local_dir=model_dir)
    model = CosyVoice2(model_dir, load_jit=False, load_trt=False, load_vllm=False, fp16=False, setting=setting, llm_run_id=llm_run_id,
                       flow_run_id=flow_run_id, hifigan_run_id=hifigan_run_id, final=(True if final is None else final), backbone=backbone)
    return Cosy2EU(model)
