"""The three stages chained on one GPU: text ids + prompt -> speech tokens -> mel -> waveform.

This is the body of `CosyVoice2Model.tts(..., stream=False)` (cosyvoice/cli/model.py:336-401): `llm_job` (:118-139) then
`token2wav(finalize=True)` (:300-334), for a batch of utterances.  The reference runs one utterance per call and a
Python thread per LLM; here the utterances of a batch share every LLM decode step (one hipGraph replay per step for
the whole batch), the flow runs over the packed ragged batch, and HiFT runs per utterance.
"""
import torch

from .flow import FlowEngine
from .hift import HiftEngine, HiftPool
from .llm import LLMEngine, MODE_GREEDY, MODE_RAS


class Synthesizer:
    sample_rate = 24000

    def __init__(self, llm_sd, flow_sd, hift_sd, device='cuda:0', max_batch=1, max_text=256, max_prompt_tokens=750,
                 max_new_tokens=1536):
        self.device = torch.device(device)
        max_pos = max_text + max_prompt_tokens + max_new_tokens + 8
        self.llm = LLMEngine(llm_sd, device, max_seqs=max(1, max_batch), max_pos=max_pos, max_out=max_new_tokens)
        max_len = 2 * (max_prompt_tokens + max_new_tokens)
        self.flow = FlowEngine(flow_sd, device, max_utts=max_batch, max_len=max_len)
        self.hift_pool = HiftPool(hift_sd, device, max_frames=2 * max_new_tokens, n=1 if max_batch == 1 else 4)
        self.hift = self.hift_pool.engines[0]
        self.max_batch = max_batch

    def tokens(self, reqs, mode=MODE_GREEDY, seed=0, force_len=None):
        """reqs: list of dicts with text [1,Lt], prompt_text [1,Lp] (may be empty), llm_prompt_speech_token [1,P'] (may be
        empty: cross-lingual mode, cli/frontend.py:515-522).  Returns list of token lists."""
        return self.llm.generate([(r['text'], r['prompt_text'], r['llm_prompt_speech_token']) for r in reqs], mode=mode,
                                 seed=seed, force_len=force_len)

    def token2wav(self, reqs, toks, speed=1.0, noise=None):
        """cli/model.py:300-334 with finalize=True, token_offset=0: flow -> (speed) -> hift.  reqs carry
        flow_prompt_speech_token [1,P], prompt_speech_feat [1,2P,80], flow_embedding [1,192]."""
        utts = [dict(token=torch.tensor(t, dtype=torch.int32).unsqueeze(0), prompt_token=r['flow_prompt_speech_token'],
                     prompt_feat=r['prompt_speech_feat'], embedding=r['flow_embedding']) for r, t in zip(reqs, toks)]
        mels = self.flow.inference_batch(utts, streaming=False, finalize=True)
        wavs = []
        for i, mel in enumerate(mels):
            if speed != 1.0:                                                      # cli/model.py:328-330
                mel = torch.nn.functional.interpolate(mel, size=int(mel.shape[2] / speed), mode='linear')
            wav, _ = self.hift.inference(mel, None, noise=None if noise is None else noise[i])
            wavs.append(wav)
        return wavs

    def synthesize(self, reqs, mode=MODE_RAS, seed=0, force_len=None, speed=1.0):
        toks = self.tokens(reqs, mode, seed, force_len)
        return self.token2wav(reqs, toks, speed), toks


def synthetic_request(seed=1986, text_len=50, prompt_len=255, device='cuda:0', zero_shot=True, prompt_text_len=20):
    """SURVEY.md §8(d) inputs, already resident on the device."""
    from . import synth
    inp = synth.synthetic_inputs(seed=seed, text_len=text_len, prompt_len=prompt_len, prompt_text_len=prompt_text_len if zero_shot else 0)
    d = torch.device(device)
    e0 = torch.zeros(1, 0, dtype=torch.int32)
    return dict(text=inp['text'].to(d), prompt_text=inp['prompt_text'].to(d),
                llm_prompt_speech_token=(inp['prompt_token'] if zero_shot else e0).to(d),
                flow_prompt_speech_token=inp['prompt_token'].to(d), prompt_speech_feat=inp['prompt_feat'].to(d),
                flow_embedding=inp['embedding'].to(d))
