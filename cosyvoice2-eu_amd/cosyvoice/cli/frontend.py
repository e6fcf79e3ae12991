"""Prompt / text frontend (cosyvoice/cli/frontend.py of the reference).  OUT OF THE HOT PATH ("next #1"):
it runs once per prompt on third-party models (Qwen tokenizer, whisper log-mel -> speech_tokenizer_v2.onnx,
kaldi fbank -> campplus.onnx).  This module keeps the interface the API layer calls; the two feature extractors and the prompt mel
run on the device (cv2amd/prompt.py), the two ONNX graphs through onnxruntime when it is installed; `PrecomputedFrontEnd` serves pre-extracted prompts (the reference's own `spk2info` mechanism,
cli/cosyvoice.py:70-76) and is what the tests and the benchmark use.
"""
import os

import torch


class FrontEndUnavailable(RuntimeError):
    pass


class PrecomputedFrontEnd:
    """Frontend over pre-extracted prompts: spk2info[id] holds the dict frontend_zero_shot would build (prompt_text tokens,
    speech tokens, prompt mel, embeddings).  Text -> ids goes through `tokenize` (any callable str -> list[int])."""

    def __init__(self, tokenize, spk2info=None):
        self.tokenize = tokenize
        self.spk2info = dict(spk2info or {})

    def text_normalize(self, text, split=True, text_frontend=True, multilingual=True, pack_mode='sentence', target_token_len=512):
        """cli/frontend.py:419-480.  A generator passes through untouched (bistream input); text_frontend=False or an empty string
        means no normalisation and no splitting; otherwise sentence-level normalisation, then `split_paragraph` packs the sentences
        into segments of at most ~80 tokens (60 minimum, a tail under 20 tokens joins its predecessor), so the LLM never sees more
        text per request than it was trained on; punctuation-only segments are dropped."""
        from collections.abc import Generator
        from cosyvoice.utils import frontend_utils as U
        if isinstance(text, Generator):
            return [text]
        if text_frontend is False or text == '':
            return [text] if split else text
        text = text.strip()
        if not text:
            return [''] if split else ''
        sents = U.split_sentences(text) if multilingual else [text]
        parser = getattr(self, 'inflect_parser', None)              # frontend.py:122: inflect.engine(); the restatement when absent
        normalized = [U.normalize_sentence(s, U.detect_lang(s), parser) for s in sents]
        tok = self.tokenize
        if pack_mode == 'paragraph':
            n = int(target_token_len)
            segs = U.split_paragraph(' '.join(normalized).strip(), tok, 'en', token_max_n=n, token_min_n=max(1, int(0.75 * n)),
                                     merge_len=max(1, int(0.25 * n)), comma_split=False)
        else:
            segs = []
            for s in normalized:
                segs.extend(U.split_paragraph(s, tok, 'en', token_max_n=80, token_min_n=60, merge_len=20, comma_split=False))
        texts = [t for t in segs if not U.is_only_punctuation(t)]
        return texts if split else ' '.join(texts)

    def _extract_text_token(self, text):
        ids = self.tokenize(text)
        t = torch.tensor([ids], dtype=torch.int32)
        return t, torch.tensor([t.shape[1]], dtype=torch.int32)

    def frontend_sft(self, tts_text, spk_id):
        """frontend.py:485-489: text ids + the stored speaker embedding of `spk_id` (spk2info.pt of an SFT model dir).  A CosyVoice2 model
        dir ships no spk2info: an unknown id raises KeyError exactly as the reference's `self.spk2info[spk_id]` does."""
        tts_text_token, tts_text_token_len = self._extract_text_token(tts_text)
        embedding = self.spk2info[spk_id]['embedding']
        return {'text': tts_text_token, 'text_len': tts_text_token_len, 'llm_embedding': embedding, 'flow_embedding': embedding}

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_speech_16k, resample_rate, zero_shot_spk_id):
        if zero_shot_spk_id == '':
            raise FrontEndUnavailable('prompt feature extraction needs onnxruntime + whisper + the model_dir ONNX files; '
                                      'use add_zero_shot_spk() with pre-extracted features or install them')
        model_input = dict(self.spk2info[zero_shot_spk_id])
        model_input['text'], model_input['text_len'] = self._extract_text_token(tts_text)
        return model_input

    def frontend_cross_lingual(self, tts_text, prompt_speech_16k, resample_rate, zero_shot_spk_id):
        model_input = self.frontend_zero_shot(tts_text, '', prompt_speech_16k, resample_rate, zero_shot_spk_id)
        for k in ('prompt_text', 'prompt_text_len', 'llm_prompt_speech_token', 'llm_prompt_speech_token_len'):
            model_input.pop(k, None)              # frontend.py:517-521: the LLM sees the text only
        return model_input


    def frontend_instruct2(self, tts_text, instruct_text, prompt_speech_16k, resample_rate, zero_shot_spk_id):
        """frontend.py:533-537: zero-shot input whose prompt text is the instruction + '<|endofprompt|>', without the LLM's prompt
        speech tokens.  As in the reference, a registered speaker id takes its stored prompt text (frontend.py:509-510)."""
        model_input = self.frontend_zero_shot(tts_text, instruct_text + '<|endofprompt|>', prompt_speech_16k, resample_rate, zero_shot_spk_id)
        for k in ('llm_prompt_speech_token', 'llm_prompt_speech_token_len'):
            model_input.pop(k, None)
        return model_input


    def frontend_vc(self, source_speech_16k, prompt_speech_16k, resample_rate):
        """frontend.py:539-549.  Without the ONNX speech tokenizer the two arguments name pre-extracted entries: `source_speech_16k`
        = a [1, n] tensor of source speech tokens (or a key of spk2info whose flow_prompt_speech_token is used), `prompt_speech_16k`
        = a key of spk2info holding the prompt's tokens / mel / embedding."""
        if not isinstance(prompt_speech_16k, str):
            raise FrontEndUnavailable('frontend_vc on raw audio needs onnxruntime + whisper + the model_dir ONNX files')
        p = self.spk2info[prompt_speech_16k]
        src = self.spk2info[source_speech_16k]['flow_prompt_speech_token'] if isinstance(source_speech_16k, str) else source_speech_16k
        src = src.reshape(1, -1).to(torch.int32)
        return {'source_speech_token': src, 'source_speech_token_len': torch.tensor([src.shape[1]], dtype=torch.int32),
                'flow_prompt_speech_token': p['flow_prompt_speech_token'], 'flow_prompt_speech_token_len': p['flow_prompt_speech_token_len'],
                'prompt_speech_feat': p['prompt_speech_feat'], 'prompt_speech_feat_len': p['prompt_speech_feat_len'],
                'flow_embedding': p['flow_embedding']}


class CosyVoiceFrontEnd(PrecomputedFrontEnd):
    """The reference's frontend on its own third-party stack (frontend.py:38-127).  Raises FrontEndUnavailable at construction
    when that stack (onnxruntime, whisper, transformers tokenizer files) is not installed."""

    def __init__(self, model_dir, allowed_special='all'):
        try:
            import onnxruntime
            from transformers import AutoTokenizer
        except ImportError as e:
            raise FrontEndUnavailable('CosyVoiceFrontEnd needs onnxruntime and transformers: {}'.format(e))
        self._speech = None                                                      # cv2amd.prompt.SpeechFeatures (device), built on first use
        tok_dir = os.path.join(model_dir, 'CosyVoice-BlankEN')
        self._tok = AutoTokenizer.from_pretrained(tok_dir)
        special = {'eos_token': '<|endoftext|>', 'pad_token': '<|endoftext|>',
                   'additional_special_tokens': ['<|im_start|>', '<|im_end|>', '<|endofprompt|>', '[breath]', '<strong>', '</strong>', '[noise]',
                                                 '[laughter]', '[cough]', '[clucking]', '[accent]', '[quick_breath]', '<laughter>', '</laughter>',
                                                 '[hissing]', '[sigh]', '[vocalized-noise]', '[lipsmack]', '[mn]']}
        self._tok.add_special_tokens(special)                                   # tokenizer/tokenizer.py:244-265
        super().__init__(lambda t: self._tok([t], return_tensors='pt')['input_ids'][0].tolist())
        self._prompt, self._prompt_rate = None, None                             # cv2amd.prompt.PromptFeatures, built on first use
        opt = onnxruntime.SessionOptions()
        opt.graph_optimization_level = onnxruntime.GraphOptimizationLevel.ORT_ENABLE_ALL
        opt.intra_op_num_threads = 1
        self.campplus_session = onnxruntime.InferenceSession(os.path.join(model_dir, 'campplus.onnx'), sess_options=opt, providers=['CPUExecutionProvider'])
        self.speech_tokenizer_session = onnxruntime.InferenceSession(os.path.join(model_dir, 'speech_tokenizer_v2.onnx'), sess_options=opt,
                                                                     providers=['CPUExecutionProvider'])
        spk = os.path.join(model_dir, 'spk2info.pt')
        if os.path.exists(spk):
            self.spk2info = torch.load(spk, map_location='cpu')

    def _speech_features(self):
        if self._speech is None:
            from cv2amd.prompt import SpeechFeatures
            self._speech = SpeechFeatures(self._feature_device())
        return self._speech

    @staticmethod
    def _feature_device():
        """the process's current device (one rank per GPU calls torch.cuda.set_device(LOCAL_RANK) first): the extractors' tables, buffers and
        launches live there, not on cuda:0"""
        return torch.device('cuda', torch.cuda.current_device())

    def _extract_speech_token(self, speech):                                    # frontend.py:262-274
        assert speech.shape[1] / 16000 <= 30, 'do not support extract speech token for audio longer than 30s'
        feat = self._speech_features().whisper_log_mel(speech)                   # whisper.log_mel_spectrogram(speech, n_mels=128) on the device
        ort_in = self.speech_tokenizer_session.get_inputs()
        tok = self.speech_tokenizer_session.run(None, {ort_in[0].name: feat.detach().cpu().numpy(),
                                                       ort_in[1].name: __import__('numpy').array([feat.shape[2]], dtype='int32')})[0].flatten().tolist()
        t = torch.tensor([tok], dtype=torch.int32)
        return t, torch.tensor([t.shape[1]], dtype=torch.int32)

    def _extract_spk_embedding(self, speech):                                   # frontend.py:276-283
        feat = self._speech_features().kaldi_fbank(speech)                       # kaldi.fbank(...) - mean over frames, on the device
        emb = self.campplus_session.run(None, {self.campplus_session.get_inputs()[0].name: feat.unsqueeze(0).cpu().numpy()})[0].flatten().tolist()
        return torch.tensor([emb])

    def frontend_vc(self, source_speech_16k, prompt_speech_16k, resample_rate):
        if isinstance(prompt_speech_16k, str):
            return super().frontend_vc(source_speech_16k, prompt_speech_16k, resample_rate)
        z = self.frontend_zero_shot('', '', prompt_speech_16k, resample_rate, '')
        src, src_len = self._extract_speech_token(source_speech_16k)
        return {'source_speech_token': src, 'source_speech_token_len': src_len,
                'flow_prompt_speech_token': z['flow_prompt_speech_token'], 'flow_prompt_speech_token_len': z['flow_prompt_speech_token_len'],
                'prompt_speech_feat': z['prompt_speech_feat'], 'prompt_speech_feat_len': z['prompt_speech_feat_len'], 'flow_embedding': z['flow_embedding']}

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_speech_16k, resample_rate, zero_shot_spk_id):
        if zero_shot_spk_id != '':
            return super().frontend_zero_shot(tts_text, prompt_text, prompt_speech_16k, resample_rate, zero_shot_spk_id)
        text, text_len = self._extract_text_token(tts_text)
        ptext, ptext_len = self._extract_text_token(prompt_text)
        # frontend.py:497-498: Resample(16000, resample_rate) + feat_extractor, both on the device (cv2_resample, cv2_melspec)
        if self._prompt is None or self._prompt_rate != resample_rate:
            from cv2amd.prompt import PromptFeatures
            self._prompt, self._prompt_rate = PromptFeatures(self._feature_device(), sr=resample_rate), resample_rate
        feat = self._prompt.prompt_feat(prompt_speech_16k).cpu()
        tok, tok_len = self._extract_speech_token(prompt_speech_16k)
        n = min(int(feat.shape[1] / 2), tok.shape[1])                            # frontend.py:498-502: force feat = 2 x token
        feat, tok = feat[:, :2 * n], tok[:, :n]
        emb = self._extract_spk_embedding(prompt_speech_16k)
        return {'text': text, 'text_len': text_len, 'prompt_text': ptext, 'prompt_text_len': ptext_len,
                'llm_prompt_speech_token': tok, 'llm_prompt_speech_token_len': torch.tensor([n], dtype=torch.int32),
                'flow_prompt_speech_token': tok, 'flow_prompt_speech_token_len': torch.tensor([n], dtype=torch.int32),
                'prompt_speech_feat': feat, 'prompt_speech_feat_len': torch.tensor([2 * n], dtype=torch.int32),
                'llm_embedding': emb, 'flow_embedding': emb}
