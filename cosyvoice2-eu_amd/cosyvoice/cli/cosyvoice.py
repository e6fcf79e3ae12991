"""`CosyVoice2` — the API surface of cosyvoice/cli/cosyvoice.py:144-294 kept for callers (evaluation pipeline,
run_inference.py, cosyvoice2_eu): same constructor arguments, same generator methods yielding
{'tts_speech': float32 CPU tensor [1, n]}, `.sample_rate`, `add_zero_shot_spk`.  The body is new: checkpoints are read
straight from `.pt` files (no hyperpyyaml object graph; the architecture is fixed) into the HIP
engines.
"""
import logging
import os
import time

import torch

from cosyvoice.cli.frontend import CosyVoiceFrontEnd, FrontEndUnavailable, PrecomputedFrontEnd
from cosyvoice.cli.model import CosyVoice2Model


class CosyVoice2:
    def __init__(self, model_dir, load_jit=False, load_trt=False, load_vllm=False, fp16=False, trt_concurrent=1, setting='original',
                 llm_run_id=None, flow_run_id=None, hifigan_run_id=None, final=False, backbone=None, frontend=None, state_dicts=None):
        """model_dir: directory with llm/flow/hift checkpoints (+ tokenizer and ONNX frontend models).
        Extra keyword-only hooks of this build: `frontend` (object with the CosyVoiceFrontEnd interface) and `state_dicts`
        (llm, flow, hift) to construct the model without files (synthetic weights, tests)."""
        self.model_dir = model_dir
        self.fp16 = fp16
        self.sample_rate = 24000                                            # conf/cosyvoice2.yaml:8
        if load_jit or load_trt or load_vllm:
            logging.warning('load_jit / load_trt / load_vllm are NVIDIA-path accelerators of the reference; ignored on MI355X')
        if state_dicts is None and not os.path.exists(model_dir):
            raise ValueError('model_dir {} does not exist'.format(model_dir))
        # cosyvoice2.yaml (cli/cosyvoice.py:176-226): required next to the checkpoints, read for the run-time hyper-parameters and
        # checked against the fixed architecture; `backbone` selects qwen_pretrain_path exactly as the reference does
        from cv2amd.config import Config
        config = Config()
        if state_dicts is None:
            hyper_yaml_path = '{}/cosyvoice2.yaml'.format(model_dir)
            if not os.path.exists(hyper_yaml_path):
                raise ValueError('{} not found!'.format(hyper_yaml_path))
            overrides, qwen = {}, None
            blank = os.path.join(model_dir, 'CosyVoice-BlankEN')
            if backbone is not None:
                if backbone == 'blanken':
                    qwen = blank if os.path.exists(blank) else None
                elif backbone.startswith('hf:'):
                    qwen = backbone[3:]
                elif backbone.startswith('local:'):
                    qwen = backbone[6:]
                else:
                    qwen = backbone
            elif os.path.exists(blank) and not os.environ.get('COSYV2_IGNORE_BLANKEN'):
                qwen = blank
            if qwen:
                overrides['qwen_pretrain_path'] = qwen
                logging.info('Using LLM backbone: {}'.format(qwen))
            config = Config.from_file(hyper_yaml_path, overrides)
            self.sample_rate = config.sample_rate
        self.config = config
        # checkpoint selection, cli/cosyvoice.py:240-265
        if final:
            tokens = {'llm', 'flow', 'hifigan'}
        elif setting == 'original':
            tokens = set()
        else:
            tokens = set(setting.split('_'))
            invalid = tokens - {'llm', 'flow', 'hifigan'}
            if invalid:
                raise ValueError('setting should be one of "original", "llm", "flow", "hifigan", "llm_flow", "llm_hifigan", '
                                 '"flow_hifigan", "llm_flow_hifigan", but got {}'.format(setting))
        chosen = {}
        for key, run_id in (('llm', llm_run_id), ('flow', flow_run_id), ('hift', hifigan_run_id)):
            token = key if key != 'hift' else 'hifigan'
            if final or (token in tokens and run_id is not None):
                suffix = '' if final else '-{}'.format(run_id)
            else:
                suffix = '-original'
            chosen[key] = '{}/{}{}.pt'.format(model_dir, key, suffix)
        self.model = CosyVoice2Model(fp16=fp16, config=config)
        if state_dicts is not None:
            self.model.load_state_dicts(*state_dicts)
        else:
            print('Loading CosyVoice2 with\n\tLLM: {}\n\tFlow: {}\n\tHiFT: {}'.format(chosen['llm'], chosen['flow'], chosen['hift']))
            self.model.load(chosen['llm'], chosen['flow'], chosen['hift'])
        if frontend is not None:
            self.frontend = frontend
        else:
            try:
                self.frontend = CosyVoiceFrontEnd(model_dir)
            except FrontEndUnavailable as e:
                logging.warning('prompt frontend unavailable (%s); only pre-extracted prompts (add_zero_shot_spk / spk2info) will work', e)
                self.frontend = PrecomputedFrontEnd(lambda t: (_ for _ in ()).throw(FrontEndUnavailable('no tokenizer')))

    def list_available_spks(self):
        return list(self.frontend.spk2info.keys())

    def add_zero_shot_spk(self, prompt_text, prompt_speech_16k, zero_shot_spk_id):
        assert zero_shot_spk_id != '', 'do not use empty zero_shot_spk_id'
        model_input = self.frontend.frontend_zero_shot('', prompt_text, prompt_speech_16k, self.sample_rate, '')
        del model_input['text']
        del model_input['text_len']
        self.frontend.spk2info[zero_shot_spk_id] = model_input
        return True

    def save_spkinfo(self):
        torch.save(self.frontend.spk2info, '{}/spk2info.pt'.format(self.model_dir))

    def _run(self, model_input, stream, speed, label):
        start_time = time.time()
        logging.info('synthesis text {}'.format(label))
        for model_output in self.model.tts(**model_input, stream=stream, speed=speed):
            speech_len = model_output['tts_speech'].shape[1] / self.sample_rate
            logging.info('yield speech len {}, rtf {}'.format(speech_len, (time.time() - start_time) / speech_len))
            yield model_output
            start_time = time.time()

    def inference_sft(self, tts_text, spk_id, stream=False, speed=1.0, text_frontend=True):
        """cli/cosyvoice.py:81-90 (inherited by CosyVoice2): a speaker of spk2info by id, no prompt.  A CosyVoice2 model dir has no
        spk2info.pt, so an unknown `spk_id` raises the reference's KeyError (frontend.py:487)."""
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_sft(i, spk_id)
            yield from self._run(model_input, stream, speed, i)

    def inference_zero_shot(self, tts_text, prompt_text, prompt_speech_16k, zero_shot_spk_id='', stream=False, speed=1.0, text_frontend=True):
        prompt_text = self.frontend.text_normalize(prompt_text, split=False, text_frontend=text_frontend)
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            if isinstance(i, str) and len(i) < 0.5 * len(prompt_text):
                logging.warning('synthesis text {} too short than prompt text {}, this may lead to bad performance'.format(i, prompt_text))
            model_input = self.frontend.frontend_zero_shot(i, prompt_text, prompt_speech_16k, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, stream, speed, i)

    def inference_cross_lingual(self, tts_text, prompt_speech_16k, zero_shot_spk_id='', stream=False, speed=1.0, text_frontend=True):
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_cross_lingual(i, prompt_speech_16k, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, stream, speed, i)

    def inference_vc(self, source_speech_16k, prompt_speech_16k, stream=False, speed=1.0):
        """cli/cosyvoice.py:132-139: voice conversion = the source utterance's speech tokens through flow + HiFT with the prompt's voice."""
        model_input = self.frontend.frontend_vc(source_speech_16k, prompt_speech_16k, self.sample_rate)
        yield from self._run(model_input, stream, speed, 'vc')

    def inference_instruct(self, *args, **kwargs):
        raise NotImplementedError('inference_instruct is not implemented for CosyVoice2!')

    def inference_instruct2(self, tts_text, instruct_text, prompt_speech_16k, zero_shot_spk_id='', stream=False, speed=1.0, text_frontend=True):
        """cli/cosyvoice.py:284-295: same three stages, the instruction rides in the prompt-text slot."""
        for i in self.frontend.text_normalize(tts_text, split=True, text_frontend=text_frontend):
            model_input = self.frontend.frontend_instruct2(i, instruct_text, prompt_speech_16k, self.sample_rate, zero_shot_spk_id)
            yield from self._run(model_input, stream, speed, i)
