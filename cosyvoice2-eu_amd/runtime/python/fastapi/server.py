"""HTTP server of the reference (runtime/python/fastapi/server.py:40-81) over the MI355X CosyVoice2: every endpoint streams the yielded
chunks as int16 PCM.  Form / file uploads need python-multipart (as in the reference); `build_app` raises a clear error without it."""
import argparse
import os
import sys

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT_DIR, '..'))
sys.path.insert(0, os.path.join(ROOT_DIR, '..', '..', '..'))
import wire  # noqa: E402


def generate_data(model_output):
    for i in model_output:
        yield wire.pcm16(i['tts_speech'])


def build_app(cosyvoice):
    from fastapi import FastAPI, UploadFile, Form, File
    from fastapi.responses import StreamingResponse
    from fastapi.middleware.cors import CORSMiddleware
    from cosyvoice.utils.file_utils import load_wav
    app = FastAPI()
    app.add_middleware(CORSMiddleware, allow_origins=['*'], allow_credentials=True, allow_methods=['*'], allow_headers=['*'])

    @app.get('/inference_zero_shot')
    @app.post('/inference_zero_shot')
    async def inference_zero_shot(tts_text: str = Form(), prompt_text: str = Form(), prompt_wav: UploadFile = File()):
        return StreamingResponse(generate_data(cosyvoice.inference_zero_shot(tts_text, prompt_text, load_wav(prompt_wav.file, 16000))))

    @app.get('/inference_cross_lingual')
    @app.post('/inference_cross_lingual')
    async def inference_cross_lingual(tts_text: str = Form(), prompt_wav: UploadFile = File()):
        return StreamingResponse(generate_data(cosyvoice.inference_cross_lingual(tts_text, load_wav(prompt_wav.file, 16000))))

    @app.get('/inference_instruct2')
    @app.post('/inference_instruct2')
    async def inference_instruct2(tts_text: str = Form(), instruct_text: str = Form(), prompt_wav: UploadFile = File()):
        return StreamingResponse(generate_data(cosyvoice.inference_instruct2(tts_text, instruct_text, load_wav(prompt_wav.file, 16000))))

    @app.get('/inference_sft')
    @app.post('/inference_sft')
    async def inference_sft(tts_text: str = Form(), spk_id: str = Form()):
        # CosyVoice2 has no SFT speakers of its own; a registered zero-shot speaker id plays that role (cli/cosyvoice.py:70-76)
        return StreamingResponse(generate_data(cosyvoice.inference_zero_shot(tts_text, '', None, zero_shot_spk_id=spk_id)))
    return app


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--port', type=int, default=50000)
    ap.add_argument('--model_dir', type=str, required=True)
    args = ap.parse_args()
    import uvicorn
    from cosyvoice.cli.cosyvoice import CosyVoice2
    uvicorn.run(build_app(CosyVoice2(args.model_dir, final=True)), host='0.0.0.0', port=args.port)
