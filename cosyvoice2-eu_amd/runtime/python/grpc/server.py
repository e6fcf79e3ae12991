"""gRPC server of the reference (runtime/python/grpc/server.py:33-76) over the MI355X CosyVoice2: rpc Inference(Request) -> stream of
Response{tts_audio = int16 PCM}.  No generated stubs: the messages are (de)serialised by runtime/python/wire.py."""
import argparse
import logging
import os
import sys
from concurrent import futures

ROOT_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT_DIR, '..'))
sys.path.insert(0, os.path.join(ROOT_DIR, '..', '..', '..'))
import wire  # noqa: E402


class CosyVoiceServiceImpl:
    def __init__(self, cosyvoice):
        self.cosyvoice = cosyvoice

    def Inference(self, request, context):
        import torch
        kind, f = request
        if kind == 'sft_request':
            out = self.cosyvoice.inference_sft(f['tts_text'], f['spk_id'])
        elif kind == 'zero_shot_request':
            out = self.cosyvoice.inference_zero_shot(f['tts_text'], f['prompt_text'], torch.from_numpy(wire.pcm16_to_float(f['prompt_audio']).copy()))
        elif kind == 'cross_lingual_request':
            out = self.cosyvoice.inference_cross_lingual(f['tts_text'], torch.from_numpy(wire.pcm16_to_float(f['prompt_audio']).copy()))
        else:
            out = self.cosyvoice.inference_instruct(f['tts_text'], f['spk_id'], f['instruct_text'])
        for i in out:
            yield wire.pcm16(i['tts_speech'])


def make_server(cosyvoice, port, max_conc=4):
    import grpc
    impl = CosyVoiceServiceImpl(cosyvoice)
    handler = grpc.method_handlers_generic_handler('cosyvoice.CosyVoice', {
        'Inference': grpc.unary_stream_rpc_method_handler(impl.Inference, request_deserializer=wire.decode_request,
                                                          response_serializer=wire.encode_response)})
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_conc), maximum_concurrent_rpcs=max_conc)
    server.add_generic_rpc_handlers((handler,))
    bound = server.add_insecure_port('0.0.0.0:{}'.format(port))
    return server, bound


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--port', type=int, default=50000)
    ap.add_argument('--max_conc', type=int, default=4)
    ap.add_argument('--model_dir', type=str, required=True)
    args = ap.parse_args()
    logging.basicConfig(level=logging.DEBUG, format='%(asctime)s %(levelname)s %(message)s')
    from cosyvoice.cli.cosyvoice import CosyVoice2
    srv, _ = make_server(CosyVoice2(args.model_dir, final=True), args.port, args.max_conc)
    srv.start()
    logging.info('server listening on 0.0.0.0:{}'.format(args.port))
    srv.wait_for_termination()
