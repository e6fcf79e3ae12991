"""Wire formats of the reference's serving shells, dependency-free.

* PCM: every yielded chunk is sent as little-endian int16 = float waveform * 2**15 (runtime/python/fastapi/server.py:40-43,
  grpc/server.py:69-72).
* gRPC: service `cosyvoice.CosyVoice`, rpc `Inference(Request) returns (stream Response)` (runtime/python/grpc/cosyvoice.proto:8-42).
  The four request kinds are a oneof of small messages holding strings and one bytes field; the protobuf encoding of those is
  restated here (varint keys, length-delimited fields), so the server needs no generated *_pb2 module (grpc_tools is not part of
  the image).  tests/test_host_cpu.py checks this codec against google.protobuf on the same schema.
"""
import numpy as np

KINDS = {1: ('sft_request', {1: 'spk_id', 2: 'tts_text'}),
         2: ('zero_shot_request', {1: 'tts_text', 2: 'prompt_text', 3: 'prompt_audio'}),
         3: ('cross_lingual_request', {1: 'tts_text', 2: 'prompt_audio'}),
         4: ('instruct_request', {1: 'tts_text', 2: 'spk_id', 3: 'instruct_text'})}
BYTES_FIELDS = {'prompt_audio'}


def pcm16(wave):
    """float tensor / array [1, n] in [-1, 1) -> int16 little-endian bytes (the reference's `(x * 2**15).astype(np.int16)`)."""
    a = wave.numpy() if hasattr(wave, 'numpy') else np.asarray(wave)
    return (a * (2 ** 15)).astype(np.int16).tobytes()


def pcm16_to_float(buf):
    """grpc/server.py:49-50: int16 bytes -> float32 [1, n] / 2**15."""
    return (np.frombuffer(buf, dtype=np.int16).astype(np.float32) / (2 ** 15))[None, :]


def _varint(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, i):
    shift, n = 0, 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, i
        shift += 7


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        field, wt = key >> 3, key & 7
        if wt == 2:
            n, i = _read_varint(buf, i)
            yield field, bytes(buf[i:i + n])
            i += n
        elif wt == 0:
            _, i = _read_varint(buf, i)
        elif wt == 1:
            i += 8
        elif wt == 5:
            i += 4
        else:
            raise ValueError('unsupported protobuf wire type {}'.format(wt))


def encode_request(kind, **fields):
    """kind in ('sft_request', 'zero_shot_request', 'cross_lingual_request', 'instruct_request')."""
    num = next(k for k, (name, _) in KINDS.items() if name == kind)
    names = {v: k for k, v in KINDS[num][1].items()}
    inner = b''
    for name in sorted(fields, key=lambda n: names[n]):
        v = fields[name]
        v = v if name in BYTES_FIELDS else v.encode('utf-8')
        if len(v):                                         # proto3 omits default (empty) values
            inner += _ld(names[name], v)
    return _ld(num, inner)


def decode_request(buf):
    """-> (kind, dict of fields); absent strings / bytes default to empty, as proto3 does.  The last oneof member on the wire wins."""
    kind, out = None, {}
    for field, payload in _fields(buf):
        if field in KINDS:
            kind, names = KINDS[field]
            out = {n: (b'' if n in BYTES_FIELDS else '') for n in names.values()}
            for f, p in _fields(payload):
                if f in names:
                    out[names[f]] = p if names[f] in BYTES_FIELDS else p.decode('utf-8')
    return kind, out


def encode_response(tts_audio):
    return _ld(1, tts_audio) if len(tts_audio) else b''


def decode_response(buf):
    audio = b''
    for field, payload in _fields(buf):
        if field == 1:
            audio = payload
    return audio
