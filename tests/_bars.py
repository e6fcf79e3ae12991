"""Every tolerance of the GPU parity tests goes through bar(): the measured value is recorded beside its limit
(gpurun_out/bars.jsonl on the GPU box), so that a bar can be set from what is measured plus a stated margin and a
loosened bar shows up in the record."""
import json
import os

_OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')


def bar(name, value, limit, msg=''):
    value = float(value)
    try:
        os.makedirs(_OUT, exist_ok=True)
        with open(os.path.join(_OUT, 'bars.jsonl'), 'a') as f:
            f.write(json.dumps({'bar': name, 'value': value, 'limit': limit}) + '\n')
    except OSError:
        pass
    assert value < limit, f'{name}: {value:.3e} >= {limit:.3e} {msg}'
