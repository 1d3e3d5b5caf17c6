"""Key-addressed deterministic weights, shared by make_golden.py (reference models on the CPU
oracle) and the tests (the build's own models on the HIP path).

The 21.8 M parameters cannot be committed as a fixture, and seeding torch's global RNG would
tie the fixture to module construction order.  Instead every state_dict entry is filled from
a generator seeded by crc32(key): any model exposing the reference's state_dict keys and
shapes (network/spvcnn.py, network/minkunet.py) gets bit-identical weights.
"""
import zlib

import torch


def fill_state_dict(model, seed=7122):
    sd = model.state_dict()
    out = {}
    for key in sorted(sd.keys()):
        t = sd[key]
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + seed) % (2 ** 31))
        if not t.is_floating_point():
            out[key] = torch.zeros_like(t)          # num_batches_tracked
            continue
        shape = tuple(t.shape)
        r = torch.rand(shape, generator=g, dtype=torch.float32)
        if key.endswith('running_var'):
            v = 0.5 + r
        elif key.endswith('running_mean'):
            v = (r - 0.5) * 0.2
        elif key.endswith('.kernel'):
            fan = shape[-2] * (shape[0] if len(shape) == 3 else 1)
            v = (r * 2 - 1) * (3.0 / fan) ** 0.5
        elif key.endswith('weight') and len(shape) == 2:      # nn.Linear
            v = (r * 2 - 1) * (3.0 / shape[1]) ** 0.5
        elif key.endswith('weight'):                          # BatchNorm gamma
            v = 0.75 + 0.5 * r
        else:                                                 # biases / BatchNorm beta
            v = (r - 0.5) * 0.2
        out[key] = v.to(t.dtype)
    model.load_state_dict(out, strict=True)
    return model


def state_dict_signature(model):
    return [(k, tuple(v.shape), str(v.dtype).replace('torch.', ''))
            for k, v in model.state_dict().items()]
