"""Deterministic, name-keyed fills shared by the golden generator (reference side) and the tests (oracle / HIP side).

Values depend only on (tag, key, shape) through numpy's RandomState seeded with a CRC of the name, so the
reference's modules, the oracle's state dicts and riders_amd's modules all receive identical weights without
committing them.
"""
import zlib

import numpy as np
import torch


def _rs(name):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)


def rand_array(name, shape, scale=1.0, lo=-1.0):
    """uniform [lo, 1) * scale, float32."""
    a = _rs(name).uniform(lo, 1.0, size=tuple(int(s) for s in shape)).astype(np.float32)
    return a * np.float32(scale)


def fill_value(tag, key, shape):
    name = tag + "/" + key
    leaf = key.split(".")[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_mean":
        return rand_array(name, shape, 0.1)
    if leaf == "running_var":
        return rand_array(name, shape, 0.5) + np.float32(1.0)
    is_norm = "batch_norm" in key or "norm1" in key or "norm2" in key or ".bn" in key
    if leaf == "weight" and is_norm:
        return rand_array(name, shape, 0.5) + np.float32(1.0)
    if leaf == "bias":
        return rand_array(name, shape, 0.2)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return rand_array(name, shape, 1.7 / np.sqrt(max(fan_in, 1)))


def fill_state_dict(module, tag):
    """Overwrite every parameter and buffer of `module` in place; returns the (name -> tensor) dict used."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        kk = k[7:] if k.startswith("module.") else k
        new[k] = torch.from_numpy(fill_value(tag, kk, tuple(v.shape))).to(v.dtype)
    module.load_state_dict(new)
    return new


def make_state_dict(keys_shapes, tag):
    """{key: shape} -> {key: tensor} with the same values fill_state_dict would write."""
    return {k: torch.from_numpy(fill_value(tag, k, tuple(s))) for k, s in keys_shapes.items()}
