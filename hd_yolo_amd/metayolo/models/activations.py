"""Activation factory (reference: metayolo/models/activations.py:106-126).  Only SiLU and Identity have HIP
kernels on the hot path; the other names resolve to torch modules so configs still parse, and the planner
rejects them loudly if they are actually reached."""
import torch.nn as nn

_BY_NAME = {'relu': nn.ReLU, 'gelu': nn.GELU, 'glu': nn.GLU, 'silu': nn.SiLU, 'mish': nn.Mish}


def _get_activation_fn(activation):
    if activation is True:
        return nn.SiLU()
    if not activation:
        return nn.Identity()
    if isinstance(activation, nn.Module):
        return activation
    if isinstance(activation, str) and activation in _BY_NAME:
        return _BY_NAME[activation]()
    raise RuntimeError(f"activation: {activation} is not registered.")
