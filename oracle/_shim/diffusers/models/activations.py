import torch


def get_activation(name):
    assert name in ('swish', 'silu')
    return torch.nn.SiLU()
