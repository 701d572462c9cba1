"""Deterministic network weights shared by tests/golden/make_golden_r2.py (which fills the REFERENCE's networks with
them) and tests/test_agents.py (which fills hope_amd.policy's): fixtures then only need to hold inputs and outputs."""
import math

import torch


def det_fill(module_or_state_dict, salt=0.0):
    """every tensor t (in state_dict order, index j) becomes  s * sin(0.37 * i + 1.7 * j + salt)  over its flat index i,
    with s = 1.2 / sqrt(fan_in) for >= 2-D tensors, 0.05 for biases; LayerNorm-style 1-D 'weight' tensors get 1 + that."""
    sd = module_or_state_dict if isinstance(module_or_state_dict, dict) else module_or_state_dict.state_dict()
    with torch.no_grad():
        for j, (name, t) in enumerate(sd.items()):
            i = torch.arange(t.numel(), dtype=torch.float64)
            w = torch.sin(0.37 * i + 1.7 * j + salt)
            if t.dim() >= 2:
                fan_in = t[0].numel()
                w = w * (1.2 / math.sqrt(fan_in))
            else:
                w = w * 0.05
                if name.endswith('norm.weight'):
                    w = w + 1.0
            t.copy_(w.view(t.shape).to(t.dtype))
    return module_or_state_dict


def probe(module, k=6):
    """a small fingerprint of every parameter tensor: [sum, abs-sum, first k flat entries] in float64"""
    out = []
    for _, p in module.named_parameters():
        f = p.detach().double().flatten()
        head = torch.zeros(k, dtype=torch.float64)
        head[:min(k, f.numel())] = f[:k]
        out.append(torch.cat([torch.stack([f.sum(), f.abs().sum()]), head]))
    return torch.stack(out)
