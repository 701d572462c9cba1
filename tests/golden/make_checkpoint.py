#!/usr/bin/env python3
"""Writes tests/golden/ref_checkpoint.pt in the reference's params-only checkpoint format (sac_agent.py:339-358):
a dict of state_dicts + 'log' + 'state_norm' + 'optimizer', where 'state_norm' is an instance of the REFERENCE's own
class model.state_norm.StateNorm (numpy only, importable here) fed with seeded observations.  The networks are small
stand-ins: the loader is about the container format and the pickled class path, not the architecture.
Run from the repo root:  python tests/golden/make_checkpoint.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, '/root/reference/src')
from model.state_norm import StateNorm   # noqa: E402  (the reference's class, so the pickle carries its module path)


def main():
    rng = np.random.default_rng(0)
    shapes = {'lidar': (120,), 'target': (5,), 'action_mask': (42,)}
    sn = StateNorm(shapes)
    obs_log = []
    for _ in range(50):
        obs = {'lidar': rng.uniform(0, 10, 120).astype(np.float32), 'target': rng.normal(size=5).astype(np.float32),
               'action_mask': rng.uniform(size=42).astype(np.float32), 'img': None}
        obs_log.append({k: (v.copy() if v is not None else None) for k, v in obs.items()})
        sn.state_norm(obs, update=True)
    torch.manual_seed(0)
    actor, critic = torch.nn.Linear(8, 4), torch.nn.Linear(8, 1)
    opt = (torch.optim.Adam(actor.parameters()), torch.optim.Adam(critic.parameters()), torch.optim.Adam(critic.parameters()))
    ckpt = {'actor_net': actor.state_dict(), 'critic_net1': critic.state_dict(), 'log': torch.zeros(1, 2) - 0.5,
            'state_norm': sn, 'optimizer': opt}
    torch.save(ckpt, os.path.join(HERE, 'ref_checkpoint.pt'))
    np.savez(os.path.join(HERE, 'ref_checkpoint_expect.npz'), mean_lidar=sn.state_mean['lidar'], std_lidar=sn.state_std['lidar'],
             mean_target=sn.state_mean['target'], std_target=sn.state_std['target'], n_state=sn.n_state,
             actor_weight=actor.state_dict()['weight'].numpy())
    print('wrote ref_checkpoint.pt', os.path.getsize(os.path.join(HERE, 'ref_checkpoint.pt')), 'bytes')


if __name__ == '__main__':
    main()
