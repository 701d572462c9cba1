"""Reads the checkpoints the reference's agents write without the reference on the path.

SACAgent.save / PPOAgent.save (src/model/agent/sac_agent.py:339-358, ppo_agent.py) torch.save a dict:
    <net name>: state_dict ...,  'log': log_std,  'state_norm': model.state_norm.StateNorm instance,
    'optimizer': tuple of torch optimizers
The StateNorm instance (and, for full-object saves, config classes) is pickled by module path, so a plain
torch.load needs `model.state_norm` importable.  load_hope_checkpoint maps those paths onto local stand-ins and
returns tensors plus a BatchedStateNorm initialised from the stored running statistics.
"""
import pickle

import numpy as np
import torch

from .agent_glue import BatchedStateNorm


class RefStateNorm:
    """attribute bag for a pickled model.state_norm.StateNorm (state_norm.py:7-20)"""
    observation_shape = None
    update_modal = None
    n_state = 0
    fixed = False


class _Opaque:
    """any other reference-side object stored in a checkpoint (configs, agents): attributes only"""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'state': state})


# Globals a HOPE checkpoint legitimately refers to besides the reference's own classes: tensor / storage rebuilders, the
# optimizer classes of the 'optimizer' tuple, containers and numpy scalars / arrays -- an EXPLICIT list of names.  Anything
# else is refused, so that loading an untrusted file cannot import and call arbitrary code (weights_only=False is needed for
# the StateNorm instance).  No module prefixes: e.g. torch._utils._import_dotted_name returns any importable callable and
# torch.storage._load_from_bytes is an unrestricted torch.load, both reachable through a whole-module allowance.
_ALLOWED = {
    'collections.OrderedDict', 'collections.defaultdict', 'collections.deque', 'builtins.dict', 'builtins.list',
    'builtins.tuple', 'builtins.set', 'builtins.frozenset', 'builtins.int', 'builtins.float', 'builtins.bool', 'builtins.str',
    'builtins.complex', 'builtins.slice', 'builtins.range', 'builtins.bytes', 'builtins.bytearray', '_codecs.encode',
    # tensors and storages
    'torch._utils._rebuild_tensor_v2', 'torch._utils._rebuild_tensor', 'torch._utils._rebuild_parameter',
    'torch._utils._rebuild_parameter_with_state', 'torch._tensor._rebuild_from_type_v2', 'torch.nn.parameter.Parameter',
    'torch.storage.UntypedStorage', 'torch.storage.TypedStorage', 'torch.Tensor', 'torch.Size', 'torch.device', 'torch.dtype',
    'torch.serialization._get_layout', 'torch.FloatStorage', 'torch.DoubleStorage', 'torch.LongStorage', 'torch.IntStorage',
    'torch.HalfStorage', 'torch.BoolStorage', 'torch.ByteStorage', 'torch.BFloat16Storage', 'torch.float32', 'torch.float64',
    'torch.int64', 'torch.int32', 'torch.uint8',
    # optimizer objects (sac_agent.py:339-355 / ppo_agent.py:351-371 store them whole)
    'torch.optim.adam.Adam', 'torch.optim.adamw.AdamW', 'torch.optim.sgd.SGD', 'torch.optim.rmsprop.RMSprop',
    # numpy arrays / scalars inside the StateNorm instance
    'numpy.dtype', 'numpy.ndarray', 'numpy.core.multiarray._reconstruct', 'numpy.core.multiarray.scalar',
    'numpy._core.multiarray._reconstruct', 'numpy._core.multiarray.scalar', 'numpy.core.numeric._frombuffer',
    'numpy._core.numeric._frombuffer'}


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod == 'model.state_norm' and name == 'StateNorm':
            return RefStateNorm
        if mod.split('.')[0] in ('model', 'configs', 'env', 'evaluation', 'train'):
            return type(name, (_Opaque,), {'__module__': mod})
        full = f'{mod}.{name}'.replace('__builtin__.', 'builtins.')       # protocol-2 pickles name the py2 module
        if full in _ALLOWED:
            return super().find_class(mod, name)
        raise pickle.UnpicklingError(f'hope_amd.checkpoint: refusing to load global {full!r} from a checkpoint')


class _shim_pickle:
    """the slice of the pickle module torch.load uses"""
    __name__ = 'pickle'
    Unpickler = _Unpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def state_norm_from_ref(ref, device='cpu'):
    """model.state_norm.StateNorm (running mean / S / std per modality, n_state, fixed) -> BatchedStateNorm"""
    mean, S, std = getattr(ref, 'state_mean', None), getattr(ref, 'S', None), getattr(ref, 'state_std', None)
    if not mean or S is None or std is None:               # older saves without running statistics
        return None
    modal = tuple(k for k, v in (ref.update_modal or {}).items() if v and k in mean)
    shapes = {k: int(np.prod(np.shape(ref.state_mean[k]))) for k in modal}
    sn = BatchedStateNorm(shapes=shapes, update_modal=modal, device=device)
    for k in modal:
        sn.mean[k] = torch.as_tensor(np.asarray(ref.state_mean[k], dtype=np.float64), device=device)
        sn.S[k] = torch.as_tensor(np.asarray(ref.S[k], dtype=np.float64), device=device)
        sn.std[k] = torch.as_tensor(np.asarray(ref.state_std[k], dtype=np.float64), device=device)
    sn.n_state = int(ref.n_state)
    sn.fixed = bool(getattr(ref, 'fixed', False))
    return sn


def load_hope_checkpoint(path, device='cpu'):
    """-> {'state_dicts': {name: state_dict}, 'log_std': tensor | None, 'state_norm': BatchedStateNorm | None,
           'raw': the unpickled dict}"""
    raw = torch.load(path, map_location=device, pickle_module=_shim_pickle, weights_only=False)
    if not isinstance(raw, dict):
        raise ValueError('expected the params-only checkpoint format (a dict); got ' + type(raw).__name__)
    out = {'state_dicts': {}, 'log_std': None, 'state_norm': None, 'raw': raw}
    for k, v in raw.items():
        if k == 'log':
            out['log_std'] = v.detach() if torch.is_tensor(v) else v
        elif k == 'state_norm':
            out['state_norm'] = state_norm_from_ref(v, device) if isinstance(v, RefStateNorm) else None
        elif isinstance(v, dict) and all(torch.is_tensor(t) for t in v.values()):
            out['state_dicts'][k] = v
    return out
