"""hope_amd -- MI355X-native batched parking simulator for HOPE (env-step hot path only).

Public surface:
  hope_amd.ParkingBatch              N parallel scenes on one GPU (tensor in / tensor out)
  hope_amd.CarParking / CarParkingWrapper   N=1 look-alikes of the reference env classes
  hope_amd.scenes                    DLP scene pool + Normal/Complex/Extrem generator (host side)
  hope_amd.tables                    ActionMask / lidar tables (host side, numpy)
"""
from .build import build_extension, lib_path  # noqa: F401
from ._lib import load_library, HopeError  # noqa: F401


def __getattr__(name):   # lazy: torch is only needed for the device-side classes
    if name == 'ParkingBatch':
        from .batch_env import ParkingBatch
        return ParkingBatch
    if name in ('CarParking', 'CarParkingWrapper', 'Status'):
        from . import env
        return getattr(env, name)
    raise AttributeError(name)
