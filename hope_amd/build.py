"""Builds hope_amd/libhope_env.so for gfx950 with hipcc (in-tree, so the .so travels to the GPU box)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
_ROOT = os.path.dirname(_HERE)
SOURCES = ['hope_env.hip', 'hope_rs.hip', 'hope_bev.hip', 'hope_scenegen.cpp']
HEADERS = ['hope_math.h', 'hope_dev.h', 'hope_internal.h', 'hope_step_kernel.h', 'hope_obs_pair.h', 'hope_motion_pair.h', os.path.join(_ROOT, 'include', 'hope_env.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared',
         '-Wno-unused-value', '-pthread']


def lib_path():
    # HOPE_AMD_LIB: another build of the same ABI (A/B timing of two kernel versions in one gpurun call)
    return os.environ.get('HOPE_AMD_LIB') or os.path.join(_HERE, 'libhope_env.so')


def _stale(out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(_CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_extension(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU present.  Returns the path of the .so."""
    out = lib_path()
    if not force and not _stale(out):
        return out
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    extra = os.environ.get('HOPE_BUILD_DEFS', '').split()          # A/B builds: HOPE_BUILD_DEFS="-DHOPE_MASK_MG=8" HOPE_AMD_LIB=...
    cmd = [hipcc] + FLAGS + extra + ['-I' + os.path.join(_ROOT, 'include'), '-I' + _CSRC] + \
          [os.path.join(_CSRC, s) for s in SOURCES] + ['-o', out]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return out
