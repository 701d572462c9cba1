"""ctypes binding of include/hope_env.h.  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

from .build import lib_path

# mirror of include/hope_env.h
LIDAR_NUM, N_ACTION, N_ITER, UPSAMPLE, TARGET_DIM, RS_MAX_SEG = 120, 42, 10, 10, 5, 5
F_OBS_F64, F_ACTION_F64, F_PROFILE, F_IMAGE, F_OVERLAP = 0x1, 0x2, 0x4, 0x8, 0x10
STAGE_MOTION, STAGE_OBS, STAGE_REWARD, STAGE_RS, STAGE_ALL = 0x1, 0x2, 0x4, 0x8, 0xF
ACTION_PHYSICAL = 0x10
ACTION_RESCALE_F32 = 0x80
AUTO_REDRAW = 0x100
DEFER_RS = 0x200
STAGE_IMG = 0x40
IMG_SIZE, IMG_CHANNELS, TRAJ_RENDER_LEN = 64, 3, 20
AUTO_RESET = 0x20
KERNELS = ('k_kinematics', 'k_env_step', 'k_rs_words', 'k_rs_validate', 'k_bev_image', 'k_bev_prep', 'k_rs_compact', 'k_post', 'k_rs_segs', 'k_rs_screen')
ABI_VERSION = 8

EXPORTS = ['hope_env_create', 'hope_env_destroy', 'hope_last_error', 'hope_abi_version', 'hope_env_upload_tables',
           'hope_env_set_scenes', 'hope_env_step', 'hope_env_wait_rs', 'hope_env_last_step', 'hope_env_wait_rs_step', 'hope_env_download_n_obst', 'hope_env_queue_check', 'hope_env_reset_obs', 'hope_env_download_state',
           'hope_env_upload_state', 'hope_env_restart', 'hope_env_set_pool', 'hope_env_pool_staging', 'hope_env_commit_pool', 'hope_env_commit_pool_relaxed', 'hope_env_pool_staging_ready', 'hope_env_pool_generation', 'hope_env_redraw', 'hope_env_set_redraw_seed', 'hope_env_download_pool_index', 'hope_env_set_dlp_cases', 'hope_env_pool_overflow', 'hope_env_set_draw_class', 'hope_env_download_scenes', 'hope_env_download_pool_state', 'hope_env_restore_maps', 'hope_env_kernel_ms', 'hope_env_kernel_union_ms', 'hope_env_profile_kernels', 'hope_debug_math', 'hope_debug_traffic', 'hope_debug_mask_lut', 'hope_debug_rs_prof', 'hope_debug_rs_log', 'hope_debug_rs_filter_stats', 'hope_debug_rs_filter_dump', 'hope_debug_step_prof', 'hope_debug_census', 'hope_scenegen_generate', 'hope_scenegen_default_threads', 'hope_env_num_scenes', 'hope_env_max_obstacles', 'hope_env_device_arch']


class HopeError(RuntimeError):
    pass


class StepOut(C.Structure):
    _fields_ = [('lidar', C.c_void_p), ('action_mask', C.c_void_p), ('target', C.c_void_p), ('reward', C.c_void_p),
                ('reward_info', C.c_void_p), ('status', C.c_void_p), ('done', C.c_void_p), ('pose', C.c_void_p),
                ('rs_word', C.c_void_p), ('rs_lengths', C.c_void_p), ('img', C.c_void_p)]


_lib = None


def load_library():
    """dlopen hope_amd/libhope_env.so (built by hope_amd.build.build_extension / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64.  Import it FIRST so that this library
    # binds to the HIP runtime torch already initialised (one runtime per process); loading in the other
    # order leaves the process with two runtimes and "no ROCm-capable device" from the second one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = lib_path()
    if not os.path.exists(path):
        raise HopeError(f'{path} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    L = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise HopeError(f'{path} does not export {name}')
    L.hope_last_error.restype = C.c_char_p
    L.hope_env_device_arch.restype = C.c_char_p
    L.hope_env_device_arch.argtypes = [C.c_void_p]
    L.hope_env_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_uint32]
    L.hope_env_destroy.argtypes = [C.c_void_p]
    L.hope_env_upload_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_set_scenes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
    L.hope_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(StepOut), C.c_void_p]
    L.hope_env_wait_rs.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_last_step.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_wait_rs_step.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.hope_env_download_n_obst.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_queue_check.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_reset_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(StepOut), C.c_void_p]
    L.hope_env_download_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_upload_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_restart.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_set_pool.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_pool_staging.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_void_p)] * 5
    L.hope_env_commit_pool.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.hope_env_commit_pool_relaxed.argtypes = [C.c_void_p, C.c_int]
    L.hope_env_redraw.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.hope_env_set_redraw_seed.argtypes = [C.c_void_p, C.c_uint64]
    L.hope_env_download_pool_index.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_set_dlp_cases.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.hope_env_set_draw_class.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.hope_env_pool_overflow.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_download_scenes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_download_pool_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_env_restore_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
    L.hope_env_pool_staging_ready.argtypes = [C.c_void_p]
    L.hope_env_pool_generation.argtypes = [C.c_void_p, C.c_void_p]
    L.hope_env_kernel_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.hope_env_kernel_union_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.hope_env_profile_kernels.argtypes = [C.c_void_p, C.c_uint32]
    L.hope_debug_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_debug_traffic.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p]
    L.hope_debug_mask_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.hope_scenegen_generate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.hope_env_num_scenes.argtypes = [C.c_void_p]
    L.hope_env_max_obstacles.argtypes = [C.c_void_p]
    if L.hope_abi_version() != ABI_VERSION:
        raise HopeError(f'ABI mismatch: library {L.hope_abi_version()} vs binding {ABI_VERSION}')
    _lib = L
    return L


def check(rc, what=''):
    if rc != 0:
        msg = load_library().hope_last_error().decode(errors='replace')
        raise HopeError(f'{what} failed (code {rc}): {msg}')
