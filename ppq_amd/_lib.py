"""ctypes loader for ``libppq_hip.so`` -- the C-ABI shared library declared in ``include/ppq_hip.h``.

The HIP library IS the product path: there is no CPU or PyTorch fallback anywhere in
``ppq_amd``.  If the library has not been built (``python __graft_entry__.py`` or
``make -C ppq_amd/csrc``) importing this module raises ImportError, loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPQHIP_LIBRARY: developer override used by tools/variants.sh to A/B kernel builds on the GPU box
LIB_PATH = os.environ.get('PPQHIP_LIBRARY') or os.path.join(_HERE, 'libppq_hip.so')

c_f32p = ctypes.c_void_p     # device pointers travel as integers (tensor.data_ptr())
c_i32p = ctypes.c_void_p
c_f64p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_flt = ctypes.c_float
c_vp = ctypes.c_void_p

ABI_VERSION = 4              # == PPQHIP_ABI_VERSION of include/ppq_hip.h; a library of any other version is refused below


class ProfEntry(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char * 48), ('launches', ctypes.c_int64),
                ('total_ms', ctypes.c_double), ('total_bytes', ctypes.c_double)]


# name -> (restype, argtypes); mirrors include/ppq_hip.h one to one
PROTOTYPES = {
    'ppqhip_last_error': (ctypes.c_char_p, []),
    'ppqhip_version': (c_int, []),
    'ppqhip_device_arch': (c_int, [ctypes.c_char_p, c_int]),
    'ppqhip_fq_linear_t': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_vp]),
    'ppqhip_fq_linear_c': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp]),
    'ppqhip_to_int_t': (c_int, [c_f32p, c_f32p, c_f32p, c_vp, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    'ppqhip_to_int_c': (c_int, [c_f32p, c_f32p, c_f32p, c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_vp]),
    'ppqhip_fq_linear_t_bwd': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int,
                                       c_vp]),
    'ppqhip_fq_linear_c_bwd': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int,
                                       c_int, c_int, c_vp]),
    'ppqhip_fq_linear_c_bwd_multi': (c_int, [c_vp, c_int, c_int, c_vp]),
    'ppqhip_fq_linear_t_bwd_partials': (c_i64, [c_i64]),
    'ppqhip_fq_linear_t_bwd_main': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_int, c_vp]),
    'ppqhip_lsq_finish_multi': (c_int, [c_vp, c_int, c_vp]),
    'ppqhip_fq_float_t': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_int, c_int, c_flt, c_flt, c_int, c_vp]),
    'ppqhip_fq_float_c': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int, c_int, c_flt, c_flt,
                                  c_int, c_vp]),
    'ppqhip_fq_float_c_bwd': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int,
                                      c_int, c_flt, c_flt, c_int, c_vp]),
    'ppqhip_hist_workspace_bytes': (c_i64, [c_i64, c_i64]),
    'ppqhip_hist_sym_t': (c_int, [c_f32p, c_i64, c_flt, c_int, c_i32p, c_i64, c_vp, c_vp]),
    'ppqhip_hist_asym_t': (c_int, [c_f32p, c_i64, c_flt, c_flt, c_int, c_i32p, c_i64, c_vp, c_vp]),
    'ppqhip_hist_sym_c': (c_int, [c_f32p, c_i64, c_i64, c_i64, c_flt, c_int, c_i32p, c_i64, c_vp]),
    'ppqhip_hist_sym_c_scales': (c_int, [c_f32p, c_i64, c_i64, c_i64, c_f32p, c_int, c_i32p, c_i64, c_vp]),
    'ppqhip_hist_asym_c_ranges': (c_int, [c_f32p, c_i64, c_i64, c_i64, c_f32p, c_f32p, c_int, c_i32p, c_i64, c_vp]),
    'ppqhip_quantile_workspace_bytes': (c_i64, [c_i64]),
    'ppqhip_quantile_t': (c_int, [c_f32p, c_i64, c_flt, c_f32p, c_vp, c_vp, c_vp]),
    'ppqhip_isotone_t': (c_int, [c_f32p, c_i64, c_f32p, c_vp, c_vp]),
    'ppqhip_minmax_workspace_bytes': (c_i64, [c_i64]),
    'ppqhip_minmax_t': (c_int, [c_f32p, c_i64, c_f32p, c_vp, c_vp]),
    'ppqhip_minmax_c': (c_int, [c_f32p, c_i64, c_i64, c_i64, c_f32p, c_f32p, c_vp]),
    'ppqhip_minmax_c_multi': (c_int, [c_vp, c_int, c_vp]),
    'ppqhip_fq_linear_multi_table_bytes': (c_i64, [c_int]),
    'ppqhip_fq_linear_multi': (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'ppqhip_float_scale_search_table_bytes': (c_i64, [c_int]),
    'ppqhip_float_scale_search': (c_int, [c_vp, c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    'ppqhip_fq_float_multi_table_bytes': (c_i64, [c_int]),
    'ppqhip_fq_float_multi': (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_vp]),
    'ppqhip_quantile_multi_workspace_bytes': (c_i64, [c_int, c_i64]),
    'ppqhip_quantile_t_multi': (c_int, [c_vp, c_int, c_flt, c_vp, c_vp]),
    'ppqhip_quantile_debug_layout': (None, [c_vp]),
    'ppqhip_quantile_hot_layout': (None, [c_vp]),
    'ppqhip_minmax_t_slots_multi': (c_int, [c_vp, c_int, c_vp]),
    'ppqhip_hist_t_rows_multi': (c_int, [c_vp, c_int, c_int, c_int, c_i64, c_vp]),
    'ppqhip_channel_sum': (c_int, [c_f32p, c_i64, c_i64, c_i64, c_f64p, c_vp]),
    'ppqhip_hist_rows': (c_i64, []),
    'ppqhip_hist_sym_t_rows': (c_int, [c_f32p, c_i64, c_flt, c_int, c_i32p, c_i64, c_vp]),
    'ppqhip_hist_asym_t_rows': (c_int, [c_f32p, c_i64, c_flt, c_flt, c_int, c_i32p, c_i64, c_vp]),
    'ppqhip_hist_rows_finish': (c_int, [c_i32p, c_i64, c_i32p, c_vp]),
    'ppqhip_check_bin_rule': (c_int, [c_f32p, c_i64, c_flt, c_flt, c_int, c_vp, c_vp]),
    'ppqhip_minmax_slots': (c_i64, []),
    'ppqhip_minmax_t_slots': (c_int, [c_f32p, c_i64, c_f32p, c_vp]),
    'ppqhip_minmax_slots_finish': (c_int, [c_f32p, c_f32p, c_vp]),
    'ppqhip_mse_loss_host': (c_flt, [ctypes.POINTER(ctypes.c_int64), c_i64, c_int, c_int, c_int]),
    'ppqhip_mse_search_workspace_bytes': (c_i64, [c_i64]),
    'ppqhip_mse_search': (c_int, [c_i32p, c_i64, c_i64, c_f64p, c_f64p, c_int, c_int, c_int, c_i32p, c_vp, c_vp]),
    'ppqhip_kl_num_candidates': (c_i64, [c_i64, c_int]),
    'ppqhip_kl_losses': (c_int, [c_i32p, c_i64, c_i64, c_int, c_f64p, c_vp]),
    'ppqhip_tensor_clip_t': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_vp]),
    'ppqhip_tensor_clip_c': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_vp]),
    'ppqhip_rounding_loss': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_vp]),
    'ppqhip_rounding_loss_bwd': (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_i64, c_i64, c_int, c_int,
                                         c_int, c_vp]),
    'ppqhip_prof_enable': (c_int, [c_int]),
    'ppqhip_prof_collect': (c_int, [ctypes.POINTER(ProfEntry), c_int]),
    'ppqhip_prof_event_overhead_us': (ctypes.c_double, [c_vp, c_int]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: the HIP kernel library has not been built.  Run '
        '`python -c "import __graft_entry__ as g; g.build()"` or `make -C ppq_amd/csrc` (needs hipcc, '
        'targets gfx950).  ppq_amd has no CPU / PyTorch fallback by design.')

# One HIP runtime per process: libppq_hip.so must bind to the SAME libamdhip64 that PyTorch-ROCm
# uses (its wheel bundles one, same SONAME as /opt/rocm's), otherwise stream handles and device
# pointers handed over from torch mean nothing to our launches ("no ROCm-capable device").
# Importing torch first and pre-loading its runtime makes the dynamic linker resolve our
# DT_NEEDED libamdhip64.so.7 to the copy that is already mapped.
import torch  # noqa: E402

_torch_hip = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
if os.path.exists(_torch_hip):
    ctypes.CDLL(_torch_hip, mode=ctypes.RTLD_GLOBAL)

try:
    lib = ctypes.CDLL(LIB_PATH)
except OSError as e:   # pragma: no cover - broken ROCm install
    raise ImportError(f'cannot load {LIB_PATH}: {e}') from e

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)       # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


# The library is git-ignored and built separately: a stale copy would be called with shifted arguments.
if lib.ppqhip_version() != ABI_VERSION:
    raise ImportError(f'{LIB_PATH} has ABI version {lib.ppqhip_version()}, this package needs {ABI_VERSION}: rebuild it '
                      '(`make -C ppq_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`)')


def last_error() -> str:
    msg = lib.ppqhip_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''
