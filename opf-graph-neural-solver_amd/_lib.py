"""ctypes binding of libgns_hip.so (include/gns_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GNS_ERRORS = {1: 'GNS_EINVAL (bad argument)', 2: 'GNS_EUNSUPPORTED (no compiled kernel holds this model: latent_dim <= 20, hidden_dim <= 14, K <= 64; narrower models run zero-padded on the next wider kernel)',
              3: 'GNS_ETOPOLOGY (bus id out of range or not a valid line index)', 4: 'GNS_ESIZE (buffer too small)',
              5: 'GNS_ELAUNCH (HIP launch error)'}


class GnsConfig(ctypes.Structure):
    _fields_ = [('n_bus', ctypes.c_int32), ('n_line', ctypes.c_int32), ('n_gen', ctypes.c_int32), ('K', ctypes.c_int32),
                ('latent_dim', ctypes.c_int32), ('hidden_dim', ctypes.c_int32), ('multiple_phi', ctypes.c_int32),
                ('gamma', ctypes.c_float)]


def library_path():
    # GNS_LIB selects an alternative build (ablation / A-B timing runs); the default is the shipped library
    return os.environ.get('GNS_LIB') or os.path.join(_HERE, 'libgns_hip.so')


def load_library():
    """Load (once) and type the C-ABI.  Raises OSError with build instructions if the .so is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path) and not os.environ.get('GNS_LIB') and os.environ.get('GNS_NO_AUTOBUILD') != '1':
        # a source-only checkout: build in-tree once (hipcc cross-compiles for gfx950; ~1 minute); never a fallback path
        import shutil
        import subprocess
        if shutil.which('make') and (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
            env = dict(os.environ, PATH=os.environ.get('PATH', '') + ':/opt/rocm/bin')
            subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc'), '-j4'], check=False, env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(path):
        raise OSError(f'{path} not found: build it with `make -C {os.path.join(_HERE, "csrc")}` '
                      '(hipcc --offload-arch=gfx950) or `python -c "import __graft_entry__ as g; g.build()"`. '
                      'There is no CPU fallback for the GNS hot path.')
    lib = ctypes.CDLL(path)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    cfgp = ctypes.POINTER(GnsConfig)
    lib.gns_version.restype = ctypes.c_char_p
    lib.gns_version.argtypes = []
    lib.gns_param_count.argtypes = [cfgp, ctypes.POINTER(i64)]
    lib.gns_config_supported.argtypes = [cfgp]
    lib.gns_topology_bytes.argtypes = [i32, i32, i32, ctypes.POINTER(sz)]
    lib.gns_prepare_topology.argtypes = [i32, i32, i32, vp, vp, vp, vp, sz]
    lib.gns_workspace_bytes.argtypes = [cfgp, i64, ctypes.c_int, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.gns_forward.argtypes = [cfgp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, sz, ctypes.c_int, vp]
    lib.gns_uses_packed_inputs.argtypes = [cfgp, i64, ctypes.c_int]
    lib.gns_prepack_bytes.argtypes = [cfgp, i64, ctypes.POINTER(sz)]
    lib.gns_prepack.argtypes = [cfgp, vp, vp, vp, vp, i64, vp, sz, vp]
    lib.gns_backward.argtypes = [cfgp, vp, vp, vp, vp, vp, i64, vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.gns_adam_step.argtypes = [vp, vp, vp, vp, i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, i64, vp]
    lib.gns_adam_step_dev.argtypes = [vp, vp, vp, vp, i64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, vp]
    lib.gns_team_status_offset.argtypes = [cfgp, i64, ctypes.c_int, ctypes.POINTER(sz)]
    lib.gns_team_status.argtypes = [cfgp, i64, vp, sz, ctypes.c_int, ctypes.POINTER(ctypes.c_int), vp]
    lib.gns_profile_enable.argtypes = [ctypes.c_int]
    lib.gns_profile_read.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    lib.gns_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.gns_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    for f in ('gns_profile_enable', 'gns_profile_read', 'gns_param_count', 'gns_config_supported', 'gns_topology_bytes', 'gns_prepare_topology',
              'gns_workspace_bytes', 'gns_forward', 'gns_backward', 'gns_profile_enable', 'gns_profile_read',
              'gns_set_option', 'gns_get_option', 'gns_prepack_bytes', 'gns_prepack', 'gns_uses_packed_inputs', 'gns_adam_step',
              'gns_adam_step_dev', 'gns_team_status', 'gns_team_status_offset'):
        getattr(lib, f).restype = ctypes.c_int
    _LIB = lib
    return lib


EXPORTS = ('gns_version', 'gns_param_count', 'gns_config_supported', 'gns_topology_bytes', 'gns_prepare_topology',
           'gns_workspace_bytes', 'gns_forward', 'gns_backward', 'gns_profile_enable', 'gns_profile_read',
           'gns_set_option', 'gns_get_option', 'gns_prepack_bytes', 'gns_prepack', 'gns_uses_packed_inputs', 'gns_adam_step',
           'gns_adam_step_dev', 'gns_team_status', 'gns_team_status_offset')


def set_option(name: str, value: int) -> None:
    """Process-wide tuning knob of the library (include/gns_hip.h, "configuration")."""
    rc = load_library().gns_set_option(name.encode(), int(value))
    if rc != 0:
        raise ValueError(f'gns_set_option({name!r}, {value}) rejected: {GNS_ERRORS.get(rc, rc)}')


def get_option(name: str) -> int:
    v = ctypes.c_int()
    rc = load_library().gns_get_option(name.encode(), ctypes.byref(v))
    if rc != 0:
        raise ValueError(f'unknown option {name!r}')
    return v.value
