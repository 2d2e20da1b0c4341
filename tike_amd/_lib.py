"""ctypes binding of the C ABI declared in include/tike_amd.h.

The shared library is built in-tree (``tike_amd/csrc/libtike_amd.so``) by
``tike_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is no CPU
fallback: if the library is missing, importing this module raises.
"""
import ctypes
import os
import re

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must
# be loaded BEFORE libtike_amd.so so that both bind to the same runtime: with
# the opposite order the process holds two runtimes and every launch from
# this library fails with hipErrorNoDevice (100).
import torch  # noqa: F401  (load order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TIKE_AMD_LIB: alternative build of the same library (A/B tuning runs)
LIB_PATH = os.environ.get("TIKE_AMD_LIB") or os.path.join(
    _HERE, "csrc", "libtike_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tike_amd.h")

# the header version this binding's prototypes were written against
ABI_VERSION = 11
ERR_ARG = 1000001
ERR_UNSUPPORTED = 1000002
ERR_COMM = 2000000
COMM_ID_BYTES = 128

if not os.path.isfile(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP library first "
        "(`make -C tike_amd/csrc` or `python -c 'import __graft_entry__ as g; "
        "g.build()'`). tike_amd has no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float
_d = ctypes.c_double

_PROTOTYPES = {
    "tike_abi_version": [],
    "tike_init": [],
    "tike_set_deterministic": [_i, _p, _l],
    "tike_patch_fwd": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "tike_patch_adj": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "tike_conv_fwd": [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "tike_conv_adj": [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "tike_conv_adj_probe": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "tike_fft2": [_p, _p, _l, _i, _i, _f, _p],
    "tike_fft2_supported": [_i],
    "tike_fft2_general": [_p, _p, _l, _i, _i, _f, _i, _i, _p],
    "tike_pfa_supported": [_i, _i, _i],
    "tike_pfa_fwd_gather": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i,
                            _i, _i, _i, _p],
    "tike_pfa_fwd_subtiles_supported": [_i, _i, _i],
    "tike_pfa_fwd_subtiles": [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _i,
                              _i, _i, _i, _i, _p],
    "tike_pfa_fft2": [_p, _p, _l, _i, _i, _p],
    "tike_pfa_combine_gradient": [_p, _p, _p, _p, _i, _i, _i, _f, _i, _f, _l, _i,
                                  _p],
    "tike_pfa_inv_products": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _p, _f,
                              _i, _i, _i, _i, _f, _p],
    "tike_gen_supported": [_i, _i, _i],
    "tike_gen_fwd_rows": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i,
                          _i, _i, _i, _p],
    "tike_gen_cols_gradient": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _f,
                               _l, _p],
    "tike_gen_inv_rows_gradients": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p,
                                    _p, _f, _i, _i, _i, _i, _f, _p],
    "tike_fresnel_spect_prop": [_p, _p, _p, _l, _i, _i, _f, _f, _p],
    "tike_fft2_pass1": [_p, _p, _l, _i, _i, _p],
    "tike_fft2_pass2_inplace": [_p, _l, _i, _i, _f, _p],
    "tike_slice_step": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "tike_fft2_pass2_intensity": [_p, _p, _l, _i, _i, _i, _f, _i, _p],
    "tike_fresnel_colpass": [_p, _p, _i, _p, _l, _i, _f, _p],
    "tike_ifft2_pass2_products": [_p, _p, _p, _p, _i, _p, _p, _f, _p, _i, _i, _i,
                                  _i, _i, _i, _f, _p],
    "tike_ptycho_fwd": [_p, _p, _p, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i,
                        _i, _i, _f, _i, _p],
    "tike_ptycho_adj": [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i,
                        _i, _f, _i, _p],
    "tike_ifft2_crop": [_p, _p, _p, _l, _i, _i, _f, _p],
    "tike_ptycho_fwd_intensity": [_p, _p, _p, _i, _p, _p, _i, _i, _p, _p, _p,
                                  _i, _i, _i, _i, _i, _i, _f, _p],
    "tike_gradient_scale": [_p, _p, _p, _p, _p, _i, _i, _i, _f, _l, _p],
    "tike_ifft2_crop_scaled": [_p, _p, _i, _p, _p, _l, _i, _i, _f, _p],
    "tike_ifft2_crop_scaled_modes": [_p, _p, _p, _p, _i, _p, _p, _l, _i, _i, _f,
                                     _p],
    "tike_poisson_steps": [_p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p],
    "tike_poisson_steps_handoff": [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _f,
                                   _f, _l, _f, _f, _p],
    "tike_scale_modes": [_p, _p, _p, _l, _i, _p],
    "tike_ptycho_fwd_intensity_only": [_p, _p, _p, _i, _p, _p, _i, _i, _p, _p,
                                       _i, _i, _i, _i, _i, _i, _f, _p],
    "tike_ptycho_fwd_gradient_scale": [_p, _p, _p, _i, _p, _p, _i, _i, _p, _p,
                                       _p, _p, _p, _p, _p, _i, _i, _i, _i, _i,
                                       _i, _f, _i, _f, _l, _p],
    "tike_fwd_pass1": [_p, _p, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i,
                       _i, _i, _i, _p],
    "tike_fwd_gradient_scale": [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _f,
                                _i, _f, _l, _p],
    "tike_grad_ifft2_pass1": [_p, _p, _p, _p, _i, _p, _l, _i, _f, _p],
    "tike_fwd_grad_ifft2_pass1": [_p, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _f,
                                  _l, _p],
    "tike_fwd_grad_ifft2_pass1_slices": [_p, _p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _f,
                                  _l, _p, _i, _p],
    "tike_ifft2_pass1_scaled": [_p, _p, _p, _p, _i, _p, _l, _i, _p],
    "tike_ifft2_pass2_gradients": [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _f,
                                   _i, _i, _i, _f, _p],
    "tike_ifft2_pass2_gradients_scaled": [_p, _p, _p, _p, _p, _i, _i, _p, _p,
                                          _p, _f, _i, _i, _i, _f, _p, _p],
    "tike_ifft2_pass2_eigen_fits": [_i, _i, _i],
    "tike_ifft2_pass2_gradients_modes": [_p, _p, _p, _p, _p, _i, _i, _p, _p,
                                         _p, _f, _i, _i, _i, _f, _i, _i, _i, _p],
    "tike_poisson_steps_grad_ifft2_pass1": [_p, _p, _i, _p, _p, _p, _p, _p, _i,
                                            _i, _i, _f, _f, _l, _f, _f, _p],
    "tike_object_update_precond": [_p, _p, _p, _f, _p, _p, _p, _l, _p],
    "tike_lstsq_step_sums": [_p, _p, _i, _f, _p, _p],
    "tike_lstsq_step_solve": [_p, _i, _f, _p, _d, _i, _i, _p, _p],
    "tike_probe_update": [_p, _p, _p, _p, _f, _l, _p],
    "tike_eigen_weights0": [_p, _p, _i, _i, _i, _i, _p, _p],
    "tike_eigen_proj_mean": [_p, _i, _p, _l, _p, _l, _i, _p, _p],
    "tike_eigen_normalise": [_p, _p, _d, _f, _i, _p, _p, _p],
    "tike_eigen_dsum": [_p, _i, _l, _p, _p],
    "tike_eigen_weights": [_p, _i, _l, _p, _d, _p, _l, _p, _i, _p, _p],
    "tike_grad_ifft2_crop": [_p, _p, _p, _p, _i, _p, _p, _l, _i, _i, _f, _f,
                             _p],
    "tike_position_sums": [_p, _p, _i, _p, _p, _p, _i, _i, _p, _i, _p, _p, _i,
                           _i, _i, _p],
    "tike_farplane_gradient": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _l,
                               _p],
    "tike_intensity": [_p, _p, _l, _i, _l, _p],
    "tike_cost_each_pattern": [_p, _p, _p, _l, _l, _i, _p],
    "tike_objective_grad": [_p, _p, _p, _p, _l, _i, _l, _i, _p],
    "tike_lstsq_gradients": [_p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i,
                             _i, _i, _i, _i, _p],
    "tike_varying_probe": [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p],
    "tike_scatter_patches": [_p, _p, _p, _i, _i, _i, _i, _p],
    "tike_eigen_position_sums": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i,
                                 _i, _p],
    "tike_eigen_pixel_update": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _i, _i,
                                _i, _p],
    "tike_probe_grad": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "tike_probe_preconditioner": [_p, _p, _p, _i, _i, _i, _i, _p],
    "tike_psi_preconditioner": [_p, _p, _p, _i, _i, _i, _i, _p],
    "tike_scatter_amplitudes": [_p, _p, _p, _i, _i, _i, _i, _p],
    "tike_lstsq_step_stats": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p,
                              _p, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "tike_eigen_pixel_update1": [_p, _p, _p, _p, _p, _p, _l, _p, _p, _i, _i, _i,
                                 _p, _p, _f, _p, _p, _p, _i, _i, _p],
    "tike_lstsq_tail_mid": [_p, _p, _i, _p, _f, _p, _i, _f, _p, _d, _i, _i, _p,
                            _p],
    "tike_eigen_position_sums1": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p,
                                  _i, _i, _p],
    "tike_lstsq_tail_finish": [_p, _p, _d, _p, _p, _p, _p, _f, _l, _p, _l, _i,
                               _i, _p, _p, _i, _i, _p],
    "tike_lstsq_chunk_gradients": [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _i,
                                   _f, _l, _p, _p, _p, _p, _p, _p, _p, _p, _f,
                                   _p, _i, _i, _i, _i, _i, _f, _f, _p],
    "tike_cgrad_direction": [_p, _p, _p, _p, _l, _i, _p, _i, _d, _p, _p, _p],
    "tike_cgrad_line_search": [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i,
                               _i, _i, _i, _i, _f, _d, _p, _p, _i, _p],
    "tike_cgrad_line_search_linear": [_i, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p,
                                      _p, _i, _i, _i, _i, _i, _i, _f, _d, _p, _i,
                                      _p, _p],
    "tike_comm_unique_id": [_p],
    "tike_comm_create": [_p, _i, _i, ctypes.POINTER(_p)],
    "tike_comm_destroy": [_p],
    "tike_comm_allreduce_sum": [_p, _p, _l, _i, _p],
    "tike_comm_broadcast": [_p, _p, _l, _i, _p],
    "tike_cluster_farthest_fill": [_p, _l, _p, _i, _l],
    "tike_cluster_swap_sweep": [_p, _l, _i, _p, _p, _p, _p, _p],
}


def build_id():
    """sha256 of the sources the loaded library was built from
    (`tike_build_id`; '' for a library older than round 6)."""
    try:
        fn = lib.tike_build_id
    except AttributeError:
        return ""
    fn.restype = ctypes.c_char_p
    fn.argtypes = []
    return fn().decode()


def declared_symbols():
    """Names of every function declared in include/tike_amd.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"^int\s+(tike_\w+)\s*\(", text, flags=re.M)))


for _name, _args in _PROTOTYPES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError as e:
        raise ImportError(
            f"{LIB_PATH} does not export {_name}: it was built from another "
            "include/tike_amd.h; rebuild it (`make -C tike_amd/csrc`).") from e
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int


def _check_abi_version():
    """Refuse a library built from another header: its entries would take
    this binding's positional arguments for something else."""
    got = lib.tike_abi_version()
    m = re.search(r"^#define\s+TIKE_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(),
                  flags=re.M) if os.path.isfile(HEADER_PATH) else None
    if got != ABI_VERSION or (m and int(m.group(1)) != ABI_VERSION):
        raise ImportError(
            f"{LIB_PATH} reports ABI version {got}, include/tike_amd.h "
            f"{m.group(1) if m else '?'}, this binding expects {ABI_VERSION}: "
            "rebuild the library (`make -C tike_amd/csrc`).")


_check_abi_version()


def check(rc, what=""):
    """Translate a C-ABI return code into the reference's error behaviour."""
    if rc == 0:
        return
    if rc == ERR_ARG:
        raise ValueError(f"{what}: incompatible shapes / arguments")
    if rc == ERR_UNSUPPORTED:
        raise ValueError(f"{what}: unsupported size for the HIP path")
    if rc >= ERR_COMM:
        raise RuntimeError(f"{what}: RCCL error {rc - ERR_COMM}")
    raise RuntimeError(f"{what}: HIP error {rc}")


# TIKE_DETERMINISTIC=1: fixed-order sums instead of float atomics (bit-identical
# iterates from run to run; include/tike_amd.h `tike_set_deterministic`).  The
# scratch buffer the library needs is allocated here, on the device in use
# when the first tensor goes to the GPU (tike_amd._arrays.current_device).
DETERMINISTIC = os.environ.get("TIKE_DETERMINISTIC", "0") == "1"
DETERMINISTIC_SCRATCH_MIB = int(os.environ.get("TIKE_DETERMINISTIC_MIB", "256"))
_det_scratch = None


def ensure_deterministic():
    """Hand the library its scratch buffer (no-op unless TIKE_DETERMINISTIC=1).
    The buffer lives on ONE device -- the library's pointer is process-wide --
    and its users are ordered by ONE stream: when the current device is no
    longer the buffer's (torch.cuda.set_device, a second Reconstruction on
    another GPU) the buffer is re-allocated there and registered again, after
    the work in flight on the old one has finished.  Two devices driven
    concurrently from one process are not supported in this mode (one process
    per GPU is the design)."""
    global _det_scratch
    if not DETERMINISTIC:
        return
    current = torch.cuda.current_device()
    if _det_scratch is not None and _det_scratch.device.index == current:
        return
    if _det_scratch is not None:
        torch.cuda.synchronize(_det_scratch.device)
    _det_scratch = torch.empty(DETERMINISTIC_SCRATCH_MIB << 20,
                               dtype=torch.uint8,
                               device=torch.device("cuda", current))
    check(lib.tike_set_deterministic(1, _det_scratch.data_ptr(),
                                     _det_scratch.numel()),
          "tike_set_deterministic")
