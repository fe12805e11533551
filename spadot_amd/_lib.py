"""ctypes loader for the in-tree HIP libraries.

The product has no CPU path: a missing library is a hard error that tells the user how to
build it (``python -m spadot_amd.csrc.build``); nothing here falls back to numpy/torch.
"""
import ctypes
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_CACHE = {}


class NativeLibraryMissing(RuntimeError):
    pass


def _load(name):
    if name in _CACHE:
        return _CACHE[name]
    # torch bundles its own libamdhip64.so.7; importing it first makes this library bind to the
    # SAME HIP runtime (same SONAME), so torch device pointers and streams are valid in our kernels.
    import torch  # noqa: F401

    path = os.path.join(_CSRC, name)
    if not os.path.exists(path):
        raise NativeLibraryMissing(
            f"{path} is missing: build the HIP extension with `python -m spadot_amd.csrc.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    _CACHE[name] = lib
    return lib


class _Checked:
    """A library entry whose call is refused unless it carries exactly the declared number of arguments.

    ctypes only rejects a cdecl call with too FEW arguments; one with too many goes through, and a C entry that has grown
    parameters the Python call site does not pass yet reads whatever the stack holds in their place -- on the GPU that
    is a kernel dereferencing a host address (the round-3 `k_sgemm_small` fault).  The header, the
    argtypes here and every call site must agree; tests/test_abi_cpu.py checks the first two against each other on CPU."""
    __slots__ = ("fn", "n", "name")

    def __init__(self, fn, name):
        self.fn, self.n, self.name = fn, len(fn.argtypes), name

    @property
    def argtypes(self):
        return self.fn.argtypes

    @property
    def restype(self):
        return self.fn.restype

    def __call__(self, *args):
        if len(args) != self.n:
            raise TypeError(f"{self.name} takes {self.n} arguments (include/*.h), {len(args)} given")
        return self.fn(*args)


def _seal(lib):
    """Every spadot_* entry that has argtypes becomes a _Checked wrapper; one without argtypes is an error here."""
    for name, fn in list(vars(lib).items()):
        if name.startswith("spadot_") and not isinstance(fn, _Checked):
            if fn.argtypes is None and name not in ("spadot_ot_version", "spadot_model_version"):
                raise RuntimeError(f"{name}: no argtypes declared in spadot_amd/_lib.py")
            if fn.argtypes is not None:
                setattr(lib, name, _Checked(fn, name))


class OTConfig(ctypes.Structure):
    """struct spadot_ot_config (include/spadot_ot.h)."""
    _fields_ = [("lambda1", ctypes.c_double), ("lambda2", ctypes.c_double), ("epsilon", ctypes.c_double),
                ("epsilon0", ctypes.c_double), ("tolerance", ctypes.c_double), ("tau", ctypes.c_double),
                ("batch_size", ctypes.c_int), ("max_iter", ctypes.c_int)]


class OTInfo(ctypes.Structure):
    """struct spadot_ot_info (include/spadot_ot.h)."""
    _fields_ = [("gap", ctypes.c_double), ("stage_iters", ctypes.c_int * 6), ("absorbs", ctypes.c_int),
                ("gap_checks", ctypes.c_int)]


class OTSmallProblem(ctypes.Structure):
    """struct spadot_ot_small_problem (include/spadot_ot.h part C)."""
    _fields_ = [("x_dev", ctypes.c_void_p), ("y_dev", ctypes.c_void_p), ("C_dev", ctypes.c_void_p),
                ("G_dev", ctypes.c_void_p), ("plan_dev", ctypes.c_void_p), ("gamma_rownorm_dev", ctypes.c_void_p),
                ("I", ctypes.c_int), ("J", ctypes.c_int)]


class OTSmallInfo(ctypes.Structure):
    """struct spadot_ot_small_info (include/spadot_ot.h part C)."""
    _fields_ = [("gap", ctypes.c_double), ("stage_iters", ctypes.c_int * 6), ("absorbs", ctypes.c_int),
                ("gap_checks", ctypes.c_int), ("status", ctypes.c_int), ("reserved", ctypes.c_int)]


class WeightImage(ctypes.Structure):
    """struct spadot_weight_image (include/spadot_model.h)."""
    _fields_ = [("offset", ctypes.c_longlong), ("rows", ctypes.c_int), ("K", ctypes.c_int), ("Kp", ctypes.c_int),
                ("reserved", ctypes.c_int), ("image", ctypes.c_void_p)]


class WeightImages(ctypes.Structure):
    """struct spadot_weight_images (include/spadot_model.h)."""
    _fields_ = [("n", ctypes.c_int), ("reserved", ctypes.c_int), ("w", WeightImage * 8)]


def ot_lib():
    """libspadot_ot.so with argtypes/restypes of include/spadot_ot.h part B set."""
    lib = _load("libspadot_ot.so")
    if getattr(lib, "_spadot_ready", False):
        return lib
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.spadot_ot_version.restype = ctypes.c_char_p
    lib.spadot_ot_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci, vp]
    lib.spadot_ot_create.restype = ci
    lib.spadot_ot_destroy.argtypes = [vp]
    lib.spadot_ot_destroy.restype = None
    lib.spadot_ot_ld.argtypes = [vp]
    lib.spadot_ot_ld.restype = ci
    lib.spadot_ot_fused_geometry.argtypes = [vp, ctypes.POINTER(ci)]
    lib.spadot_ot_fused_geometry.restype = None
    lib.spadot_ot_matrix_dev.argtypes = [vp, ci]
    lib.spadot_ot_matrix_dev.restype = vp
    lib.spadot_ot_vector_dev.argtypes = [vp, ci]
    lib.spadot_ot_vector_dev.restype = vp
    lib.spadot_ot_vector_host.argtypes = [vp, ci, vp]
    lib.spadot_ot_vector_host.restype = ci
    lib.spadot_ot_matrix_host.argtypes = [vp, ci, vp]
    lib.spadot_ot_matrix_host.restype = ci
    lib.spadot_ot_set_cost_dev.argtypes = [vp, vp, ci, ci]
    lib.spadot_ot_set_cost_dev.restype = ci
    lib.spadot_ot_set_cost_host.argtypes = [vp, vp]
    lib.spadot_ot_set_cost_host.restype = ci
    lib.spadot_ot_set_cost_from_latents_dev.argtypes = [vp, vp, vp, ci, ci]
    lib.spadot_ot_set_cost_from_latents_dev.restype = ci
    lib.spadot_ot_solve.argtypes = [vp, vp, ctypes.POINTER(OTConfig), ctypes.POINTER(OTInfo)]
    lib.spadot_ot_solve.restype = ci
    lib.spadot_ot_plan_dev.argtypes = [vp, vp, ci, ci]
    lib.spadot_ot_plan_dev.restype = ci
    lib.spadot_ot_plan_group_sums_dev.argtypes = [vp, vp, ci, vp]
    lib.spadot_ot_plan_group_sums_dev.restype = ci
    lib.spadot_ot_plan_host.argtypes = [vp, vp]
    lib.spadot_ot_plan_host.restype = ci
    lib.spadot_ot_plan_rowsums_host.argtypes = [vp, vp]
    lib.spadot_ot_plan_rowsums_host.restype = ci
    lib.spadot_ot_run_iterations.argtypes = [vp, ctypes.POINTER(OTConfig), cd, ci, ctypes.POINTER(ctypes.c_float)]
    lib.spadot_ot_run_iterations.restype = ci
    lib.spadot_ot_run_tau_flag.argtypes = [vp, ci]
    lib.spadot_ot_run_tau_flag.restype = ci
    lib.spadot_ot_run_checked.argtypes = [vp, ctypes.POINTER(OTConfig), cd, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_float)]
    lib.spadot_ot_run_checked.restype = ci
    lib.spadot_ot_time_kernels.argtypes = [vp, ctypes.POINTER(OTConfig), cd, ci, ctypes.POINTER(ctypes.c_float)]
    lib.spadot_ot_time_kernels.restype = ci
    lib.spadot_ot_small_max.argtypes = []
    lib.spadot_ot_small_max.restype = ci
    lib.spadot_ot_small_solve.argtypes = [ci, ctypes.POINTER(OTSmallProblem), ci, ci, ctypes.POINTER(OTConfig), vp, vp]
    lib.spadot_ot_small_solve.restype = ci
    _seal(lib)
    lib._spadot_ready = True
    return lib


def model_lib():
    """libspadot_model.so with argtypes/restypes of include/spadot_model.h set."""
    lib = _load("libspadot_model.so")
    if getattr(lib, "_spadot_ready", False):
        return lib
    vp, ci, cd, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_longlong
    lib.spadot_model_version.restype = ctypes.c_char_p
    sig = {
        "spadot_gat_forward": [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp],
        "spadot_gat_backward_target": [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp],
        "spadot_gat_backward_source": [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "spadot_gat_logits": [vp, ci, vp, vp, ci, ci, ci, vp, vp, vp],
        "spadot_gat_att_grad": [vp, ci, vp, vp, ci, ci, ci, vp, ci, vp, vp, vp, ci, vp],
        "spadot_gat_alpha": [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp],
        "spadot_gat_softmax_backward": [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp],
        "spadot_gat_ds_src": [vp, vp, vp, ci, ci, vp, vp],
        "spadot_gat_mfma_supported": [ci, ci, ci, ci],
        "spadot_gat_tail_supported": [ci, ci, ci],
        "spadot_gat_tail_wvec": [vp, ci, vp, vp, ci, ci, ci, vp, ci, vp, vp, vp, vp],
        "spadot_gat_tail_logits": [vp, ci, ci, vp, vp, vp, ci, ci, ci, vp, vp],
        "spadot_gat_tail_aggregate": [vp, ci, ci, vp, vp, vp, ci, ci, ci, vp, vp, vp],
        "spadot_gat_tail_headmean": [vp, ci, vp, ci, ci, ci, vp, vp],
        "spadot_gat_tail_colsum_rows": [vp, ci, ci, ci, vp, vp],
        "spadot_gat_tail_scale_colsum": [vp, ci, ci, ci, cd, vp, vp, vp],
        "spadot_gat_tail_edge_backward": [vp, ci, ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp],
        "spadot_gat_tail_source_backward": [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, vp, ci, cd, vp],
        "spadot_gat_tail_dwvec_rows": [ci],
        "spadot_gat_tail_dwvec": [vp, ci, ci, vp, vp, ci, ci, ci, ci, vp, vp],
        "spadot_gat_tail_wvec_backward": [vp, ci, vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, vp, vp],
        "spadot_gat_aggregate": [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp],
        "spadot_gat_edge_dot": [vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, ci, vp],
        "spadot_gemm_tn_bf16": [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp],
        "spadot_gemm_nn_bf16": [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp],
        "spadot_enc_fused_supported": [ci, ci, ci, ci],
        "spadot_recon_fb_supported": [ci, ci, ci],
        "spadot_recon_fb": [vp, vp, vp, ci, vp, vp, ci, ci, ci, cd, vp, vp, vp, vp, vp, vp],
        "spadot_sum_parts": [vp, ci, cd, vp, vp],
        "spadot_enc_bn_map": [vp, vp, vp, vp, vp, vp, vp, ci, ci, cd, cd, cd, vp, vp, vp, vp, ci, vp, vp],
        "spadot_enc_bn_fc": [vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, cd, cd, cd, vp, vp, vp, vp, vp, ci, vp, vp],
        "spadot_enc_sum_z": [vp, ci, vp, ci, ci, vp, vp],
        "spadot_svgp_pre2_partials": [vp, ci, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp],
        "spadot_gemm_nn_bf16_masked": [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, cd, vp],
        "spadot_gemm_tn_bf16_split": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp],
        "spadot_gemm_wgrad_bf16": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp],
        "spadot_gemm_wgrad_bf16_tiled": [vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp],
        "spadot_mlp_chain_supported": [ci, vp],
        "spadot_headfc_backward": [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp],
        "spadot_bias_sqerr_forward": [vp, vp, vp, ci, ci, cd, vp, vp, vp],
        "spadot_bias_sqerr_backward": [vp, vp, vp, vp, ci, ci, cd, vp, vp, vp],
        "spadot_mlp_chain_workspace": [ci, ci, vp, vp, vp],
        "spadot_mlp_chain_forward": [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "spadot_mlp_chain_forward_bf16": [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "spadot_mlp_chain_backward": [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "spadot_mlp_chain_backward_add": [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "spadot_kernel_matrix": [vp, vp, ci, ci, ci, cd, ci, ci, vp, vp],
        "spadot_spd_inverse_logdet": [vp, ci, ci, vp, vp, vp],
        "spadot_rowdot_forward": [vp, vp, ci, ci, ci, ci, vp, vp],
        "spadot_rowdot_backward": [vp, vp, ci, ci, ci, ci, vp, vp],
        "spadot_elbo_forward": [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp],
        "spadot_elbo_backward": [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp],
        "spadot_bn_act_forward": [vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, cd, cd, cd, vp, vp, vp, vp],
        "spadot_bn_act_backward": [vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, cd, vp, vp, vp, vp],
        "spadot_ln_act_forward": [vp, vp, vp, ci, ci, cd, cd, vp, vp, vp, vp],
        "spadot_ln_act_backward": [vp, vp, vp, vp, vp, vp, ci, ci, cd, vp, vp, vp, vp],
        "spadot_svgp_post_forward": [vp] * 9 + [ci, ci, ci, cd, cd, cd] + [vp] * 7,
        "spadot_svgp_post_backward": [vp] * 13 + [ci, ci, ci, cd, cd] + [vp] * 8,
        "spadot_svgp_grad_tail": [vp] * 11 + [ci, ci, cd] + [vp] * 4,
        "spadot_svgp_q1t": [vp, vp, vp, vp, vp, ci, ci, ci, vp, vp],
        "spadot_svgp_post_pm_pv": [vp, vp, vp, ci, ci, cd, vp, vp, vp],
        "spadot_svgp_pre": [vp, ci, ci, vp, vp, vp, vp, vp],
        "spadot_svgp_pre2": [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "spadot_svgp_mid": [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp],
        "spadot_spd_inverse_logdet2": [vp, ci, ci, ci, vp, vp, vp, vp, vp],
        "spadot_latent_head_forward": [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "spadot_latent_head_backward": [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp],
        "spadot_cluster_losses_forward": [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp],
        "spadot_cluster_losses_backward": [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp],
        "spadot_cluster_losses_fb": [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "spadot_mix_losses_forward": [vp, vp, vp, vp],
        "spadot_mix_losses_backward": [vp, vp, vp, vp],
        "spadot_sqerr_forward": [vp, vp, ll, cd, ci, vp, vp, vp],
        "spadot_sqerr_backward": [vp, vp, vp, ll, cd, ci, vp, vp],
        "spadot_kmeans_assign": [vp, vp, ci, ci, ci, ci, vp, vp],
        "spadot_sgemm_small": [ci, vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, ll, ll, ll, vp],
        "spadot_lloyd_step": [vp, vp, ci, ci, ci, ci, cd, vp, vp, vp, vp, ci, vp],
        "spadot_lloyd_step_groups": [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp],
        "spadot_colsum": [vp, ci, ci, vp, vp],
        "spadot_cast_rows_multi": [vp, vp, vp, vp, vp, ci, vp],
        "spadot_knn": [vp, ci, ci, ci, vp, vp],
        "spadot_stamp": [vp, ci, vp],
        "spadot_grad_sumsq": [vp, ll, vp, vp, vp],
        "spadot_clip_adamw_dev": [vp, vp, vp, vp, ll, cd, cd, cd, cd, cd, cd, vp, vp, vp, vp, vp, vp],
        "spadot_clip_adamw_images_dev": [vp, vp, vp, vp, ll, cd, cd, cd, cd, cd, cd, vp, vp, vp, vp, ctypes.POINTER(WeightImages), vp],
        "spadot_grad_norm_step_dev": [vp, ll, vp, vp, vp, vp],
        "spadot_adamw_range_dev": [vp, vp, vp, vp, ll, ll, cd, cd, cd, cd, cd, cd, vp, vp, vp, ctypes.POINTER(WeightImages), vp],
        "spadot_adamw_step": [vp, vp, vp, vp, vp, ll, cd, cd, cd, cd, cd, cd, ci, vp],
        "spadot_adamw_step_dev": [vp, vp, vp, vp, vp, ll, cd, cd, cd, cd, cd, cd, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ci
    lib.spadot_recon_fb_workspace.argtypes = [ci, ci, ci]
    lib.spadot_recon_fb_workspace.restype = ll
    lib.spadot_enc_fused_workspace.argtypes = [ci, ci, ci, ci]
    lib.spadot_enc_fused_workspace.restype = ll
    lib.spadot_gemm_bf16_split_workspace.argtypes = [ci, ci, ci, ci]
    lib.spadot_gemm_bf16_split_workspace.restype = ll
    lib.spadot_gemm_wgrad_bf16_workspace.argtypes = [ci, ci, ci, ci]
    lib.spadot_gemm_wgrad_bf16_workspace.restype = ll
    lib.spadot_gemm_wgrad_bf16_workspace_tiled.argtypes = [ci, ci, ci, ci, ci]
    lib.spadot_gemm_wgrad_bf16_workspace_tiled.restype = ll
    _seal(lib)
    lib._spadot_ready = True
    return lib
