"""ctypes binding of libvihds_hip.so (C ABI: include/vihds_hip.h).

This is the stub a vi-hds maintainer would add: the reference is pure Python/PyTorch, so the
foreign-function boundary is ctypes; torch is only used by callers for device memory and streams.  There is no
CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import sys

VIHDS_MAX_SLOTS = 64

MODELS = {
    # models.LOOKUP key (reference models/__init__.py:19-35) -> enum vihds_model
    "dr_constant": 0,
    "dr_constant_v2": 1,
    "auto_constant": 2,
    "prpr_constant": 3,
    "relay_constant": 4,
    "degrader_constant": 5,
    "dr_constant_precisions": 6,
    "dr_constant_precisions_v2": 7,
    "auto_constant_precisions": 8,
    "prpr_constant_precisions": 9,
    "relay_constant_precisions": 10,
    "degrader_constant_precisions": 11,
    "dr_blackbox": 12,
    "inducer_constant": 13,
    "inducer_constant_precisions": 14,
    "debug_constant": 15,
}
E_UNSUPPORTED = -2  # VIHDS_E_UNSUPPORTED (include/vihds_hip.h)
SOLVERS = {"modeuler": 0, "modeulerwhile": 1, "euler": 2, "midpoint": 3, "rk4": 4, "dopri5": 5, "bosh3": 6,
           "adaptive_heun": 7, "dopri8": 8}
# torchdiffeq's adaptive pairs (vihds_rk_adaptive.hpp); "dopri8" runs an 8th-order Dormand-Prince pair with Hairer's
# DOP853 coefficients, not torchdiffeq's own 8(7) tableau (vihds_dop853_tableau.hpp)
ADAPTIVE_SOLVERS = ("dopri5", "bosh3", "adaptive_heun", "dopri8")

_c_float_p = ctypes.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class OdeProblem(ctypes.Structure):
    """struct vihds_ode_problem"""

    _fields_ = [
        ("model", ctypes.c_int),
        ("solver", ctypes.c_int),
        ("B", ctypes.c_int),
        ("S", ctypes.c_int),
        ("T", ctypes.c_int),
        ("C", ctypes.c_int),
        ("D", ctypes.c_int),
        ("n_rows", ctypes.c_int),
        ("slot_row", ctypes.c_int * VIHDS_MAX_SLOTS),
        ("n_hidden_prec", ctypes.c_int),
        ("n_hidden_states", ctypes.c_int),
        ("n_latent_states", ctypes.c_int),
        ("n_const", ctypes.c_int),
        ("init_latent", ctypes.c_float),
        ("init_prec", ctypes.c_float),
        ("logp_grad_broadcast", ctypes.c_int),
        ("kernel_variant", ctypes.c_int),
    ]


_LIB = None
_P = ctypes.c_void_p
_I = ctypes.c_int

class IwaeJob(ctypes.Structure):
    """struct vihds_iwae_job (include/vihds_hip.h): the IWAE loss evaluated inside vihds_theta_bwd"""

    _fields_ = [("logp", ctypes.c_void_p), ("log_p", ctypes.c_void_p), ("log_q", ctypes.c_void_p),
                ("n_iwae_total", ctypes.c_int), ("log_w", ctypes.c_void_p), ("lse", ctypes.c_void_p),
                ("loss", ctypes.c_void_p), ("ticket", ctypes.c_void_p)]


class ThetaOpts(ctypes.Structure):
    """struct vihds_theta_opts (include/vihds_hip.h)"""

    _fields_ = [("q_rows", ctypes.c_void_p), ("q_prec_is_log", ctypes.c_int), ("rng", ctypes.c_void_p),
                ("S_total", ctypes.c_int), ("s_offset", ctypes.c_int), ("g_theta_scale", ctypes.c_void_p),
                ("iwae", ctypes.POINTER(IwaeJob))]


class Conditioner(ctypes.Structure):
    """struct vihds_conditioner (include/vihds_hip.h)"""

    _fields_ = [("E", ctypes.c_int), ("first_row", ctypes.c_int), ("w_mean", ctypes.c_float), ("w_std", ctypes.c_float),
                ("z", ctypes.c_void_p), ("rng", ctypes.c_void_p), ("relevance", ctypes.c_void_p),
                ("is_default", ctypes.c_void_p)]


class OffsetLayer(ctypes.Structure):
    """struct vihds_offset_layer (include/vihds_hip.h)"""

    _fields_ = [("n", ctypes.c_int), ("src_row", ctypes.c_int), ("dst_row", ctypes.c_int), ("W", ctypes.c_void_p),
                ("bias", ctypes.c_void_p)]


class EncoderShape(ctypes.Structure):
    """struct vihds_encoder_shape (include/vihds_hip.h)"""

    _fields_ = [(n, ctypes.c_int) for n in ("B", "C_in", "L", "F", "K", "pool", "H", "n_tr", "D", "nl", "l_tr", "l_dv",
                                            "ng", "g_tr", "g_dv", "ngl", "nc")]


class GramRect(ctypes.Structure):
    """struct vihds_gram_rect (include/vihds_hip.h)"""

    _fields_ = [(n, ctypes.c_int) for n in ("a0", "na", "b0", "nb", "dest0", "dest_stride_a", "dest_stride_b")]


ADAM_MAX_TENSORS = 32


class AdamTensors(ctypes.Structure):
    """struct vihds_adam_tensors (include/vihds_hip.h)"""

    _fields_ = [("n", ctypes.c_int), ("size", ctypes.c_int * ADAM_MAX_TENSORS),
                ("param", ctypes.c_void_p * ADAM_MAX_TENSORS), ("grad", ctypes.c_void_p * ADAM_MAX_TENSORS)]


TAIL_MAX_EXTRA = 4


class TailTensor(ctypes.Structure):
    """struct vihds_tail_tensor (include/vihds_hip.h)"""

    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("grad_src", ctypes.c_void_p),
                ("map", ctypes.c_void_p), ("size", ctypes.c_int), ("nparts", ctypes.c_int),
                ("part_stride", ctypes.c_longlong), ("mv_offset", ctypes.c_int)]


class StepTailArgs(ctypes.Structure):
    """struct vihds_step_tail_args (include/vihds_hip.h)"""

    _fields_ = [("P", ctypes.c_int), ("S", ctypes.c_int), ("kind", ctypes.c_void_p), ("q_all", ctypes.c_void_p),
                ("q_rows", ctypes.c_void_p), ("p_mu", ctypes.c_void_p), ("p_prec", ctypes.c_void_p),
                ("clip_lo", ctypes.c_void_p), ("clip_hi", ctypes.c_void_p), ("u", ctypes.c_void_p),
                ("g_theta_unit", ctypes.c_void_p), ("iwae", IwaeJob), ("g_all", ctypes.c_void_p),
                ("delta_obs", ctypes.c_void_p), ("inputs", ctypes.c_void_p), ("dev1hot", ctypes.c_void_p),
                ("lin_w", ctypes.c_void_p), ("local_w", ctypes.c_void_p), ("pooled", ctypes.c_void_p),
                ("hidden", ctypes.c_void_p), ("g_pre", ctypes.c_void_p), ("g_conv", ctypes.c_void_p),
                ("param", ctypes.c_void_p * 8), ("grad", ctypes.c_void_p * 8), ("mv_offset", ctypes.c_int * 8),
                ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("state", ctypes.c_void_p), ("lr_dev", ctypes.c_void_p),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                # ABI 13: pre-weighted gradients, shifted gradient rows, decoder-side tensors, dr_blackbox's offset layer
                ("g_theta_weighted", ctypes.c_int), ("g_shift_lo", ctypes.c_int), ("g_shift_n", ctypes.c_int),
                ("g_shift", ctypes.c_int), ("n_extra", ctypes.c_int), ("extra", TailTensor * TAIL_MAX_EXTRA),
                ("off_n", ctypes.c_int), ("off_row0", ctypes.c_int), ("off_w", ctypes.c_void_p), ("off_b", ctypes.c_void_p),
                ("off_gw", ctypes.c_void_p), ("off_gb", ctypes.c_void_p), ("off_mv_w", ctypes.c_int),
                ("off_mv_b", ctypes.c_int), ("off_rowsum", ctypes.c_void_p), ("phase", ctypes.c_int),
                ("rng_advance", ctypes.c_void_p)]


_PROTOTYPES = {
    "vihds_abi_version": (_I, []),
    "vihds_last_error": (ctypes.c_char_p, []),
    "vihds_debug_newton_hist": (_I, [_P]),
    "vihds_model_n_states": (_I, [_I]),
    "vihds_model_n_species": (_I, [_I]),
    "vihds_model_n_slots": (_I, [_I]),
    "vihds_model_slot_name": (ctypes.c_char_p, [_I, _I]),
    "vihds_model_n_weights": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_fwd": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 10),
    "vihds_ode_bwd": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 14),
    "vihds_ode_bwd_elbo": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 14),
    "vihds_ode_adaptive_workspace_floats": (ctypes.c_longlong, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_adaptive_grid": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 5 + [ctypes.c_float, ctypes.c_float] + [_P] * 2
                                + [_I, _P, _P]),
    "vihds_ode_logp_grad": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 8),
    "vihds_theta_ode_logp_grad": (_I, [ctypes.POINTER(OdeProblem), _I] + [_P] * 8 + [ctypes.POINTER(ThetaOpts),
                                                                                     ctypes.POINTER(Conditioner)]
                                  + [_P] * 10),
    "vihds_theta_ode_fwd": (_I, [ctypes.POINTER(OdeProblem), _I] + [_P] * 8 + [ctypes.POINTER(ThetaOpts), ctypes.POINTER(OffsetLayer)]
                            + [_P] * 12),
    "vihds_rng_advance": (_I, [_P, _P]),
    "vihds_ode_bwd_aux_floats": (ctypes.c_longlong, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_bwd_reduces_weights": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_traj_layout": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_adaptive_tape_floats": (ctypes.c_longlong, [ctypes.POINTER(OdeProblem), _I]),
    "vihds_ode_adaptive_fwd": (_I, [ctypes.POINTER(OdeProblem), _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _I, _P, _P, _P]),
    "vihds_ode_adaptive_bwd": (_I, [ctypes.POINTER(OdeProblem), _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "vihds_ode_fwd_summaries_supported": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_fwd_summaries_workspace_floats": (ctypes.c_longlong, [ctypes.POINTER(OdeProblem)]),
    "vihds_ode_fwd_summaries": (_I, [ctypes.POINTER(OdeProblem)] + [_P] * 13),
    "vihds_ode_adaptive_fwd_w": (_I, [ctypes.POINTER(OdeProblem), _P, _P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _I, _P, _P, _P]),
    "vihds_ode_adaptive_bwd_w": (_I, [ctypes.POINTER(OdeProblem), _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "vihds_blackbox_dump_fields": (_I, []),
    "vihds_problem_n_states": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_problem_n_slots": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_problem_dump_fields": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_blackbox_gram_on_chip": (_I, [ctypes.POINTER(OdeProblem)]),
    "vihds_blackbox_tail_offset_floats": (ctypes.c_longlong, [ctypes.POINTER(OdeProblem)]),
    "vihds_blackbox_gram_reduce": (_I, [ctypes.POINTER(OdeProblem), _P, _P, _P]),
    "vihds_theta_fwd": (_I, [_I, _I, _I] + [_P] * 11 + [ctypes.POINTER(ThetaOpts), _P]),
    "vihds_theta_bwd": (_I, [_I, _I, _I] + [_P] * 13 + [ctypes.POINTER(ThetaOpts), _P]),
    "vihds_iwae_fwd": (_I, [_I, _I] + [_P] * 7),
    "vihds_iwae_bwd": (_I, [_I, _I] + [_P] * 5),
    "vihds_iwae_loss_fwd": (_I, [_I, _I, _I] + [_P] * 12),
    "vihds_iwae_loss_unit_grad": (_I, [_I, _I, _I]),
    "vihds_iwae_combine": (_I, [_I, _I, _I, _I] + [_P] * 7),
    "vihds_iwae_loss_bwd": (_I, [_I, _I] + [_P] * 6),
    "vihds_device_condition": (_I, [_I] * 6 + [ctypes.c_float, ctypes.c_float] + [_P] * 7),
    "vihds_encoder_fwd": (_I, [ctypes.POINTER(EncoderShape)] + [_P] * 16),
    "vihds_encoder_bwd": (_I, [ctypes.POINTER(EncoderShape)] + [_P] * 19),
    "vihds_gram_scratch_floats": (ctypes.c_longlong, [ctypes.c_longlong, _I, _P]),
    "vihds_gram_blocks": (_I, [_I, ctypes.c_longlong, _I] + [_P] * 5),
    "vihds_blackbox_tail_grads": (_I, [_P] * 8),
    "vihds_offset_rows_fwd": (_I, [_I] * 7 + [_P] * 5),
    "vihds_offset_rows_bwd": (_I, [_I] * 8 + [_P] * 4),
    "vihds_gather_batch": (_I, [_I] * 6 + [_P] * 8 + [_P]),
    "vihds_adam_step": (_I, [ctypes.POINTER(AdamTensors), _P, _P, _P, _P] + [ctypes.c_float] * 5 + [_P, _P]),
    "vihds_step_tail": (_I, [ctypes.POINTER(EncoderShape), ctypes.POINTER(StepTailArgs), _P]),
    "vihds_step_tail_supported": (_I, [ctypes.POINTER(EncoderShape), _I, _I]),
    "vihds_iw_summaries": (_I, [_I] * 5 + [_P] * 5 + [ctypes.POINTER(ctypes.c_int)] + [_P] * 5),
    "vihds_iw_summaries_plan": (_I, [_I]),
    "vihds_iw_summaries_states": (_I, [_I] * 6 + [_P] * 4 + [ctypes.POINTER(ctypes.c_int)] + [_P] * 5),
}


def library_path():
    here = os.path.dirname(os.path.abspath(__file__))
    return os.environ.get("VIHDS_HIP_LIB", os.path.join(os.path.dirname(here), "lib", "libvihds_hip.so"))


def lib():
    """Load (once) and return the shared library; raise loudly if it is not there."""
    global _LIB
    if _LIB is None:
        import torch  # noqa: F401  -- make sure torch's HIP runtime is the one already mapped in this process

        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "libvihds_hip.so not found at %s -- build it with `make -C vi-hds_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % path
            )
        handle = ctypes.CDLL(path)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        if handle.vihds_abi_version() != 14:
            raise RuntimeError("libvihds_hip.so ABI version mismatch")
        _LIB = handle
    return _LIB


def exported_symbols():
    return sorted(_PROTOTYPES)


def check(rc, what):
    if rc != 0:
        msg = lib().vihds_last_error().decode()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg))


def model_slots(model_key):
    """Reference parameter names in the kernel's slot order for a model."""
    L = lib()
    m = MODELS[model_key]
    n = L.vihds_model_n_slots(m)
    if n < 0:
        raise RuntimeError("model '%s' is not supported by this build of libvihds_hip.so" % model_key)
    return [L.vihds_model_slot_name(m, s).decode() for s in range(n)]


BLACKBOX_BUILTIN = (2, 25, 20, 12)  # n_latent_species, n_hidden_decoder, n_hidden_decoder_precisions, n_z + n_x + n_y


def blackbox_variant_path(L, HS, HP, NLAT):
    return os.path.join(os.path.dirname(library_path()), "libvihds_bb_%d_%d_%d_%d.so" % (L, HS, HP, NLAT))


def ensure_blackbox_variant(L, HS, HP, NLAT):
    """dr_blackbox kernels are compiled per network size (csrc/vihds_bb_variant.hpp): the ICML sizes are part of
    libvihds_hip.so, any other set is a side library next to it.  Build it here (hipcc, one to several minutes, once --
    it stays in vi-hds_amd/lib/) when it is missing; raise loudly when that is not possible or switched off
    (VIHDS_BLACKBOX_JIT=0).  No CPU fallback."""
    key = (int(L), int(HS), int(HP), int(NLAT))
    if key == BLACKBOX_BUILTIN:
        return None
    path = blackbox_variant_path(*key)
    if os.path.exists(path):
        # a side library of an earlier layout of the BbVariant record (no `vihds_bb_variant_v2`): rebuild it
        with open(path, "rb") as f:
            stale = b"vihds_bb_variant_v2" not in f.read()
        if stale and os.environ.get("VIHDS_BLACKBOX_JIT", "1") != "0" and "VIHDS_HIP_LIB" not in os.environ:
            os.remove(path)
    if not os.path.exists(path):
        import subprocess

        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
        cmd = ["make", "-C", csrc, "blackbox", "L=%d" % key[0], "HS=%d" % key[1], "HP=%d" % key[2], "NLAT=%d" % key[3]]
        if (os.environ.get("VIHDS_BLACKBOX_JIT", "1") == "0" or "VIHDS_HIP_LIB" in os.environ
                or not os.path.exists(os.path.join(csrc, "sized", "ode_dr_blackbox_sized.hip"))):
            raise RuntimeError("dr_blackbox at sizes %s needs %s (build: %s)" % (key, path, " ".join(cmd)))
        # one builder at a time (several ranks of one job meet here together): an exclusive lock on a file next to the
        # library, and a second look once it is held
        import fcntl

        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(os.path.join(os.path.dirname(path), ".blackbox_build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not os.path.exists(path):
                    sys.stderr.write("[vihds] building the dr_blackbox kernels for sizes %s: %s\n" % (key, " ".join(cmd)))
                    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                    if res.returncode != 0 or not os.path.exists(path):
                        raise RuntimeError("building %s failed:\n%s" % (path, res.stdout[-2000:]))
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return path


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
