"""ctypes binding of libsaev_amd.so (the C ABI declared in include/saev_amd.h).

The product path has no CPU fallback: if the shared library is missing or a HIP device is not
present, everything here raises.  Build the library with ``make`` (or ``__graft_entry__.build()``).
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = pathlib.Path(os.environ.get("SAEV_AMD_LIB", _HERE / "libsaev_amd.so"))

ABI_VERSION = 10


class SaevCfg(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("d_sae", C.c_int32), ("top_k", C.c_int32), ("k_aux", C.c_int32),
        ("alpha", C.c_float), ("dead_threshold_tokens", C.c_int64),
        ("normalize_w_dec", C.c_int32), ("remove_parallel_grads", C.c_int32),
        ("max_batch", C.c_int32), ("encoder_mode", C.c_int32), ("aux_dead_cap", C.c_int32),
        ("shard_world", C.c_int32), ("bound_mode", C.c_int32), ("max_backward_rows", C.c_int32),
    ]


class SaevDebugCfg(C.Structure):
    """Route switches (include/saev_amd.h: saev_debug_cfg); all zero = shipped defaults."""

    _fields_ = [(n, C.c_int32) for n in ("struct_size", "dw_route", "enc_mfma", "fused_chain", "ngroups", "enc_wgs", "refresh_first",
                                         "refresh_every", "aux_small_max", "fwd_route", "dead_lag", "csc_route", "fin_route", "prep_route", "aux_dense_route", "aux_small_route", "own_check", "enc_rot", "group_route", "aux_split_route", "aux_wide_route")]


class SaevLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("off_W_dec", "off_b_dec", "off_W_enc", "off_b_enc", "n_total", "chunk_a", "chunk_b")]


class SaevStepStats(C.Structure):
    _fields_ = [
        ("mse", C.c_float), ("aux", C.c_float), ("l0", C.c_float), ("l1", C.c_float),
        ("grad_norm", C.c_float), ("upper", C.c_float), ("n_dead", C.c_int32),
        ("n_overflow_rows", C.c_int32), ("cand_max", C.c_int32), ("dense_route", C.c_int32), ("sse", C.c_double), ("sum_sq", C.c_double),
    ]


class SaevError(RuntimeError):
    pass


P = C.c_void_p
_SIGNATURES = {
    "saev_abi_version": (C.c_int, []),
    "saev_last_error": (C.c_char_p, [P]),
    "saev_layout": (C.c_int, [C.POINTER(SaevCfg), C.POINTER(SaevLayout)]),
    "saev_create": (C.c_int, [C.POINTER(SaevCfg), C.c_int, C.POINTER(P)]),
    "saev_create_ex": (C.c_int, [C.POINTER(SaevCfg), C.POINTER(SaevDebugCfg), C.c_int, C.POINTER(P)]),
    "saev_destroy": (None, [P]),
    "saev_bind": (C.c_int, [P, P, P, P, P]),
    "saev_bind_tracker": (C.c_int, [P, P, P]),
    "saev_tracker_touched": (C.c_int, [P]),
    "saev_set_prefixes": (C.c_int, [P, P, C.c_int32]),
    "saev_share_x": (C.c_int, [P, P]),
    "saev_toks_since_active": (P, [P]),
    "saev_fired_flags": (P, [P]),
    "saev_stats_device": (P, [P]),
    "saev_read_stats": (C.c_int, [P, C.POINTER(SaevStepStats), P]),
    "saev_normalize_w_dec": (C.c_int, [P, P]),
    "saev_encode_dense": (C.c_int, [P, P, C.c_int32, P, P]),
    "saev_topk_dense": (C.c_int, [P, P, C.c_int32, C.c_int32, P, P, P, P]),
    "saev_encode_topk": (C.c_int, [P, P, C.c_int32, P, P, P]),
    "saev_scatter_dense": (C.c_int, [P, P, P, C.c_int32, C.c_int32, P, P]),
    "saev_decode_sparse": (C.c_int, [P, P, P, C.c_int32, C.c_int32, P, C.c_int32, P, P]),
    "saev_remove_parallel_grads": (C.c_int, [P, P]),
    "saev_gather_rows": (C.c_int, [P, P, P, C.c_int32, P, P]),
    "saev_step_forward": (C.c_int, [P, P, C.c_int32, C.c_int64, C.c_int32, P]),
    "saev_step_dead": (C.c_int, [P, C.c_int64, P]),
    "saev_last_aux_route": (C.c_int, [P]),
    "saev_scratch_bytes": (C.c_int64, [P, C.c_int32]),
    "saev_dead_readbacks": (C.c_int64, [P]),
    "saev_step_backward": (C.c_int, [P, P]),
    "saev_backward_begin": (C.c_int, [P, P]),
    "saev_backward_rows": (C.c_int, [P, C.c_int32, C.c_int32, P]),
    "saev_backward_rows_part": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, P]),
    "saev_backward_end": (C.c_int, [P, P]),
    "saev_copy_step_state": (C.c_int, [P, C.c_int32, P, P, P, P]),
    "saev_backward_override": (C.c_int, [P, P, P, P, P, C.c_int32]),
    "saev_aux_compact_rows": (C.c_int32, [P]),
    "saev_aux_compact_export": (C.c_int, [P, P, P]),
    "saev_aux_compact_import": (C.c_int, [P, P, P]),
    "saev_trust_gradients": (C.c_int, [P, C.c_int32]),
    "saev_grad_w_enc_t": (P, [P]),
    "saev_bind_w_enc_t": (C.c_int, [P, P]),
    "saev_step_tail": (C.c_int, [P, C.c_float, C.c_float, C.c_float, C.c_int64, P]),
    "saev_tail_prepare": (C.c_int, [P, C.c_int32, P]),
    "saev_tail_apply": (C.c_int, [P, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_int32, P]),
    "saev_sumsq_device": (P, [P]),
    "saev_bind_sumsq": (C.c_int, [P, P]),
    "saev_wdec_ready_event": (C.c_int, [P, P]),
    "saev_wenc_ready_event": (C.c_int, [P, P]),
    "saev_train_step": (C.c_int, [P, P, C.c_int32, C.c_float, C.c_float, C.c_int64, P]),
    "saev_train_step_gather": (C.c_int, [P, P, P, P, C.c_int32, C.c_float, C.c_float, C.c_int64, P]),
    "saev_params_touched": (C.c_int, [P]),
    "saev_comm_unique_id": (C.c_int, [P]),
    "saev_comm_init": (C.c_int, [P, P, C.c_int32, C.c_int32]),
    "saev_comm_world": (C.c_int, [P]),
    "saev_comm_destroy": (C.c_int, [P]),
    "saev_train_step_dp": (C.c_int, [P, P, C.c_int32, C.c_float, C.c_float, C.c_int64, P]),
    "saev_last_idx": (P, [P]),
    "saev_last_val": (P, [P]),
    "saev_last_x_hat": (P, [P]),
    "saev_copy_last": (C.c_int, [P, C.c_int32, P, P, P, P]),
    "saev_bound_state": (C.c_int, [P, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_float), P]),
    "saev_enable_kernel_timing": (C.c_int, [P, C.c_int32]),
    "saev_last_encoder_ms": (C.c_float, [P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load libsaev_amd.so and attach argument types.  Raises if the library is not built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise SaevError(
                f"{LIB_PATH} not found: build the HIP extension first (`make` in the repo root or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
            )
        # torch first: PyTorch-ROCm carries its own libamdhip64, and a process must hold ONE HIP runtime.  Loaded after torch, this
        # library's libamdhip64.so dependency resolves to the copy torch has mapped; loaded before it, the system's copy comes in
        # and torch's follows -- two runtimes, and every call here (hipSetDevice first of all) fails on pointers / devices of the
        # other one (`python __graft_entry__.py smoke`: build() loaded the library before smoke() imported torch).
        import torch  # noqa: F401

        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = lib.saev_abi_version()
        if got != ABI_VERSION:
            raise SaevError(f"libsaev_amd.so ABI version {got} != expected {ABI_VERSION}")
        _lib = lib
    return _lib


def check(lib: C.CDLL, ctx, rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.saev_last_error(ctx)
        raise SaevError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
