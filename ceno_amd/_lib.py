"""ctypes loader for libceno_hip.so (the C ABI declared in include/ceno_hip.h).

The product path has no CPU fallback: if the HIP extension is missing the import of the
library fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libceno_hip.so")
PROVER_LIB_PATH = os.environ.get("CENO_PROVER_LIB") or os.path.join(HERE, "libceno_prover.so")  # (override: the sanitizer build, ceno_amd/build.py --sanitize)

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
vpp = C.POINTER(C.c_void_p)


class SumcheckPlan(C.Structure):
    _fields_ = [
        ("num_mles", C.c_int),
        ("num_terms", C.c_int),
        ("term_coeffs", u64p),
        ("term_offsets", u32p),
        ("term_mle_idx", u32p),
        ("num_groups", C.c_int),
        ("group_term_offsets", u32p),
        ("group_term_idx", u32p),
        ("common_offsets", u32p),
        ("common_mle_idx", u32p),
        ("max_num_vars", C.c_int),
        ("max_degree", C.c_int),
    ]


_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libceno_hip.so; raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m ceno_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the prover kernels."
        )
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    sz = C.c_size_t
    i = C.c_int
    sig = {
        "ceno_hip_init": (i, [i, sz, vpp]),
        "ceno_hip_destroy": (None, [vp]),
        "ceno_hip_last_error": (C.c_char_p, [vp]),
        "ceno_hip_version": (C.c_char_p, []),
        "ceno_hip_make_current": (i, [vp]),
        "ceno_hip_device": (i, [vp]),
        "ceno_hip_stream_create": (i, [vp, vpp]),
        "ceno_hip_stream_create_lane": (i, [vp, i, vpp]),
        "ceno_hip_stream_destroy": (i, [vp, vp]),
        "ceno_hip_stream_adopt": (i, [vp, vp]),
        "ceno_hip_stream_bind": (i, [vp, vp]),
        "ceno_hip_selftest_field": (i, [vp, i, vp, C.c_size_t, u64p]),
        "ceno_hip_stream_sync": (i, [vp, vp]),
        "ceno_hip_mem_info": (i, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "ceno_hip_mem_trim": (i, [vp]),
        "ceno_hip_mem_peak": (sz, [vp, i]),
        "ceno_hip_mem_book": (i, [vp, sz]),
        "ceno_hip_mem_unbook": (i, [vp, sz]),
        "ceno_hip_mem_booked": (sz, [vp]),
        "ceno_hip_mem_booked_peak": (sz, [vp, i]),
        "ceno_hip_mle_alloc": (i, [vp, i, i, vpp]),
        "ceno_hip_mle_upload": (i, [vp, u64p, i, i, vp, vpp]),
        "ceno_hip_mle_wrap": (i, [vp, vp, i, i, vpp]),
        "ceno_hip_mle_view_chunk": (i, [vp, vp, i, sz, vpp]),
        "ceno_hip_mle_filter_even_odd": (i, [vp, vp, i, vp, vpp]),
        "ceno_hip_mle_download": (i, [vp, vp, u64p, vp]),
        "ceno_hip_mle_free": (i, [vp, vp]),
        "ceno_hip_mle_num_vars": (i, [vp]),
        "ceno_hip_mle_is_ext": (i, [vp]),
        "ceno_hip_mle_device_ptr": (vp, [vp]),
        "ceno_hip_mle_fill_splitmix": (i, [vp, vp, C.c_uint64, C.c_uint64, vp]),
        "ceno_hip_mle_fill_zero": (i, [vp, vp, vp]),
        "ceno_hip_mle_evaluate": (i, [vp, vp, u64p, u64p, vp]),
        "ceno_hip_mle_fix_variables": (i, [vp, vp, u64p, i, vp, vpp]),
        "ceno_hip_eq_build": (i, [vp, u64p, i, u64p, vp, vpp]),
        "ceno_hip_selector_build": (i, [vp, i, u64p, i, sz, sz, u32p, i, i, vp, vpp]),
        "ceno_hip_mle_evaluate_prefix_batch": (i, [vp, i, vpp, u64p, i, vp, u64p]),
        "ceno_hip_lincomb_base_batch": (i, [vp, i, u32p, vpp, u64p, vp, vpp, vpp]),
        "ceno_hip_selector_build_batch": (i, [vp, i, C.POINTER(i), C.POINTER(u64p), C.POINTER(i), C.POINTER(sz), C.POINTER(sz), vp, vpp]),
        "ceno_hip_rotation_next_base_mle": (i, [vp, vp, i, vp, vpp]),
        "ceno_hip_rotation_selector_build": (i, [vp, u64p, i, i, i, vp, vpp]),
        "ceno_hip_wit_infer": (i, [vp, vpp, i, u64p, u32p, u32p, i, u32p, i, i, vp, vpp]),
        "ceno_hip_sumcheck_begin": (i, [vp, vpp, C.POINTER(SumcheckPlan), vp, vpp]),
        "ceno_hip_sumcheck_begin_eq": (i, [vp, vpp, C.POINTER(SumcheckPlan), i, C.POINTER(i), C.POINTER(u64p), C.POINTER(sz), C.POINTER(sz), vp, vpp]),
        "ceno_hip_sumcheck_eq_components": (i, [vp]),
        "ceno_hip_stat_eq_launches": (C.c_uint64, [vp]),
        "ceno_hip_plan_report": (C.c_char_p, [vp]),
        "ceno_hip_ext_sum_blocks": (i, [vp, vp, i, C.c_size_t, vp, vp]),
        "ceno_hip_sumcheck_round": (i, [vp, vp, u64p, u64p]),
        "ceno_hip_sumcheck_round_dev": (i, [vp, vp, u64p, vp]),
        "ceno_hip_sumcheck_finish": (i, [vp, vp, u64p, u64p]),
        "ceno_hip_sumcheck_rounds_done": (i, [vp]),
        "ceno_hip_sumcheck_estimate_memory": (sz, [i, i, C.POINTER(i), i, i]),
        "ceno_hip_sumcheck_set_pipelined": (i, [vp, vp, i]),
        "ceno_hip_sumcheck_table": (i, [vp, vp, i, C.POINTER(u64p), C.POINTER(i), C.POINTER(i)]),
        "ceno_hip_sumcheck_table_host": (i, [vp, vp, i, u64p, C.c_size_t, C.POINTER(i)]),
        "ceno_hip_sumcheck_tables_host": (i, [vp, vp, i, C.POINTER(i), C.POINTER(u64p), C.POINTER(sz), C.POINTER(i)]),
        "ceno_hip_sumcheck_free": (i, [vp, vp]),
        "ceno_hip_tower_build_prod": (i, [vp, vpp, i, sz, u64p, vp, vpp]),
        "ceno_hip_tower_build_logup": (i, [vp, vpp, vpp, i, sz, u64p, vp, vpp]),
        "ceno_hip_tower_from_last_layer": (i, [vp, vpp, i, vp, vpp]),
        "ceno_hip_tower_num_vars": (i, [vp]),
        "ceno_hip_tower_layer": (i, [vp, vp, i, i, vpp]),
        "ceno_hip_tower_layer_ptr": (vp, [vp, i, i]),
        "ceno_hip_tower_build_many": (i, [vp, vp, i, vp, vpp]),
        "ceno_hip_wit_infer_many": (i, [vp, vp, i, vp]),
        "ceno_hip_tower_build_many_virtual": (i, [vp, vp, i, vp, i, vp, vpp]),
        "ceno_hip_tower_out_evals": (i, [vp, vp, u64p, vp]),
        "ceno_hip_tower_download_top": (i, [vp, vp, i, u64p, vp]),
        "ceno_hip_tower_top_layers": (i, [vp]),
        "ceno_hip_tower_prefetch_tops": (i, [vp, vpp, i, i, vp]),
        "ceno_hip_tower_num_limbs": (i, [vp]),
        "ceno_hip_tower_free": (i, [vp, vp]),
        "ceno_hip_tower_layer_sumcheck_begin": (i, [vp, vpp, i, vpp, i, i, u64p, u64p, vp, vpp]),
        "ceno_hip_ntt_batch": (i, [vp, vp, i, i, i, vp]),
        "ceno_hip_rs_encode": (i, [vp, vp, i, i, i, vp, vp]),
        "ceno_hip_transpose": (i, [vp, vp, sz, sz, vp, vp]),
        "ceno_hip_witgen_add": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_sub": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_addi": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_jal": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp, vp]),
        "ceno_hip_witgen_auipc": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp, vp]),
        "ceno_hip_witgen_slt": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_slti": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_branch_cmp": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_branch_eq": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_sumcheck_fused_eq_rounds": (i, [vp]),
        "ceno_hip_sumcheck_set_claim": (i, [vp, vp, u64p]),
        "ceno_hip_witgen_shift_r": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp, vp]),
        "ceno_hip_witgen_shift_i": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp, vp]),
        "ceno_hip_witgen_load_sub": (i, [vp, vp, i, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_mul": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_div": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_sh": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_sb": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_jalr": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_lw": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_sw": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_lui": (i, [vp, vp, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp]),
        "ceno_hip_witgen_logic_i": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp]),
        "ceno_hip_witgen_logic_r": (i, [vp, vp, i, vp, sz, vp, sz, C.c_uint64, C.c_uint32, C.c_uint32, vp, sz, vp, vp, vp, vp]),
        "ceno_hip_witgen_session_begin": (i, [vp, vpp, C.POINTER(sz), i, vp]),
        "ceno_hip_witgen_session_end": (i, [vp, vp]),
        "ceno_hip_lk_to_mlt_column": (i, [vp, vp, sz, vp, sz, vp]),
        "ceno_hip_tower_cohort_max_vars": (i, []),
        "ceno_hip_tower_cohort_capacity": (i, [vp]),
        "ceno_hip_tower_cohort_begin": (i, [vp, vp, i, vp, vpp]),
        "ceno_hip_tower_cohort_open": (i, [vp, vp, i, vp, vpp]),
        "ceno_hip_tower_cohort_set_job": (i, [vp, i, vp]),
        "ceno_hip_tower_cohort_launch": (i, [vp, vp]),
        "ceno_hip_tower_cohort_try_message": (i, [vp, i, i, u64p]),
        "ceno_hip_tower_cohort_send_challenge": (i, [vp, i, i, u64p]),
        "ceno_hip_tower_cohort_try_final": (i, [vp, i, u64p]),
        "ceno_hip_tower_cohort_round_times": (i, [vp, i, i, u64p]),
        "ceno_hip_tower_cohort_abort": (i, [vp]),
        "ceno_hip_tower_cohort_end": (i, [vp, vp]),
        "ceno_hip_poseidon2_set_constants": (i, [vp, u64p, u64p, u64p]),
        "ceno_hip_poseidon2_is_pinned": (i, [vp]),
        "ceno_hip_poseidon2_permute": (i, [vp, vp, sz, vp]),
        "ceno_hip_merkle_commit": (i, [vp, vp, i, i, vp, vpp]),
        "ceno_hip_merkle_root": (i, [vp, vp, u64p, vp]),
        "ceno_hip_merkle_open": (i, [vp, vp, sz, u64p, vp]),
        "ceno_hip_merkle_free": (i, [vp, vp]),
        "ceno_hip_batch_columns": (i, [vp, vp, sz, i, u64p, vp, i, vp]),
        "ceno_hip_batch_columns_multi": (i, [vp, i, vpp, C.POINTER(sz), C.POINTER(i), u64p, vpp, C.POINTER(i), vp]),
        "ceno_hip_basefold_fold_commit": (i, [vp, vp, i, u64p, vp, vp, vp, C.POINTER(vp)]),
        "ceno_hip_basefold_commit_codeword": (i, [vp, vp, i, vp, C.POINTER(vp)]),
        "ceno_hip_basefold_fold": (i, [vp, vp, i, u64p, vp, vp, vp]),
        "ceno_hip_gather": (i, [vp, vp, sz, i, i, vp, sz, i, i, vp, vp]),
        "ceno_hip_basefold_query_rounds": (i, [vp, C.POINTER(vp), C.POINTER(vp), i, vp, sz, vp, vp]),
        "ceno_hip_open_rounds_begin": (i, [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(i), vp, C.POINTER(vp)]),
        "ceno_hip_open_rounds_round": (i, [vp, vp, u64p, u64p]),
        "ceno_hip_open_rounds_finish": (i, [vp, vp, u64p, u64p]),
        "ceno_hip_open_rounds_done": (i, [vp]),
        "ceno_hip_open_rounds_free": (None, [vp, vp]),
        "ceno_hip_merkle_open_batch": (i, [vp, vp, vp, sz, i, vp, vp]),
        "ceno_hip_pow_grind_duplex": (i, [vp, u64p, i, u64p, vp]),
        "ceno_hip_lane_stream": (i, [vp, i, vpp]),
        "ceno_hip_debug_state": (i, [vp, C.POINTER(i), C.POINTER(i)]),
        "ceno_hip_host_timing_dump": (None, [C.c_char_p]),
        "ceno_hip_mmcs_commit": (i, [vp, C.POINTER(vp), C.POINTER(i), C.POINTER(i), i, vp, vpp]),
        "ceno_hip_mmcs_commit_over": (i, [vp, vp, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), i, vp, vpp]),
        "ceno_hip_mmcs_opening_words": (sz, [vp]),
        "ceno_hip_mmcs_open_batch": (i, [vp, vp, vp, sz, i, vp, sz, vp]),
        "ceno_hip_prof_reset": (i, [vp]),
        "ceno_hip_prof_enable": (i, [vp, i]),
        "ceno_hip_prof_get": (i, [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    L._ceno_missing = missing
    L._ceno_sig = sig
    _lib = L
    return L


def declared_symbols():
    """every function name include/ceno_hip.h declares (parsed from the header)"""
    import re

    hdr = os.path.join(os.path.dirname(HERE), "include", "ceno_hip.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ceno_hip_[a-z0-9_]+)\s*\(", txt)))
