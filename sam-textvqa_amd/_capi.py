"""ctypes binding of libsam_hip.so — the C-ABI declared in include/sam_hip.h.

The library is the product: if it is missing or a call fails this module raises; nothing here
(or anywhere in the package) falls back to a CPU or eager-PyTorch implementation."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libsam_hip.so")

_vp, _i, _i64, _u64, _f, _u = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_uint

class LnFuse(C.Structure):
    """mirror of `sam_ln_fuse` (include/sam_hip.h)"""
    _fields_ = [("gamma", _vp), ("beta", _vp), ("eps", _f), ("y", _vp), ("ldy", _i64), ("mean", _vp), ("rstd", _vp), ("done", C.c_int32),
                ("xws", _vp), ("xws_bytes", _i64)]


class GemmDesc(C.Structure):
    """mirror of `sam_gemm_desc` (include/sam_hip.h)"""
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("a_kcontig", C.c_int32), ("b_kcontig", C.c_int32),
                ("c_is_f32", C.c_int32), ("accumulate", C.c_int32), ("epilogue", C.c_int32),
                ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
                ("bias", _vp), ("residual", _vp), ("ldr", _i64),
                ("aux_out", _vp), ("aux_in", _vp), ("ld_aux", _i64),
                ("p_drop", _f), ("seed", _u64), ("offset", _u64), ("split_k", C.c_int32), ("bias_grad", _vp), ("ws", _vp), ("ws_bytes", _i64), ("force_tile", C.c_int32),
                ("defer_reduce", C.c_int32), ("split_k_used", C.c_int32), ("ln", C.POINTER(LnFuse))]


EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROPOUT_RES, EPI_DGELU, EPI_BIAS_GELU_GRAD, EPI_MUL_AUX = range(7)

# name -> argtypes (all return int status except where noted)
SIGNATURES = {
    "sam_attn_fwd": [_vp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _f, _u64, _u64, _vp, _vp, _vp, _vp],
    "sam_attn_fwd_rows": [_vp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp],
    "sam_attn_bwd": [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    "sam_attn_words_per_row": [_i],
    "sam_attn_fwd_train": [_vp, _vp, _i64, _i64, _i, _i, _i, _i, _f, _f, _u64, _u64, _vp, _vp, _vp, _vp, _vp],
    "sam_attn_bwd_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "sam_attn_bwd_fused_max_n": [],
    "sam_mask_bits_prefix_lm": [_vp, _i, _i, _i, _i, _vp, _vp],
    "sam_mask_bits_from_additive": [_vp, _i, _i, _i, _vp, _vp],
    "sam_mask_bits_spatial": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _u, _vp, _vp],
    "sam_abi_version": [],
    "sam_mask_bits_from_int8_bhnn": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "sam_spatial_relation_tensor": [_vp, _i, _i, _i, C.c_double, _vp, _vp],
    "sam_gemm_bf16": [C.POINTER(GemmDesc), _vp],
    "sam_gemm_splitk_reduce": [_vp, _i, _i, _i, _vp, _i64, _vp, _vp],
    "sam_gemm_bf16_grouped": [C.POINTER(GemmDesc), _i, _vp],
    "sam_gemm_grouped_ws_bytes": [C.POINTER(GemmDesc), _i],
    "sam_layernorm_fwd": [_vp, _i, _i64, _vp, _vp, _f, _i, _i, _vp, _i64, _vp, _vp, _vp],
    "sam_layernorm_bwd": [_vp, _i64, _vp, _i, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp, _i64, _f, _u64, _u64, _vp, _vp, _vp, _i, _vp, _vp],
    "sam_layernorm_bwd_ws_bytes": [_i],
    "sam_layernorm_bwd_partial_rows": [_i],
    "sam_layernorm_bwd_finalize_batch": [C.c_void_p, _i, _i, _vp],
    "sam_colsum_ws_bytes": [_i],
    "sam_colsum_bf16": [_vp, _i64, _i, _i, _vp, _i, _vp, _vp],
    "sam_bce_loss": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _i64, _vp, _i64, _vp],
    "sam_ptr_scores_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _i64, _i64, _vp],
    "sam_ptr_scores_bwd": [_vp, _i64, _i64, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    "sam_embedding_bwd": [_vp, _i64, _vp, _i, _i, _i, _i64, _vp, _i64, _vp, _vp],
    "sam_embedding_bwd_sorted": [_vp, _i64, _vp, _i, _i, _i, _i64, _vp, _i64, _vp, _vp],
    "sam_l2norm_pack_bf16": [_vp, _i64, _i, _i, _i, _f, _vp, _i64, _i, _i, _vp],
    "sam_embed_sum_fwd": [_vp, _i64, _vp, _i, _vp, _i64, _i, _vp, _i64, _vp, _i, _i, _i, _vp, _i64, _vp],
    "sam_embed_sum_bwd_ws_bytes": [_i, _i, _i],
    "sam_embed_sum_bwd": [_vp, _i64, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _i64, _vp, _vp],
    "sam_gather2_add_fwd": [_vp, _i64, _i, _vp, _i64, _i, _vp, _i, _i, _i, _vp, _i64, _f, _u64, _u64, _vp, _i64, _vp],
    "sam_gather2_add_bwd": [_vp, _i64, _i, _i, _vp, _i, _i, _i, _vp, _i64, _vp, _i64, _f, _u64, _u64, _vp, _i64, _vp],
    "sam_sumsq_ws_bytes": [],
    "sam_sumsq_f32": [_vp, _i64, C.c_void_p, _vp, _vp, _vp],
    "sam_adam_step": [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), C.POINTER(_f), _i, _f, _f, _f, _i64, _vp, _f, C.c_void_p, _vp],
    "sam_adam_step_dev": [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), _i, _f, _f, _f, _vp, _vp, _f, C.c_void_p, _vp],
    "sam_adam_step_range": [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64), _i, _f, _f, _f, _vp, _vp, _f, C.c_void_p, _i64, _i64, _i, _vp, _i, _vp],
    "sam_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "sam_pack_masks_u8": [_vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "sam_add_dropout_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _f, _u64, _u64, _vp],
    "sam_set_rng_state": [_vp],
    "sam_step_advance": [_vp, _u64, _vp, C.c_void_p, _vp, _vp],
    "sam_input_encoder_fwd": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _f, _u64, _u64, _vp, _i64, _vp, _vp],
    "sam_input_encoder_bwd_ws_bytes": [_i, _i],
    "sam_input_encoder_bwd": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _f, _u64, _u64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp],
    "sam_attn_fwd_dec": [_vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "sam_attn_fwd_dec_shared": [_vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "sam_attn_dec_row": [_vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _i64, _vp],
    "sam_greedy_pick": [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp],
    "sam_beam_step": [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sam_beam_step_split": [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sam_beam_step_ws_bytes": [_i, _i],
    "sam_greedy_decode_ws_bytes": [_i, _i, _i],
    "sam_copy_blocks": [C.c_void_p, _i, _vp],
    "sam_ge_u8": [_vp, _i64, _i64, _vp, _vp],
    "sam_greedy_decode_steps": [C.c_void_p, _vp, _i64, _vp],
    "sam_gemm_ln_ws_bytes": [_i, _i],
    "sam_attn_probs": [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp],
    "sam_rowvec_bf16": [_i, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i, _vp],
    "sam_set_cu_reserve": [_i],
    "sam_get_cu_reserve": [],
    "sam_debug_cu_hog": [_i, C.c_double, _vp],
}
NO_STATUS = {"sam_set_rng_state", "sam_get_cu_reserve", "sam_gemm_ln_ws_bytes", "sam_layernorm_bwd_partial_rows", "sam_gemm_grouped_ws_bytes", "sam_attn_words_per_row", "sam_attn_bwd_fused_max_n", "sam_abi_version", "sam_layernorm_bwd_ws_bytes", "sam_colsum_ws_bytes", "sam_sumsq_ws_bytes", "sam_embed_sum_bwd_ws_bytes", "sam_input_encoder_bwd_ws_bytes", "sam_greedy_decode_ws_bytes", "sam_beam_step_ws_bytes"}
RET_I64 = {"sam_gemm_grouped_ws_bytes", "sam_gemm_ln_ws_bytes", "sam_layernorm_bwd_ws_bytes", "sam_colsum_ws_bytes", "sam_sumsq_ws_bytes", "sam_embed_sum_bwd_ws_bytes", "sam_input_encoder_bwd_ws_bytes", "sam_greedy_decode_ws_bytes", "sam_beam_step_ws_bytes"}

_lib = None


class SparseRows(C.Structure):
    """mirror of `sam_sparse_rows` (include/sam_hip.h)"""
    _fields_ = [("lo", _i64), ("hi", _i64), ("row_len", C.c_int32), ("touched", _vp)]


class LrSchedule(C.Structure):
    """mirror of `sam_lr_schedule` (include/sam_hip.h)"""
    _fields_ = [("base_lr", C.c_double * 8), ("nseg", C.c_int32), ("warmup_iters", _i64), ("warmup_factor", C.c_double), ("n_decay", C.c_int32),
                ("decay_iters", _i64 * 4), ("lr_decay", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double)]


class CopyDesc(C.Structure):
    """mirror of `sam_copy_desc` (include/sam_hip.h)"""
    _fields_ = [("src", _vp), ("dst", _vp), ("batches", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("src_batch_stride", _i64), ("src_row_stride", _i64),
                ("dst_batch_stride", _i64), ("dst_row_stride", _i64), ("src_f32", C.c_int32), ("dst_f32", C.c_int32), ("accumulate", C.c_int32)]


class DecodeLayer(C.Structure):
    """mirror of `sam_decode_layer` (include/sam_hip.h)"""
    _fields_ = [(n, _vp) for n in ("wqkv", "wo", "w1", "w2", "bqkv", "bo", "b1", "b2", "ln1_g", "ln1_b", "ln2_g", "ln2_b", "qkv", "allow")] + \
               [("allow_stride_b", _i64), ("allow_stride_h", _i64)]


class DecodeDesc(C.Structure):
    """mirror of `sam_decode_desc` (include/sam_hip.h)"""
    _fields_ = [(n, C.c_int32) for n in ("n_layers", "B", "N", "n_enc", "S", "H", "D", "F", "V", "No", "t_begin", "t_end")] + \
               [(n, C.c_float) for n in ("scale", "ln_eps", "emb_ln_eps", "ptr_scale")] + \
               [("layers", C.POINTER(DecodeLayer)), ("pos_emb", _vp), ("type_emb", _vp), ("emb_ln_g", _vp), ("emb_ln_b", _vp), ("ld_pos", _i64), ("ld_type", _i64),
                ("ans_ln", _vp), ("ocr_ln", _vp), ("wc", _vp), ("bc", _vp), ("wq", _vp), ("bq", _vp), ("ptr_k", _vp), ("ocr_mask", _vp),
                ("prev_inds", _vp), ("fixed_scores", _vp), ("ld_fixed", _i64), ("ocr_scores", _vp), ("seq_out", _vp)]


class LnFinalizeItem(C.Structure):
    _fields_ = [("ws", _vp), ("rows", C.c_int32), ("accumulate", C.c_int32), ("dgamma", _vp), ("dbeta", _vp), ("dbias", _vp)]


class SamHipError(RuntimeError):
    pass


def lib():
    global _lib, LIB_PATH
    if _lib is None:
        alt = os.environ.get("SAM_HIP_LIB")          # tuning: load an alternative build of the same ABI (A/B runs inside one process tree)
        if alt:
            if not os.path.exists(alt):
                raise SamHipError("SAM_HIP_LIB=%s does not exist" % alt)
            LIB_PATH = alt
        # (re)build when the sources changed or the library is missing; a no-op (digest compare) otherwise.  A failed build raises: an older
        # libsam_hip.so is never loaded in place of the sources in the tree (changed signatures against an old binary = silent corruption).
        want = None
        if not alt:
            from . import _build
            try:
                _build.build()
            except Exception as e:
                raise SamHipError("libsam_hip.so could not be (re)built from the sources in the tree (%s); there is no fallback path" % e)
            want = _build._digest()
        if not os.path.exists(LIB_PATH):
            raise SamHipError("libsam_hip.so not built (%s): run `python __graft_entry__.py` or "
                              "sam_textvqa_amd._build.build(); there is no fallback path" % LIB_PATH)
        # torch bundles its own libamdhip64; load it FIRST so libsam_hip.so binds to the same HIP runtime instance that
        # owns torch's device allocations and streams (a second runtime copy sees "no ROCm-capable device")
        import torch
        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        l.sam_last_error.restype = C.c_char_p
        l.sam_build_digest.restype = C.c_char_p
        have = l.sam_build_digest().decode()
        if want is not None and have != want:
            raise SamHipError("libsam_hip.so was built from other sources (digest %s..., tree %s...): rebuild with `python __graft_entry__.py`" % (have[:12], want[:12]))
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _i64 if name in RET_I64 else _i
        _lib = l
    return _lib


profiler = None   # bench.py sets this to a list to collect (name, meta, start_event, end_event) per C-ABI call


def call(name, *args, meta=None):
    """invoke a status-returning entry point; non-zero -> SamHipError with the library's message"""
    l = lib()
    if profiler is not None and name not in NO_STATUS:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(l, name)(*args)
        e1.record()
        profiler.append((name, meta or {}, e0, e1))
    else:
        rc = getattr(l, name)(*args)
    if name not in NO_STATUS and rc != 0:
        raise SamHipError("%s failed (rc=%d): %s" % (name, rc, l.sam_last_error().decode()))
    return rc


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = None


def stream_handle():
    """hipStream_t of torch's current stream on the current device (the raw-handle getter: ~1 us instead of ~10 us for building a
    torch.cuda.Stream object, 500 times per step)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        get, dev = getattr(torch._C, "_cuda_getCurrentRawStream", None), getattr(torch._C, "_cuda_getDevice", None)
        _raw_stream = (lambda: get(dev())) if get and dev else (lambda: torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(_raw_stream())
