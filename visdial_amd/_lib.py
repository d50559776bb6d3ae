"""ctypes binding of libvisdial_hip.so (the C ABI declared in include/visdial_hip.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing or a
call fails this module raises -- it never routes to the oracle or to torch ops.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VD_LIB_PATH") or os.path.join(_HERE, "libvisdial_hip.so")   # override: A/B builds

ABI_VERSION = 2    # include/visdial_hip.h VD_ABI_VERSION this binding was written against

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_u64 = C.c_uint64

class Lstm2Fwd(C.Structure):
    _fields_ = [("T", C.c_int), ("N", C.c_int), ("tok_mask", C.c_void_p), ("Wh1", C.c_void_p), ("Wx2", C.c_void_p),
                ("b2", C.c_void_p), ("Wh2", C.c_void_p), ("gates1", C.c_void_p), ("h1", C.c_void_p),
                ("c1", C.c_void_p), ("gates2", C.c_void_p), ("h2", C.c_void_p), ("c2", C.c_void_p),
                ("nact", C.c_void_p)]


class Lstm2Bwd(C.Structure):
    _fields_ = [("T", C.c_int), ("N", C.c_int), ("Wh1", C.c_void_p), ("Wx2", C.c_void_p), ("Wh2", C.c_void_p),
                ("gates1", C.c_void_p), ("c1", C.c_void_p), ("gates2", C.c_void_p), ("c2", C.c_void_p),
                ("dh_last2", C.c_void_p), ("dh1_seq", C.c_void_p), ("dc1", C.c_void_p), ("dc2", C.c_void_p),
                ("nact", C.c_void_p)]


class ModelParams(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('vocabSize', 'embedSize', 'rnnHiddenSize', 'imgFeatureSize', 'imgSpatialSize',
                                         'commonEmbeddingSize', 'numAttentionLayers', 'maxQuesCount', 'numOptions')] + \
               [(k, C.c_float) for k in ('learningRate', 'lrDecayRate', 'minLRate')] + \
               [('seed', C.c_uint64), ('lstmBf16', C.c_int32), ('useStreams', C.c_int32), ('numLayers', C.c_int32),
                ('imgEmbedSize', C.c_int32), ('dropout', C.c_float)]


class Batch(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ('B', 'Tq', 'Th', 'To')] + \
               [(k, C.c_void_p) for k in ('ques_fwd', 'hist', 'img_feat', 'options', 'answer_ind')] + [('Ta', C.c_int32)] + \
               [(k, C.c_void_p) for k in ('answer_in', 'answer_out', 'option_in', 'option_out')]


# name -> argtypes  (return type is int unless listed in _RESTYPE)
PROTOTYPES = {
    "vd_last_error": [],
    "vd_abi_version": [],
    "vd_device_count": [C.POINTER(C.c_int)],
    "vd_set_device": [_i],
    "vd_device_info": [_i, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "vd_malloc": [C.POINTER(C.c_void_p), _l],
    "vd_free": [_p],
    "vd_memset": [_p, _i, _l, _p],
    "vd_memcpy_h2d": [_p, _p, _l, _p],
    "vd_memcpy_d2h": [_p, _p, _l, _p],
    "vd_memcpy_d2d": [_p, _p, _l, _p],
    "vd_stream_synchronize": [_p],
    "vd_copy_2d": [_p, _l, _p, _l, _l, _l, _p],
    "vd_mask_time_forward": [_p, _p, _p, _i, _i, _i, _p],
    "vd_mask_time_backward": [_p, _p, _p, _i, _i, _i, _p],
    "vd_log_softmax_rows": [_p, _l, _l, _i, _p],
    "vd_logsoftmax_nll": [_p, _l, _l, _i, _p, _p, _p, _i, _p],
    "vd_gemm_nt": [_p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _i, _p],
    "vd_gemm_nn": [_p, _l, _p, _l, _p, _p, _l, _i, _i, _i, _i, _p],
    "vd_gemm_tn_acc": [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p],
    "vd_gemm_tn_rows_acc": [_p, _l, _p, _p, _l, _p, _p, _l, _i, _i, _i, _p],
    "vd_colsum_acc": [_p, _l, _i, _i, _p, _p],
    "vd_lstm_forward": [_p, _l, _l, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vd_lstm_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vd_zero_inactive_rows": [_p, _l, _l, _i, _p, _i, _i, _p],
    "vd_lstm2_forward": [_p, _i, _i, _p],
    "vd_lstm2_backward": [_p, _i, _i, _p],
    "vd_lstm2_forward_flags": [_p, _i, _i, _i, _p],
    "vd_lstm2_backward_flags": [_p, _i, _i, _i, _p],
    "vd_embed_gather": [_p, _p, _p, _p, _l, _i, _f, _p],
    "vd_embed_scatter_acc": [_p, _p, _p, _p, _l, _i, _f, _p],
    "vd_token_sort": [_p, _l, _i, _p, _p, _p, _p],
    "vd_segment_rowsum_acc": [_p, _l, _p, _p, _l, _i, _p, _l, _p],
    "vd_dropout_mask": [_p, _l, _u64, _f, _p],
    "vd_dropout_apply": [_p, _p, _p, _l, _f, _p],
    "vd_tanh_backward": [_p, _p, _p, _l, _p],
    "vd_axpby": [_p, _p, _p, _l, _f, _f, _p],
    "vd_mn_attention_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vd_mn_attention_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vd_hrea_attention_forward": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vd_hrea_attention_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vd_rowdot_forward": [_p, _p, _p, _p, _i, _i, _p],
    "vd_rowdot_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _p],
    "vd_img_common_forward": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vd_img_att_forward": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vd_img_att_backward": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vd_img_tr_backward": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vd_img_common_wgrad": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p],
    "vd_score_ce": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "vd_ranks": [_p, _p, _i, _i, _p],
    "vd_clamp_adam": [_p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _f, _p],
    # model-level entry points (csrc/runtime.hip)
    "vd_model_create": [C.POINTER(ModelParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)],
    "vd_model_destroy": [_p],
    "vd_model_num_tensors": [_p],
    "vd_model_flat_size": [_p],
    "vd_model_tensor_info": [_p, _l, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "vd_model_flat_pointers": [_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)],
    "vd_model_stream": [_p],
    "vd_model_encoder_range": [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "vd_model_wait_encoder_grads": [_p, _p],
    "vd_comm_unique_id": [_p],
    "vd_comm_init": [_i, _i, _p],
    "vd_comm_info": [C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "vd_comm_destroy": [],
    "vd_comm_available": [C.POINTER(C.c_int)],
    "vd_comm_stats": [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int64)],
    "vd_comm_overlap_ms": [C.POINTER(C.c_float)],
    "vd_model_allreduce_grads": [_p],
    "vd_model_init_params": [_p, _u64],
    "vd_model_set_tensor": [_p, C.c_char_p, _p, _l],
    "vd_model_get_tensor": [_p, C.c_char_p, _i, _p, _l],
    "vd_model_set_training": [_p, _i],
    "vd_model_set_dropout_mask": [_p, C.c_char_p, _p, _l],
    "vd_model_upload_batch": [_p, C.POINTER(Batch)],
    "vd_model_forward_backward": [_p, _i],
    "vd_model_loss": [_p, C.POINTER(C.c_float)],
    "vd_model_retrieve": [_p],
    "vd_model_encode": [_p],
    "vd_model_decode_begin": [_p, _p, _i],
    "vd_model_decode_step": [_p, _p, _p],
    "vd_model_decode_select": [_p, _p, _i],
    "vd_model_update": [_p, _f],
    "vd_model_learning_rate": [_p, C.POINTER(C.c_double), _i],
    "vd_model_scores": [_p, _p, _l],
    "vd_model_ranks": [_p, _i, _p],
    "vd_model_option_rows": [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "vd_model_family_ms": [_p, C.POINTER(C.c_float)],
    "vd_model_synchronize": [_p],
}
_RESTYPE = {"vd_last_error": C.c_char_p, "vd_model_destroy": None, "vd_model_num_tensors": C.c_int64,
            "vd_model_flat_size": C.c_int64, "vd_model_stream": C.c_void_p}


class VisdialHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VisdialHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    if lib.vd_abi_version() != ABI_VERSION:
        raise VisdialHipError("%s exports ABI version %d, this binding was written against %d: rebuild the library (make -C visdial_amd/csrc)"
                              % (LIB_PATH, lib.vd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning ABI function; raise VisdialHipError with vd_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise VisdialHipError("%s failed (%d): %s" % (name, rc, lib.vd_last_error().decode()))
    return rc
