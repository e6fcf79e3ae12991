"""ctypes binding of libcv2amd.so (the C ABI declared in include/cv2_amd.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CV2_AMD_LIB') or os.path.join(_HERE, 'libcv2amd.so')   # CV2_AMD_LIB: diagnostic builds (tools/)
ABI_VERSION = 2          # include/cv2_amd.h: CV2_ABI_VERSION the ctypes mirrors of this package (lib.py, llm.py, flow.py, hift.py) were written against
_lib = None

STATE_STRIDE = 16
(ST_POS, ST_STEP, ST_NOUT, ST_DONE, ST_MINLEN, ST_MAXLEN, ST_MODE, ST_FORCE, ST_SEED_LO, ST_SEED_HI, ST_ERR, ST_LAST, ST_BIMODE, ST_NEXTFILL,
 ST_WAIT) = range(15)


class Cv2Error(RuntimeError):
    pass


class LlmDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('hidden', 'inter', 'layers', 'n_q', 'n_kv', 'vocab', 'vocab_pad', 'eos',
                                         'max_seqs', 'max_pos', 'max_out')] + [('rms_eps', C.c_float), ('max_prefill_rows', C.c_int32),
                                                                        ('top_p', C.c_float), ('top_k', C.c_int32), ('win_size', C.c_int32), ('tau_r', C.c_float)]


class LlmLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('wqkv', 'bqkv', 'wo', 'wgu', 'wdown', 'ln1', 'ln2')]


class LlmWeights(C.Structure):
    _fields_ = [('layers', C.POINTER(LlmLayer))] + [(n, C.c_void_p) for n in
                                                    ('final_norm', 'wdec', 'bdec', 'speech_emb', 'rope_cos', 'rope_sin')]


class LlmIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('state', 'out_tokens', 'logits')]


def lib():
    """Load the shared library once.  Raises Cv2Error when it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Cv2Error(f'{LIB_PATH} not found: the HIP extension is not built; there is no CPU fallback')
        L = C.CDLL(LIB_PATH)
        L.cv2_last_error.restype = C.c_char_p
        got = L.cv2_version()
        if got != ABI_VERSION:              # a stale build: its structs differ from the mirrors here (pointers would be misread silently)
            raise Cv2Error(f'{LIB_PATH} has ABI revision {got}, this package needs {ABI_VERSION}: rebuild it (python __graft_entry__.py)')
        L.cv2_llm_workspace_bytes.restype = C.c_size_t
        L.cv2_llm_workspace_bytes.argtypes = [C.POINTER(LlmDims)]
        L.cv2_llm_create.argtypes = [C.POINTER(LlmDims), C.POINTER(LlmWeights), C.POINTER(LlmIO), C.c_void_p, C.c_size_t,
                                     C.POINTER(C.c_void_p)]
        L.cv2_llm_destroy.argtypes = [C.c_void_p]
        L.cv2_llm_prefill.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.cv2_llm_decode.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cv2_llm_decode_ex.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.cv2_llm_decode_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.cv2_llm_one_launch_step.argtypes = [C.c_void_p]
        L.cv2_llm_debug_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.cv2_llm_debug_skip_publish.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.cv2_llm_debug_sample.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.cv2_flow_debug_attn_dma.argtypes = [C.c_int32]
        L.cv2_flow_debug_graph.argtypes = [C.c_int32]
        L.cv2_hift_debug_modes.argtypes = [C.c_int32, C.c_int32]
        L.cv2_hift_debug_precision.argtypes = [C.c_int32]
        L.cv2_hift_debug_f0.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.cv2_llm_extend.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cv2_llm_prefill_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
        L.cv2_llm_extend_batch.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
        L.cv2_skinny_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise Cv2Error(lib().cv2_last_error().decode())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'cv2amd expects contiguous tensors'
    return C.c_void_p(t.data_ptr())


EXPORTS = ['cv2_last_error', 'cv2_version', 'cv2_llm_workspace_bytes', 'cv2_llm_create', 'cv2_llm_destroy',
           'cv2_llm_prefill', 'cv2_llm_prefill_batch', 'cv2_llm_extend', 'cv2_llm_extend_batch', 'cv2_llm_decode', 'cv2_llm_decode_ex', 'cv2_llm_decode_rows', 'cv2_llm_one_launch_step', 'cv2_llm_debug_ptrs', 'cv2_llm_debug_skip_publish', 'cv2_llm_debug_sample', 'cv2_skinny_gemm', 'cv2_gemm_bf16',
           'cv2_flow_workspace_bytes', 'cv2_flow_create', 'cv2_flow_destroy', 'cv2_flow_inference', 'cv2_flow_inference_chunk', 'cv2_flow_cache_bytes', 'cv2_flow_cache_copy', 'cv2_flow_debug_attn_dma', 'cv2_flow_debug_graph', 'cv2_flow_estimator',
           'cv2_flow_encoder', 'cv2_hift_workspace_bytes', 'cv2_hift_create', 'cv2_hift_destroy', 'cv2_hift_inference', 'cv2_hift_inference_batch', 'cv2_hift_debug_modes', 'cv2_hift_debug_precision', 'cv2_hift_debug_f0',
           'cv2_fade_in_out', 'cv2_interp_linear', 'cv2_melspec', 'cv2_resample', 'cv2_framefeat', 'cv2_whisper_post', 'cv2_sub_col_mean', 'cv2_dbg_act', 'cv2_dbg_pre']
