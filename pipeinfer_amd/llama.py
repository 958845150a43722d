"""Python mirror of the llama.h / pipeinfer.h boundary (ctypes over libpipeinfer_amd.so): same names, arguments and
return codes as the C API, so tests and bench.py read like callers of the reference's llama.h."""
import ctypes as C

import numpy as np

from . import lib as L

llama_token = C.c_int32
llama_pos = C.c_int32
llama_seq_id = C.c_int32


class llama_batch(C.Structure):
    _fields_ = [("n_tokens", C.c_int32), ("token", C.POINTER(C.c_int32)), ("embd", C.POINTER(C.c_float)), ("pos", C.POINTER(C.c_int32)),
                ("n_seq_id", C.POINTER(C.c_int32)), ("seq_id", C.POINTER(C.POINTER(C.c_int32))), ("logits", C.POINTER(C.c_int8)),
                ("all_pos_0", C.c_int32), ("all_pos_1", C.c_int32), ("all_seq_id", C.c_int32), ("batch_id", C.c_int32), ("max_n_seq", C.c_int32)]


class llama_model_params(C.Structure):
    _fields_ = [("n_gpu_layers", C.c_int32), ("main_gpu", C.c_int32), ("tensor_split", C.POINTER(C.c_float)), ("progress_callback", C.c_void_p),
                ("progress_callback_user_data", C.c_void_p), ("vocab_only", C.c_bool), ("use_mmap", C.c_bool), ("use_mlock", C.c_bool)]


class llama_context_params(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("n_ctx", C.c_uint32), ("n_batch", C.c_uint32), ("n_threads", C.c_uint32), ("n_threads_batch", C.c_uint32),
                ("rope_scaling_type", C.c_int8), ("rope_freq_base", C.c_float), ("rope_freq_scale", C.c_float), ("yarn_ext_factor", C.c_float),
                ("yarn_attn_factor", C.c_float), ("yarn_beta_fast", C.c_float), ("yarn_beta_slow", C.c_float), ("yarn_orig_ctx", C.c_uint32),
                ("mul_mat_q", C.c_bool), ("f16_kv", C.c_bool), ("logits_all", C.c_bool), ("embedding", C.c_bool)]


class pipeinfer_spec_params(C.Structure):
    _fields_ = [("n_predict", C.c_int32), ("n_draft", C.c_int32), ("n_parallel", C.c_int32), ("p_accept", C.c_float), ("p_split", C.c_float),
                ("p_recovery", C.c_float), ("p_decay", C.c_float), ("temp", C.c_float), ("ignore_eos", C.c_int32), ("mode", C.c_int32),
                ("max_inflight", C.c_int32), ("min_spec_iters", C.c_int32), ("verbose", C.c_int32),
                ("adaptive_window", C.c_int32), ("adaptive_pause", C.c_int32)]


class pipeinfer_spec_result(C.Structure):
    _fields_ = [("n_input", C.c_int32), ("n_predict", C.c_int32), ("n_drafted", C.c_int32), ("n_accept", C.c_int32), ("n_accept_clean", C.c_int32),
                ("n_drafted_tokens", C.c_int32), ("n_runs_spec", C.c_int32), ("n_runs_nonspec", C.c_int32), ("n_runs_canceled", C.c_int32),
                ("n_draft_paused", C.c_int32), ("n_draft_catchups", C.c_int32),
                ("t_enc_s", C.c_double), ("t_dec_s", C.c_double), ("ttft_s", C.c_double), ("itl_s", C.c_double), ("decode_tps", C.c_double), ("encode_tps", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class pipeinfer_stats(C.Structure):
    _fields_ = [("n_calls", C.c_uint64), ("gpu_ms", C.c_double), ("alg_bytes", C.c_uint64), ("weight_bytes", C.c_uint64), ("kv_bytes_per_cell", C.c_uint64),
                ("n_calls_1", C.c_uint64), ("gpu_ms_1", C.c_double), ("alg_bytes_1", C.c_uint64), ("n_segments", C.c_uint64), ("n_segments_skipped", C.c_uint64),
                ("n_cancel_runs", C.c_uint64), ("n_cancel_runs_started", C.c_uint64), ("cancel_gpu_ms", C.c_double),
                ("n_passes", C.c_uint64), ("n_pass_runs", C.c_uint64), ("n_pass_tokens", C.c_uint64), ("n_coalesced_passes", C.c_uint64), ("weight_copy_bytes", C.c_uint64), ("n_reaction_windows", C.c_uint64)]


_api = None


def api():
    global _api
    if _api is None:
        lib = L.load()
        vp, i32 = C.c_void_p, C.c_int32
        lib.llama_model_default_params.restype = llama_model_params
        lib.llama_context_default_params.restype = llama_context_params
        lib.llama_backend_init.argtypes = [C.c_bool]
        lib.llama_load_model_from_file.restype = vp
        lib.llama_load_model_from_file.argtypes = [C.c_char_p, llama_model_params]
        lib.llama_free_model.argtypes = [vp]
        lib.llama_new_context_with_model.restype = vp
        lib.llama_new_context_with_model.argtypes = [vp, llama_context_params]
        lib.llama_free.argtypes = [vp]
        lib.llama_get_model.restype = vp
        lib.llama_get_model.argtypes = [vp]
        for n in ("llama_n_ctx", "llama_n_vocab", "llama_n_ctx_train", "llama_n_embd", "llama_token_bos", "llama_token_eos", "llama_token_nl",
                  "llama_add_bos_token", "llama_node_id", "llama_get_kv_cache_used_cells", "llama_get_kv_cache_token_count", "pipeinfer_context_prepare"):
            getattr(lib, n).argtypes = [vp]
        lib.llama_decode.argtypes = [vp, llama_batch]
        lib.llama_start_async_decode.restype = vp
        lib.llama_start_async_decode.argtypes = [vp, llama_batch]
        lib.llama_finish_async_decode.argtypes = [vp, llama_batch, vp]
        lib.llama_mpi_iprobe.restype = C.c_bool
        lib.llama_mpi_iprobe.argtypes = [vp]
        lib.llama_cancel_run.argtypes = [vp, C.POINTER(i32), C.c_int]
        lib.llama_get_logits.restype = C.POINTER(C.c_float)
        lib.llama_get_logits.argtypes = [vp]
        lib.llama_get_logits_ith.restype = C.POINTER(C.c_float)
        lib.llama_get_logits_ith.argtypes = [vp, i32]
        lib.llama_kv_cache_clear.argtypes = [vp]
        lib.llama_kv_cache_seq_rm.argtypes = [vp, i32, i32, i32]
        lib.llama_kv_cache_seq_cp.argtypes = [vp, i32, i32, i32, i32]
        lib.llama_kv_cache_seq_keep.argtypes = [vp, i32]
        lib.pipeinfer_kv_cache_seq_rm_range.argtypes = [vp, i32, i32, i32, i32]
        lib.pipeinfer_kv_cache_seq_cp_range.argtypes = [vp, i32, i32, i32, i32, i32]
        lib.pipeinfer_kv_cache_seq_rm_cp_range.argtypes = [vp, i32, i32, i32, i32, i32]
        lib.llama_tokenize.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(i32), C.c_int, C.c_bool, C.c_bool]
        lib.llama_token_to_piece.argtypes = [vp, i32, C.c_char_p, C.c_int]
        lib.llama_token_get_text.restype = C.c_char_p
        lib.llama_token_get_text.argtypes = [vp, i32]
        lib.pipeinfer_context_error.restype = C.c_char_p
        lib.pipeinfer_context_error.argtypes = [vp]
        lib.pipeinfer_run_first_cell.argtypes = [vp]; lib.pipeinfer_run_first_cell.restype = C.c_int32
        lib.pipeinfer_get_topk.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pipeinfer_get_stats.argtypes = [vp, C.POINTER(pipeinfer_stats)]
        lib.pipeinfer_reset_stats.argtypes = [vp]
        lib.pipeinfer_spec_default_params.restype = pipeinfer_spec_params
        lib.pipeinfer_speculative.argtypes = [vp, vp, C.POINTER(pipeinfer_spec_params), C.POINTER(i32), i32, C.POINTER(i32), i32, C.POINTER(pipeinfer_spec_result)]
        lib.llama_print_timings.argtypes = [vp]
        # stand-alone kv cells
        lib.pi_kvcells_new.restype = vp
        lib.pi_kvcells_new.argtypes = [C.c_int]
        lib.pi_kvcells_free.argtypes = [vp]
        lib.pi_kvcells_decode.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        lib.pi_kvcells_seq_rm.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        lib.pi_kvcells_seq_cp.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.pi_kvcells_seq_keep.argtypes = [vp, C.c_int]
        lib.pi_kvcells_clear.argtypes = [vp]
        lib.pi_kvcells_seq_rm_range.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.pi_kvcells_seq_cp_range.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.pi_kvcells_seq_rm_cp_range.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        for n in ("pi_kvcells_head", "pi_kvcells_used", "pi_kvcells_n_attend"):
            getattr(lib, n).argtypes = [vp]
        lib.pi_kvcells_pos.argtypes = [vp, C.c_int]
        lib.pi_kvcells_has.argtypes = [vp, C.c_int, C.c_int]
        lib.pi_kvcells_nseq.argtypes = [vp, C.c_int]
        lib.pi_kvcells_mask.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        _api = lib
    return _api


class Batch:
    """owns the arrays of a llama_batch (what llama_batch_init + llama_batch_add build in the reference, common.cpp:991-1012)"""

    def __init__(self, tokens, pos, seq_ids, batch_id=0, logits=True):
        n = len(tokens)
        self.tok = np.asarray(tokens, np.int32)
        self.pos = np.asarray(pos, np.int32)
        self.nseq = np.array([len(s) for s in seq_ids], np.int32)
        self.rows = [np.asarray(s, np.int32) for s in seq_ids]
        self.ptrs = (C.POINTER(C.c_int32) * n)(*[r.ctypes.data_as(C.POINTER(C.c_int32)) for r in self.rows])
        self.lg = np.full(n, 1 if logits else 0, np.int8)
        b = llama_batch()
        b.n_tokens = n
        b.token = self.tok.ctypes.data_as(C.POINTER(C.c_int32))
        b.pos = self.pos.ctypes.data_as(C.POINTER(C.c_int32))
        b.n_seq_id = self.nseq.ctypes.data_as(C.POINTER(C.c_int32))
        b.seq_id = C.cast(self.ptrs, C.POINTER(C.POINTER(C.c_int32)))
        b.logits = self.lg.ctypes.data_as(C.POINTER(C.c_int8))
        b.batch_id = batch_id
        b.max_n_seq = max(1, int(self.nseq.max()) if n else 1)
        self.c = b


class Model:
    def __init__(self, path, device=0):
        self.lib = api()
        p = self.lib.llama_model_default_params()
        p.main_gpu = device
        self.h = self.lib.llama_load_model_from_file(path.encode(), p)
        if not self.h:
            raise L.PipeInferError(f"cannot load {path}")
        self.n_vocab = self.lib.llama_n_vocab(self.h)

    def tokenize(self, text, add_bos=True, special=False):
        raw = text.encode("utf-8")
        buf = (C.c_int32 * (len(raw) + 8))()
        n = self.lib.llama_tokenize(self.h, raw, len(raw), buf, len(buf), add_bos, special)
        if n < 0:                                    # -n = tokens needed (byte fallback can take several tokens per character), like common.cpp:1086-1094
            buf = (C.c_int32 * (-n))()
            n = self.lib.llama_tokenize(self.h, raw, len(raw), buf, len(buf), add_bos, special)
        assert n >= 0
        return list(buf[:n])

    def token_to_piece(self, tok):
        buf = C.create_string_buffer(64)
        n = self.lib.llama_token_to_piece(self.h, tok, buf, 64)
        if n < 0:                                      # -n = the length the piece needs (llama.cpp:10112-10150), like the reference's common.cpp helper
            buf = C.create_string_buffer(-n)
            n = self.lib.llama_token_to_piece(self.h, tok, buf, -n)
        return buf.raw[:max(n, 0)]

    def free(self):
        if self.h:
            self.lib.llama_free_model(self.h)
            self.h = None


class Context:
    def __init__(self, model, n_ctx=512, n_batch=512, logits_all=True):
        self.lib = api()
        self.model = model
        p = self.lib.llama_context_default_params()
        p.n_ctx, p.n_batch, p.logits_all = n_ctx, n_batch, logits_all
        self.h = self.lib.llama_new_context_with_model(model.h, p)
        if not self.h:
            raise L.PipeInferError("llama_new_context_with_model failed")
        self.n_vocab = model.n_vocab

    def set_priority(self, priority):
        self.lib.pipeinfer_context_set_priority.argtypes = [C.c_void_p, C.c_int]
        if self.lib.pipeinfer_context_set_priority(self.h, int(priority)) != 0:
            raise L.PipeInferError(self.lib.pipeinfer_context_error(self.h).decode())

    def set_cu_range(self, lo, hi):
        self.lib.pipeinfer_context_set_cu_range.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if self.lib.pipeinfer_context_set_cu_range(self.h, int(lo), int(hi)) != 0:
            raise L.PipeInferError(self.lib.pipeinfer_context_error(self.h).decode())

    def set_segments(self, layers_per_segment):
        self.lib.pipeinfer_context_set_segments.argtypes = [C.c_void_p, C.c_int]
        if self.lib.pipeinfer_context_set_segments(self.h, int(layers_per_segment)) != 0:
            raise L.PipeInferError(self.lib.pipeinfer_context_error(self.h).decode())

    def set_coalescing(self, max_tokens):
        self.lib.pipeinfer_context_set_coalescing.argtypes = [C.c_void_p, C.c_int]
        if self.lib.pipeinfer_context_set_coalescing(self.h, int(max_tokens)) != 0:
            raise L.PipeInferError(self.lib.pipeinfer_context_error(self.h).decode())

    def set_run_tap(self, fn):
        """fn(run_id, batch_id, n_tokens, meta int32 array (a copy), logits [n_tokens][n_vocab] (a copy) or None): at every start (logits None) and finish of a run"""
        TAP = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_float))
        nv = self.n_vocab

        def _tap(user, run_id, batch_id, n, meta, nbytes, logits):
            block = np.frombuffer(C.string_at(meta, (nbytes + 3) // 4 * 4), dtype=np.int32).copy()
            rows = np.ctypeslib.as_array(logits, shape=(n, nv)).copy() if logits else None
            fn(run_id, batch_id, n, block, rows)
        self._tap = TAP(_tap) if fn else None                                   # (kept alive with the context)
        self.lib.pipeinfer_context_set_run_tap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.pipeinfer_context_set_run_tap.restype = None
        self.lib.pipeinfer_context_set_run_tap(self.h, C.cast(self._tap, C.c_void_p) if fn else None, None)

    def set_graphs(self, enable):
        self.lib.pipeinfer_context_set_graphs.argtypes = [C.c_void_p, C.c_int]
        self.lib.pipeinfer_context_set_graphs.restype = None
        self.lib.pipeinfer_context_set_graphs(self.h, int(enable))

    def prepare(self):
        if self.lib.pipeinfer_context_prepare(self.h) != 0:
            raise L.PipeInferError(self.lib.pipeinfer_context_error(self.h).decode())

    def decode(self, tokens, pos, seq_ids, batch_id=0):
        b = Batch(tokens, pos, seq_ids, batch_id)
        rc = self.lib.llama_decode(self.h, b.c)
        if rc != 0:
            raise L.PipeInferError(f"llama_decode rc={rc}: {self.lib.pipeinfer_context_error(self.h).decode()}")
        return self.logits(len(tokens))

    def start(self, tokens, pos, seq_ids, batch_id=0):
        b = Batch(tokens, pos, seq_ids, batch_id)
        t = self.lib.llama_start_async_decode(self.h, b.c)
        if not t:
            raise L.PipeInferError(f"llama_start_async_decode failed: {self.lib.pipeinfer_context_error(self.h).decode()}")
        return (t, b)

    def finish(self, ticket):
        t, b = ticket
        assert self.lib.llama_finish_async_decode(self.h, b.c, t) == 1
        return self.logits(b.c.n_tokens)

    def cancel(self, batch_ids):
        arr = (C.c_int32 * len(batch_ids))(*batch_ids)
        self.lib.llama_cancel_run(self.h, arr, len(batch_ids))

    def iprobe(self):
        return bool(self.lib.llama_mpi_iprobe(self.h))

    def logits(self, n):
        p = self.lib.llama_get_logits(self.h)
        return np.ctypeslib.as_array(p, shape=(n, self.n_vocab)).copy()

    def topk(self, i):
        ids = (C.c_int32 * 8)(); lg = (C.c_float * 8)(); s = C.c_float()
        assert self.lib.pipeinfer_get_topk(self.h, i, ids, lg, C.byref(s)) == 0
        return list(ids), list(lg), s.value

    def seq_rm(self, s, p0, p1):
        self.lib.llama_kv_cache_seq_rm(self.h, s, p0, p1)

    def seq_cp(self, a, b, p0, p1):
        self.lib.llama_kv_cache_seq_cp(self.h, a, b, p0, p1)

    def seq_keep(self, s):
        self.lib.llama_kv_cache_seq_keep(self.h, s)

    def kv_clear(self):
        self.lib.llama_kv_cache_clear(self.h)

    def stats(self):
        s = pipeinfer_stats()
        self.lib.pipeinfer_get_stats(self.h, C.byref(s))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def reset_stats(self):
        self.lib.pipeinfer_reset_stats(self.h)

    def free(self):
        if self.h:
            self.lib.llama_free(self.h)
            self.h = None


def speculative(ctx_tgt, ctx_dft, prompt, **kw):
    """runs the PipeInfer driver; returns (generated tokens, result dict)"""
    lib = api()
    p = lib.pipeinfer_spec_default_params()
    for k, v in kw.items():
        setattr(p, k, v)
    prompt = np.asarray(prompt, np.int32)
    out = np.zeros(p.n_predict + 16, np.int32)
    r = pipeinfer_spec_result()
    n = lib.pipeinfer_speculative(ctx_tgt.h, ctx_dft.h if ctx_dft is not None else None, C.byref(p), prompt.ctypes.data_as(C.POINTER(C.c_int32)), prompt.size,
                                  out.ctypes.data_as(C.POINTER(C.c_int32)), out.size, C.byref(r))
    if n < 0:
        raise L.PipeInferError(f"pipeinfer_speculative rc={n}")
    return list(out[:n]), r.as_dict()
