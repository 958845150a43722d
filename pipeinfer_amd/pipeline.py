"""Multi-process pipeline: one process per GPU.  This file is only the LAUNCHER: it reads the torchrun environment, agrees on a job name,
loads each rank's share of the model and then hands over to the library — the per-run work (header rings, metadata table, peer stores of the
activations, result ring, cancel table, the stage loop with its sender thread) is csrc/engine/transport.cpp behind include/pipeinfer.h
(`pipeinfer_transport_*`).  No Python and no torch.distributed call happens per run or per hop; torch.distributed (gloo) is used for the
rendezvous, for barriers around the timed region and to gather the stages' statistics.

Replaces the reference's MPI plumbing for the speculative hot path (ggml-mpi.c:186-310 `ggml_mpi_sync_pipelined[_back]`, :541-557 layer ranges,
:602-700 graph split; llama.cpp:5588-5760 phase 0/1 of llama_decode_internal; speculative.cpp:65-120 communicator split) with a layout
that fits one node of MI355X:

    rank 0          HEAD: the PipeInfer driver (C++, pipeinfer_speculative), the draft model on GPU 0, and a *proxy* target context that
                    keeps the KV-cell table and writes one metadata block per run into the shared run table
    rank 1..N-1     target STAGES: contiguous layer ranges, each with the KV cache of its own layers; activations [n_tok][n_embd] fp32 are
                    stored by the producing stage into a ring in the next stage's HBM (hipIpc peer mapping, one xGMI copy per hop);
                    the last stage writes logits + top-8 into the result ring the head reads

There is no collective on the data path: the exchange is point-to-point, one hop per stage boundary per run, the reference's ring.
Throughput scaling comes from keeping >= N-1 runs in flight (continuous speculation), not from a shorter single run.  Early cancellation
(llama_cancel_run, llama.cpp:9981-9993) is an entry in a shared table: a stage drops a cancelled run before starting it or between two of
its layer segments and only forwards the header.

The compute of a stage is pipeinfer_stage_run_abortable on a stage context of the library; the CPU tests plug the oracle in through
`struct pipeinfer_stage_backend` to exercise the transport and the C++ driver without a GPU.
"""
import ctypes as C
import os
import sys
import time
import uuid

import numpy as np

# struct pipeinfer_remote_ops (include/pipeinfer.h): any other transport can sit behind these; the GPU stage test chains contexts in-process
START_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
CANCEL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)
BACKEND_RUN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)


class RemoteOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("start", START_FN), ("poll", POLL_FN), ("wait", WAIT_FN), ("cancel", CANCEL_FN)]


class TransportParams(C.Structure):
    _fields_ = [("name", C.c_char_p), ("rank", C.c_int32), ("world", C.c_int32), ("n_embd", C.c_int32), ("n_vocab", C.c_int32),
                ("n_batch", C.c_int32), ("n_ctx", C.c_int32), ("device", C.c_int32), ("timeout_ms", C.c_int32)]


class TransportStats(C.Structure):
    _fields_ = [("runs", C.c_uint64), ("skipped", C.c_uint64), ("cut_short", C.c_uint64), ("cancel_notes", C.c_uint64),
                ("wait_in_ms", C.c_double), ("compute_ms", C.c_double), ("send_ms", C.c_double),
                ("received", C.c_uint64), ("wait_out_ms", C.c_double), ("hop_gpu_ms", C.c_double), ("peer_access", C.c_int32), ("device_flags", C.c_int32),
                ("flag_runs", C.c_uint64), ("flag_wait_gpu_ms", C.c_double), ("flag_seen_us", C.c_double), ("hop_store_kernel", C.c_int32), ("ring_alloc", C.c_int32), ("passes_of_several_runs", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class StageBackend(C.Structure):
    _fields_ = [("user", C.c_void_p), ("run", BACKEND_RUN_FN)]


def split_layers(n_layer, n_stage, weights=None):
    """[(l0, l1)] per stage: ggml_mpi_split_range with the reference's weights (equal by default)"""
    from . import lib as L
    lib = L.load()
    w = (C.c_float * n_stage)(*(weights or [1.0 / n_stage] * n_stage))
    out = (C.c_int32 * (2 * n_stage))()
    lib.pipeinfer_split_layers.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    lib.pipeinfer_split_layers.restype = None
    lib.pipeinfer_split_layers(n_layer, w, n_stage, out)
    return [(out[2 * i], out[2 * i + 1]) for i in range(n_stage)]


def split_layers_balanced(n_layer, n_stage, tail_layers=1.5):
    """[(l0, l1)] per stage with (nearly) equal WORK: the reference's equal weights give the last stage the truncation remainder on top of
    final norm + lm_head (80 layers over 7 stages: 11 x 6 + 14), which then bounds the pipeline.  Here every stage gets floor/ceil of an even
    share after charging the last stage `tail_layers` layers' worth for the lm_head (215 MB of Q6_K vs ~500 MB per 70B layer: ~0.45; the
    default leaves slack for the top-k and the logits copy)."""
    share = (n_layer + tail_layers) / n_stage
    cuts, acc = [0], 0.0
    for i in range(n_stage - 1):
        acc += share
        cuts.append(min(n_layer, max(cuts[-1] + 1, int(round(acc)))))
    cuts.append(n_layer)
    return [(cuts[i], cuts[i + 1]) for i in range(n_stage)]


def job_name():
    """shared-memory name of this job: a nonce from rank 0 (so that a stale segment of an earlier job can never be picked up), agreed on
    through torch.distributed when it is up, else PIPEINFER_JOB"""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            box = [f"pipeinfer-{os.environ.get('MASTER_PORT', '0')}-{uuid.uuid4().hex[:12]}"]
            dist.broadcast_object_list(box, src=0)
            return box[0]
    except ImportError:
        pass
    return os.environ.get("PIPEINFER_JOB", "pipeinfer-job")


class Transport:
    """pipeinfer_transport_* (include/pipeinfer.h).  Opening is collective over the job's ranks."""

    def __init__(self, name, rank, world, n_embd, n_vocab, n_batch, n_ctx, device=-1, timeout_ms=0):
        from . import lib as L
        self.lib = lib = L.load()
        lib.pipeinfer_transport_open.argtypes = [C.POINTER(TransportParams)]
        lib.pipeinfer_transport_open.restype = C.c_void_p
        lib.pipeinfer_transport_close.argtypes = [C.c_void_p]
        lib.pipeinfer_transport_close.restype = None
        lib.pipeinfer_transport_error.argtypes = [C.c_void_p]
        lib.pipeinfer_transport_error.restype = C.c_char_p
        lib.pipeinfer_transport_device_mode.argtypes = [C.c_void_p]
        lib.pipeinfer_transport_attach_head.argtypes = [C.c_void_p, C.c_void_p]
        lib.pipeinfer_transport_shutdown.argtypes = [C.c_void_p]
        lib.pipeinfer_transport_get_stats.argtypes = [C.c_void_p, C.POINTER(TransportStats)]
        lib.pipeinfer_transport_get_stats.restype = None
        lib.pipeinfer_transport_serve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(TransportStats)]
        self.rank, self.world, self.n_embd, self.n_vocab, self.n_batch, self.n_ctx = rank, world, n_embd, n_vocab, n_batch, n_ctx
        p = TransportParams(name.encode(), rank, world, n_embd, n_vocab, n_batch, n_ctx, device, timeout_ms)
        self.h = lib.pipeinfer_transport_open(C.byref(p))
        if not self.h:
            raise RuntimeError(f"pipeinfer_transport_open failed on rank {rank}")
        self.device_mode = bool(lib.pipeinfer_transport_device_mode(self.h))
        self._keep = None

    def error(self):
        return self.lib.pipeinfer_transport_error(self.h).decode()

    def attach_head(self, ctx):
        if self.lib.pipeinfer_transport_attach_head(self.h, ctx.h) != 0:
            raise RuntimeError("pipeinfer_transport_attach_head: " + self.error())

    def shutdown(self):
        if self.lib.pipeinfer_transport_shutdown(self.h) != 0:
            raise RuntimeError("pipeinfer_transport_shutdown: " + self.error())

    def stats(self):
        st = TransportStats()
        self.lib.pipeinfer_transport_get_stats(self.h, C.byref(st))
        return st.as_dict()

    def serve(self, ctx=None, backend=None):
        """stage loop until the head shuts the pipeline down.  ctx: a stage Context of the library (the product), or backend: a callable
        run(n_tokens, meta int32[], x_in float32[n][n_embd] | None, x_out | None) -> None | (ids, vals, sums, logits)  (tests)."""
        st = TransportStats()
        be = None
        if backend is not None:
            n_embd, n_vocab, n_batch = self.n_embd, self.n_vocab, self.n_batch
            failure = []

            def run(user, n, meta, meta_bytes, x_in, x_out, logits, ids, vals, sums):
                try:
                    block = np.frombuffer(C.string_at(meta, meta_bytes), dtype=np.int32)
                    xi = np.ctypeslib.as_array(C.cast(x_in, C.POINTER(C.c_float)), (n, n_embd)) if x_in else None
                    xo = np.ctypeslib.as_array(C.cast(x_out, C.POINTER(C.c_float)), (n, n_embd)) if x_out else None
                    res = backend(n, block, xi, xo)
                    if res is not None and logits:
                        i8, v8, sm, lg = res
                        C.memmove(ids, np.ascontiguousarray(i8, np.int32).ctypes.data, 32 * n)
                        C.memmove(vals, np.ascontiguousarray(v8, np.float32).ctypes.data, 32 * n)
                        C.memmove(sums, np.ascontiguousarray(sm, np.float32).ctypes.data, 4 * n)
                        C.memmove(logits, np.ascontiguousarray(lg, np.float32).ctypes.data, 4 * n * n_vocab)
                    return 0
                except Exception as e:                   # noqa: BLE001 - never let an exception cross the C boundary
                    failure.append(repr(e))
                    print(f"[pipeline] stage backend failed: {e!r}", file=sys.stderr)
                    return -1
            be = StageBackend(None, BACKEND_RUN_FN(run))
            self._keep = be
        rc = self.lib.pipeinfer_transport_serve(self.h, ctx.h if ctx is not None else None, C.byref(be) if be is not None else None, C.byref(st))
        if rc != 0:
            raise RuntimeError("pipeinfer_transport_serve: " + self.error())
        return st.as_dict()

    def close(self):
        if self.h:
            self.lib.pipeinfer_transport_close(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------------------------------------------
# stage side
# ---------------------------------------------------------------------------------------------------------------------------
class HipStage:
    """layers [l0, l1) of a GGUF model on this process's GPU (pipeinfer_stage_run)"""

    def __init__(self, model_path, l0, l1, is_first, is_last, n_ctx, n_batch, device_index=0, seg_layers=0):
        from . import llama as LL
        self.LL = LL
        self.model = LL.Model(model_path, device=device_index)
        self.ctx = LL.Context(self.model, n_ctx=n_ctx, n_batch=n_batch)
        lib = self.ctx.lib
        lib.pipeinfer_context_set_stage.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        lib.pipeinfer_stage_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if lib.pipeinfer_context_set_stage(self.ctx.h, l0, l1, int(is_first), int(is_last)) != 0:
            raise RuntimeError(lib.pipeinfer_context_error(self.ctx.h).decode())
        if seg_layers > 0:
            self.ctx.set_segments(seg_layers)          # a cancelled run then stops between segments (pipeinfer_stage_run_abortable)
        self.ctx.prepare()
        self.lib, self.is_first, self.is_last = lib, is_first, is_last
        self.n_vocab = self.model.n_vocab
        lib.llama_n_embd.argtypes = [C.c_void_p]
        self.n_embd = lib.llama_n_embd(self.model.h)
        self.n_batch = n_batch
        self.logits = np.zeros((n_batch, self.n_vocab), np.float32)
        self.ids = np.zeros((n_batch, 8), np.int32)
        self.vals = np.zeros((n_batch, 8), np.float32)
        self.sums = np.zeros(n_batch, np.float32)

    def run(self, n_tokens, meta, x_in, x_out):
        """meta: int32 numpy block; x_in / x_out: device buffers or None.  Returns (ids, vals, sums, logits) on the last stage."""
        ptr = lambda b: None if b is None else (b.data_ptr() if hasattr(b, "data_ptr") else b.ptr)      # a torch tensor or any object with a device address in .ptr
        rc = self.lib.pipeinfer_stage_run(self.ctx.h, n_tokens, meta.ctypes.data, meta.nbytes, ptr(x_in), ptr(x_out),
                                          self.logits.ctypes.data, self.ids.ctypes.data, self.vals.ctypes.data, self.sums.ctypes.data)
        if rc != 0:
            raise RuntimeError("pipeinfer_stage_run: " + self.lib.pipeinfer_context_error(self.ctx.h).decode())
        if self.is_last:
            return self.ids[:n_tokens], self.vals[:n_tokens], self.sums[:n_tokens], self.logits[:n_tokens]
        return None

    def stats(self):
        return self.ctx.stats()

    def close(self):
        self.ctx.free()
        self.model.free()


RING_ALLOC = {0: "none", 1: "coarse-grained (hipMalloc)", 2: "fine-grained", 3: "uncached"}


def stage_family_tables(lib, backend, n_ctx, B):
    """A stage's share of the N > 1 roofline, the same accounting as N = 1 (bench.py: roofline_tables): launch counts and algorithmic bytes = this stage's captured graphs x
    their replays (every run it served), aborted launches from the device-side counters; durations = HIP-event samples of a graphs-off replay of the pass widths it served,
    on its own context and its own GPU, after the serve loop (synthetic tokens, n_kv = 256: the kernels and launch shapes are those of the runs).  Returns (counts, samples, aborted)."""
    ctx = backend.ctx
    counts = B.launch_rows(lib, ctx)
    aborted = B.aborted_launches(lib, ctx)
    by_width = {}
    for r in counts:
        by_width[r["n_tok"]] = by_width.get(r["n_tok"], 0) + r["launches"]
    widths = sorted(sorted(by_width, key=lambda w: -by_width[w])[:6])        # the six widths that carry most launches; the others are priced with the nearest one
    n_embd, words, n_kv = backend.n_embd, (n_ctx + 31) // 32, min(256, n_ctx - 64)
    x_in = lib.pi_malloc(4 * 64 * n_embd); x_out = lib.pi_malloc(4 * 64 * n_embd)
    lib.pi_memset(x_in, 0, 4 * 64 * n_embd, None); lib.pi_device_sync()
    ctx.set_graphs(0)
    lib.pi_kernel_timing(1, 3)
    try:
        for w in widths:
            if w > 64:
                continue
            meta = np.zeros(2 * w + 4 + w * words, np.int32)
            meta[:w] = 3 + np.arange(w); meta[w:2 * w] = n_kv - w + np.arange(w)
            meta[2 * w:2 * w + 4] = (n_kv - w, n_kv, w, 0)
            mask = np.zeros((w, words), np.uint32)
            for i in range(w):                                                # token i sees cells [0, n_kv - w + i]
                vis = n_kv - w + i + 1
                mask[i, :vis // 32] = 0xFFFFFFFF
                if vis % 32:
                    mask[i, vis // 32] = (1 << (vis % 32)) - 1
            meta[2 * w + 4:] = mask.view(np.int32).ravel()
            for _ in range(3):
                rc = lib.pipeinfer_stage_run(ctx.h, w, meta.ctypes.data, meta.nbytes, None if backend.is_first else x_in, None if backend.is_last else x_out,
                                             backend.logits.ctypes.data, backend.ids.ctypes.data, backend.vals.ctypes.data, backend.sums.ctypes.data)
                if rc != 0:
                    raise RuntimeError("pipeinfer_stage_run (roofline replay): " + lib.pipeinfer_context_error(ctx.h).decode())
        lib.pi_device_sync()
        samples = B.timing_rows(lib)
    finally:
        lib.pi_kernel_timing(0, 8)
        ctx.set_graphs(1)
        lib.pi_free(x_in); lib.pi_free(x_out)
    return counts, samples, aborted


# ---------------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N  (N > 1)
# ---------------------------------------------------------------------------------------------------------------------------
def bench_main(args, rank, world):
    """BASELINE configs[2]: Llama-2-70B Q4_K_M target layer-split over ranks 1..N-1, TinyLlama draft + driver on rank 0.
    One rank per GPU; rank 0 prints the JSON line.  Work per run is fixed, so scaling is strong."""
    import json
    import torch
    import torch.distributed as dist
    from . import lib as L, llama as LL
    from .tools import synth
    import bench as B

    # gloo only: rendezvous, barriers and the statistics gather.  torch never touches the GPU in this process (no import-order constraint
    # between torch's HIP runtime state and libpipeinfer_amd.so), and nothing below is called per run.
    if not dist.is_initialized():
        # gloo announces its mesh ("[Gloo] Rank 0 is connected to ...") on STDOUT, which belongs to the one JSON line: park fd 1 on stderr meanwhile
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    lib = L.load()
    n_dev = max(1, lib.pi_device_count())
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    shared_gpu = world > n_dev
    if shared_gpu and os.environ.get("PIPEINFER_ALLOW_SHARED_GPU") != "1":
        # one rank per GPU is the contract; doubling ranks up on a device silently would turn a scaling run into something else
        raise SystemExit(f"[bench rank {rank}] {world} ranks but only {n_dev} GPU(s) visible: refusing to share devices (PIPEINFER_ALLOW_SHARED_GPU=1 overrides, tests only)")
    local %= n_dev
    L.check(lib.pi_device_init(local), "pi_device_init")
    # the default workload of N > 1 is BASELINE configs[2] (70B target); PIPEINFER_KEEP_WORKLOAD=1 keeps an explicit --workload 7b
    tspec, tft, dspec, dft = B.WORKLOADS[args.workload]
    n_layer = synth.SPECS[tspec][1]
    n_stage = world - 1
    # the bench splits by work; PIPEINFER_REFERENCE_SPLIT=1 uses the reference's ggml_mpi_split_range with equal weights instead
    ref_split = bool(os.environ.get("PIPEINFER_REFERENCE_SPLIT"))
    ranges = split_layers(n_layer, n_stage) if ref_split else split_layers_balanced(n_layer, n_stage)
    cache = args.cache
    os.makedirs(cache, exist_ok=True)

    # the whole model's file, when an N = 1 run of this box has left it in the cache: every rank maps it and uploads only its own layer range (the loader is lazy per
    # range) — a second 41 GB of stage files would not fit the GPU box's /tmp beside it (the tensors are the same bytes: same seed, same generator)
    full_path = B.model_path(cache, tspec, tft, 0x5EED, args.signal, args.noise)
    use_full = os.path.exists(full_path)

    def stage_file(tag, only):
        if use_full:
            return full_path
        path = os.path.join(cache, f"{tspec}-{tft}-bigram-g{args.signal}-n{args.noise}-{tag}of{n_stage}.gguf")
        made.append(path)
        if not os.path.exists(path):
            t0 = time.time()
            synth.write_model(path + f".tmp{rank}", tspec, tft, seed=0x5EED, structure="bigram", noise=args.noise, signal=args.signal, only=only)
            os.replace(path + f".tmp{rank}", path)
            print(f"[bench rank {rank}] generated {path} in {time.time() - t0:.1f}s", file=sys.stderr)
        return path

    made = []                                       # stage files of THIS split: removed at the end (a 70B target is 41 GB per split)
    n_embd, n_vocab = synth.SPECS[tspec][2], synth.SPECS[tspec][6]
    if rank == 0:
        tgt_path = stage_file("head", lambda name: False)              # header + vocabulary only: the head never reads weights
        dft_path = B.ensure_model(cache, dspec, dft, 0xD4AF, args.signal, args.noise, flip=args.draft_miss)
        mt, md = LL.Model(tgt_path, device=local), LL.Model(dft_path, device=local)
        ct, cd = LL.Context(mt, n_ctx=args.n_ctx, n_batch=512), LL.Context(md, n_ctx=args.n_ctx, n_batch=512)
        cd.prepare()
    else:
        si = rank - 1
        l0, l1 = ranges[si]
        is_first, is_last = si == 0, si == n_stage - 1

        def mine(name):
            if name.startswith("blk."):
                return l0 <= int(name.split(".")[1]) < l1
            return (is_first and name.startswith("token_embd")) or (is_last and (name.startswith("output") or name.startswith("output_norm")))
        path = stage_file(f"stage{si}", mine)
        backend = HipStage(path, l0, l1, is_first, is_last, args.n_ctx, 512, device_index=local, seg_layers=int(os.environ.get("PIPEINFER_SEG_LAYERS", str(max(1, (l1 - l0 + 3) // 4)))))     # four segments per stage run: a cancelled run stops within a quarter of the stage
        if n_stage == 1:
            # --gpus 2: the one stage is the whole target — it carries the runs waiting in its ring in ONE pass over the weights, like the in-process target of N = 1
            # (engine feature, transport.cpp: run coalescing in the serve loop; every token keeps the bits it gets alone)
            backend.ctx.set_coalescing(int(os.environ.get("PIPEINFER_COALESCE", str(B.default_coalesce(args.n_draft)))))
    dist.barrier()                                  # (stage files of a 70B target take a while to write: open the transport together)
    tp = Transport(job_name(), rank, world, n_embd, n_vocab, 512, args.n_ctx, device=local)
    if rank == 0:
        tp.attach_head(ct)
    dist.barrier()

    if rank != 0:
        st = tp.serve(ctx=backend.ctx)
        stats = backend.stats()
        t = torch.tensor([stats["gpu_ms_1"], float(stats["n_calls_1"]), float(stats["alg_bytes_1"]), float(st["skipped"]), float(st["cut_short"]), st["wait_in_ms"], st["compute_ms"], st["send_ms"] + st["hop_gpu_ms"], float(st["runs"]), st["wait_out_ms"], float(st["peer_access"]),
                          float(st["device_flags"]), float(st["flag_runs"]), st["flag_wait_gpu_ms"], st["flag_seen_us"], float(st["hop_store_kernel"]), float(st["ring_alloc"]), float(st["passes_of_several_runs"])], dtype=torch.float64)    # single-token calls
        dist.gather(t, None, dst=0)
        # this stage's rows of the by-family roofline (counts of everything it served x durations of a graphs-off replay on its own GPU)
        try:
            rows = stage_family_tables(lib, backend, args.n_ctx, B)
        except Exception as e:                                                # the line still goes out; the head says which stage has no table
            print(f"[bench rank {rank}] no roofline table from this stage: {e!r}", file=sys.stderr)
            rows = None
        dist.gather_object(rows, None, dst=0)
        dist.barrier()
        tp.close()
        backend.close()
        if not os.environ.get("PIPEINFER_KEEP_STAGE_FILES"):
            for f in made:
                try:
                    os.remove(f)
                except OSError:
                    pass
        return

    spec = dict(n_predict=args.n_predict, n_draft=args.n_draft, n_parallel=args.n_parallel, p_accept=args.p_accept, p_split=args.p_split,
                ignore_eos=1, mode=args.mode, min_spec_iters=args.min_spec_iters, max_inflight=args.max_inflight, verbose=int(os.environ.get("PIPEINFER_VERBOSE", "0")),
                adaptive_window=args.adaptive_window, adaptive_pause=args.adaptive_pause)
    rng = np.random.default_rng(0xBEEF)
    prompts = [[int(x) for x in rng.integers(3, mt.n_vocab, args.n_prompt)] for _ in range(args.warmup + args.steps + 1)]

    def generation(i, **over):
        ct.kv_clear(); cd.kv_clear()
        kw = dict(spec); kw.update(over)
        return LL.speculative(ct, cd, prompts[i], **kw)

    ref_tokens, _ = generation(0, mode=2)
    spec_tokens, _ = generation(0)
    lossless = ref_tokens[:args.n_predict] == spec_tokens[:args.n_predict]
    first_diff = None if lossless else B.lossless_diagnostic(ct, prompts[0], ref_tokens, spec_tokens, args.n_predict)
    for i in range(args.warmup):
        generation(1 + i)
    L.check(lib.pi_device_sync())
    t_start = time.time()
    results = [generation(1 + args.warmup + i)[1] for i in range(args.steps)]
    L.check(lib.pi_device_sync())
    t_total = time.time() - t_start
    tp.shutdown()
    tp_device_mode = tp.device_mode
    head_stats = tp.stats()
    gathered = [torch.zeros(18, dtype=torch.float64) for _ in range(world)]
    dist.gather(torch.zeros(18, dtype=torch.float64), gathered, dst=0)
    stage_rows = [None] * world
    dist.gather_object(None, stage_rows, dst=0)
    dist.barrier()
    tp.close()
    if not os.environ.get("PIPEINFER_KEEP_STAGE_FILES"):
        for f in made:
            try:
                os.remove(f)
            except OSError:
                pass

    main_sum = B.summarize(results)
    main_sum["runs_skipped_by_stages"] = int(head_stats["skipped"])
    stage_stats = [{"rank": r, "layers": list(ranges[r - 1]), "calls": int(gathered[r][1]), "gpu_ms_per_call": round(float(gathered[r][0]) / max(1.0, float(gathered[r][1])), 4),
                    "GBps": round(float(gathered[r][2]) / max(1e-9, float(gathered[r][0]) * 1e-3) / 1e9, 1),
                    "runs_skipped": int(gathered[r][3]), "runs_cut_short": int(gathered[r][4]), "wait_for_input_ms": round(float(gathered[r][5]), 1),
                    "compute_ms": round(float(gathered[r][6]), 1), "hop_ms": round(float(gathered[r][7]), 3), "runs": int(gathered[r][8]),     # hop: GPU time of the in-stream store into the next stage's ring (last stage: logits into the result slot) + host time to publish the header
                    "wait_for_slot_ms": round(float(gathered[r][9]), 1), "peer_access": int(gathered[r][10]),
                    # device-side arrival flags: runs are announced early and enqueued behind a device wait; flag_seen_us = flag stamped by the producing stream -> seen by
                    # the waiting kernel (one device clock: meaningful between stages that share a device), wait_for_input_gpu_ms = GPU time spent waiting for input
                    "device_flags": bool(gathered[r][11]), "flag_runs": int(gathered[r][12]), "wait_for_input_gpu_ms": round(float(gathered[r][13]), 1),
                    "flag_seen_us_per_run": round(float(gathered[r][14]) / max(1.0, float(gathered[r][12])), 2) if gathered[r][12] > 0 else None,
                    # decided and logged at the transport's set-up, per rank: how this stage stores its activations into the next stage's ring, and what memory its own ring is
                    "hop_store": "kernel" if gathered[r][15] else "memcpy", "ring_alloc": RING_ALLOC.get(int(gathered[r][16]), "?"),
                    "passes_of_several_runs": int(gathered[r][17])} for r in range(1, world)]                      # (run coalescing on a stage that is the whole target: --gpus 2)
    slow = max(stage_stats, key=lambda s: s["gpu_ms_per_call"])
    # the same by-family roofline as N = 1: every stage's launch counts (graphs x replays, aborted launches apart) priced with the durations of a graphs-off replay on its
    # own GPU; families add up over the stages; the dominant family is the one the stages spent most GPU time in.  HBM traffic: the committed single-GPU PMC pass of the same
    # kernel sources (bytes fetched per algorithmic byte of a family do not depend on which stage launches it)
    per_stage, stage_inst = {}, {}
    for r in range(1, world):
        if stage_rows[r] is not None:
            per_stage[r], stage_inst[r] = B.roofline_tables(*stage_rows[r])
    by_family = B.merge_family_tables(list(per_stage.values()))
    dom_name = next(iter(by_family), None)
    tr, traffic_src = B.committed_traffic("70b" if args.workload == "70b" else args.workload, dom_name) if dom_name else (None, "no stage sent a table")
    sched = {0: "PipeInfer (asynchronous continuous speculation, examples/speculative)", 1: "synchronous tree speculation (examples/speculative_orig)", 2: "target only"}[args.mode]
    out = {"metric": B.METRIC if args.workload == "70b" else f"accepted tokens/sec + acceptance rate, {tspec} {tft} target / {dspec} draft", "value": main_sum["tokens_per_s"], "unit": "tokens/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_total / args.steps, 2), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int8 x int4/6 dot, fp32 accumulate (f16 KV)", "data": "synthetic",
           # how a stage boundary is crossed: "peer" = the producing stage stores into a ring in the consuming stage's HBM (hipIpc mapping, xGMI), "host" = rings in shared host memory
           "transport": "peer" if tp_device_mode else "host", "devices_shared": bool(shared_gpu),
           "hop_ms": {str(s_["rank"]): s_["hop_ms"] for s_ in stage_stats}, "hop_us_per_run": {str(s_["rank"]): round(1e3 * s_["hop_ms"] / max(1, s_["runs"]), 2) for s_ in stage_stats},
           "wait_for_input_ms": {str(s_["rank"]): s_["wait_for_input_ms"] for s_ in stage_stats},
           # how a run crosses a stage boundary: "device flag" = announced early, the consuming stage's run is already enqueued behind a device-side wait when the producing
           # stream writes the arrival flag behind its peer store (no host thread in between); "host header" = the header is published by the host after the run
           "hop": "device flag" if all(s_["device_flags"] for s_ in stage_stats) else "host header",
           "flag_seen_us_per_run": {str(s_["rank"]): s_["flag_seen_us_per_run"] for s_ in stage_stats},
           "wait_for_input_gpu_ms": {str(s_["rank"]): s_["wait_for_input_gpu_ms"] for s_ in stage_stats},
           "config": {"workload": f"{tspec} {tft} target layer-split over {n_stage} GPU(s) + {dspec} {dft} draft on GPU0", "schedule": sched,
                      "n_prompt": args.n_prompt, "n_predict": args.n_predict, "n_draft": args.n_draft, "n_parallel": args.n_parallel, "p_accept": args.p_accept,
                      "p_split": args.p_split, "n_ctx": args.n_ctx, "sampling": "greedy", "layer_ranges": [list(r) for r in ranges],
                      "layer_split": "reference ggml_mpi_split_range, equal weights" if ref_split else "balanced by work (lm_head charged to the last stage)",
                      "engine": {"stage_run_coalescing_max_tokens": int(os.environ.get("PIPEINFER_COALESCE", str(B.default_coalesce(args.n_draft)))) if n_stage == 1 else 0},
                      "synthetic_structure": f"bigram signal={args.signal} noise={args.noise}, draft wrong on {args.draft_miss:.0%} of the vocabulary"},
           "acceptance": {k: v for k, v in main_sum.items() if k != "tokens_per_s"},
           "lossless_vs_target_only": bool(lossless), "first_difference": first_diff,
           "roofline": None}
    if dom_name:
        dom = by_family[dom_name]
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernels"], "family": dom_name, "achieved": dom["achieved_GBps"], "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                           "traffic": round(tr["fetch_over_alg"] * dom["alg_bytes_per_launch"]) if tr else None, "traffic_fetch_over_alg": tr["fetch_over_alg"] if tr else None, "traffic_source": traffic_src,
                           "kernel_src_sha16": B.kernel_source_sha(), "avg_launch_us": dom["avg_launch_us"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"], "launches": dom["launches"],
                           "aborted_launches": dom["aborted_launches"], "share_of_gpu_time": dom["share_of_gpu_time"], "by_family": by_family,
                           "by_family_per_stage": {str(r): t for r, t in per_stage.items()}, "per_instance_per_stage": {str(r): t for r, t in stage_inst.items()},
                           "how": "per stage: launch counts and bytes of everything the stage served (captured graphs x replays, launches aborted on entry apart) x HIP-event durations of a graphs-off "
                                  "replay of the same pass widths on the stage's own GPU after the serve loop; families summed over the stages" + (" (ranks share one device here: the replays ran side by side)" if shared_gpu else ""),
                           "slowest_stage_decode_call": {"rank": slow["rank"], "gpu_ms_per_call": slow["gpu_ms_per_call"], "GBps": slow["GBps"], "frac": round(slow["GBps"] / B.HBM_PEAK_GBS, 4)},
                           "stages": stage_stats}
    else:
        out["roofline"] = {"bound": "hbm", "kernel": None, "achieved": slow["GBps"], "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(slow["GBps"] / B.HBM_PEAK_GBS, 4), "traffic": None,
                           "note": "no stage sent a launch table: achieved = the slowest stage's single-token calls (whole stage call)", "stages": stage_stats}
    if not args.no_cpu_baseline:
        try:
            full = B.ensure_model(cache, tspec, tft, 0x5EED, args.signal, args.noise) if n_layer <= 32 else None
            out["cpu_baseline"] = B.cpu_baseline(tspec, tft, full, cache, args.signal, args.noise, args.cpu_tokens)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": repr(e)}
    B.emit(out, getattr(args, "detail", None))
