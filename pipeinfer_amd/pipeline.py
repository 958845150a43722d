"""Multi-process pipeline: one process per GPU, torch.distributed as the transport (backend "nccl" = RCCL over xGMI for the
activations, gloo for the small host-side control messages).

Replaces the reference's MPI plumbing for the speculative hot path (ggml-mpi.c:186-310 `ggml_mpi_sync_pipelined[_back]`,
:541-557 layer ranges, :602-700 graph split; llama.cpp:5588-5760 phase 0/1 of llama_decode_internal; speculative.cpp:65-120
communicator split) with a layout that fits one node of MI355X:

    rank 0          HEAD: the PipeInfer driver (C++, pipeinfer_speculative), the draft model on GPU 0, and a *proxy* target
                    context that keeps the KV-cell table and ships one metadata block per run (include/pipeinfer.h)
    rank 1..N-1     target STAGES: contiguous layer ranges (pipeinfer_split_layers = --mpi-layer-split), each with the KV cache
                    of its own layers; activations [n_tok][n_embd] fp32 travel stage -> stage by send/recv on device memory;
                    the last stage returns logits + top-8 to the head

There is no collective on the data path: the exchange is point-to-point, one message per stage boundary per run, exactly the
reference's ring.  Throughput scaling comes from keeping >= N-1 runs in flight (continuous speculation), not from
a shorter single run.  Early cancellation (llama_cancel_run, llama.cpp:9981-9993) is a tiny message from the head to every
stage; a stage skips the layers of a cancelled run and only forwards its header.

The compute of a stage sits behind a two-method backend (`run`, `close`): `HipStage` is the product (pipeinfer_stage_run in
libpipeinfer_amd.so); the CPU tests plug the oracle in to exercise this file and the C++ driver without a GPU.
"""
import collections
import ctypes as C
import os
import queue
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

HDR_WORDS = 8
INLINE_WORDS = 2048                 # run messages are one fixed-size block: header + metadata when it fits (verify batches always do)
KIND_RUN, KIND_SHUTDOWN = 1, 2
TAG_RUN, TAG_RESULT, TAG_CANCEL, TAG_ACT = 11, 12, 13, 14
F_SKIPPED, F_LONG = 1, 2           # F_LONG: the metadata follows as a second message (prompt-sized batches)

# struct pipeinfer_remote_ops (include/pipeinfer.h)
START_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
CANCEL_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)


class RemoteOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("start", START_FN), ("poll", POLL_FN), ("wait", WAIT_FN), ("cancel", CANCEL_FN)]


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


def run_message(run_seq, batch_id, n_tokens, words, flags, gen, meta):
    """header + (when it fits) the metadata block of a run, as one fixed-size int32 message"""
    msg = torch.zeros(HDR_WORDS + INLINE_WORDS, dtype=torch.int32)
    long_ = words > INLINE_WORDS
    msg[:HDR_WORDS] = torch.tensor([KIND_RUN, run_seq, batch_id, n_tokens, words, flags | (F_LONG if long_ else 0), gen, 0], dtype=torch.int32)
    if not long_:
        msg[HDR_WORDS:HDR_WORDS + words] = meta[:words]
    return msg


def result_words(n_tokens, n_vocab):
    """one message per run from the last stage to the head: header | top-8 ids | their logits | sum exp | logits (32-bit words)"""
    return HDR_WORDS + n_tokens * (8 + 8 + 1 + n_vocab)


class Links:
    """the two process groups of a run: `ctrl` carries host tensors (gloo), `data` device tensors (nccl = RCCL) — on a CPU-only
    box both are the same gloo group and activations are host tensors"""

    def __init__(self, ctrl=None, data=None, device="cpu", via_host=False):
        self.ctrl, self.data, self.device, self.via_host = ctrl, data, device, via_host

    def send_ctrl(self, t, dst, tag):
        dist.send(t, dst, group=self.ctrl, tag=tag)

    def recv_ctrl(self, t, src, tag):
        dist.recv(t, src, group=self.ctrl, tag=tag)

    def irecv_ctrl(self, t, src, tag):
        return dist.irecv(t, src, group=self.ctrl, tag=tag)

    def send_data(self, t, dst):
        if t.is_cuda and self.via_host:                 # several ranks on one GPU (RCCL needs one GPU per rank): stage through the host
            dist.send(t.cpu(), dst, group=self.ctrl, tag=TAG_ACT)
        elif t.is_cuda:
            dist.send(t, dst, group=self.data)
            torch.cuda.current_stream().synchronize()
        else:
            dist.send(t, dst, group=self.ctrl, tag=TAG_ACT)

    def recv_data(self, t, src):
        if t.is_cuda and self.via_host:
            h = torch.empty(t.shape, dtype=t.dtype)
            dist.recv(h, src, group=self.ctrl, tag=TAG_ACT)
            t.copy_(h)
            torch.cuda.current_stream().synchronize()
        elif t.is_cuda:
            dist.recv(t, src, group=self.data)
            torch.cuda.current_stream().synchronize()       # the stage kernels run on the library's own stream
        else:
            dist.recv(t, src, group=self.ctrl, tag=TAG_ACT)


def init_links():
    """process groups from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)"""
    use_gpu = torch.cuda.is_available()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # more ranks than GPUs (single-GPU dry runs of the N > 1 path): everything over gloo, activations staged through the host
    shared_gpu = use_gpu and (world > torch.cuda.device_count() or os.environ.get("PIPEINFER_FORCE_GLOO"))
    if not dist.is_initialized():
        dist.init_process_group("nccl" if use_gpu and not shared_gpu else "gloo")
    if use_gpu:
        local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if shared_gpu:
            return Links(ctrl=None, data=None, device=f"cuda:{local}", via_host=True)
        links = Links(ctrl=dist.new_group(backend="gloo"), data=None, device=f"cuda:{local}")
        probe_data_path(links, int(os.environ.get("RANK", "0")), world)
        return links
    return Links(ctrl=None, data=None, device="cpu")


def probe_data_path(links, rank, world, timeout_s=120.0):
    """one small device tensor down the stage chain 1 -> 2 -> ... -> world-1 before any run: creates the RCCL point-to-point
    communicators outside the timed region and, if the device path is unusable on this node (exception or no answer), moves
    every rank to the host-staged path together (the agreement is an all-reduce on the gloo group)"""
    state = {"ok": True}

    def chain():
        try:
            t = torch.full((256,), float(rank), dtype=torch.float32, device=links.device)
            if rank >= 2:
                dist.recv(t, rank - 1, group=links.data)
                if t.is_cuda:
                    torch.cuda.synchronize()
                state["ok"] = bool(t[0].item() == float(rank - 1))
                t.fill_(float(rank))
            if 1 <= rank < world - 1:
                dist.send(t, rank + 1, group=links.data)
                if t.is_cuda:
                    torch.cuda.synchronize()
        except Exception as e:                           # noqa: BLE001 - any transport failure means "use the host path"
            print(f"[pipeline] rank {rank}: device send/recv probe failed ({type(e).__name__}: {e}); staging activations through the host", file=sys.stderr)
            state["ok"] = False

    th = threading.Thread(target=chain, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        print(f"[pipeline] rank {rank}: device send/recv probe timed out after {timeout_s:.0f} s; staging activations through the host", file=sys.stderr)
        state["ok"] = False
    flag = torch.tensor([1 if state["ok"] else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=links.ctrl)
    if int(flag[0]) == 0:
        links.via_host = True


# ---------------------------------------------------------------------------------------------------------------------------
# head side
# ---------------------------------------------------------------------------------------------------------------------------
class RemoteTarget:
    """transport behind the head's proxy context: implements struct pipeinfer_remote_ops over torch.distributed"""

    def __init__(self, links, first_stage, last_stage, stage_ranks, n_vocab):
        self.links, self.first, self.last, self.stages, self.n_vocab = links, first_stage, last_stage, list(stage_ranks), n_vocab
        self.pending = collections.deque()              # (run_seq, n_tokens, result buffer, arrived) in start order
        # a posted receive only completes inside wait() (gloo), so a helper thread waits on the receives in order and flags
        # arrival; llama_mpi_iprobe (poll) then is a flag test
        self.waitq = queue.Queue()
        self.waiter = threading.Thread(target=self._waiter, daemon=True)
        self.waiter.start()
        self.keep = []                                  # in-flight cancel sends
        self.sends = collections.deque()                # in-flight run messages (tensors must outlive the sends)
        self.n_started = self.n_results = self.n_skipped = self.n_cancel_msgs = 0
        self.gen = 0                                    # batch ids restart with every generation: cancel notes carry (generation, batch id)
        self.ops = RemoteOps(None, START_FN(self._start), POLL_FN(self._poll), WAIT_FN(self._wait), CANCEL_FN(self._cancel))

    def attach(self, ctx):
        lib = ctx.lib
        lib.pipeinfer_context_set_remote.argtypes = [C.c_void_p, C.c_void_p]
        lib.pipeinfer_context_set_remote.restype = None
        lib.pipeinfer_context_set_remote(ctx.h, C.byref(self.ops))

    def _waiter(self):
        while True:
            item = self.waitq.get()
            if item is None:
                return
            req, arrived = item
            req.wait()
            arrived.set()

    def _start(self, user, run_seq, batch_id, n_tokens, meta, meta_bytes):
        try:
            words = (meta_bytes + 3) // 4
            payload = torch.from_numpy(np.frombuffer(C.string_at(meta, words * 4), dtype=np.int32).copy())
            hdr = run_message(run_seq, batch_id, n_tokens, words, 0, self.gen, payload)
            # the receive for this run's result is posted BEFORE the run leaves: the last stage can then always deliver, so a
            # full pipeline drains instead of dead-locking against the head's next (blocking) send
            buf = torch.empty(result_words(n_tokens, self.n_vocab), dtype=torch.float32)
            arrived = threading.Event()
            self.waitq.put((self.links.irecv_ctrl(buf, self.last, TAG_RESULT), arrived))
            self.pending.append((run_seq, n_tokens, buf, arrived))
            # non-blocking: a gloo send only completes once the first stage has posted the matching receive, i.e. after it has
            # finished the previous run — the driver must keep drafting meanwhile (messages to one peer keep their order)
            w1 = dist.isend(hdr, self.first, group=self.links.ctrl, tag=TAG_RUN)
            w2 = dist.isend(payload, self.first, group=self.links.ctrl, tag=TAG_RUN) if words > INLINE_WORDS else None
            self.sends.append((w1, w2, hdr, payload))
            while self.sends and self.sends[0][0].is_completed() and (self.sends[0][1] is None or self.sends[0][1].is_completed()):
                self.sends.popleft()
            self.n_started += 1
            return 0
        except Exception as e:                          # never let an exception cross the C boundary
            print(f"[pipeline] start failed: {e!r}", file=sys.stderr)
            return -1

    def _poll(self, user):
        return int(bool(self.pending) and self.pending[0][3].is_set())

    def _wait(self, user, run_seq, logits, ids, vals, sums):
        try:
            seq, n, buf, arrived = self.pending.popleft()
            arrived.wait()
            h = buf[:HDR_WORDS].view(torch.int32).tolist()
            if h[1] != run_seq or seq != run_seq:
                raise RuntimeError(f"result for run {h[1]} while waiting for {run_seq}")
            computed = not (h[5] & F_SKIPPED)
            if computed:
                base = buf.data_ptr() + 4 * HDR_WORDS
                C.memmove(ids, base, 4 * 8 * n)
                C.memmove(vals, base + 4 * 8 * n, 4 * 8 * n)
                C.memmove(sums, base + 4 * 16 * n, 4 * n)
                C.memmove(logits, base + 4 * 17 * n, 4 * n * self.n_vocab)
                self.n_results += 1
            else:
                self.n_skipped += 1
            return 1 if computed else 0
        except Exception as e:
            print(f"[pipeline] wait failed: {e!r}", file=sys.stderr)
            return -1

    def _cancel(self, user, batch_id):
        if batch_id == -2:
            self.gen += 1
            return
        for r in self.stages:
            t = torch.tensor([self.gen, batch_id], dtype=torch.int32)
            self.keep.append((dist.isend(t, r, group=self.links.ctrl, tag=TAG_CANCEL), t))
            self.n_cancel_msgs += 1
        self.keep = [(w, t) for w, t in self.keep if not w.is_completed()]

    def shutdown(self):
        """stops the stage loops (after every started run has been waited for)"""
        assert not self.pending, "runs still in flight"
        for w1, w2, _, _ in self.sends:
            w1.wait()
            if w2 is not None:
                w2.wait()
        self.sends.clear()
        bye = torch.zeros(HDR_WORDS + INLINE_WORDS, dtype=torch.int32)
        bye[0] = KIND_SHUTDOWN
        self.links.send_ctrl(bye, self.first, TAG_RUN)
        for r in self.stages:                           # completes the cancel receive every stage keeps posted
            dist.send(torch.tensor([-1, -1], dtype=torch.int32), r, group=self.links.ctrl, tag=TAG_CANCEL)
        for w, _ in self.keep:
            w.wait()
        self.waitq.put(None)


# ---------------------------------------------------------------------------------------------------------------------------
# stage side
# ---------------------------------------------------------------------------------------------------------------------------
class HipStage:
    """layers [l0, l1) of a GGUF model on this process's GPU (pipeinfer_stage_run)"""

    def __init__(self, model_path, l0, l1, is_first, is_last, n_ctx, n_batch, device_index=0):
        from . import llama as LL
        self.LL = LL
        self.model = LL.Model(model_path, device=device_index)
        self.ctx = LL.Context(self.model, n_ctx=n_ctx, n_batch=n_batch)
        lib = self.ctx.lib
        lib.pipeinfer_context_set_stage.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        lib.pipeinfer_stage_run.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if lib.pipeinfer_context_set_stage(self.ctx.h, l0, l1, int(is_first), int(is_last)) != 0:
            raise RuntimeError(lib.pipeinfer_context_error(self.ctx.h).decode())
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
        """meta: int32 numpy block; x_in / x_out: device tensors or None.  Returns (ids, vals, sums, logits) on the last stage."""
        rc = self.lib.pipeinfer_stage_run(self.ctx.h, n_tokens, meta.ctypes.data, meta.nbytes,
                                          x_in.data_ptr() if x_in is not None else None, x_out.data_ptr() if x_out is not None else None,
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


def stage_loop(links, backend, head, prev, nxt, n_embd, n_vocab, n_batch):
    """serves runs until the head shuts the pipeline down.  prev == head on the first stage, nxt is None on the last."""
    is_first, is_last = prev == head, nxt is None
    dev = links.device
    x_in = None if is_first else torch.empty((n_batch, n_embd), dtype=torch.float32, device=dev)
    x_out = None if is_last else torch.empty((n_batch, n_embd), dtype=torch.float32, device=dev)
    cancelled, clock = set(), threading.Lock()

    def cancel_listener():                              # early cancellation (speculative.cpp:1184-1230 -> llama_cancel_run)
        cbuf = torch.zeros(2, dtype=torch.int32)
        while True:
            links.recv_ctrl(cbuf, head, TAG_CANCEL)
            if int(cbuf[1]) < 0:                        # the head's shutdown note
                return
            with clock:
                cancelled.add((int(cbuf[0]), int(cbuf[1])))
    listener = threading.Thread(target=cancel_listener, daemon=True)
    listener.start()
    hdr = torch.zeros(HDR_WORDS + INLINE_WORDS, dtype=torch.int32)
    n_runs = n_skipped = 0
    while True:
        links.recv_ctrl(hdr, prev, TAG_RUN)
        h = hdr[:HDR_WORDS].tolist()
        if h[0] == KIND_SHUTDOWN:
            if not is_last:
                links.send_ctrl(hdr, nxt, TAG_RUN)
            break
        _, run_seq, batch_id, n, words, flags, gen = h[:7]
        if flags & F_LONG:
            meta = torch.empty(words, dtype=torch.int32)
            links.recv_ctrl(meta, prev, TAG_RUN)
        else:
            meta = hdr[HDR_WORDS:HDR_WORDS + words].clone()
        skipped = bool(flags & F_SKIPPED)
        if not is_first and not skipped:
            links.recv_data(x_in[:n], prev)
        with clock:
            if (gen, batch_id) in cancelled and not os.environ.get("PIPEINFER_NO_SKIP"):
                skipped = True
        res = None
        if not skipped:
            res = backend.run(n, meta.numpy(), None if is_first else x_in[:n], None if is_last else x_out[:n])
            n_runs += 1
        else:
            n_skipped += 1
        if not is_last:
            out_msg = run_message(run_seq, batch_id, n, words, F_SKIPPED if skipped else 0, gen, meta)
            links.send_ctrl(out_msg, nxt, TAG_RUN)
            if words > INLINE_WORDS:
                links.send_ctrl(meta, nxt, TAG_RUN)
            if not skipped:
                links.send_data(x_out[:n], nxt)
        else:
            out_hdr = np.array([KIND_RUN, run_seq, batch_id, n, words, F_SKIPPED if skipped else 0, gen, 0], np.int32)
            msg = np.zeros(result_words(n, n_vocab), np.float32)            # fixed size per n: the head posted the receive already
            msg[:HDR_WORDS] = out_hdr.view(np.float32)
            if not skipped:
                ids, vals, sums, logits = res
                o = HDR_WORDS
                msg[o:o + 8 * n] = np.ascontiguousarray(ids, np.int32).reshape(-1).view(np.float32); o += 8 * n
                msg[o:o + 8 * n] = np.asarray(vals, np.float32).reshape(-1); o += 8 * n
                msg[o:o + n] = np.asarray(sums, np.float32).reshape(-1); o += n
                msg[o:] = np.asarray(logits, np.float32).reshape(-1)
            links.send_ctrl(torch.from_numpy(msg), head, TAG_RESULT)
    listener.join()
    return {"runs": n_runs, "skipped": n_skipped}


# ---------------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N  (N > 1)
# ---------------------------------------------------------------------------------------------------------------------------
def bench_main(args, rank, world):
    """BASELINE configs[2]: Llama-2-70B Q4_K_M target layer-split over ranks 1..N-1, TinyLlama draft + driver on rank 0.
    One rank per GPU; rank 0 prints the JSON line.  Work per run is fixed, so scaling is strong."""
    import json
    from . import lib as L, llama as LL
    from .tools import synth
    import bench as B

    links = init_links()
    local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, torch.cuda.device_count())
    lib = L.load()
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

    def stage_file(tag, only):
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
        remote = RemoteTarget(links, 1, world - 1, range(1, world), mt.n_vocab)
        remote.attach(ct)
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
        backend = HipStage(path, l0, l1, is_first, is_last, args.n_ctx, 512, device_index=local)
    dist.barrier(group=links.ctrl)

    if rank != 0:
        st = stage_loop(links, backend, 0, 0 if rank == 1 else rank - 1, None if rank == world - 1 else rank + 1, n_embd, n_vocab, 512)
        stats = backend.stats()
        t = torch.tensor([stats["gpu_ms_1"], float(stats["n_calls_1"]), float(stats["alg_bytes_1"]), 0.0], dtype=torch.float64)    # single-token calls
        dist.gather(t, None, dst=0, group=links.ctrl)
        dist.barrier(group=links.ctrl)
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
    remote.shutdown()
    gathered = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
    dist.gather(torch.zeros(4, dtype=torch.float64), gathered, dst=0, group=links.ctrl)
    dist.barrier(group=links.ctrl)
    if not os.environ.get("PIPEINFER_KEEP_STAGE_FILES"):
        for f in made:
            try:
                os.remove(f)
            except OSError:
                pass

    main_sum = B.summarize(results)
    main_sum["runs_skipped_by_stages"] = remote.n_skipped
    stage_stats = [{"rank": r, "layers": list(ranges[r - 1]), "calls": int(gathered[r][1]), "gpu_ms_per_call": round(float(gathered[r][0]) / max(1.0, float(gathered[r][1])), 4),
                    "GBps": round(float(gathered[r][2]) / max(1e-9, float(gathered[r][0]) * 1e-3) / 1e9, 1)} for r in range(1, world)]
    slow = max(stage_stats, key=lambda s: s["gpu_ms_per_call"])
    # dominant kernel = the single-token Q4_K mat-vec of every stage; its HBM traffic comes from the committed single-GPU PMC pass of the same
    # kernel sources (the per-launch traffic of a mat-vec does not depend on which stage launches it)
    traffic, traffic_src = B.committed_traffic(args.workload, "Q4_K_nt1")
    sched = {0: "PipeInfer (asynchronous continuous speculation, examples/speculative)", 1: "synchronous tree speculation (examples/speculative_orig)", 2: "target only"}[args.mode]
    out = {"metric": B.METRIC if args.workload == "70b" else f"accepted tokens/sec + acceptance rate, {tspec} {tft} target / {dspec} draft", "value": main_sum["tokens_per_s"], "unit": "tokens/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_total / args.steps, 2), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int8 x int4/6 dot, fp32 accumulate (f16 KV)", "data": "synthetic",
           "config": {"workload": f"{tspec} {tft} target layer-split over {n_stage} GPU(s) + {dspec} {dft} draft on GPU0, RCCL send/recv", "schedule": sched,
                      "n_prompt": args.n_prompt, "n_predict": args.n_predict, "n_draft": args.n_draft, "n_parallel": args.n_parallel, "p_accept": args.p_accept,
                      "p_split": args.p_split, "n_ctx": args.n_ctx, "sampling": "greedy", "layer_ranges": [list(r) for r in ranges],
                      "layer_split": "reference ggml_mpi_split_range, equal weights" if ref_split else "balanced by work (lm_head charged to the last stage)",
                      "synthetic_structure": f"bigram signal={args.signal} noise={args.noise}, draft wrong on {args.draft_miss:.0%} of the vocabulary"},
           "acceptance": {k: v for k, v in main_sum.items() if k != "tokens_per_s"},
           "lossless_vs_target_only": bool(lossless), "first_difference": first_diff,
           "roofline": {"bound": "hbm", "kernel": "k_mmvq_kq<Q4_K,NT=1> inside the slowest stage's forward pass", "achieved": slow["GBps"], "peak": B.HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(slow["GBps"] / B.HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "note": "achieved = algorithmic bytes of the slowest stage's single-token calls / their GPU time (whole stage call, not one kernel)", "stages": stage_stats}}
    if not args.no_cpu_baseline:
        try:
            full = B.ensure_model(cache, tspec, tft, 0x5EED, args.signal, args.noise) if n_layer <= 32 else None
            out["cpu_baseline"] = B.cpu_baseline(tspec, tft, full, cache, args.signal, args.noise, args.cpu_tokens)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": repr(e)}
    print(json.dumps(out))
