// Stage-to-stage transport of a one-node pipeline, inside the library (include/pipeinfer.h, "in-library transport").
//
// Replaces, for the speculative hot path, the reference's MPI message plumbing (ggml-mpi.c:186-310 ggml_mpi_sync_pipelined[_back]:
// per run one message of tokens / positions / sequence ids per hop, then the activations, then the logits back to rank 0;
// llama.cpp:9981-9993 llama_cancel_run -> ggml_mpi_cancel_run: one small message to every worker) with what one node of MI355X offers:
//
//   * run METADATA is not forwarded hop by hop: the head writes each run's block once into a table in a POSIX shared-memory segment
//     that every stage maps; what travels from stage to stage is a 32-byte header in a single-producer/single-consumer ring;
//   * ACTIVATIONS [n_tokens][n_embd] fp32 are stored by the producing stage straight into a ring in the consuming stage's HBM
//     (hipIpc memory handle opened once; one device-to-device copy over xGMI per hop on the producer's copy stream).  The consumer
//     never posts a receive: slot k of its ring is free as soon as it has consumed header k, so the next run's activations arrive
//     while it still computes the current one.  The producer reserves the slot before it starts the run and its compute stream stores the
//     activations there right behind its last kernel — "direct peer store + flag" (SURVEY 2.4): no staging buffer, no sender thread;
//   * LOGITS + the device top-8 of the last stage go into a result ring in the shared segment (D2H straight into the slot);
//   * CANCEL notes are entries of one shared append-only table — no message per stage; a stage looks at it before a run and,
//     through pipeinfer_stage_run_abortable, between the segments of the run it is computing.
//
// If the peer mapping cannot be set up (or the probe block does not arrive intact) every rank falls back, together, to activation
// rings in the shared segment (D2H + H2D per hop); processes without a GPU (the CPU tests, with the oracle as the stage backend) use
// the same host rings.  Nothing here is a collective: the exchange is point-to-point, one hop per stage boundary per run.
#include "engine.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_set>

namespace {

constexpr uint32_t TP_MAGIC = 0x54504950u;       // "PIPT"
constexpr int TP_MAX_WORLD = 64;
constexpr int TP_RUN_TABLE = 256;                // metadata blocks of runs in flight (started, result not yet consumed)
constexpr int TP_IN1_SLOTS = 256;                // headers the head may queue at the first stage (no activations attached)
constexpr int TP_X_SLOTS = 4;                    // activation slots per stage boundary
constexpr int TP_RES_SLOTS = 4;                  // result slots (last stage -> head)
constexpr int TP_CANCEL_CAP = 8192;
enum { KIND_RUN = 1, KIND_SHUTDOWN = 2 };
enum { F_SKIPPED = 1 };
enum ring_alloc_kind { RING_NONE = 0, RING_COARSE = 1, RING_FINEGRAINED = 2, RING_UNCACHED = 3 };
const char * ring_alloc_name(int k) { return k == RING_FINEGRAINED ? "fine-grained" : k == RING_UNCACHED ? "uncached" : k == RING_COARSE ? "coarse-grained (hipMalloc)" : "none"; }

struct alignas(64) ring_ctl {
    std::atomic<uint64_t> head;                  // messages produced
    char pad0[56];
    std::atomic<uint64_t> tail;                  // messages consumed
    char pad1[56];
};
struct msg_hdr { int32_t kind, run_seq, batch_id, n_tokens, flags, gen, pad0, pad1; };

struct stage_block {                             // index = rank
    ring_ctl in;                                 // producer: rank - 1 (the head for rank 1); consumer: rank
    msg_hdr slots[TP_IN1_SLOTS];                 // ranks >= 2 use the first TP_X_SLOTS
    struct alignas(16) xflag_t { std::atomic<uint32_t> value; uint32_t pad; uint64_t stamp; } xflag[TP_X_SLOTS], xprobe;      // arrival flags of this stage's activation slots (runtime.hip: pi_xflag); xprobe: the set-up probe of that path
    std::atomic<uint32_t> ipc_ready, probe_written, joined, setup_done, flags_done;
    std::atomic<uint32_t> probe2_armed;                   // coherence probe of the real hop (store + flag against a consumer that has the slot cached and its reader enqueued)
    std::atomic<int32_t> ring_alloc;             // how this rank's activation ring was allocated (ring_alloc_kind)
    std::atomic<int32_t> device;                 // HIP device ordinal of the rank's process (-1: none)
    char ipc_handle[64];                         // hipIpcMemHandle_t
};
struct shm_layout {
    std::atomic<uint32_t> magic;
    int32_t world, n_embd, n_vocab, n_batch, n_ctx;
    uint64_t meta_stride, x_slot_bytes, res_slot_bytes, off_meta, off_x, off_res, total;
    std::atomic<int32_t> fatal;                  // any rank sets it: every wait gives up
    std::atomic<int32_t> device_fail;            // the peer-store path is unusable on this node: host rings for everybody
    std::atomic<int32_t> flags_off;              // some rank cannot use device-side arrival flags: every rank publishes its headers from the host
    std::atomic<uint64_t> n_cancel;
    std::atomic<uint64_t> cancel[TP_CANCEL_CAP]; // (generation << 32) | batch id
    // what a coalescing stage sees of the head's driver thread (context.cpp: wait_for_reaction, for the in-process feeder): calls it has made, whether it is blocked waiting
    // for a result, and its call count right after it consumed its latest result
    std::atomic<uint64_t> head_epoch, head_epoch_after_consume;
    std::atomic<int32_t> head_waiting;
    ring_ctl res;
    stage_block st[TP_MAX_WORLD];
};
static_assert(sizeof(hipIpcMemHandle_t) <= 64, "hipIpcMemHandle_t does not fit its slot");

int64_t now_us() { return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct pipeinfer_transport {
    std::string name, error;
    int rank = 0, world = 0, n_embd = 0, n_vocab = 0, n_batch = 0, n_ctx = 0, device = -1, timeout_ms = 120000;
    bool creator = false, device_mode = false, set_hop_mode = false;
    uint8_t * base = nullptr; size_t total = 0;
    shm_layout * L = nullptr;
    // stage side
    float * my_ring = nullptr;                   // device mode: this stage's activation ring in its HBM (ranks >= 2)
    float * peer_ring = nullptr;                 // device mode: the next stage's ring, mapped through hipIpc
    hipStream_t copy_stream = nullptr;
    bool res_registered = false;
    // device-side arrival flags: the control block of the segment page-locked and mapped into this process's device address space
    bool flags_on = false, ctl_registered = false; uint8_t * ctl_dev = nullptr;
    void * flag_dev(int rank_of_consumer, uint64_t k) const { return ctl_dev + ((uint8_t *) &L->st[rank_of_consumer].xflag[k % TP_X_SLOTS] - base); }
    int peer_access = -1;                        // hipDeviceCanAccessPeer(this device, next stage's device); -1: not asked
    int ring_alloc = RING_NONE;                  // this stage's own ring
    int64_t device_wait_us = 600ll * 1000 * 1000; // bound of a run's device-side wait for its activations (its own, generous: a slow but healthy upstream must not kill the job; a cancel, a failure or a dead pipeline end the wait early)
    // head side
    int32_t gen = 0;
    uint64_t started = 0, consumed = 0;
    pipeinfer_transport_stats head_stats{};
    // cancel view of a stage (compute thread only)
    uint64_t cancel_cursor = 0; std::unordered_set<uint64_t> cancelled; uint64_t cur_key = ~0ull;
    std::vector<uint64_t> cur_keys;              // the members of the pass being computed (a stage that carries several runs at once)

    bool fail(const std::string & what) {
        error = what;
        fprintf(stderr, "[pipeinfer transport, rank %d] %s\n", rank, what.c_str());
        if (L) L->fatal.store(1);
        return false;
    }
    // spins, then sleeps in 50 us steps; false on timeout (timeout_ms < 0: none) or when any rank has declared the pipeline dead
    template <class F> bool wait_for(F pred, int tmo_ms) {
        for (int i = 0; i < 4000; i++) { if (pred()) return true; __builtin_ia32_pause(); }
        const int64_t t0 = now_us();
        while (true) {
            if (pred()) return true;
            if (L->fatal.load(std::memory_order_relaxed)) return false;
            if (tmo_ms >= 0 && now_us() - t0 > (int64_t) tmo_ms * 1000) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    int in_cap(int r) const { return r == 1 ? TP_IN1_SLOTS : TP_X_SLOTS; }
    uint8_t * meta_of(int32_t run_seq) const { return base + L->off_meta + (size_t) ((uint32_t) run_seq % TP_RUN_TABLE) * L->meta_stride; }
    float * host_x_slot(int r, uint64_t k) const { return (float *) (base + L->off_x + ((size_t) (r - 2) * TP_X_SLOTS + k % TP_X_SLOTS) * L->x_slot_bytes); }
    uint8_t * res_slot(uint64_t k) const { return base + L->off_res + (size_t) (k % TP_RES_SLOTS) * L->res_slot_bytes; }
    // fixed offsets inside a result slot, sized for n_batch tokens
    int32_t * res_ids(uint8_t * s) const { return (int32_t *) (s + 64); }
    float * res_vals(uint8_t * s) const { return (float *) (s + 64 + (size_t) n_batch * 32); }
    float * res_sums(uint8_t * s) const { return (float *) (s + 64 + (size_t) n_batch * 64); }
    float * res_logits(uint8_t * s) const { return (float *) (s + 64 + (size_t) n_batch * 68); }

    void scan_cancels() {
        const uint64_t n = L->n_cancel.load(std::memory_order_acquire);
        for (; cancel_cursor < n; cancel_cursor++) cancelled.insert(L->cancel[cancel_cursor % TP_CANCEL_CAP].load(std::memory_order_relaxed));
    }
    bool is_cancelled(int32_t g, int32_t batch_id) { scan_cancels(); return cancelled.count(((uint64_t) (uint32_t) g << 32) | (uint32_t) batch_id) > 0; }
};

namespace {

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool map_segment(pipeinfer_transport * tp, const pipeinfer_transport_params & P) {
    const size_t words = (size_t) (P.n_ctx + 31) / 32;
    const size_t meta_stride = round_up((size_t) P.n_batch * 8 + 16 + (size_t) P.n_batch * words * 4, 64);
    const size_t x_slot = round_up((size_t) P.n_batch * P.n_embd * 4, 4096);
    const size_t res_slot = round_up(64 + (size_t) P.n_batch * (68 + (size_t) P.n_vocab * 4), 4096);
    const size_t off_meta = round_up(sizeof(shm_layout), 4096);
    const size_t off_x = off_meta + round_up(meta_stride * TP_RUN_TABLE, 4096);
    const size_t off_res = off_x + x_slot * TP_X_SLOTS * (size_t) std::max(0, P.world - 2);
    const size_t total = off_res + res_slot * TP_RES_SLOTS;
    int fd = -1;
    if (tp->rank == 0) {
        (void) shm_unlink(tp->name.c_str());
        fd = shm_open(tp->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return tp->fail("shm_open(create " + tp->name + ") failed: " + strerror(errno));
        if (ftruncate(fd, (off_t) total) != 0) { close(fd); return tp->fail(std::string("ftruncate of the shared segment failed: ") + strerror(errno)); }
        tp->creator = true;
    } else {
        const int64_t t0 = now_us();
        while (true) {
            fd = shm_open(tp->name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t) sb.st_size >= total) break;      // pages are sparse: the size appears with the creator's ftruncate
            if (fd >= 0) { close(fd); fd = -1; }
            if (now_us() - t0 > (int64_t) tp->timeout_ms * 1000) return tp->fail("the head's shared segment " + tp->name + " did not appear");
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void * m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return tp->fail(std::string("mmap of the shared segment failed: ") + strerror(errno));
    tp->base = (uint8_t *) m; tp->total = total; tp->L = (shm_layout *) m;
    shm_layout * L = tp->L;
    if (tp->rank == 0) {
        L->world = P.world; L->n_embd = P.n_embd; L->n_vocab = P.n_vocab; L->n_batch = P.n_batch; L->n_ctx = P.n_ctx;
        L->meta_stride = meta_stride; L->x_slot_bytes = x_slot; L->res_slot_bytes = res_slot; L->off_meta = off_meta; L->off_x = off_x; L->off_res = off_res; L->total = total;
        L->magic.store(TP_MAGIC, std::memory_order_release);
    } else {
        const int64_t t0 = now_us();
        while (L->magic.load(std::memory_order_acquire) != TP_MAGIC) {
            if (now_us() - t0 > (int64_t) tp->timeout_ms * 1000) return tp->fail("the shared segment was never initialised by the head");
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (L->world != P.world || L->n_embd != P.n_embd || L->n_vocab != P.n_vocab || L->n_batch != P.n_batch || L->n_ctx != P.n_ctx || L->total != total)
            return tp->fail("this rank's transport parameters differ from the head's");
    }
    return true;
}

// peer-store path: ranks >= 2 export a ring in their HBM, their upstream neighbour maps it and sends a probe block through it
void setup_device_rings(pipeinfer_transport * tp) {
    shm_layout * L = tp->L;
    const int r = tp->rank, W = tp->world;
    auto bad = [&](const char * what, hipError_t e) { fprintf(stderr, "[pipeinfer transport, rank %d] %s: %s — activations will be staged through host memory\n", r, what, hipGetErrorString(e)); L->device_fail.store(1); };
    hipError_t e = hipSetDevice(tp->device);
    if (e != hipSuccess) { bad("hipSetDevice", e); return; }
    if (r >= 1 && (e = hipStreamCreateWithFlags(&tp->copy_stream, hipStreamNonBlocking)) != hipSuccess) { bad("hipStreamCreate", e); return; }
    if (r >= 2) {
        hipIpcMemHandle_t h;
        // The ring is written by ANOTHER device (the upstream stage's k_hop_store over xGMI) and read by this one, with a device-side flag in between — no host
        // synchronisation and no runtime-inserted cache maintenance stands between the remote write and this device's first read of a re-used slot.  Ordinary hipMalloc
        // memory is coarse-grained: this device may keep lines of the slot's previous contents in its L2, coherent with other agents only at kernel / synchronisation
        // boundaries the runtime knows about.  Fine-grained device memory is coherent with other agents at system scope DURING a kernel (a peer's release store followed by
        // this device's system-scope acquire of the flag — k_flag_publish / k_abort_begin_wait — makes the data visible); 128 KB per slot, its bandwidth is irrelevant.
        // PIPEINFER_RING_ALLOC = finegrained (default) | uncached | coarse.
        const char * want = getenv("PIPEINFER_RING_ALLOC");
        int kind = want && !strcmp(want, "coarse") ? RING_COARSE : want && !strcmp(want, "uncached") ? RING_UNCACHED : RING_FINEGRAINED;
        e = hipSuccess;
        if (kind != RING_COARSE) {
            e = hipExtMallocWithFlags((void **) &tp->my_ring, L->x_slot_bytes * TP_X_SLOTS, kind == RING_UNCACHED ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
            if (e == hipSuccess && (e = hipIpcGetMemHandle(&h, tp->my_ring)) != hipSuccess) { (void) hipFree(tp->my_ring); tp->my_ring = nullptr; }      // (must be exportable)
            if (e != hipSuccess) { (void) hipGetLastError(); fprintf(stderr, "[pipeinfer transport, rank %d] no %s activation ring (%s): falling back to hipMalloc (coarse-grained)\n", r, ring_alloc_name(kind), hipGetErrorString(e)); tp->my_ring = nullptr; kind = RING_COARSE; }
        }
        if (kind == RING_COARSE) e = hipMalloc((void **) &tp->my_ring, L->x_slot_bytes * TP_X_SLOTS);
        tp->ring_alloc = kind; L->st[r].ring_alloc.store(kind);
        fprintf(stderr, "[pipeinfer transport, rank %d] activation ring on device %d: %s memory, %d slots of %zu bytes\n", r, tp->device, ring_alloc_name(kind), TP_X_SLOTS, (size_t) L->x_slot_bytes);
        if (e != hipSuccess) { bad("allocation of the activation ring", e); tp->my_ring = nullptr; }
        else if ((e = hipMemset(tp->my_ring, 0, L->x_slot_bytes * TP_X_SLOTS)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) bad("hipMemset of the activation ring", e);
        else if ((e = hipIpcGetMemHandle(&h, tp->my_ring)) != hipSuccess) bad("hipIpcGetMemHandle", e);
        else memcpy(L->st[r].ipc_handle, &h, sizeof h);
        L->st[r].ipc_ready.store(1, std::memory_order_release);                 // (also on failure: the neighbour must not wait for it)
    }
    if (r >= 1 && r + 1 < W) {                                                  // map the next stage's ring and write the probe block
        if (!tp->wait_for([&] { return L->st[r + 1].ipc_ready.load(std::memory_order_acquire) != 0; }, tp->timeout_ms)) L->device_fail.store(1);
        if (!L->device_fail.load()) {
            // say what the hardware offers for this pair before relying on it (the lazy flag of hipIpcOpenMemHandle below would enable it silently — or not)
            const int nd = L->st[r + 1].device.load();
            if (nd >= 0 && nd != tp->device) {
                int can = 0;
                e = hipDeviceCanAccessPeer(&can, tp->device, nd);
                tp->peer_access = e == hipSuccess ? can : 0;
                hipError_t pe = hipSuccess;
                if (tp->peer_access) { pe = hipDeviceEnablePeerAccess(nd, 0); if (pe == hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); pe = hipSuccess; } }
                fprintf(stderr, "[pipeinfer transport, rank %d] device %d -> device %d (rank %d): hipDeviceCanAccessPeer = %d, hipDeviceEnablePeerAccess: %s\n", r, tp->device, nd, r + 1,
                        tp->peer_access, tp->peer_access ? hipGetErrorString(pe) : "not attempted");
                if (!tp->peer_access || pe != hipSuccess) { fprintf(stderr, "[pipeinfer transport, rank %d] no peer access to the next stage's device: activations will be staged through host memory\n", r); L->device_fail.store(1); }
            } else if (nd == tp->device) {
                fprintf(stderr, "[pipeinfer transport, rank %d] the next stage (rank %d) shares device %d: the hop is a copy inside one HBM\n", r, r + 1, nd);
            }
        }
        if (!L->device_fail.load()) {
            hipIpcMemHandle_t h; memcpy(&h, L->st[r + 1].ipc_handle, sizeof h);
            if ((e = hipIpcOpenMemHandle((void **) &tp->peer_ring, h, hipIpcMemLazyEnablePeerAccess)) != hipSuccess) { bad("hipIpcOpenMemHandle", e); tp->peer_ring = nullptr; }
            else {
                float probe[256];
                for (int i = 0; i < 256; i++) probe[i] = (float) (r * 1000 + i);
                // (through the hop's own store - a kernel of this device writing the peer mapping - so that the probe tests what the runs will use)
                float * scratch = nullptr;
                if ((e = hipMalloc((void **) &scratch, sizeof probe)) != hipSuccess || (e = hipMemcpy(scratch, probe, sizeof probe, hipMemcpyHostToDevice)) != hipSuccess) bad("probe scratch", e);
                else if (!pi::hop_store(tp->peer_ring, scratch, sizeof probe, tp->copy_stream)) bad("probe store into the next stage's ring", hipGetLastError());
                else if ((e = hipStreamSynchronize(tp->copy_stream)) != hipSuccess) bad("probe store into the next stage's ring", e);
                if (scratch) (void) hipFree(scratch);
            }
        }
        L->st[r + 1].probe_written.store(1, std::memory_order_release);
    }
    if (r >= 2) {
        if (!tp->wait_for([&] { return L->st[r].probe_written.load(std::memory_order_acquire) != 0; }, tp->timeout_ms)) L->device_fail.store(1);
        if (!L->device_fail.load() && tp->my_ring) {
            float got[256];
            if ((e = hipMemcpy(got, tp->my_ring, sizeof got, hipMemcpyDeviceToHost)) != hipSuccess) bad("read-back of the probe block", e);
            else for (int i = 0; i < 256; i++) if (got[i] != (float) ((r - 1) * 1000 + i)) { fprintf(stderr, "[pipeinfer transport, rank %d] the probe block did not arrive intact in the peer-mapped ring\n", r); L->device_fail.store(1); break; }
        }
    }
}

// ---- head side: struct pipeinfer_remote_ops over the shared rings -------------------------------------------------------------------------
int head_start(void * user, int32_t run_seq, int32_t batch_id, int32_t n_tokens, const void * meta, size_t meta_bytes) {
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    shm_layout * L = tp->L;
    if (meta_bytes > L->meta_stride || n_tokens > tp->n_batch) { tp->fail("run metadata larger than the transport was opened for (n_batch / n_ctx)"); return -1; }
    ring_ctl & in = L->st[1].in;
    const uint64_t h = in.head.load(std::memory_order_relaxed);
    if (tp->started - tp->consumed >= (uint64_t) TP_RUN_TABLE || h - in.tail.load(std::memory_order_acquire) >= (uint64_t) TP_IN1_SLOTS) {
        tp->fail("more than " + std::to_string(TP_RUN_TABLE) + " runs in flight: the driver never waits for its oldest run"); return -1;
    }
    memcpy(tp->meta_of(run_seq), meta, meta_bytes);
    msg_hdr & m = L->st[1].slots[h % TP_IN1_SLOTS];
    m.kind = KIND_RUN; m.run_seq = run_seq; m.batch_id = batch_id; m.n_tokens = n_tokens; m.flags = 0; m.gen = tp->gen;
    in.head.store(h + 1, std::memory_order_release);
    L->head_epoch.fetch_add(1, std::memory_order_release);
    tp->started++;
    tp->head_stats.runs++;
    return 0;
}
int head_poll(void * user) {
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    tp->L->head_epoch.fetch_add(1, std::memory_order_release);
    return tp->L->res.head.load(std::memory_order_acquire) > tp->L->res.tail.load(std::memory_order_relaxed) ? 1 : 0;
}
int head_wait(void * user, int32_t run_seq, float * logits, int32_t * ids, float * vals, float * sums) {
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    shm_layout * L = tp->L;
    const uint64_t t = L->res.tail.load(std::memory_order_relaxed);
    const int64_t t0 = now_us();
    L->head_epoch.fetch_add(1, std::memory_order_release);
    L->head_waiting.fetch_add(1, std::memory_order_release);
    const bool arrived = tp->wait_for([&] { return L->res.head.load(std::memory_order_acquire) > t; }, tp->timeout_ms);
    L->head_waiting.fetch_sub(1, std::memory_order_release);
    if (!arrived) { tp->fail("no result for run " + std::to_string(run_seq) + " (a stage died or timed out)"); return -1; }
    tp->head_stats.wait_in_ms += (now_us() - t0) / 1e3;
    uint8_t * s = tp->res_slot(t);
    const msg_hdr m = *(const msg_hdr *) s;
    if (m.run_seq != run_seq) { tp->fail("result of run " + std::to_string(m.run_seq) + " while waiting for " + std::to_string(run_seq)); return -1; }
    const bool computed = !(m.flags & F_SKIPPED);
    if (computed) {
        const size_t n = (size_t) m.n_tokens;
        memcpy(ids, tp->res_ids(s), n * 32); memcpy(vals, tp->res_vals(s), n * 32); memcpy(sums, tp->res_sums(s), n * 4);
        memcpy(logits, tp->res_logits(s), n * (size_t) tp->n_vocab * 4);
    } else tp->head_stats.skipped++;
    L->res.tail.store(t + 1, std::memory_order_release);
    L->head_epoch_after_consume.store(L->head_epoch.fetch_add(1, std::memory_order_acq_rel) + 1, std::memory_order_release);
    tp->consumed++;
    tp->head_stats.received = tp->consumed;
    return computed ? 1 : 0;
}
void head_cancel(void * user, int32_t batch_id) {
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    if (batch_id == -2) { tp->gen++; return; }                                  // a new generation: batch ids restart
    shm_layout * L = tp->L;
    const uint64_t n = L->n_cancel.load(std::memory_order_relaxed);
    L->cancel[n % TP_CANCEL_CAP].store(((uint64_t) (uint32_t) tp->gen << 32) | (uint32_t) batch_id, std::memory_order_relaxed);
    L->n_cancel.store(n + 1, std::memory_order_release);
    tp->head_stats.cancel_notes++;
}

int abort_probe(void * user) {                                                  // between the segments of the run a stage is computing
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    tp->scan_cancels();
    return tp->cancelled.count(tp->cur_key) > 0 ? 1 : 0;
}
int abort_probe_all(void * user) {                                              // ... of a pass that carries several runs: it stops when ALL of them are cancelled
    pipeinfer_transport * tp = (pipeinfer_transport *) user;
    tp->scan_cancels();
    for (uint64_t k : tp->cur_keys) if (!tp->cancelled.count(k)) return 0;
    return 1;
}

}  // namespace

extern "C" pipeinfer_transport * pipeinfer_transport_open(const struct pipeinfer_transport_params * P) {
    if (!P || !P->name || P->world < 2 || P->world > TP_MAX_WORLD || P->rank < 0 || P->rank >= P->world || P->n_batch <= 0 || P->n_embd <= 0 || P->n_vocab <= 0 || P->n_ctx <= 0) {
        fprintf(stderr, "pipeinfer_transport_open: bad parameters\n"); return nullptr;
    }
    pipeinfer_transport * tp = new pipeinfer_transport();
    tp->name = P->name[0] == '/' ? P->name : std::string("/") + P->name;
    tp->rank = P->rank; tp->world = P->world; tp->n_embd = P->n_embd; tp->n_vocab = P->n_vocab; tp->n_batch = P->n_batch; tp->n_ctx = P->n_ctx;
    tp->device = P->device; if (P->timeout_ms > 0) tp->timeout_ms = P->timeout_ms;
    if (!map_segment(tp, *P)) { pipeinfer_transport_close(tp); return nullptr; }
    shm_layout * L = tp->L;
    if (P->device < 0 || getenv("PIPEINFER_TRANSPORT_HOST")) L->device_fail.store(1);   // a rank without a GPU (or the test switch): host rings for everybody
    L->st[tp->rank].device.store(P->device);
    L->st[tp->rank].joined.store(1, std::memory_order_release);
    for (int r = 0; r < tp->world; r++)
        if (!tp->wait_for([&] { return L->st[r].joined.load(std::memory_order_acquire) != 0; }, tp->timeout_ms)) { tp->fail("rank " + std::to_string(r) + " never joined the transport"); pipeinfer_transport_close(tp); return nullptr; }
    if (tp->creator) (void) shm_unlink(tp->name.c_str());                       // everybody has it mapped: the name can go (nothing is left behind after a crash)
    if (!L->device_fail.load() && P->device >= 0) setup_device_rings(tp);
    L->st[tp->rank].setup_done.store(1, std::memory_order_release);
    for (int r = 0; r < tp->world; r++)
        if (!tp->wait_for([&] { return L->st[r].setup_done.load(std::memory_order_acquire) != 0; }, tp->timeout_ms)) { tp->fail("rank " + std::to_string(r) + " never finished the transport set-up"); pipeinfer_transport_close(tp); return nullptr; }
    tp->device_mode = P->device >= 0 && !L->device_fail.load();
    if (const char * dw = getenv("PIPEINFER_DEVICE_WAIT_MS")) if (atoll(dw) > 0) tp->device_wait_us = atoll(dw) * 1000;
    if (P->device >= 0 && tp->rank >= 1) {
        // the hop's store, decided here once and said per rank: a kernel of the producing stream when both ends are device memory (this stage's HBM -> the peer mapping of
        // the next stage's ring), hipMemcpyAsync when the rings are in host memory.  Nothing is left to hipPointerGetAttributes on an IPC mapping.
        const bool by_memcpy = getenv("PIPEINFER_HOP_STORE") && !strcmp(getenv("PIPEINFER_HOP_STORE"), "memcpy");
        pi::hop_store_set_mode(tp->device_mode ? 1 : 2); tp->set_hop_mode = true;
        fprintf(stderr, "[pipeinfer transport, rank %d] hop store: %s (%s rings%s)\n", tp->rank, tp->device_mode && !by_memcpy ? "kernel of the producing stream (k_hop_store)" : "hipMemcpyAsync",
                tp->device_mode ? "device" : "host", by_memcpy ? ", PIPEINFER_HOP_STORE=memcpy" : "");
    }
    // Arrival flags (stage ranks, device mode; PIPEINFER_TRANSPORT_FLAGS=0: the host-published headers of round 3): the segment's control block is page-locked and mapped, so
    // that the producing stage's stream can write a flag and the consuming stage's stream can poll it.  A rank that cannot map it says so and every rank uses the host path.
    if (tp->device_mode && tp->rank >= 1 && !(getenv("PIPEINFER_TRANSPORT_FLAGS") && atoi(getenv("PIPEINFER_TRANSPORT_FLAGS")) == 0)) {
        void * dp = nullptr;
        if (hipSetDevice(P->device) == hipSuccess && hipHostRegister(tp->base, round_up(sizeof(shm_layout), 4096), hipHostRegisterMapped) == hipSuccess) {
            tp->ctl_registered = true;
            if (hipHostGetDevicePointer(&dp, tp->base, 0) == hipSuccess && dp) tp->ctl_dev = (uint8_t *) dp;
        }
        if (!tp->ctl_dev) { (void) hipGetLastError(); fprintf(stderr, "[pipeinfer transport, rank %d] the control block cannot be mapped for device access: headers are published by the host\n", tp->rank); L->flags_off.store(1); }
        // Probe the path on this node before any run depends on it (like the ring probe above): each stage's device writes the next stage's probe flag and each stage's device
        // polls its own.  A poll that never sees the write (a topology where a device does not observe another device's stores to mapped host memory) costs 3 s here, once,
        // and sends every rank to the host path - instead of one time-out per run.
        if (tp->ctl_dev) {
            const uint32_t probe_ticket = 0x2bad5eed;
            auto probe_dev = [&](int rank_of_consumer) { return (void *) (tp->ctl_dev + ((uint8_t *) &L->st[rank_of_consumer].xprobe - tp->base)); };
            bool ok = true;
            if (tp->rank + 1 < tp->world) ok = pi_op_flag_publish(probe_dev(tp->rank + 1), probe_ticket, nullptr, 0, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess;
            if (ok && tp->rank >= 2) {
                int32_t * scratch = nullptr;                      // {abort word, time-out flag}
                ok = hipMalloc((void **) &scratch, 2 * sizeof(int32_t)) == hipSuccess && hipMemset(scratch, 0, 2 * sizeof(int32_t)) == hipSuccess;
                int32_t seen[2] = {0, 1};
                ok = ok && pi_op_abort_begin_wait(scratch, 1, nullptr, probe_dev(tp->rank), probe_ticket, 3000000, scratch + 1, nullptr, nullptr) == 0
                        && hipDeviceSynchronize() == hipSuccess && hipMemcpy(seen, scratch, sizeof(seen), hipMemcpyDeviceToHost) == hipSuccess && seen[1] == 0 && seen[0] == (1 << 1);
                if (scratch) (void) hipFree(scratch);
            }
            if (!ok) { (void) hipGetLastError(); fprintf(stderr, "[pipeinfer transport, rank %d] the arrival-flag probe failed on this node: headers are published by the host\n", tp->rank); L->flags_off.store(1); }
            // Second probe — the REAL pattern of a hop, not only the flag (the first probe says nothing about the activations): the consumer reads slot 0 of its ring twice, so that
            // whatever its caches keep of the slot's old contents is there, and enqueues the wait kernel with the slot's reader (the same k_hop_store a run takes its input with)
            // behind it; only then does the producer store a new pattern through the peer mapping and publish the flag from its stream.  What the reader sees must be the new
            // pattern: no host synchronisation and no kernel launch stands between the remote write and the read, exactly as in a run.  A stale or partial block sends every
            // rank to the host-published headers.
            constexpr int PN = 1024; const uint32_t ticket2 = 0x3c0ffee5;
            float * got_dev = nullptr; int32_t * word2 = nullptr; bool armed = false;
            if (ok && tp->rank >= 2 && tp->my_ring) {
                armed = hipMalloc((void **) &got_dev, 2 * PN * sizeof(float)) == hipSuccess && hipMalloc((void **) &word2, 64) == hipSuccess && hipMemset(word2, 0, 64) == hipSuccess
                     && pi::hop_store(got_dev + PN, tp->my_ring, PN * sizeof(float), nullptr) && pi::hop_store(got_dev + PN, tp->my_ring, PN * sizeof(float), nullptr) && hipDeviceSynchronize() == hipSuccess
                     && pi_op_abort_begin_wait(word2, 1, nullptr, probe_dev(tp->rank), ticket2, 3000000, word2 + 1, nullptr, nullptr) == 0
                     && pi::hop_store(got_dev, tp->my_ring, PN * sizeof(float), nullptr);
                if (!armed) (void) hipGetLastError();
            }
            if (tp->rank >= 2) L->st[tp->rank].probe2_armed.store(armed ? 1 : 2, std::memory_order_release);
            if (tp->rank + 1 < tp->world) {                                      // producer side of the boundary behind this stage
                const int c = tp->rank + 1;
                (void) tp->wait_for([&] { return L->st[c].probe2_armed.load(std::memory_order_acquire) != 0; }, tp->timeout_ms);
                if (ok && tp->peer_ring && L->st[c].probe2_armed.load() == 1) {
                    std::vector<float> pat(PN); for (int i = 0; i < PN; i++) pat[i] = (float) (tp->rank * 7777 + i) + 0.5f;
                    float * pat_dev = nullptr;
                    std::this_thread::sleep_for(std::chrono::microseconds(500));   // (the consumer's wait kernel is spinning by now; not required, it makes the probe the harder case)
                    const bool sent = hipMalloc((void **) &pat_dev, PN * sizeof(float)) == hipSuccess && hipMemcpy(pat_dev, pat.data(), PN * sizeof(float), hipMemcpyHostToDevice) == hipSuccess
                                   && pi::hop_store(tp->peer_ring, pat_dev, PN * sizeof(float), tp->copy_stream) && pi_op_flag_publish(probe_dev(c), ticket2, nullptr, 0, tp->copy_stream) == 0
                                   && hipStreamSynchronize(tp->copy_stream) == hipSuccess;
                    if (!sent) { (void) hipGetLastError(); fprintf(stderr, "[pipeinfer transport, rank %d] the hop probe could not be sent\n", tp->rank); }
                    if (pat_dev) (void) hipFree(pat_dev);
                }
            }
            if (armed) {
                std::vector<float> got(PN); int32_t seen[2] = {0, 1};
                bool fresh = hipDeviceSynchronize() == hipSuccess && hipMemcpy(got.data(), got_dev, PN * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(seen, word2, sizeof seen, hipMemcpyDeviceToHost) == hipSuccess && seen[1] == 0;
                int bad_at = -1;
                for (int i = 0; fresh && i < PN; i++) if (got[i] != (float) ((tp->rank - 1) * 7777 + i) + 0.5f) { fresh = false; bad_at = i; }
                if (getenv("PIPEINFER_TEST_HOP_PROBE_FAIL")) { fresh = false; bad_at = 0; }          // tests: the fall-back every rank takes together when the probe fails
                if (!fresh) {
                    (void) hipGetLastError();
                    fprintf(stderr, "[pipeinfer transport, rank %d] the hop probe FAILED (%s): a reader enqueued behind the device-side wait did not see the block the upstream device stored before its flag — "
                                    "activation ring: %s memory.  Headers are published by the host instead.\n", tp->rank, bad_at >= 0 ? "stale or partial data" : "the flag never arrived", ring_alloc_name(tp->ring_alloc));
                    L->flags_off.store(1);
                } else fprintf(stderr, "[pipeinfer transport, rank %d] hop probe ok: store + device flag from device %d seen fresh by a reader enqueued behind the wait (ring: %s memory)\n", tp->rank, L->st[tp->rank - 1].device.load(), ring_alloc_name(tp->ring_alloc));
            }
            if (got_dev) (void) hipFree(got_dev);
            if (word2) (void) hipFree(word2);
        } else if (tp->rank >= 2) L->st[tp->rank].probe2_armed.store(2, std::memory_order_release);
    } else if (tp->rank >= 1) { L->flags_off.store(1); L->st[tp->rank].probe2_armed.store(2, std::memory_order_release); }
    L->st[tp->rank].flags_done.store(1, std::memory_order_release);
    for (int r = 1; r < tp->world; r++)
        if (!tp->wait_for([&] { return L->st[r].flags_done.load(std::memory_order_acquire) != 0; }, tp->timeout_ms)) { tp->fail("rank " + std::to_string(r) + " never finished the flag set-up"); pipeinfer_transport_close(tp); return nullptr; }
    tp->flags_on = tp->device_mode && tp->ctl_dev && !L->flags_off.load();
    if (P->device >= 0 && tp->rank == tp->world - 1) {                          // the last stage copies D2H straight into the result slots: pin them
        if (hipSetDevice(P->device) == hipSuccess && hipHostRegister(tp->base + L->off_res, L->res_slot_bytes * TP_RES_SLOTS, hipHostRegisterDefault) == hipSuccess) tp->res_registered = true;
        else (void) hipGetLastError();
    }
    return tp;
}

extern "C" void pipeinfer_transport_close(pipeinfer_transport * tp) {
    if (!tp) return;
    if (tp->set_hop_mode) pi::hop_store_set_mode(0);                          // the process-wide store mode was this transport's: later stage chains / transports ask the runtime (or set their own) again
    if (tp->res_registered) (void) hipHostUnregister(tp->base + tp->L->off_res);
    if (tp->ctl_registered) (void) hipHostUnregister(tp->base);
    if (tp->peer_ring) (void) hipIpcCloseMemHandle(tp->peer_ring);
    if (tp->my_ring) (void) hipFree(tp->my_ring);
    if (tp->copy_stream) (void) hipStreamDestroy(tp->copy_stream);
    if (tp->creator && tp->L && !tp->L->st[tp->world - 1].joined.load()) (void) shm_unlink(tp->name.c_str());
    if (tp->base) munmap(tp->base, tp->total);
    delete tp;
}

extern "C" const char * pipeinfer_transport_error(pipeinfer_transport * tp) { return tp ? tp->error.c_str() : "no transport"; }
extern "C" int pipeinfer_transport_device_mode(pipeinfer_transport * tp) { return tp && tp->device_mode ? 1 : 0; }

extern "C" int pipeinfer_transport_attach_head(pipeinfer_transport * tp, struct llama_context * ctx) {
    if (!tp || !ctx || tp->rank != 0) return -1;
    pipeinfer_remote_ops ops{};
    ops.user = tp; ops.start = head_start; ops.poll = head_poll; ops.wait = head_wait; ops.cancel = head_cancel;
    pipeinfer_context_set_remote(ctx, &ops);
    return 0;
}

extern "C" int pipeinfer_transport_shutdown(pipeinfer_transport * tp) {
    if (!tp || tp->rank != 0) return -1;
    if (tp->started != tp->consumed) { tp->fail("shutdown with runs still in flight"); return -1; }
    shm_layout * L = tp->L;
    ring_ctl & in = L->st[1].in;
    const uint64_t h = in.head.load(std::memory_order_relaxed);
    if (!tp->wait_for([&] { return h - in.tail.load(std::memory_order_acquire) < (uint64_t) TP_IN1_SLOTS; }, tp->timeout_ms)) return -1;
    msg_hdr & m = L->st[1].slots[h % TP_IN1_SLOTS];
    memset(&m, 0, sizeof m); m.kind = KIND_SHUTDOWN;
    in.head.store(h + 1, std::memory_order_release);
    return 0;
}

extern "C" void pipeinfer_transport_get_stats(pipeinfer_transport * tp, struct pipeinfer_transport_stats * out) { if (tp && out) *out = tp->head_stats; }

// ---- stage side -----------------------------------------------------------------------------------------------------------------------------
extern "C" int pipeinfer_transport_serve(pipeinfer_transport * tp, struct llama_context * ctx, const struct pipeinfer_stage_backend * backend, struct pipeinfer_transport_stats * out) {
    if (!tp || tp->rank < 1 || (!ctx && !backend)) return -1;
    shm_layout * L = tp->L;
    const int r = tp->rank;
    const bool is_first = r == 1, is_last = r == tp->world - 1, dev = tp->device_mode;
    pipeinfer_transport_stats st{};
    if (dev && hipSetDevice(tp->device) != hipSuccess) { tp->fail("hipSetDevice failed"); return -1; }

    st.peer_access = tp->peer_access;
    // host rings (no GPU in this process, or no peer mapping): a stage context still computes on its device and its stream copies straight into / out of the host slots
    // (hipMemcpyDefault); slots are page-locked once so that those copies are asynchronous
    bool x_registered = false;
    if (ctx && !dev && tp->world > 2) {
        if (hipHostRegister(tp->base + L->off_x, L->x_slot_bytes * TP_X_SLOTS * (size_t) (tp->world - 2), hipHostRegisterDefault) == hipSuccess) x_registered = true; else (void) hipGetLastError();
    }
    const size_t xb = L->x_slot_bytes;

    // kernel-granular cancel: while a run computes, a watcher follows the cancel table and stops it through the run's device-side abort word
    std::atomic<uint64_t> watch_key{~0ull}; std::atomic<bool> watch_stop{false};
    std::thread watcher;
    if (ctx) watcher = std::thread([&] {
        if (tp->device >= 0) (void) hipSetDevice(tp->device);
        uint64_t cursor = 0; std::unordered_set<uint64_t> seen;
        while (!watch_stop.load(std::memory_order_relaxed)) {
            const uint64_t n = L->n_cancel.load(std::memory_order_acquire);
            for (; cursor < n; cursor++) seen.insert(L->cancel[cursor % TP_CANCEL_CAP].load(std::memory_order_relaxed));
            // (key, run id) must belong together: the serve loop publishes the key before stage_run takes a fresh id and withdraws it after stage_run has zeroed
            // the id again, and ids never repeat — the same non-zero id on both sides of the key load means the key was that run's throughout
            const int32_t rid = ctx->stage_run_id.load(std::memory_order_acquire);
            const uint64_t key = watch_key.load(std::memory_order_acquire);
            if (rid != 0 && key != ~0ull && ctx->stage_run_id.load(std::memory_order_acquire) == rid && (seen.count(key) || L->fatal.load(std::memory_order_relaxed))) pi::stage_abort_run(ctx, rid);     // (a dead pipeline: whatever waits on the device for input that will never come is released)
            std::this_thread::sleep_for(std::chrono::microseconds(key != ~0ull ? 30 : 200));
        }
    });

    ring_ctl & in = L->st[r].in;
    const int cap = tp->in_cap(r);
    int rc_all = 0;
    // publishes header m downstream (next stage's ring slot h, or the result ring): the activations / logits it announces have landed (the stream was synchronised)
    auto publish = [&](const msg_hdr & m, uint64_t h, uint8_t * rs) {
        if (is_last) { *(msg_hdr *) rs = m; L->res.head.store(h + 1, std::memory_order_release); }
        else { L->st[r + 1].slots[h % TP_X_SLOTS] = m; L->st[r + 1].in.head.store(h + 1, std::memory_order_release); }
    };
    // Run coalescing on a stage that is the WHOLE target (world == 2: draft + driver on GPU 0, the target on GPU 1; pipeinfer_context_set_coalescing on the stage context):
    // the runs whose headers wait in the ring are carried by ONE pass over the weights, exactly like the in-process feeder's (context.cpp: feeder_main) — the non-speculative
    // run for a freshly sampled token and the speculative runs drafted meanwhile cost one weight stream instead of one each.  Members are consecutive headers (results leave in
    // start order), at most TP_RES_SLOTS of them (each owns a result slot), up to coalesce_max tokens, none in conflict over a cell (stage_pass_conflict).  Before composing a
    // pass the stage gives the head's driver the feeder's reaction window: until it blocks on a result, or has consumed one of the last pass's and called since, or 10 ms.
    // (A backend that is not a context of this library — the CPU tests' oracle — computes a pass member by member, in start order: the same results by construction; what the
    //  tests then exercise is the gathering, the conflict rule, the result order and the reaction window.  PIPEINFER_STAGE_COALESCE=<tokens> switches it on for such a backend.)
    const int cmax_backend = backend && getenv("PIPEINFER_STAGE_COALESCE") ? atoi(getenv("PIPEINFER_STAGE_COALESCE")) : 0;
    const bool coalescing = is_first && is_last && ((ctx && !backend && ctx->coalesce_max.load() > 1) || cmax_backend > 1);
    bool pass_pending = false; uint64_t pass_first_res = 0; int64_t last_pass_us = 0;
    // the reaction window is bounded by PIPEINFER_REACTION_US (10 ms) AND by three quarters of what the last pass took: a stage whose passes take a millisecond does not sit
    // out ten waiting for a driver that is busy rather than blocked (set explicitly, the variable is the bound whatever the pass takes)
    static const bool reaction_fixed = getenv("PIPEINFER_REACTION_US") != nullptr;
    static const int64_t reaction_us = getenv("PIPEINFER_REACTION_US") ? atoll(getenv("PIPEINFER_REACTION_US")) : 10000;
    // tests: a stage that is slow to pick its input up, so that the head's runs DO queue behind one another (a tiny model on a fast GPU may never see two headers waiting)
    static const int64_t test_latency_us = getenv("PIPEINFER_TEST_STAGE_LATENCY_US") ? atoll(getenv("PIPEINFER_TEST_STAGE_LATENCY_US")) : 0;
    while (coalescing) {
        const uint64_t t = in.tail.load(std::memory_order_relaxed);
        int64_t t0 = now_us();
        if (!tp->wait_for([&] { return in.head.load(std::memory_order_acquire) > t; }, -1)) { rc_all = -1; break; }
        st.wait_in_ms += (now_us() - t0) / 1e3;
        if (test_latency_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(test_latency_us));
        if (pass_pending) {
            pass_pending = false;
            const int64_t w0 = now_us();
            const int64_t window_us = reaction_fixed ? reaction_us : std::min<int64_t>(reaction_us, std::max<int64_t>(500, last_pass_us * 3 / 4));
            for (int spin = 0; ; spin++) {
                if (L->head_waiting.load(std::memory_order_acquire) > 0) break;
                if (L->res.tail.load(std::memory_order_acquire) > pass_first_res && L->head_epoch.load(std::memory_order_acquire) > L->head_epoch_after_consume.load(std::memory_order_acquire)) break;
                if (now_us() - w0 > window_us || L->fatal.load(std::memory_order_relaxed)) break;
                if (spin < 2000) __builtin_ia32_pause(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
            }
        }
        msg_hdr first = L->st[r].slots[t % cap];
        if (first.kind == KIND_SHUTDOWN) { in.tail.store(t + 1, std::memory_order_release); break; }
        // the members: consecutive headers that are here NOW
        const int cmax = backend ? cmax_backend : ctx->coalesce_max.load();
        const size_t words = (size_t) (tp->n_ctx + 31) / 32;
        std::vector<msg_hdr> hdr; std::vector<bool> live; std::vector<const void *> metas; std::vector<int32_t> ntok;
        int tokens = 0;
        const uint64_t avail = in.head.load(std::memory_order_acquire) - t;
        for (uint64_t k = 0; k < avail && hdr.size() < (size_t) TP_RES_SLOTS; k++) {
            const msg_hdr h = L->st[r].slots[(t + k) % cap];
            if (h.kind != KIND_RUN) break;
            const bool skip = (h.flags & F_SKIPPED) != 0 || tp->is_cancelled(h.gen, h.batch_id);
            if (!skip) {
                if (!metas.empty() && (tokens + h.n_tokens > cmax || pi::stage_pass_conflict(tp->n_ctx, (int) metas.size(), metas.data(), ntok.data(), tp->meta_of(h.run_seq), h.n_tokens))) break;
                metas.push_back(tp->meta_of(h.run_seq)); ntok.push_back(h.n_tokens); tokens += h.n_tokens;
            }
            hdr.push_back(h); live.push_back(!skip);
        }
        const size_t nm = hdr.size();
        t0 = now_us();
        const uint64_t oh = L->res.head.load(std::memory_order_relaxed);
        if (!tp->wait_for([&] { return oh + nm - L->res.tail.load(std::memory_order_acquire) <= (uint64_t) TP_RES_SLOTS; }, -1)) { rc_all = -1; break; }
        st.wait_out_ms += (now_us() - t0) / 1e3;
        st.received += nm;
        int rc = 1;                                                             // (no live member: nothing to compute)
        if (!metas.empty()) {
            std::vector<float *> lg, tv, ts; std::vector<int32_t *> ti;
            tp->cur_keys.clear();
            for (size_t k = 0; k < nm; k++) if (live[k]) {
                uint8_t * rs = tp->res_slot(oh + k);
                lg.push_back(tp->res_logits(rs)); ti.push_back(tp->res_ids(rs)); tv.push_back(tp->res_vals(rs)); ts.push_back(tp->res_sums(rs));
                tp->cur_keys.push_back(((uint64_t) (uint32_t) hdr[k].gen << 32) | (uint32_t) hdr[k].batch_id);
            }
            if (tp->cur_keys.size() == 1) { tp->cur_key = tp->cur_keys[0]; watch_key.store(tp->cur_key, std::memory_order_release); }      // one run: kernel-granular cancel as before
            t0 = now_us();
            if (backend) {
                rc = 0;
                for (size_t k = 0; k < metas.size() && rc == 0; k++) {
                    const size_t mb = (size_t) ntok[k] * 8 + 16 + (size_t) ntok[k] * words * 4;
                    rc = backend->run(backend->user, ntok[k], metas[k], mb, nullptr, nullptr, lg[k], ti[k], tv[k], ts[k]);
                    if (rc == 1) rc = -1;                                       // (a backend does not drop single members of a pass)
                }
            } else rc = pi::stage_run_pass(ctx, (int) metas.size(), metas.data(), ntok.data(), lg.data(), ti.data(), tv.data(), ts.data(), abort_probe_all, tp);
            last_pass_us = now_us() - t0;
            st.compute_ms += last_pass_us / 1e3;
            watch_key.store(~0ull, std::memory_order_release); tp->cur_key = ~0ull;
            if (rc < 0) { tp->fail(std::string("stage run failed: ") + (ctx ? pipeinfer_context_error(ctx) : "backend error")); rc_all = -1; break; }
            if (rc == 1) st.cut_short += metas.size(); else if (metas.size() > 1) st.passes_of_several_runs++;
        }
        in.tail.store(t + nm, std::memory_order_release);
        for (size_t k = 0; k < nm; k++) {                                       // results leave in start order
            msg_hdr m = hdr[k];
            // A member the head cancelled AFTER the gather was computed with the pass (a pass of several runs stops only when all of them are cancelled) - but it is reported like
            // the one-run-per-pass loop reports a run its cancel reached: skipped.  The head never reads a cancelled run's rows, and the stage's runs / skipped counters and the
            // head's skipped count then mean the same thing in both loops (what differs is GPU time: the one-run loop stops that run's kernels, a pass carries it to the end).
            const bool computed = live[k] && rc == 0 && !(metas.size() > 1 && tp->is_cancelled(m.gen, m.batch_id));
            m.flags = computed ? 0 : F_SKIPPED;
            if (computed) st.runs++; else st.skipped++;
            *(msg_hdr *) tp->res_slot(oh + k) = m;
            L->res.head.store(oh + k + 1, std::memory_order_release);
        }
        // (a gather whose members were all cancelled already put nothing on the GPU: there is nothing for the driver to react to, and no window behind it -
        //  the in-process feeder's rule, context.cpp: feed_skips_cancelled)
        pass_pending = !metas.empty(); pass_first_res = oh;
    }
    while (!coalescing) {
        const uint64_t t = in.tail.load(std::memory_order_relaxed);
        int64_t t0 = now_us();
        if (!tp->wait_for([&] { return in.head.load(std::memory_order_acquire) > t; }, -1)) { rc_all = -1; break; }
        st.wait_in_ms += (now_us() - t0) / 1e3;
        msg_hdr m = L->st[r].slots[t % cap];
        // reserve the outgoing slot BEFORE the run: the stage's own stream stores into it, and slot k of a consumer is free once it has consumed header k
        uint8_t * rs = nullptr; uint64_t oh = 0;
        t0 = now_us();
        if (is_last) {
            if (m.kind != KIND_SHUTDOWN) {
                oh = L->res.head.load(std::memory_order_relaxed);
                if (!tp->wait_for([&] { return oh - L->res.tail.load(std::memory_order_acquire) < (uint64_t) TP_RES_SLOTS; }, -1)) { rc_all = -1; break; }
                rs = tp->res_slot(oh);
            }
        } else {
            ring_ctl & nx = L->st[r + 1].in;
            oh = nx.head.load(std::memory_order_relaxed);
            if (!tp->wait_for([&] { return oh - nx.tail.load(std::memory_order_acquire) < (uint64_t) TP_X_SLOTS; }, -1)) { rc_all = -1; break; }
        }
        st.wait_out_ms += (now_us() - t0) / 1e3;
        if (m.kind == KIND_SHUTDOWN) {
            in.tail.store(t + 1, std::memory_order_release);
            if (!is_last) publish(m, oh, nullptr);
            break;
        }
        st.received++;
        const int n = m.n_tokens;
        const size_t words = (size_t) (tp->n_ctx + 31) / 32, meta_bytes = (size_t) n * 8 + 16 + (size_t) n * words * 4;
        bool skipped = (m.flags & F_SKIPPED) != 0;
        if (!skipped && tp->is_cancelled(m.gen, m.batch_id)) skipped = true;    // the head cancelled it before this stage started on it
        // Arrival flags (device mode): the header only ANNOUNCES a run — it is passed on at once, before this stage has computed anything, so that every stage behind has the
        // run enqueued (behind a device-side wait for its activations) by the time they arrive; the activations' arrival is the flag this stage's stream writes behind its
        // peer store.  A header that says "skipped" still means what it meant: nobody computes it, no flag is written or waited for.  What a header cannot say any more is
        // that a run was cut short DURING its computation: that travels in the flag's abort bit, from stream to stream.
        const bool flags = tp->flags_on && ctx && !backend;
        bool announced = false;
        if (flags && !is_last) { m.flags = skipped ? F_SKIPPED : 0; publish(m, oh, nullptr); announced = true; }
        if (!skipped) {
            const float * x_in = is_first ? nullptr : (dev ? (const float *) ((uint8_t *) tp->my_ring + (t % TP_X_SLOTS) * xb) : tp->host_x_slot(r, t));
            float * xo = is_last ? nullptr : (dev ? (float *) ((uint8_t *) tp->peer_ring + (oh % TP_X_SLOTS) * xb) : tp->host_x_slot(r + 1, oh));
            tp->cur_key = ((uint64_t) (uint32_t) m.gen << 32) | (uint32_t) m.batch_id;
            watch_key.store(tp->cur_key, std::memory_order_release);
            t0 = now_us();
            int rc;
            const double hop0 = ctx ? ctx->stat_hop_ms : 0.0;
            pi::pi_stage_hop hop;
            if (flags) {
                if (!is_first) { hop.in_flag = tp->flag_dev(r, t); hop.in_ticket = (uint32_t) (t + 1); }
                if (!is_last) { hop.out_flag = tp->flag_dev(r + 1, oh); hop.out_ticket = (uint32_t) (oh + 1); }
                hop.wait_timeout_us = tp->device_wait_us;
            }
            if (backend) rc = backend->run(backend->user, n, tp->meta_of(m.run_seq), meta_bytes, x_in, xo, rs ? tp->res_logits(rs) : nullptr, rs ? tp->res_ids(rs) : nullptr, rs ? tp->res_vals(rs) : nullptr, rs ? tp->res_sums(rs) : nullptr);
            else rc = pi::stage_run(ctx, n, tp->meta_of(m.run_seq), meta_bytes, x_in, xo, rs ? tp->res_logits(rs) : nullptr, rs ? tp->res_ids(rs) : nullptr, rs ? tp->res_vals(rs) : nullptr, rs ? tp->res_sums(rs) : nullptr, abort_probe, tp,
                                    flags ? &hop : nullptr);
            st.compute_ms += (now_us() - t0) / 1e3;
            if (ctx) st.hop_gpu_ms += ctx->stat_hop_ms - hop0;
            watch_key.store(~0ull, std::memory_order_release);
            tp->cur_key = ~0ull;
            if (rc < 0) {
                // (the stage behind may be waiting on the device for this run's flag: release it, marked aborted)
                if (announced) { L->st[r + 1].xflag[oh % TP_X_SLOTS].stamp = 0; L->st[r + 1].xflag[oh % TP_X_SLOTS].value.store((((uint32_t) (oh + 1) & 0x7fffffffu) << 1) | 1u, std::memory_order_release); }
                tp->fail(std::string("stage run failed: ") + (ctx ? pipeinfer_context_error(ctx) : "backend error")); rc_all = -1; break;
            }
            if (rc == 1) { skipped = true; st.cut_short++; }
            else st.runs++;
        }
        if (skipped) st.skipped++;
        t0 = now_us();
        in.tail.store(t + 1, std::memory_order_release);                        // the incoming activation slot is free again
        if (!announced) {
            m.flags = skipped ? F_SKIPPED : 0;
            publish(m, oh, rs);                                                 // what the header announces has landed: stage_run returns with its stream idle
        }
        st.send_ms += (now_us() - t0) / 1e3;
    }
    if (rc_all != 0 && !is_last) {                                              // tell the stages behind this one: nothing else will come
        ring_ctl & nx = L->st[r + 1].in;
        const uint64_t h = nx.head.load(std::memory_order_relaxed);
        if (h - nx.tail.load(std::memory_order_acquire) < (uint64_t) TP_X_SLOTS) { msg_hdr bye{}; bye.kind = KIND_SHUTDOWN; publish(bye, h, nullptr); }
    }
    watch_stop.store(true);
    if (watcher.joinable()) watcher.join();
    if (x_registered) (void) hipHostUnregister(tp->base + L->off_x);
    st.cancel_notes = tp->cancel_cursor;
    st.device_flags = tp->flags_on && ctx && !backend ? 1 : 0;
    st.hop_store_kernel = pi::hop_store_mode() == 1 && !(getenv("PIPEINFER_HOP_STORE") && !strcmp(getenv("PIPEINFER_HOP_STORE"), "memcpy")) ? 1 : 0;
    st.ring_alloc = tp->ring_alloc;
    if (ctx) { st.flag_runs = ctx->stat_flag_runs; st.flag_wait_gpu_ms = ctx->stat_flag_wait_ms; st.flag_seen_us = ctx->stat_flag_seen_us; }
    if (out) *out = st;
    return rc_all;
}
