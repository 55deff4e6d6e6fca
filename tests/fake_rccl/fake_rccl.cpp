// tests/fake_rccl/fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped and never loaded by the product on its own.
//
// A stand-in for the ten RCCL entry points csrc/exchange.hip binds through dlsym, so that the multi-rank code of
// libtelomeri_hip.so (grouped ncclSend / ncclRecv into fixed places, the summary all-gathers, the owners' broadcasts)
// can be EXECUTED by several processes that share ONE GPU: real RCCL refuses two ranks on one device, and the builder's
// boxes have one.  Selected with TLM_RCCL_LIB=<this .so>; without that variable the library loads librccl.so.1.
//
// Transport: files under /dev/shm named by the communicator's id, and a barrier in a shared control page.  Every call
// is executed synchronously -- the stream is drained first, device buffers are copied through host memory -- which is a
// legal (if slow) schedule of the stream-ordered operations RCCL would queue.  Group semantics as in RCCL: between
// ncclGroupStart and the matching ncclGroupEnd operations are only recorded; the outermost ncclGroupEnd runs them all:
//   publish (my sends, my broadcasts as root, my all-gather shares) -> barrier -> consume (my receives, ...) -> barrier
//   -> tidy (a send nobody received is found here) -> barrier.
// Pairing is CHECKED, which real RCCL does not do for you: the k-th ncclSend of rank a to rank b inside a group must meet
// the k-th ncclRecv of b from a with the same byte count; an unmatched send or receive, or a size mismatch, is an error
// on both sides instead of a hang.  A rank that fails or times out (FAKE_RCCL_TIMEOUT_S, default 120) raises a flag in
// the control page and every peer's pending or next call fails too (ncclRemoteError).
//
// FAKE_RCCL_ASYNC=1 (round 4): the calls behave like RCCL's towards the CALLER -- they return at once, the operation runs
// in stream order on the caller's stream and completes later.  Every call / outermost ncclGroupEnd records an event on its
// stream, queues a one-lane GATE kernel behind it (it holds the stream until the batch is done, as the collective's own
// kernel would) and hands the batch to the communicator's worker thread; the worker waits FAKE_RCCL_ASYNC_DELAY_MS (default
// 30) on purpose, then for the event, moves the bytes over a stream of its own, and opens the gate.  A caller that reads a
// result on the host before it has synchronised the stream, that frees or reuses a buffer early, or that launches a
// consumer on another stream without an event, finds stale data here, on ONE GPU.  Errors of a batch surface at the next
// call (as RCCL's asynchronous errors do): the failure tests run in the synchronous mode.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#ifdef FAKE_RCCL_HOST_ONLY
#include <hip/hip_runtime_api.h>
#else
#include <hip/hip_runtime.h>
#endif
#include <rccl/rccl.h>

// FAKE_RCCL_HOST_ONLY (tests/fake_rccl/Makefile target `host`): the same library with buffers in HOST memory and no HIP
// call at all, so that the stand-in's own logic -- pairing, size checks, group batching, barriers, failure propagation --
// is tested on machines without a GPU (tests/test_fake_rccl_host.py).
#ifdef FAKE_RCCL_HOST_ONLY
#define hipStreamSynchronize(s) hipSuccess
#define hipMemcpy(dst, src, n, kind) (std::memcpy((dst), (src), (n)), hipSuccess)
#endif

namespace {

struct Control {                    // one page in /dev/shm/<id>.ctl, zero-filled at creation
    std::atomic<uint32_t> arrived, generation, failed, members;
};

struct Job;
struct Comm {
    int rank = 0, world = 1;
    Control *ctl = nullptr;
    std::string base;               // /dev/shm/<id>
    uint64_t seq = 0;               // batches executed so far: the same on every rank (collective call order)
    double timeout_s = 120;
    // FAKE_RCCL_ASYNC: batches are executed by a worker thread behind a gate kernel on the caller's stream
    bool async = false;
    int delay_ms = 30;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job *> jobs;
    bool stop = false;
    std::atomic<int> async_error{0};        // first failed batch (sticky): reported by the next call
    uint32_t *gate_host = nullptr, *gate_dev = nullptr;     // host-mapped word the gate kernels watch
    uint32_t gate_seq = 0;
    hipStream_t copy_stream = nullptr;
    void *stage = nullptr, *stage_dev = nullptr;    // pinned, device-mapped staging buffer of the worker's copies
};

enum Kind { SEND, RECV, BCAST, ALLGATHER };
struct Op {
    Kind kind;
    Comm *comm;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;                       // peer (send / recv) or root (broadcast)
    hipStream_t stream;
};

struct Job {
    std::vector<Op> ops;
    hipEvent_t ready = nullptr;
    uint32_t gate = 0;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_pending;
thread_local char g_msg[512] = "";

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 1;
    }
}

ncclResult_t fail(Comm *c, ncclResult_t code, const char *fmt, const char *a = "", long long b = 0, long long d = 0) {
    std::snprintf(g_msg, sizeof(g_msg), fmt, a, b, d);
    std::fprintf(stderr, "fake_rccl[rank %d]: %s\n", c ? c->rank : -1, g_msg);
    if (c && c->ctl) c->ctl->failed.store(1);
    return code;
}

ncclResult_t barrier(Comm *c) {
    Control *k = c->ctl;
    if (k->failed.load()) return fail(c, ncclRemoteError, "a peer has failed%s", "");
    const uint32_t gen = k->generation.load();
    if (k->arrived.fetch_add(1) + 1 == (uint32_t) c->world) {
        k->arrived.store(0);
        k->generation.fetch_add(1);
        return ncclSuccess;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; k->generation.load() == gen; spin++) {
        if (k->failed.load()) return fail(c, ncclRemoteError, "a peer failed while this rank waited at a barrier%s", "");
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if ((spin & 1023) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s)
            return fail(c, ncclSystemError, "barrier timed out: a peer never arrived%s", "");
    }
    return ncclSuccess;
}

bool write_file(const std::string &path, const void *host, size_t bytes) {
    const int fd = ::open(path.c_str(), O_CREAT | O_EXCL | O_WRONLY | O_NOFOLLOW, 0600);
    if (fd < 0) return false;
    const char *p = (const char *) host;
    size_t left = bytes;
    while (left) {
        const ssize_t w = ::write(fd, p, left);
        if (w <= 0) { ::close(fd); return false; }
        p += w;
        left -= (size_t) w;
    }
    return ::close(fd) == 0;
}

// -1: missing, -2: size mismatch (*got = size found), 0: ok
int read_file(const std::string &path, void *host, size_t bytes, long long *got) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW);
    if (fd < 0) return -1;
    struct stat st;
    if (::fstat(fd, &st) != 0 || (size_t) st.st_size != bytes) {
        *got = (long long) st.st_size;
        ::close(fd);
        return -2;
    }
    char *p = (char *) host;
    size_t left = bytes;
    while (left) {
        const ssize_t r = ::read(fd, p, left);
        if (r <= 0) { ::close(fd); return -1; }
        p += r;
        left -= (size_t) r;
    }
    ::close(fd);
    return 0;
}

std::string piece(const Comm *c, uint64_t seq, const char *tag, long long a, long long b, long long k) {
    char buf[160];
    std::snprintf(buf, sizeof(buf), ".%llu.%s.%lld.%lld.%lld", (unsigned long long) seq, tag, a, b, k);
    return c->base + buf;
}

// device <-> host copies of a batch: synchronous calls in the synchronous mode; in the asynchronous mode the caller's stream
// is held by the gate kernel, so the worker copies on a stream of its own
#ifndef FAKE_RCCL_HOST_ONLY
// byte copy by a kernel: device memory <-> the worker's pinned, device-mapped staging buffer (or device <-> device)
__global__ void fake_rccl_copy_kernel(unsigned char *dst, const unsigned char *src, size_t n) {
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}

hipError_t batch_copy(Comm *c, void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    if (!c->async) return hipMemcpy(dst, src, n, kind);
    // NOT through the runtime's copy calls: its copies of one direction run in order behind each other (an SDMA engine, or
    // the runtime's own blit queue, is a FIFO), and the CALLER's device-to-host copy that waits behind the gate kernel --
    // in stream order, as it should -- holds every later copy, the worker's included: the batch that has to open the
    // gate never gets its bytes (first versions: the 4.5 MB summary all-gather "took" 60 005 ms on both ranks, the gate's own
    // time-out).  A kernel on the worker's stream reads and writes a pinned, device-mapped staging buffer instead.
    constexpr size_t STAGE = 4u << 20;
    if (!c->stage) {
        if (hipHostMalloc(&c->stage, STAGE, hipHostMallocMapped) != hipSuccess) return hipErrorOutOfMemory;
        if (hipHostGetDevicePointer(&c->stage_dev, c->stage, 0) != hipSuccess) return hipErrorOutOfMemory;
    }
    auto run = [&](void *d, const void *s_, size_t m) {
        hipLaunchKernelGGL(fake_rccl_copy_kernel, dim3(256), dim3(256), 0, c->copy_stream, (unsigned char *) d, (const unsigned char *) s_, m);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? hipStreamSynchronize(c->copy_stream) : e;
    };
    if (kind == hipMemcpyDeviceToDevice) return run(dst, src, n);
    for (size_t at = 0; at < n; at += STAGE) {
        const size_t m = n - at < STAGE ? n - at : STAGE;
        hipError_t e;
        if (kind == hipMemcpyDeviceToHost) {
            e = run(c->stage_dev, (const char *) src + at, m);
            if (e == hipSuccess) std::memcpy((char *) dst + at, c->stage, m);
        } else {
            std::memcpy(c->stage, (const char *) src + at, m);
            e = run((char *) dst + at, c->stage_dev, m);
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
#else
#define batch_copy(c, dst, src, n, kind) (std::memcpy((dst), (src), (n)), hipSuccess)
#endif

ncclResult_t run_batch(std::vector<Op> &ops, bool streams_are_ready = false) {
    if (ops.empty()) return ncclSuccess;
    Comm *c = ops[0].comm;
    for (const Op &o : ops)
        if (o.comm != c) return fail(c, ncclInvalidUsage, "one group spans several communicators%s", "");
    for (const Op &o : ops) {
        (void) o;
        if (!streams_are_ready && hipStreamSynchronize(o.stream) != hipSuccess) return fail(c, ncclUnhandledCudaError, "hipStreamSynchronize failed%s", "");
    }
    const uint64_t seq = c->seq++;
    std::vector<char> host;
    std::vector<std::string> mine;              // files this rank created
    std::vector<int> n_send(c->world, 0), n_recv(c->world, 0);
    int n_bcast = 0, n_gather = 0;
    ncclResult_t rc = ncclSuccess;
    // ---- publish
    for (const Op &o : ops) {
        std::string path;
        if (o.kind == SEND) path = piece(c, seq, "s", c->rank, o.peer, n_send[o.peer]++);
        else if (o.kind == BCAST) { const int k = n_bcast++; if (o.peer != c->rank) continue; path = piece(c, seq, "b", o.peer, 0, k); }
        else if (o.kind == ALLGATHER) path = piece(c, seq, "g", c->rank, 0, n_gather++);
        else continue;
        host.resize(o.bytes ? o.bytes : 1);
        if (o.bytes && batch_copy(c, host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(c, ncclUnhandledCudaError, "device-to-host copy of %s failed", path.c_str());
            break;
        }
        if (!write_file(path, host.data(), o.bytes)) {
            rc = fail(c, ncclSystemError, "cannot write %s", path.c_str());
            break;
        }
        mine.push_back(path);
    }
    static const bool verbose_steps = std::getenv("FAKE_RCCL_ASYNC_VERBOSE") != nullptr;
    const auto t_pub = std::chrono::steady_clock::now();
    if (rc == ncclSuccess) rc = barrier(c);
    const auto t_bar = std::chrono::steady_clock::now();
    // ---- consume
    n_bcast = n_gather = 0;
    for (size_t i = 0; rc == ncclSuccess && i < ops.size(); i++) {
        const Op &o = ops[i];
        if (o.kind == SEND) continue;
        const int lo = o.kind == ALLGATHER ? 0 : o.peer, hi = o.kind == ALLGATHER ? c->world : o.peer + 1;
        const int k = o.kind == RECV ? n_recv[o.peer]++ : o.kind == BCAST ? n_bcast++ : n_gather++;
        for (int r = lo; r < hi && rc == ncclSuccess; r++) {
            const std::string path = o.kind == RECV ? piece(c, seq, "s", r, c->rank, k)
                                   : o.kind == BCAST ? piece(c, seq, "b", r, 0, k) : piece(c, seq, "g", r, 0, k);
            char *dst = (char *) o.dst + (o.kind == ALLGATHER ? (size_t) r * o.bytes : 0);
            if (o.kind == BCAST && r == c->rank) {          // the root: send buffer -> receive buffer when they differ
                if (o.src != o.dst && o.bytes && batch_copy(c, o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                    rc = fail(c, ncclUnhandledCudaError, "root copy of %s failed", path.c_str());
                continue;
            }
            host.resize(o.bytes ? o.bytes : 1);
            long long got = 0;
            const int st = read_file(path, host.data(), o.bytes, &got);
            if (st == -1) rc = fail(c, ncclInvalidUsage, "no matching operation on the peer for %s", path.c_str());
            else if (st == -2) rc = fail(c, ncclInvalidArgument, "size mismatch on %s: expected %lld bytes, peer gave %lld", path.c_str(), (long long) o.bytes, got);
            else if (o.bytes && batch_copy(c, dst, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(c, ncclUnhandledCudaError, "host-to-device copy of %s failed", path.c_str());
            if (rc == ncclSuccess && o.kind == RECV) ::unlink(path.c_str());      // taken: what is left after the batch is unmatched
        }
    }
    if (verbose_steps && c->async)
        std::fprintf(stderr, "fake_rccl[rank %d]   batch seq %llu: barrier after publish %.0f ms, consume %.0f ms\n", c->rank, (unsigned long long) seq,
                     std::chrono::duration<double, std::milli>(t_bar - t_pub).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_bar).count());
    if (rc == ncclSuccess) rc = barrier(c);
    // ---- tidy; a send file that is still there was never received
    for (const std::string &p : mine) {
        const bool was_send = p.find(".s.") != std::string::npos;
        if (::unlink(p.c_str()) == 0 && was_send && rc == ncclSuccess)
            rc = fail(c, ncclInvalidUsage, "ncclSend without a matching ncclRecv: %s", p.c_str());
    }
    // ... which only the SENDER can see: one more barrier, so that the rank that never posted the receive fails as well
    if (rc == ncclSuccess) rc = barrier(c);
    return rc;
}

#ifndef FAKE_RCCL_HOST_ONLY
// Holds the caller's stream until the worker has finished batch `want` (or a minute has passed: the grid always drains).
__global__ void fake_rccl_gate_kernel(const uint32_t *flag, uint32_t want) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();              // 100 MHz
    while ((int32_t) (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
        if (wall_clock64() - t0 > 60ull * 100000000ull) break;
        __builtin_amdgcn_s_sleep(100);
    }
}

void worker_main(Comm *c) {
    for (;;) {
        Job *j = nullptr;
        {
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->stop || !c->jobs.empty(); });
            if (c->jobs.empty()) return;
            j = c->jobs.front();
            c->jobs.pop_front();
        }
        static const bool verbose = std::getenv("FAKE_RCCL_ASYNC_VERBOSE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::this_thread::sleep_for(std::chrono::milliseconds(c->delay_ms));     // on purpose: the caller is long gone
        ncclResult_t rc = ncclSuccess;
        if (hipEventSynchronize(j->ready) != hipSuccess) rc = fail(c, ncclUnhandledCudaError, "hipEventSynchronize failed%s", "");
        const double t_ready = ms();
        size_t bytes = 0;
        for (const Op &o : j->ops) bytes += o.bytes;
        if (rc == ncclSuccess) rc = run_batch(j->ops, true);
        if (verbose)
            std::fprintf(stderr, "fake_rccl[rank %d] async batch %u: %zu ops, %zu bytes; inputs ready after %.0f ms, done after %.0f ms, rc %d\n",
                         c->rank, j->gate, j->ops.size(), bytes, t_ready, ms(), (int) rc);
        if (rc != ncclSuccess) {
            int expected = 0;
            c->async_error.compare_exchange_strong(expected, (int) rc);
        }
        __atomic_store_n(c->gate_host, j->gate, __ATOMIC_RELEASE);              // the gate opens whatever happened
        (void) hipEventDestroy(j->ready);
        delete j;
    }
}

ncclResult_t run_batch_async(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    Comm *c = ops[0].comm;
    if (const int e = c->async_error.load()) return fail(c, (ncclResult_t) e, "an earlier asynchronous batch failed%s", "");
    for (const Op &o : ops)
        if (o.comm != c || o.stream != ops[0].stream) return fail(c, ncclInvalidUsage, "one group spans several communicators or streams%s", "");
    Job *j = new Job();
    j->ops.swap(ops);
    j->gate = ++c->gate_seq;
    if (hipEventCreateWithFlags(&j->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(j->ready, j->ops[0].stream) != hipSuccess) {
        delete j;
        return fail(c, ncclUnhandledCudaError, "cannot record the batch's event%s", "");
    }
    hipLaunchKernelGGL(fake_rccl_gate_kernel, dim3(1), dim3(64), 0, j->ops[0].stream, (const uint32_t *) c->gate_dev, j->gate);
    if (hipGetLastError() != hipSuccess) {
        delete j;
        return fail(c, ncclUnhandledCudaError, "cannot launch the gate kernel%s", "");
    }
    {
        std::lock_guard<std::mutex> lk(c->mu);
        c->jobs.push_back(j);
    }
    c->cv.notify_one();
    return ncclSuccess;
}
#endif

ncclResult_t dispatch(std::vector<Op> &ops) {
#ifndef FAKE_RCCL_HOST_ONLY
    if (!ops.empty() && ops[0].comm && ops[0].comm->async) return run_batch_async(ops);
#endif
    return run_batch(ops);
}

ncclResult_t submit(const Op &o) {
    if (!o.comm) return ncclInvalidArgument;
    if (o.kind != ALLGATHER && (o.peer < 0 || o.peer >= o.comm->world)) return fail(o.comm, ncclInvalidArgument, "peer out of range%s", "");
    if ((o.kind == SEND || o.kind == RECV) && o.peer == o.comm->rank) return fail(o.comm, ncclInvalidUsage, "send / receive to self%s", "");
    g_pending.push_back(o);
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_pending);
    return dispatch(ops);
}

}   // namespace

extern "C" {

const char *fake_rccl_marker(void) { return "tests/fake_rccl: not RCCL"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    unsigned char rnd[12] = {0};
    if (FILE *f = std::fopen("/dev/urandom", "rb")) {
        (void) !std::fread(rnd, 1, sizeof(rnd), f);
        std::fclose(f);
    }
    char *p = id->internal;
    p += std::snprintf(p, 32, "tlm_fakerccl_%ld_", (long) ::getpid());
    for (unsigned char b : rnd) p += std::snprintf(p, 3, "%02x", b);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    if (std::strncmp(id.internal, "tlm_fakerccl_", 13) != 0 || std::strchr(id.internal, '/')) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = nranks;
    c->base = std::string("/dev/shm/") + id.internal;
    if (const char *t = std::getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = std::atof(t);
    const std::string ctl = c->base + ".ctl";
    const int fd = ::open(ctl.c_str(), O_CREAT | O_RDWR | O_NOFOLLOW, 0600);
    if (fd < 0 || ::ftruncate(fd, 4096) != 0) {
        if (fd >= 0) ::close(fd);
        delete c;
        return ncclSystemError;
    }
    void *m = ::mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->ctl = (Control *) m;
    c->ctl->members.fetch_add(1);
    const ncclResult_t rc = barrier(c);
    if (rc != ncclSuccess) {
        ::munmap(m, 4096);
        delete c;
        return rc;
    }
#ifndef FAKE_RCCL_HOST_ONLY
    if (const char *a = std::getenv("FAKE_RCCL_ASYNC")) {
        if (std::atoi(a) != 0) {
            if (const char *d = std::getenv("FAKE_RCCL_ASYNC_DELAY_MS")) c->delay_ms = std::atoi(d);
            if (hipHostMalloc((void **) &c->gate_host, 64, hipHostMallocMapped) != hipSuccess
                || hipHostGetDevicePointer((void **) &c->gate_dev, c->gate_host, 0) != hipSuccess
                || hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
                ::munmap(m, 4096);
                delete c;
                return ncclUnhandledCudaError;
            }
            *c->gate_host = 0;
            c->async = true;
            c->worker = std::thread(worker_main, c);
        }
    }
#endif
    *comm = (ncclComm_t) c;
    return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort) {
    Comm *c = (Comm *) comm;
    if (!c) return ncclInvalidArgument;
#ifndef FAKE_RCCL_HOST_ONLY
    if (c->async) {         // what is queued is finished first (its gates must open), then the worker goes
        {
            std::lock_guard<std::mutex> lk(c->mu);
            c->stop = true;
        }
        c->cv.notify_one();
        if (c->worker.joinable()) c->worker.join();
        (void) hipStreamDestroy(c->copy_stream);
        (void) hipHostFree(c->gate_host);
        if (c->stage) (void) hipHostFree(c->stage);
    }
#endif
    if (abort) c->ctl->failed.store(1);
    if (c->ctl->members.fetch_sub(1) == 1) ::unlink((c->base + ".ctl").c_str());
    ::munmap(c->ctl, 4096);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return leave(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return leave(comm, true); }

const char *ncclGetErrorString(ncclResult_t r) {
    static thread_local char buf[640];
    static const char *names[] = {"success", "unhandled HIP error", "system error", "internal error", "invalid argument",
                                  "invalid usage", "remote error"};
    std::snprintf(buf, sizeof(buf), "fake_rccl: %s (%s)", (int) r >= 0 && (int) r < 7 ? names[(int) r] : "?", g_msg);
    return buf;
}

ncclResult_t ncclGroupStart() {
    g_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_pending);
    return dispatch(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return submit(Op{SEND, (Comm *) comm, buf, nullptr, count * type_size(t), peer, s});
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return submit(Op{RECV, (Comm *) comm, nullptr, buf, count * type_size(t), peer, s});
}
ncclResult_t ncclBroadcast(const void *src, void *dst, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s) {
    return submit(Op{BCAST, (Comm *) comm, src, dst, count * type_size(t), root, s});
}
ncclResult_t ncclAllGather(const void *src, void *dst, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    return submit(Op{ALLGATHER, (Comm *) comm, src, dst, count * type_size(t), 0, s});
}

}   // extern "C"
