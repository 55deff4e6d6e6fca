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
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
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

struct Comm {
    int rank = 0, world = 1;
    Control *ctl = nullptr;
    std::string base;               // /dev/shm/<id>
    uint64_t seq = 0;               // batches executed so far: the same on every rank (collective call order)
    double timeout_s = 120;
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

ncclResult_t run_batch(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    Comm *c = ops[0].comm;
    for (const Op &o : ops)
        if (o.comm != c) return fail(c, ncclInvalidUsage, "one group spans several communicators%s", "");
    for (const Op &o : ops) {
        (void) o;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(c, ncclUnhandledCudaError, "hipStreamSynchronize failed%s", "");
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
        if (o.bytes && hipMemcpy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            rc = fail(c, ncclUnhandledCudaError, "device-to-host copy of %s failed", path.c_str());
            break;
        }
        if (!write_file(path, host.data(), o.bytes)) {
            rc = fail(c, ncclSystemError, "cannot write %s", path.c_str());
            break;
        }
        mine.push_back(path);
    }
    if (rc == ncclSuccess) rc = barrier(c);
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
                if (o.src != o.dst && o.bytes && hipMemcpy(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess)
                    rc = fail(c, ncclUnhandledCudaError, "root copy of %s failed", path.c_str());
                continue;
            }
            host.resize(o.bytes ? o.bytes : 1);
            long long got = 0;
            const int st = read_file(path, host.data(), o.bytes, &got);
            if (st == -1) rc = fail(c, ncclInvalidUsage, "no matching operation on the peer for %s", path.c_str());
            else if (st == -2) rc = fail(c, ncclInvalidArgument, "size mismatch on %s: expected %lld bytes, peer gave %lld", path.c_str(), (long long) o.bytes, got);
            else if (o.bytes && hipMemcpy(dst, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(c, ncclUnhandledCudaError, "host-to-device copy of %s failed", path.c_str());
            if (rc == ncclSuccess && o.kind == RECV) ::unlink(path.c_str());      // taken: what is left after the batch is unmatched
        }
    }
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

ncclResult_t submit(const Op &o) {
    if (!o.comm) return ncclInvalidArgument;
    if (o.kind != ALLGATHER && (o.peer < 0 || o.peer >= o.comm->world)) return fail(o.comm, ncclInvalidArgument, "peer out of range%s", "");
    if ((o.kind == SEND || o.kind == RECV) && o.peer == o.comm->rank) return fail(o.comm, ncclInvalidUsage, "send / receive to self%s", "");
    g_pending.push_back(o);
    if (g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_pending);
    return run_batch(ops);
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
    *comm = (ncclComm_t) c;
    return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort) {
    Comm *c = (Comm *) comm;
    if (!c) return ncclInvalidArgument;
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
    return run_batch(ops);
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
