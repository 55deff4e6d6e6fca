// telomeri_amd/csrc/device_pool.hip -- the library's own device-memory cache (one per device, process lifetime).
//
// Every device buffer of the library (graph arrays, build temporaries, walk workspaces, path sets) comes from here.
// A build of the A. thaliana-shaped graph allocates ~60 buffers, 30 GB in all; hipMalloc / hipFree of GB-sized buffers
// cost milliseconds each and synchronise the device, and the HIP stream-ordered pool (hipMallocAsync on the device's
// DEFAULT pool, round 1) turned out to be both a process-wide side effect (its release threshold had to be changed)
// and slow to re-use blocks freed on another stream (a second graph build took 10x the first one).  So: blocks are
// hipMalloc'd once, kept in a size-ordered free list when released, and handed out again to requests of (nearly) the
// same size.  Ordering between the old and the new user of a block:
//   * same stream        -> stream order, nothing to do;
//   * another stream     -> the new stream waits for an event recorded on the old stream at release time;
//   * released "synced"  -> the caller has synchronised the device (graph / path-set destruction), nothing to wait for.
// Nothing here touches the device's default memory pool or any other process-wide HIP setting.  tlm_pool_trim returns
// every cached block to the driver (tlm_ctx_trim in the ABI).
#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>

#include "tlm_common.h"

namespace {

const hipStream_t DEAD_STREAM = (hipStream_t) (intptr_t) -1;     // the stream a block was used on no longer exists

struct Block {
    void *p = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;      // last user
    hipEvent_t ev = nullptr;           // recorded on `stream` at release (created on first need)
    bool ev_pending = false;
};

struct Pool {
    std::mutex mu;
    std::unordered_map<void *, Block *> live;
    std::multimap<size_t, Block *> cached;
    tlm_memory_info info{};
};

constexpr int MAX_DEVICES = 64;

Pool &pool_of(int device) {
    static Pool *pools = new Pool[MAX_DEVICES];      // never destroyed: path sets may be released during interpreter exit
    return pools[(device >= 0 && device < MAX_DEVICES) ? device : 0];
}

size_t round_size(size_t bytes) {
    if (bytes < 4096) return 4096;
    if (bytes <= (1u << 20)) return (bytes + 4095) & ~(size_t) 4095;
    return (bytes + ((1u << 21) - 1)) & ~(size_t) ((1u << 21) - 1);      // 2 MiB: the driver's large-page granularity
}

void release_block_to_driver(Block *b) {
    if (b->ev) (void) hipEventDestroy(b->ev);
    (void) hipFree(b->p);
    delete b;
}

thread_local double t_malloc_ms = 0;

}  // namespace

double tlm_pool_thread_malloc_ms() { return t_malloc_ms; }

int tlm_pool_alloc(void **out, size_t bytes, hipStream_t s) {
    *out = nullptr;
    int dev = 0;
    TLM_HIP(hipGetDevice(&dev));
    Pool &P = pool_of(dev);
    const size_t need = round_size(bytes);
    Block *b = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.cached.lower_bound(need);
        // a block may be up to 1/8 (+ 64 KiB) larger than the request: builds repeat their sizes exactly, so the slack
        // only matters for inputs that change a little between calls
        if (it != P.cached.end() && it->first <= need + need / 8 + (64u << 10)) {
            b = it->second;
            P.cached.erase(it);
            P.info.cached_bytes -= b->bytes;
            P.info.reuses++;
        }
    }
    if (b) {
        if (b->ev_pending && b->stream != s) TLM_HIP(hipStreamWaitEvent(s, b->ev, 0));
        b->ev_pending = false;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) {           // out of memory: give the cache back to the driver and try once more
            (void) hipGetLastError();
            tlm_pool_trim(dev);
            e = hipMalloc(&p, need);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        t_malloc_ms += ms;
        if (e != hipSuccess) {
            tlm_set_error("device allocation of %zu bytes failed: %s", need, hipGetErrorString(e));
            return TLM_EHIP;
        }
        b = new Block();
        b->p = p;
        b->bytes = need;
        std::lock_guard<std::mutex> lk(P.mu);
        P.info.driver_allocs++;
        P.info.driver_alloc_ms += ms;
    }
    b->stream = s;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.live[b->p] = b;
        P.info.live_bytes += b->bytes;
        if (P.info.live_bytes > P.info.peak_live_bytes) P.info.peak_live_bytes = P.info.live_bytes;
    }
    *out = b->p;
    return TLM_OK;
}

void tlm_pool_free(void *p, bool device_is_synced) {
    if (!p) return;
    hipPointerAttribute_t at;
    int dev = 0;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;
    else (void) hipGetLastError();
    Pool &P = pool_of(dev);
    Block *b = nullptr;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) return;          // not ours (never happens; a double free must not corrupt the list)
        b = it->second;
        P.live.erase(it);
        P.info.live_bytes -= b->bytes;
    }
    b->ev_pending = false;
    if (!device_is_synced) {
        if (b->stream == DEAD_STREAM) {
            (void) hipDeviceSynchronize();       // its stream is gone: whatever used the block on other streams must end
        } else {
            if (!b->ev && hipEventCreateWithFlags(&b->ev, hipEventDisableTiming) != hipSuccess) b->ev = nullptr;
            if (b->ev && hipEventRecord(b->ev, b->stream) == hipSuccess) b->ev_pending = true;
            else (void) hipStreamSynchronize(b->stream);
        }
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.cached.emplace(b->bytes, b);
    P.info.cached_bytes += b->bytes;
}

// A stream is about to be destroyed (it has been synchronised): no block may name it any more.
void tlm_pool_forget_stream(int device, hipStream_t s) {
    Pool &P = pool_of(device);
    std::lock_guard<std::mutex> lk(P.mu);
    for (auto &kv : P.live)
        if (kv.second->stream == s) kv.second->stream = DEAD_STREAM;
    for (auto &kv : P.cached)
        if (kv.second->stream == s) {
            kv.second->stream = DEAD_STREAM;
            kv.second->ev_pending = false;
        }
}

void tlm_pool_trim(int device) {
    Pool &P = pool_of(device);
    std::multimap<size_t, Block *> gone;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        gone.swap(P.cached);
        P.info.cached_bytes = 0;
    }
    if (gone.empty()) return;
    int cur = 0;
    (void) hipGetDevice(&cur);
    (void) hipSetDevice(device);
    (void) hipDeviceSynchronize();
    for (auto &kv : gone) release_block_to_driver(kv.second);
    (void) hipSetDevice(cur);
}

void tlm_pool_info(int device, tlm_memory_info *out) {
    Pool &P = pool_of(device);
    std::lock_guard<std::mutex> lk(P.mu);
    *out = P.info;
}
