// telomeri_amd/csrc/device_pool.hip -- the library's own device-memory cache (one per device, process lifetime).
//
// Every device buffer of the library (graph arrays, build temporaries, walk workspaces, path sets) comes from here.
// A build of the A. thaliana-shaped graph allocates ~60 buffers, 30 GB in all; hipMalloc / hipFree of GB-sized buffers
// cost milliseconds each and synchronise the device, and the HIP stream-ordered pool (hipMallocAsync on the device's
// DEFAULT pool, round 1) turned out to be both a process-wide side effect (its release threshold had to be changed)
// and slow to re-use blocks freed on another stream (a second graph build took 10x the first one).  So: blocks are
// hipMalloc'd once, kept in a size-ordered free list when released, and handed out again to requests of (nearly) the
// same size; a request that finds no such block but a much larger one takes its share of that one (the rest stays in the
// list as a block of its own.  Pieces are never merged again: a freed block keeps its size, so a caller that repeats its
// requests -- a graph rebuilt, a walk call repeated -- finds every block it used the last time, whatever happened in
// between; merged neighbours did not, and the driver call that then became necessary took a second right behind a
// large release).  Without that the host program allocated 30 GB afresh for its walks (0.85 s of hipMalloc, behind the
// address-space lock another thread was holding for an unmapping) while 35 GB sat in the list: the uploaded text, the
// columns and the build's temporaries, all of other sizes.  Ordering between the old and the new user of a block:
//   * same stream        -> stream order, nothing to do;
//   * another stream     -> the new stream waits for an event recorded on the old stream at release time;
//   * released "synced"  -> the caller has synchronised the device (graph / path-set destruction), nothing to wait for;
//   * a split leaves what is pending on the block pending on its rest.
// Nothing here touches the device's default memory pool or any other process-wide HIP setting.  tlm_pool_trim returns
// every cached block to the driver (tlm_ctx_trim in the ABI).
#include <algorithm>
#include <chrono>
#include <iterator>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "tlm_common.h"

namespace {

const hipStream_t DEAD_STREAM = (hipStream_t) (intptr_t) -1;     // the stream a block was used on no longer exists

struct Block;

struct Slab {                          // one driver allocation; its blocks tile it
    void *base = nullptr;
    size_t bytes = 0;
    std::map<size_t, Block *> pieces;  // by offset, live and cached alike
    size_t live = 0;                   // pieces handed out
};

struct Block {
    void *p = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;      // last user
    hipEvent_t ev = nullptr;           // recorded on `stream` at release (created on first need)
    bool ev_pending = false;
    Slab *slab = nullptr;
    bool cached = false;
    std::multimap<size_t, Block *>::iterator where;     // in Pool::cached while `cached`
    size_t offset() const { return (size_t) ((char *) p - (char *) slab->base); }
};

struct Pool {
    std::mutex mu;
    std::unordered_map<void *, Block *> live;
    std::multimap<size_t, Block *> cached;
    tlm_memory_info info{};
};

constexpr size_t SPLIT_MIN = 8u << 20;        // blocks are split only into pieces of at least this size

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

void drop_block(Block *b) {             // the bookkeeping of a block that is no more (merged away, or its slab is released)
    if (b->ev) (void) hipEventDestroy(b->ev);
    delete b;
}

void cache_insert(Pool &P, Block *b) {  // P.mu held
    b->cached = true;
    b->where = P.cached.emplace(b->bytes, b);
    P.info.cached_bytes += b->bytes;
}

void cache_remove(Pool &P, Block *b) {  // P.mu held
    P.cached.erase(b->where);
    b->cached = false;
    P.info.cached_bytes -= b->bytes;
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
            cache_remove(P, b);
            b->slab->live++;
            P.info.reuses++;
        } else if (need >= SPLIT_MIN && it != P.cached.end() && it->first >= need + SPLIT_MIN) {
            // the smallest block that holds the request with a useful rest: the request takes its head
            Block *big = it->second;
            cache_remove(P, big);
            // what is pending on the block stays pending on its rest (the event is the rest's); the head's new user is
            // either on the same stream (stream order) or made to wait for the event here, on the device
            if (big->ev_pending && big->stream != s) (void) hipStreamWaitEvent(s, big->ev, 0);
            b = new Block();
            b->p = big->p;
            b->bytes = need;
            b->stream = s;
            b->slab = big->slab;
            b->slab->live++;
            big->p = (char *) big->p + need;
            big->bytes -= need;
            big->slab->pieces.erase(b->offset());
            big->slab->pieces[b->offset()] = b;
            big->slab->pieces[big->offset()] = big;
            cache_insert(P, big);
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
        b->slab = new Slab();
        b->slab->base = p;
        b->slab->bytes = need;
        b->slab->pieces[0] = b;
        b->slab->live = 1;
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
        b->slab->live--;
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
    cache_insert(P, b);
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
    // driver allocations none of whose blocks is in use: their blocks leave the list, the allocation goes back
    std::vector<Slab *> gone;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        std::vector<Slab *> seen;
        for (auto &kv : P.cached) {
            Slab *sl = kv.second->slab;
            if (sl->live == 0 && std::find(seen.begin(), seen.end(), sl) == seen.end()
                && std::all_of(sl->pieces.begin(), sl->pieces.end(), [](const std::pair<const size_t, Block *> &pc) { return pc.second->cached; }))
                seen.push_back(sl);
        }
        for (Slab *sl : seen) {
            for (auto &pc : sl->pieces) cache_remove(P, pc.second);
            gone.push_back(sl);
        }
    }
    if (gone.empty()) return;
    int cur = 0;
    (void) hipGetDevice(&cur);
    (void) hipSetDevice(device);
    (void) hipDeviceSynchronize();
    for (Slab *sl : gone) {
        for (auto &pc : sl->pieces) drop_block(pc.second);
        (void) hipFree(sl->base);
        delete sl;
    }
    (void) hipSetDevice(cur);
}

void tlm_pool_info(int device, tlm_memory_info *out) {
    Pool &P = pool_of(device);
    std::lock_guard<std::mutex> lk(P.mu);
    *out = P.info;
}
