// common.hpp -- shared declarations of libirotavg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <unistd.h>
#include <vector>

#include "../../include/irotavg_hip.h"

namespace irh {

struct HipError {
    hipError_t e;
};

#define IRH_CHECK(expr)                                                                        \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            std::fprintf(stderr, "[irotavg_hip] %s failed: %s (%s:%d)\n", #expr,               \
                         hipGetErrorString(_e), __FILE__, __LINE__);                           \
            throw ::irh::HipError{_e};                                                         \
        }                                                                                      \
    } while (0)

// Device memory pool. A solve handle owns a few hundred buffers; hipMalloc / hipFree cost ~0.1 ms
// each and hipFree synchronises the device, which made handle churn (one-shot calls, the global
// re-solve of rot_avg on every loop closure) cost tens of milliseconds. Released blocks are cached
// per device and size and handed out again; a block released since the last device-wide
// synchronisation is "dirty" (work of its previous owner may still be in flight) and the first
// reuse of a dirty block synchronises the device once. irotavg_trim_memory() returns the cache to
// the driver; IROTAVG_POOL_LIMIT_MB bounds it (default 16384, 0 disables pooling).
struct DevPool {
    struct Block {
        void *p;
        size_t bytes;
        bool dirty;
        uint64_t epoch;  // value of DevPool::epoch when the block was released
    };
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, Block> cache;
    size_t cached = 0, limit = (size_t)16384 << 20;
    uint64_t epoch = 0;  // advances with every release
    DevPool() {
        if (const char *e = std::getenv("IROTAVG_POOL_LIMIT_MB")) limit = (size_t)std::strtoull(e, nullptr, 10) << 20;
    }
    static DevPool &get() {
        static DevPool *pool = new DevPool();  // never destroyed: no hipFree after runtime teardown
        return *pool;
    }
    // Small blocks: multiples of 512 B. Blocks of 64 KiB and more: the next step of a geometric ladder (x 1.19 per
    // step, four steps per doubling) -- the global re-solves of a growing view-graph ask for slightly more every
    // time (every array scales with the views / edges seen so far), and a request one step up would miss every block
    // the previous re-solve released (rot_avg inside the stream: 22 ms against 8 ms at a steady size; hipMalloc of
    // the handle's few hundred buffers). At most 19 % of head-room, on a 288 GB device.
    // headroom() > 1: requests of this thread are enlarged by that factor first (the handles of a GROWING view-graph,
    // resident.hip: every global re-solve is a few per cent larger than the last one; with half as much again the
    // blocks of one solve serve the following ones until the graph has grown by 50 %)
    // Scoped to the THREADS that work for the handle that asked for it (round 5; until then a process-wide switch:
    // while one session's re-solve ran, every other handle's allocations were enlarged as well): the thread that builds
    // and drives the handle holds a HeadroomScope, and so does every worker thread while it runs a job for a Graph whose
    // pool_headroom flag is set (l1ra's chains: the clones inherit the flag).
    static int &headroom_depth() {
        static thread_local int d = 0;
        return d;
    }
    struct HeadroomScope {
        bool on;
        explicit HeadroomScope(bool enable = true) : on(enable) {
            if (on) headroom_depth()++;
        }
        ~HeadroomScope() {
            if (on) headroom_depth()--;
        }
        HeadroomScope(const HeadroomScope &) = delete;
        HeadroomScope &operator=(const HeadroomScope &) = delete;
    };
    static double headroom() { return headroom_depth() > 0 ? 1.5 : 1.0; }
    static size_t round_up(size_t bytes) {
        if (headroom() > 1.0 && bytes >= (64u << 10)) bytes = (size_t)((double)bytes * headroom());
        if (bytes < (64u << 10)) return (bytes + 511) & ~(size_t)511;
        size_t p2 = (size_t)64 << 10;
        while (p2 * 2 <= bytes) p2 *= 2;
        const size_t steps[4] = {p2, p2 + p2 / 4 - p2 / 16, p2 + p2 / 2 - p2 / 12, p2 + p2 * 11 / 16};  // ~ x1, x1.19, x1.41, x1.69
        for (size_t s : steps)
            if (bytes <= s) return (s + 511) & ~(size_t)511;
        return p2 * 2;
    }
    // returns nullptr if nothing suitable is cached; *got = size of the block handed out
    void *take(int dev, size_t bytes, size_t *got) {
        bool sync = false;
        void *p = nullptr;
        uint64_t epoch_seen = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = cache.lower_bound({dev, bytes});
            if (it == cache.end() || it->first.first != dev ||
                it->first.second > (headroom() > 1.0 ? 2 * bytes : bytes + bytes / 8 + 4096))
                return nullptr;
            p = it->second.p;
            *got = it->second.bytes;
            sync = it->second.dirty;
            cached -= it->second.bytes;
            cache.erase(it);
            epoch_seen = epoch;
        }
        if (sync) {
            // The block (and every block released before this point) may still be in use by its
            // previous owner's queued work: synchronise FIRST, and only then mark as clean the
            // blocks that were already cached when the synchronisation began (a block released
            // while it ran carries a newer epoch and stays dirty). Another thread that takes a
            // dirty block meanwhile synchronises itself -- never hands it out early.
            (void)hipDeviceSynchronize();
            std::lock_guard<std::mutex> lk(mu);
            for (auto &kv : cache)
                if (kv.first.first == dev && kv.second.epoch <= epoch_seen) kv.second.dirty = false;
        }
        return p;
    }
    void give(int dev, void *p, size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cached + bytes <= limit) {
                cache.insert({{dev, bytes}, Block{p, bytes, true, ++epoch}});
                cached += bytes;
                return;
            }
        }
        (void)hipFree(p);
    }
    size_t trim() {
        std::multimap<std::pair<int, size_t>, Block> drop;
        size_t freed = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(cache);
            freed = cached;
            cached = 0;
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto &kv : drop) {
            (void)hipSetDevice(kv.first.first);
            (void)hipFree(kv.second.p);
        }
        (void)hipSetDevice(cur);
        return freed;
    }
};

// Streams are recycled the same way (hipStreamCreate / hipStreamDestroy cost 1-2 ms each): a
// stream handed back must be idle (synchronised by its owner).
struct StreamPool {
    std::mutex mu;
    std::multimap<int, hipStream_t> idle;
    static StreamPool &get() {
        static StreamPool *pool = new StreamPool();
        return *pool;
    }
    hipStream_t take() {
        int dev = 0;
        IRH_CHECK(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = idle.find(dev);
            if (it != idle.end()) {
                hipStream_t s = it->second;
                idle.erase(it);
                return s;
            }
        }
        hipStream_t s = nullptr;
        IRH_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return s;
    }
    void give(hipStream_t s, int dev = -1) {
        if (!s) return;
        if (dev < 0) (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() < 64) {
            idle.insert({dev, s});
            return;
        }
        (void)hipStreamDestroy(s);
    }
    // Three streams created back to back and kept together. The runtime multiplexes its streams onto a few hardware
    // queues (four by default), a new stream going to the least loaded one: three streams created in a row sit on
    // three different queues, three arbitrary streams of the pool often do not -- and kernels of streams that share
    // a queue do not overlap (l1ra's three concurrent solver chains: 9.9 ms per call or 13.1 ms, fixed per process,
    // depending on which streams the pool happened to hand out).
    struct Trio {
        hipStream_t s[3];
    };
    std::multimap<int, Trio> idle3;
    Trio take_trio() {
        int dev = 0;
        IRH_CHECK(hipGetDevice(&dev));
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = idle3.find(dev);
            if (it != idle3.end()) {
                Trio t = it->second;
                idle3.erase(it);
                return t;
            }
        }
        Trio t{{nullptr, nullptr, nullptr}};
        for (int c = 0; c < 3; c++) IRH_CHECK(hipStreamCreateWithFlags(&t.s[c], hipStreamNonBlocking));
        return t;
    }
    void give_trio(const Trio &t, int dev) {
        std::lock_guard<std::mutex> lk(mu);
        if (idle3.size() < 16) {
            idle3.insert({dev, t});
            return;
        }
        for (int c = 0; c < 3; c++) (void)hipStreamDestroy(t.s[c]);
    }
    void trim() {
        std::multimap<int, hipStream_t> drop;
        std::multimap<int, Trio> drop3;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(idle);
            drop3.swap(idle3);
        }
        for (auto &kv : drop) (void)hipStreamDestroy(kv.second);
        for (auto &kv : drop3)
            for (int c = 0; c < 3; c++) (void)hipStreamDestroy(kv.second.s[c]);
    }
};

// Pinned host staging blocks (hipHostMalloc) for the small device -> host read-backs that steer the
// solve (PCG done flag + scalars, score partials, reduction partials of l1decode_pd): a copy into
// pageable memory goes through the runtime's own staging and left the GPU idle ~30 us per host
// decision (kernel trace, round 2); into pinned memory it is a plain DMA. Fixed-size blocks, recycled.
struct PinPool {
    static constexpr size_t kBytes = 128 * 1024;
    std::mutex mu;
    std::vector<void *> idle;
    static PinPool &get() {
        static PinPool *pool = new PinPool();
        return *pool;
    }
    void *take() {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (!idle.empty()) {
                void *p = idle.back();
                idle.pop_back();
                return p;
            }
        }
        void *p = nullptr;
        // mapped + coherent (fine-grained): a kernel may store into the block and the host may poll it while the
        // kernel's stream is still busy (publish_parts / wait_published, solver.hip)
        static const bool plain = std::getenv("IROTAVG_PIN_DEFAULT") != nullptr;  // experiments (with IROTAVG_NO_POLL=1)
        IRH_CHECK(hipHostMalloc(&p, kBytes, plain ? hipHostMallocDefault : (hipHostMallocMapped | hipHostMallocCoherent)));
        std::memset(p, 0, kBytes);
        return p;
    }
    void give(void *p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        if (idle.size() < 64) {
            idle.push_back(p);
            return;
        }
        (void)hipHostFree(p);
    }
};

// Device buffer (pooled hipMalloc; sized once per graph, HBM-resident for the handle's life).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool own = true;  // false: a non-owning alias of another handle's buffer (read-only data)
    size_t cap_bytes = 0;
    int dev = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), own(o.own), cap_bytes(o.cap_bytes), dev(o.dev) {
        o.p = nullptr;
        o.n = 0;
        o.cap_bytes = 0;
    }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            n = o.n;
            own = o.own;
            cap_bytes = o.cap_bytes;
            dev = o.dev;
            o.p = nullptr;
            o.n = 0;
            o.cap_bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p && own) DevPool::get().give(dev, p, cap_bytes);
        p = nullptr;
        n = 0;
        cap_bytes = 0;
        own = true;
    }
    void alias(const DevBuf &o) {
        release();
        p = o.p;
        n = o.n;
        own = false;
    }
    void alloc_like(const DevBuf &o, hipStream_t s) {
        alloc(o.n);
        if (o.n) IRH_CHECK(hipMemsetAsync(p, 0, o.n * sizeof(T), s));
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        const size_t bytes = DevPool::round_up(count * sizeof(T));
        IRH_CHECK(hipGetDevice(&dev));
        size_t got = 0;
        if (void *q = DevPool::get().take(dev, bytes, &got)) {
            p = (T *)q;
            cap_bytes = got;
            return;
        }
        hipError_t e = hipMalloc((void **)&p, bytes);
        if (e != hipSuccess && DevPool::get().trim() > 0) {  // out of memory: give the cache back first
            (void)hipGetLastError();
            e = hipMalloc((void **)&p, bytes);
        }
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            IRH_CHECK(e);
        }
        cap_bytes = bytes;
    }
    void upload(const T *h, size_t count, hipStream_t s) {
        if (count > n) alloc(count);
        if (count) IRH_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void upload(const std::vector<T> &h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void zero(hipStream_t s) {
        if (n) IRH_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

constexpr int kRowBlock = 256;  // threads per workgroup of the row / edge / vector kernels (4 waves)
constexpr int kMaxParts = 512;  // dot-product partials (= max grid of a reducing kernel)
constexpr int kMaxLevels = 16;
// A pivot of a dense elimination that is not above kDeadTol x the largest diagonal entry of the
// matrix (the scale of the rounding errors in every Schur complement) is "dead": the unknown
// solves to 0. That is an isolated view, or the last view of a component that no weighted edge
// ties to a fixed view -- what SPQR's rank detection / the oracle's Cholesky decide, too.
constexpr double kDeadTol = 1e-13;
constexpr int kSellUnroll = 8;  // slice widths are multiples of this (batch size of the row loops)
constexpr int kBandMax = 4;     // coarsest operators up to this half-bandwidth are inverted by the banded kernels (dense.hip)

// per-edge flag bits (host-built; see build.cpp)
enum : uint8_t {
    EF_CJ = 1,  // A(k, j-f) = +1 present   (ral/l1_irls.cpp:770-772)
    EF_CI = 2,  // A(k, i-f) = -1 present   (ral/l1_irls.cpp:774-776)
};
// boundary-slot flag bits
enum : uint8_t {
    BF_IRLS = 1,  // contributes to A'D^2A (the make_A matrix has a coefficient for this row)
    BF_L1H = 2,   // contributes to the make_AtA matrix (ral/l1_irls.cpp:825-835)
    BF_NEG = 4,   // self loop: make_AtA ends with -1 on the diagonal (see oracle lap_fill_AtA_times)
};

// scalar block shared by the PCG kernels (device memory, doubles)
enum ScalIdx : int {
    SC_RZ0 = 0,   // rz of parity 0 (3 values, padded to 4)
    SC_RZ1 = 4,   // rz of parity 1
    SC_BB = 8,    // ||b||^2 per column
    SC_RELRES = 12,  // ||r||/||b|| per column at the last check
    // Chronopoulos-Gear recurrences of the two-launch iteration (cgcg.hip): gamma = r.u and alpha of
    // the previous iteration, two parities each
    SC_GAM0 = 16,
    SC_GAM1 = 20,
    SC_ALF0 = 24,
    SC_ALF1 = 28,
    SC_DSCALE = 32,  // scale of the (re-used) dense inverse decided on the device (dense_check_async)
    SC_COUNT = 36
};
// int flags block
enum FlagIdx : int {
    FL_DONE = 0,   // 0 running, 1 converged, 2 breakdown (non-finite scalar)
    FL_ITERS = 1,  // PCG iterations performed
    FL_STALE = 2,  // dense_check_async: the coarse operator changed non-uniformly (re-invert next time)
    FL_COUNT = 4
};

// One multigrid level. The off-diagonal part of the weighted Laplacian is held in SELL-64 with
// 2-entry interleave: rows are cut into slices of 64 consecutive rows (one wavefront), slice s is
// `sl_off[s+1] - sl_off[s]` entry-columns wide (a multiple of 8), and entries 2q, 2q+1 of the row
// in lane l sit next to each other at (sl_off[s] + 2q) * 64 + 2l (+1): one lane owns one row and
// reads its entries two at a time, so a wave-level load of `val` is one contiguous 1 KiB run of
// 16 B per lane (`col`: 512 B of 8 B per lane), the inner loop has a wave-uniform trip count and
// no cross-lane reduction is needed. Within a row the entries whose column lies in the window of
// the row's 256-row workgroup tile ([tile - 64, tile + 320)) come first -- the first `sl_near[s]`
// entry-columns of the slice hold only such entries -- so the SpMV can serve them from an LDS copy
// of that window instead of gathering from global memory. Padding: val = 0, col = a row of the slice.
__host__ __device__ inline size_t sell_pos(int o0, int k, int lane) {
    return (size_t)(o0 + (k & ~1)) * 64 + (size_t)lane * 2 + (size_t)(k & 1);
}
constexpr int kWinHalo = 64;   // rows on either side of a 256-row tile held in the LDS window
constexpr int kWinLen = 256 + 2 * kWinHalo;
constexpr int kL1Win = kWinLen / 8;  // level-1 rows under a tile window (aggregates of 8)
constexpr int kL1Ext = 64;           // level-1 rows [tile/8 - 16, tile/8 + 48): the window's rows and their neighbours
struct Level {
    int n = 0;        // rows
    int nnz = 0;      // real off-diagonal entries
    int agg = 0;      // rows per aggregate towards the next (coarser) level; 0 on the coarsest
    int nsl = 0;      // slices
    long long sell_len = 0;  // 64 * total entry-columns
    int max_near = 0;        // widest near part of a slice (entry-columns)
    int uni_w = 0;           // every slice is this wide and all of it is near (band graphs), else 0:
                             // kernels then derive sl_off / sl_near instead of loading them
    DevBuf<int> sl_off, sl_near, col;
    DevBuf<double> val;     // off-diagonal values (<= 0 for a Laplacian)
    DevBuf<double> excess;  // diag - sum|offdiag| : Dirichlet mass from fixed neighbours
    DevBuf<double> diag, idg;  // diagonal and its inverse (0 where the diagonal is 0)
    // value refresh from the finer level: coarse entry c (CSR order) sums the finer SELL
    // positions cidx[cptr[c]..cptr[c+1]) and is stored at SELL position cpos[c]
    DevBuf<int> cptr, cidx, cpos;
    DevBuf<int> crow;   // CSR row pointers of the coarse entries (k_coarse_level)
    int max_row = 0;    // longest row (entries)
    // multigrid work vectors (double4 with 3 active components)
    DevBuf<double4> b, x, y, e;
};

// Worker threads for the host phases of the graph build and of rot_avg. Spawning and joining 15 threads
// costs 0.34 ms on the GPU box's host, a build has ~17 such loops and a global re-solve in a stream runs one
// every few hundred frames: 15 persistent workers take a job through one mutex / condition variable, spin
// briefly between jobs (the loops of a build follow each other within microseconds) and otherwise sleep.
// One job at a time (a second caller, e.g. one of l1ra's host threads, spawns threads as before); a process
// forked after the pool was created does the same (the workers do not exist in the child). The pool is never
// destroyed: its threads are detached and end with the process.
class HostPool {
public:
    static constexpr int kWorkers = 15;
    static HostPool &get() {
        static HostPool *p = new HostPool();
        return *p;
    }
    // fn(t) for t = 1 .. T-1 on the workers and fn(0) on the caller; false if the pool is busy or unusable
    // (the caller then spawns threads). A job that is itself running inside a pool job -- on a worker, or
    // on the caller thread inside fn(0) -- is refused by a thread-local flag (try_lock on a mutex the
    // thread already owns would be undefined behaviour). An exception thrown by fn on a worker is caught
    // there and rethrown on the caller once every chunk has ended.
    template <class F>
    bool run(int T, F &&fn) {
        if (T - 1 > kWorkers || getpid() != pid_ || inside()) return false;
        std::unique_lock<std::mutex> one(busy_, std::try_to_lock);
        if (!one.owns_lock()) return false;
        std::exception_ptr err;
        std::mutex err_mu;
        std::function<void(int)> job = [&fn, &err, &err_mu](int t) {
            try {
                fn(t);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!err) err = std::current_exception();
            }
        };
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            want_ = T - 1;
            remaining_.store(T - 1, std::memory_order_relaxed);
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        inside() = true;
        job(0);
        inside() = false;
        for (int spin = 0; remaining_.load(std::memory_order_acquire) != 0; spin++) {
            if (spin > 2000) std::this_thread::yield();
        }
        if (err) std::rethrow_exception(err);
        return true;
    }

private:
    static bool &inside() {
        static thread_local bool in_job = false;
        return in_job;
    }
    HostPool() : pid_(getpid()) {
        for (int w = 0; w < kWorkers; w++) std::thread([this, w] { loop(w); }).detach();
    }
    void loop(int w) {
        unsigned seen = 0;
        for (;;) {
            // spin for a while, then sleep on the condition variable
            bool got = false;
            for (int spin = 0; spin < 20000; spin++) {
                if (gen_.load(std::memory_order_acquire) != seen) {
                    got = true;
                    break;
                }
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
            }
            const std::function<void(int)> *job;
            int want;
            {
                std::lock_guard<std::mutex> lk(m_);  // job_ / want_ are published under the mutex
                seen = gen_.load(std::memory_order_acquire);
                job = job_;
                want = want_;
            }
            if (w < want) {
                inside() = true;
                (*job)(w + 1);  // never throws (run() wraps fn)
                inside() = false;
                remaining_.fetch_sub(1, std::memory_order_release);
            }
        }
    }
    const pid_t pid_;
    std::mutex m_, busy_;
    std::condition_variable cv_;
    std::atomic<unsigned> gen_{0};
    std::atomic<int> remaining_{0};
    const std::function<void(int)> *job_ = nullptr;
    int want_ = 0;
};

// The host phases of the build are loops over edges, rows or slices with independent iterations: they run
// on up to 16 host threads (contiguous chunks; every result is independent of the thread count --
// where a serial loop defined an order, the order is restored by sorting on the edge id).
template <class F>
inline void parallel_for(int64_t n, int64_t min_chunk, F &&fn) {
    if (n < 2 * std::max<int64_t>(1, min_chunk)) {  // the common small case (sliding windows): no look-ups at all
        fn((int64_t)0, n, 0);
        return;
    }
    static const unsigned hw = std::thread::hardware_concurrency();
    const char *fe = getenv("IROTAVG_BUILD_THREADS");  // read every time: the thread-count test changes it
    const int forced = fe ? std::min(16, std::max(1, atoi(fe))) : 0;
    int T = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / std::max<int64_t>(1, min_chunk)));
    if (forced) T = forced;
    if (T <= 1) {
        fn((int64_t)0, n, 0);
        return;
    }
    const int64_t step = (n + T - 1) / T;
    auto chunk = [&](int t) {
        const int64_t b = std::min(n, t * step), e = std::min(n, b + step);
        if (b < e) fn(b, e, t);
    };
    if (!getenv("IROTAVG_NO_HOST_POOL") && HostPool::get().run(T, chunk)) return;
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&chunk, t]() { chunk(t); });
    chunk(0);
    for (auto &x : th) x.join();
}


}  // namespace irh
