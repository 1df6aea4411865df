// Copies between the CALLER's host arrays and device memory.
//
// The device never reads or writes the caller's pages: every such copy goes through a pinned buffer of the library's own
// (hipHostMalloc: allocated by the driver, never moved), filled / drained by the host -- several threads for the large ones, so that
// the host's memcpy keeps up with PCIe.  Why: the runtime's own ways of reaching pageable memory -- hipHostRegister on the caller's
// array (what srps_setup did until round 4), and its pinned path for large hipMemcpy calls -- map the caller's pages into the device
// while the copy runs.  When the kernel moves those pages meanwhile (numpy asks for transparent huge pages on every array of 4 MB and
// more; khugepaged then collapses recycled heap pages seconds later), the copy engine faulted on the pool's kernel: "Memory access
// fault by GPU" at the first 2 MB boundary inside the image array, or inside the mask / depth array of a set-up -- three times in
// round 4's test runs (tools/stress_upload_thp.py, DESIGN.md §4.5, docs/HISTORY.md).  A copy through a buffer that cannot move has no such window.
//
// Round 5.  (i) Every transfer takes a buffer of its OWN from a pool (round-4 advisor finding: one buffer behind one mutex, held across
// waits on work the caller had queued, serialised the set-ups of `srps --gpus N`, made one context's srps_get block every other context,
// and could deadlock a phase-by-phase sharded driver).  No lock is held while anything is copied or waited for; two transfers never
// share a buffer; a buffer goes back to the pool when its transfer has ended.  It also lets srps_setup upload the mask and the depth
// maps WHILE the images stream (they used to wait for the image transfer's lock: 2 - 3 ms of a 23 ms set-up).  (ii) A large upload is
// a pipeline: T filling threads claim 1 MB pieces from one counter and copy them into the 16 MB slot their chunk maps to with
// non-temporal stores (no read-for-ownership of the destination, nothing of the 1 GB left dirty in the caches for the DMA to snoop);
// the calling thread only issues each slot's DMA when its pieces are in and frees the slot when the DMA's event has fired.  No
// barrier couples the threads: fillers run up to three chunks ahead of the copy engine.
#include <immintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "srps_internal.h"

namespace srps {
namespace {

constexpr size_t kSlot = (size_t)2 << 20;      // small transfers: bytes per slot (37 us of PCIe, 0.2 ms of one thread's memcpy)
constexpr int kSlots = 32;                     // 64 MB pinned per buffer
constexpr size_t kBig = (size_t)16 << 20;      // large transfers: the same buffer as four slots of 16 MB, one DMA each
constexpr int kBigSlots = 4;                   // (one DMA per 16 MB: per 2 MB slot the runtime's calls cost a third of the rate)
static_assert(kBig * kBigSlots == kSlot * kSlots, "one buffer, two ways to cut it");
constexpr size_t kPiece = (size_t)1 << 20;     // what a filling thread claims at a time
constexpr int kMaxThreads = 32;

// ---- the pool: pinned buffers are expensive to make (milliseconds) and contexts come and go, so they live as long as the process ----
struct PoolBuf { char* p; size_t bytes; bool wc; };      // wc: write-combined host memory (uploads only: the host never reads it)
struct BouncePool {
    std::mutex m;                              // guards the list only: never held while a transfer runs
    std::vector<PoolBuf> free_;
    int made = 0;
} g_pool;

// a buffer of at least `bytes` (two sizes occur: 64 MB, and four slots of the large-transfer chunk)
int pool_acquire(size_t bytes, PoolBuf* out, bool wc = false) {
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        for (size_t i = 0; i < g_pool.free_.size(); ++i)
            if (g_pool.free_[i].bytes >= bytes && g_pool.free_[i].wc == wc) { *out = g_pool.free_[i]; g_pool.free_.erase(g_pool.free_.begin() + (long)i); return SRPS_OK; }
    }
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | (wc ? hipHostMallocWriteCombined : 0));
    if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (transfer buffer)", __FILE__, __LINE__);
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        ++g_pool.made;
    }
    *out = PoolBuf{(char*)p, bytes, wc};
    return SRPS_OK;
}
// Back into the pool -- up to kPoolKeep idle buffers and kPoolKeepBytes of pinned memory are kept for the next transfers; what exceeds
// that is given back to the system (round-5 advisor finding: `srps --gpus 8` with its image uploader threads pinned about a gigabyte for
// the life of the process).  Freed outside the lock.
constexpr size_t kPoolKeep = 4, kPoolKeepBytes = (size_t)320 << 20;
void pool_release(const PoolBuf& b) {
    if (!b.p) return;
    bool keep = true;
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        size_t held = 0;
        for (const PoolBuf& f : g_pool.free_) held += f.bytes;
        keep = g_pool.free_.size() < kPoolKeep && held + b.bytes <= kPoolKeepBytes;
        if (keep) g_pool.free_.push_back(b);
    }
    if (!keep) (void)hipHostFree(b.p);
}
struct Lease {                                 // a transfer's buffer, back in the pool when the transfer ends (its copies have been waited for by then)
    PoolBuf buf{nullptr, 0, false};
    char* base = nullptr;
    ~Lease() { pool_release(buf); }
};

// the slots' events of one transfer, on the device of its stream; destroyed (after a wait) when the transfer ends
struct SlotEvents {
    hipEvent_t ev[kSlots] = {};
    bool pending[kSlots] = {};
    int make(int s) {
        if (ev[s]) return SRPS_OK;
        return hipEventCreateWithFlags(&ev[s], hipEventDisableTiming) == hipSuccess ? SRPS_OK : SRPS_ERR_HIP;
    }
    int wait(int s) {
        if (!pending[s]) return SRPS_OK;
        pending[s] = false;
        return hipEventSynchronize(ev[s]) == hipSuccess ? SRPS_OK : SRPS_ERR_HIP;
    }
    ~SlotEvents() {
        for (int s = 0; s < kSlots; ++s)
            if (ev[s]) { if (pending[s]) (void)hipEventSynchronize(ev[s]); (void)hipEventDestroy(ev[s]); }
    }
};

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
// bytes per DMA of a large transfer (SRPS_XFER_CHUNK_MB: development, tools/setup_time.py): a multiple of the piece
size_t big_chunk() {
    static const size_t c = (size_t)std::max(1, std::min(env_int("SRPS_XFER_CHUNK_MB", (int)(kBig >> 20)), 64)) << 20;      // (development knob; 64 MB x 4 slots = the largest buffer the pool keeps)
    return c;
}
// streams a large transfer's DMAs alternate between (SRPS_XFER_STREAMS; 1: all on the caller's stream): consecutive copies of ONE stream
// run one after the other with a gap between them, copies of two streams overlap
int big_streams() {
    static const int n = std::max(1, std::min(env_int("SRPS_XFER_STREAMS", 1), 2));
    return n;
}

// CPUs the process may really use: its affinity mask and its cgroup's quota (cpu.max: "quota period" in microseconds, or "max").  The GPU
// boxes of this pool show 256 logical CPUs and grant 16: a set-up whose 16 filling threads, issuing thread and structure thread ran
// together was throttled by the scheduler at random -- 21 / 26 / 32 ms for the same 1 GB (gpurun_out/r5c).
int cpu_budget() {
    static const int n = [] {
        int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = std::max(1, CPU_COUNT(&set));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long quota = atol(q);
                if (quota > 0) cpus = (int)std::min<long>(cpus, std::max<long>(1, (quota + period - 1) / period));
            }
            fclose(f);
        }
        return cpus;
    }();
    return n;
}

// filling threads of a large upload: a quarter of what the process may use (the rest is the caller's: its own threads, the runtime's),
// two to eight -- same-box sweeps on 1 GB: 4 threads 19.8 - 20.2 ms and 99 ms of CPU time per set-up, 8 threads 19.8 - 20.2 ms and 171 ms,
// 16 threads 21.0, 32 threads 23 - 38 (throttled by the CPU quota).  SRPS_XFER_THREADS overrides (development: tools/setup_time.py)
int copy_threads(size_t bytes) {
    if (bytes < kBig / 2) return 0;            // below 8 MB the calling thread copies by itself (upload_small)
    int t = std::max(2, std::min(cpu_budget() / 4, 8));
    if (bytes < 4 * kBig) t = std::min(t, 4);  // the mask, a depth map: a few threads for 1 ms of copying
    t = env_int("SRPS_XFER_THREADS", t);
    return std::max(1, std::min(t, kMaxThreads));
}

// dst is 64-byte aligned (slot base + a multiple of 1 MB); src is the caller's, any alignment; n any length
__attribute__((target("avx2"))) void copy_nt(char* dst, const char* src, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
        _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
        _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();                              // the streamed lines are globally visible before the piece is reported filled
}
inline void fill(char* dst, const char* src, size_t n) {
#if !defined(__HIP_DEVICE_COMPILE__)           // host code only: the device pass of hipcc parses this file too and has no x86 builtins
    static const bool nt = (__builtin_cpu_init(), __builtin_cpu_supports("avx2")) && env_int("SRPS_XFER_NT", 1) != 0;
    if (nt) { copy_nt(dst, src, n); return; }
#endif
    memcpy(dst, src, n);
}

// the single-thread form: chunks of 2 MB through a ring of slots, the memcpy of one chunk while the previous ones cross PCIe
int upload_small(char* base, void* d_dst, const void* h_src, size_t bytes, hipStream_t st, SlotEvents& se) {
    constexpr int kRing = 4;
    const size_t nch = (bytes + kSlot - 1) / kSlot;
    for (size_t i = 0; i < nch; ++i) {
        const int s = (int)(i % kRing);
        const size_t off = i * kSlot, len = std::min(kSlot, bytes - off);
        if (se.make(s) != SRPS_OK || se.wait(s) != SRPS_OK) return SRPS_ERR_HIP;
        fill(base + (size_t)s * kSlot, (const char*)h_src + off, len);
        if (hipMemcpyAsync((char*)d_dst + off, base + (size_t)s * kSlot, len, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(se.ev[s], st) != hipSuccess) return SRPS_ERR_HIP;
        se.pending[s] = true;
    }
    return SRPS_OK;
}

// The many-thread form.  Pieces are claimed in order from `next`; piece q lies in chunk q / ppc, which uses slot chunk % kBigSlots once
// the slot's previous chunk has crossed (`crossed` = number of chunks whose DMA has completed, published by the calling thread).
// `filled[chunk % kBigSlots]` counts the pieces in; the calling thread issues chunk c's DMA when its count is complete.
int upload_big(char* base, void* d_dst, const void* h_src, size_t bytes, hipStream_t st, SlotEvents& se, int T_wanted) {
    const size_t CH = big_chunk();
    const int ppc = (int)(CH / kPiece);
    const size_t nch = (bytes + CH - 1) / CH;
    const size_t npieces = (bytes + kPiece - 1) / kPiece;
    std::atomic<size_t> next{0};
    std::atomic<long> crossed{0};
    std::atomic<int> filled[kBigSlots];
    for (auto& f : filled) f.store(0, std::memory_order_relaxed);
    std::atomic<int> stop{0};
    auto worker = [&]() {
        for (;;) {
            const size_t q = next.fetch_add(1, std::memory_order_relaxed);
            if (q >= npieces) return;
            const size_t c = q / (size_t)ppc;
            const int s = (int)(c % kBigSlots);
            for (int spins = 0; (long)c - kBigSlots >= crossed.load(std::memory_order_acquire); ++spins) {      // the slot still holds chunk c - kBigSlots
                if (stop.load(std::memory_order_relaxed)) return;
                if (spins < 4096) __builtin_ia32_pause(); else std::this_thread::yield();
            }
            const size_t off = q * kPiece, len = std::min(kPiece, bytes - off);
            fill(base + (size_t)s * CH + (off - c * CH), (const char*)h_src + off, len);
            filled[s].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 0; t < T_wanted; ++t) th.emplace_back(worker);
    } catch (...) {                            // no more threads to be had: the ones that started do the work
    }
    if (th.empty()) return upload_small(base, d_dst, h_src, bytes, st, se);      // none at all: this thread, chunk by chunk
    int rc = SRPS_OK;
    // a second stream for every other chunk: both start behind the work already queued on st (the destination may be in use there)
    hipStream_t st2 = nullptr;
    hipEvent_t ev0 = nullptr;
    if (big_streams() > 1 && nch > 1) {
        if (hipStreamCreateWithFlags(&st2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev0, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ev0, st) != hipSuccess || hipStreamWaitEvent(st2, ev0, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (st2) (void)hipStreamDestroy(st2);
            st2 = nullptr;
        }
    }
    size_t issued = 0, done = 0;
    auto pieces_of = [&](size_t c) { return (int)std::min<size_t>((size_t)ppc, npieces - c * (size_t)ppc); };
    for (int s = 0; s < kBigSlots && rc == SRPS_OK; ++s) rc = se.make(s);
    int idle = 0;
    while (rc == SRPS_OK && done < nch) {
        bool progressed = false;
        if (issued < nch) {
            const int s = (int)(issued % kBigSlots);
            if (filled[s].load(std::memory_order_acquire) >= pieces_of(issued)) {
                const size_t off = issued * CH, len = std::min(CH, bytes - off);
                filled[s].store(0, std::memory_order_relaxed);          // nobody adds to it again before `crossed` lets the slot's next chunk in
                hipStream_t cs = (st2 && (issued & 1)) ? st2 : st;
                if (hipMemcpyAsync((char*)d_dst + off, base + (size_t)s * CH, len, hipMemcpyHostToDevice, cs) != hipSuccess ||
                    hipEventRecord(se.ev[s], cs) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
                se.pending[s] = true;
                ++issued;
                progressed = true;
            }
        }
        if (done < issued) {
            const int s = (int)(done % kBigSlots);
            const hipError_t q = hipEventQuery(se.ev[s]);
            if (q == hipSuccess) {
                se.pending[s] = false;
                ++done;
                crossed.store((long)done, std::memory_order_release);
                progressed = true;
            } else if (q != hipErrorNotReady) { rc = SRPS_ERR_HIP; break; }
            else (void)hipGetLastError();
        }
        // idle: a pause, and after a run of idle rounds the thread gives its time slice away -- the filling threads work under the same CPU
        // quota this loop would otherwise burn (round-5 advisor finding)
        if (progressed) idle = 0;
        else if (++idle < 64) __builtin_ia32_pause();
        else { std::this_thread::yield(); idle = 0; }
    }
    stop.store(1);
    if (rc != SRPS_OK) next.store(npieces);    // nothing more to claim
    for (auto& x : th) x.join();
    if (st2) {                                 // everything on the second stream has been waited for (done == nch), or is waited for now
        (void)hipStreamSynchronize(st2);
        (void)hipStreamDestroy(st2);
    }
    if (ev0) (void)hipEventDestroy(ev0);
    return rc;
}

}  // namespace

// h_src -> d_dst on stream st.  Returns when the caller's array has been read AND the last slot's copy has run (the buffer goes back to
// the pool): work queued on st afterwards finds the data, and so does the host after a wait for st.
int host_upload(srps_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SRPS_OK;
    SRPS_REQUIRE(d_dst && h_src, SRPS_ERR_INVALID, "host_upload: null pointer");
    const int T = copy_threads(bytes);
    Lease buf;
    static const bool wc = env_int("SRPS_XFER_WC", 0) != 0;      // (development: write-combined staging for the large uploads)
    SRPS_TRY(pool_acquire(T == 0 ? kSlot * kSlots : std::max(kSlot * kSlots, big_chunk() * kBigSlots), &buf.buf, wc && T > 0));
    buf.base = buf.buf.p;
    (void)ctx;
    int rc;
    {
        SlotEvents se;
        rc = T == 0 ? upload_small(buf.base, d_dst, h_src, bytes, st, se) : upload_big(buf.base, d_dst, h_src, bytes, st, se, T);
        for (int s = 0; s < kSlots && rc == SRPS_OK; ++s) rc = se.wait(s);
    }                                          // (~SlotEvents waits for whatever an error path left pending: the buffer is idle when it goes back)
    if (rc != SRPS_OK) { set_error("host_upload: a HIP call failed (%s)", hipGetErrorString(hipGetLastError())); return rc; }
    return SRPS_OK;
}

// d_src -> h_dst, after the work queued on st.  Returns when the caller's array holds the data.
int host_download(srps_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SRPS_OK;
    SRPS_REQUIRE(h_dst && d_src, SRPS_ERR_INVALID, "host_download: null pointer");
    Lease buf;
    SRPS_TRY(pool_acquire(kSlot * kSlots, &buf.buf));
    char* base = buf.buf.p;
    int rc = SRPS_OK;
    {
        SlotEvents se;
        const size_t nch = (bytes + kSlot - 1) / kSlot;
        constexpr int kRing = 8;               // copies in flight ahead of the host's memcpy
        for (size_t i = 0; i < nch + kRing && rc == SRPS_OK; ++i) {
            if (i >= (size_t)kRing) {          // chunk i - kRing has arrived: hand it to the caller
                const size_t j = i - kRing, off = j * kSlot, len = std::min(kSlot, bytes - off);
                const int s = (int)(j % kRing);
                if ((rc = se.wait(s)) != SRPS_OK) break;
                memcpy((char*)h_dst + off, base + (size_t)s * kSlot, len);
            }
            if (i < nch) {
                const size_t off = i * kSlot, len = std::min(kSlot, bytes - off);
                const int s = (int)(i % kRing);
                if (se.make(s) != SRPS_OK || hipMemcpyAsync(base + (size_t)s * kSlot, (const char*)d_src + off, len, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    hipEventRecord(se.ev[s], st) != hipSuccess) { rc = SRPS_ERR_HIP; break; }
                se.pending[s] = true;
            }
        }
    }
    (void)ctx;
    if (rc != SRPS_OK) { set_error("host_download: a HIP call failed (%s)", hipGetErrorString(hipGetLastError())); return rc; }
    return SRPS_OK;
}

// how many pinned buffers the process has made so far (tests: concurrent transfers each take their own)
int xfer_buffers_made() {
    std::lock_guard<std::mutex> lk(g_pool.m);
    return g_pool.made;
}

}  // namespace srps
