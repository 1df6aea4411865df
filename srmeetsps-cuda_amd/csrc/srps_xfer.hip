// Copies between the CALLER's host arrays and device memory.
//
// The device never reads or writes the caller's pages: every such copy goes through a pinned buffer of the library's own
// (hipHostMalloc: allocated by the driver, never moved), filled / drained by the host -- several threads for the large ones, so that
// the host's memcpy keeps up with PCIe.  Why: the runtime's own ways of reaching pageable memory -- hipHostRegister on the caller's
// array (what srps_setup did until round 4), and its pinned path for large hipMemcpy calls -- map the caller's pages into the device
// while the copy runs.  When the kernel moves those pages meanwhile (numpy asks for transparent huge pages on every array of 4 MB and
// more; khugepaged then collapses recycled heap pages seconds later), the copy engine faulted on the pool's kernel: "Memory access
// fault by GPU" at the first 2 MB boundary inside the image array, or inside the mask / depth array of a set-up -- three times in
// round 4's test runs (tools/stress_upload_thp.py).  A copy through a buffer that cannot move has no such window.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "srps_internal.h"

namespace srps {
namespace {

constexpr size_t kSlot = (size_t)2 << 20;      // small transfers: bytes per slot (37 us of PCIe, 0.2 ms of one thread's memcpy)
constexpr int kSlots = 32;                     // 64 MB pinned in all
constexpr size_t kBig = (size_t)16 << 20;      // large transfers: the same buffer as four slots of 16 MB, each filled by all copying threads together
constexpr int kBigSlots = 4;                   // (one DMA per 16 MB: per 2 MB slot the runtime's calls, serialised among 16 threads, cost a third of the rate)
static_assert(kBig * kBigSlots == kSlot * kSlots, "one buffer, two ways to cut it");
constexpr int kMaxThreads = 16;

struct Bounce {
    std::mutex m;                              // one transfer at a time owns the slots
    char* base = nullptr;                      // allocated by the first transfer, kept for the life of the process (64 MB of pinned host memory:
                                               // pinning takes milliseconds, contexts come and go)
};
Bounce g_bounce;

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

int bounce_base(char** out) {
    if (!g_bounce.base) {
        void* p = nullptr;
        const hipError_t e = hipHostMalloc(&p, kSlot * kSlots, hipHostMallocPortable);
        if (e != hipSuccess) return hip_fail(e, "hipHostMalloc (transfer buffer)", __FILE__, __LINE__);
        g_bounce.base = (char*)p;
    }
    *out = g_bounce.base;
    return SRPS_OK;
}

int copy_threads(size_t bytes) {
    if (bytes < 2 * kBig) return 1;
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(2u, std::min<unsigned>(hc / 4, (unsigned)kMaxThreads));
}

// all threads of a large transfer meet here twice per 16 MB; they wait for a fraction of a millisecond at most (one DMA)
struct SpinBarrier {
    std::atomic<int> count{0}, gen{0};
    int n = 1;
    void wait() {
        const int g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            count.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
            return;
        }
        for (int spins = 0; gen.load(std::memory_order_acquire) == g; ++spins) {
            if (spins < 2048) __builtin_ia32_pause(); else std::this_thread::yield();
        }
    }
};

// the single-thread form: chunks of 2 MB through a ring of slots, the memcpy of one chunk while the previous ones cross PCIe
int upload_small(char* base, void* d_dst, const void* h_src, size_t bytes, hipStream_t st, SlotEvents& se) {
    constexpr int kRing = 4;
    const size_t nch = (bytes + kSlot - 1) / kSlot;
    for (size_t i = 0; i < nch; ++i) {
        const int s = (int)(i % kRing);
        const size_t off = i * kSlot, len = std::min(kSlot, bytes - off);
        if (se.make(s) != SRPS_OK || se.wait(s) != SRPS_OK) return SRPS_ERR_HIP;
        memcpy(base + (size_t)s * kSlot, (const char*)h_src + off, len);
        if (hipMemcpyAsync((char*)d_dst + off, base + (size_t)s * kSlot, len, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipEventRecord(se.ev[s], st) != hipSuccess) return SRPS_ERR_HIP;
        se.pending[s] = true;
    }
    return SRPS_OK;
}

// the many-thread form: every 16 MB slot is filled by all T threads (a slice each), then goes out as ONE copy, issued by thread 0
int upload_big(srps_ctx* ctx, char* base, void* d_dst, const void* h_src, size_t bytes, hipStream_t st, SlotEvents& se, int T_wanted) {
    const size_t nch = (bytes + kBig - 1) / kBig;
    SpinBarrier bar;
    std::atomic<int> err{0}, go{0};
    int T = 1;                                 // threads that take part: set before `go` (a thread that could not be started is not waited for)
    auto worker = [&](int t) {
        while (go.load(std::memory_order_acquire) == 0) std::this_thread::yield();
        if (t != 0 && hipSetDevice(ctx->device) != hipSuccess) err.store(1);
        for (size_t c = 0; c < nch; ++c) {
            const int s = (int)(c % kBigSlots);
            const size_t off = c * kBig, len = std::min(kBig, bytes - off);
            if (t == 0 && !err.load() && (se.make(s) != SRPS_OK || se.wait(s) != SRPS_OK)) err.store(1);      // the slot's previous copy has run
            bar.wait();
            if (!err.load()) {
                const size_t per = ((len + (size_t)T - 1) / (size_t)T + 63) & ~(size_t)63, b = std::min(len, per * (size_t)t), e = std::min(len, b + per);
                if (e > b) memcpy(base + (size_t)s * kBig + b, (const char*)h_src + off + b, e - b);
            }
            bar.wait();
            if (t == 0 && !err.load()) {
                if (hipMemcpyAsync((char*)d_dst + off, base + (size_t)s * kBig, len, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(se.ev[s], st) != hipSuccess) err.store(1);
                else se.pending[s] = true;
            }
        }
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < T_wanted; ++t) { th.emplace_back(worker, t); T = t + 1; }
    } catch (...) {                            // no more threads to be had: the ones that started do the work
    }
    bar.n = T;
    go.store(1, std::memory_order_release);
    worker(0);
    for (auto& x : th) x.join();
    return err.load() ? SRPS_ERR_HIP : SRPS_OK;
}

}  // namespace

// h_src -> d_dst on stream st.  Returns when the caller's array has been read AND the last slot's copy has run (the slots go back to
// the pool): work queued on st afterwards finds the data, and so does the host after a wait for st.
int host_upload(srps_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SRPS_OK;
    SRPS_REQUIRE(d_dst && h_src, SRPS_ERR_INVALID, "host_upload: null pointer");
    std::lock_guard<std::mutex> lk(g_bounce.m);
    char* base = nullptr;
    SRPS_TRY(bounce_base(&base));
    SlotEvents se;
    const int T = copy_threads(bytes);
    int rc = T == 1 ? upload_small(base, d_dst, h_src, bytes, st, se) : upload_big(ctx, base, d_dst, h_src, bytes, st, se, T);
    for (int s = 0; s < kSlots && rc == SRPS_OK; ++s) rc = se.wait(s);
    if (rc != SRPS_OK) { set_error("host_upload: a HIP call failed (%s)", hipGetErrorString(hipGetLastError())); return rc; }
    return SRPS_OK;
}

// d_src -> h_dst, after the work queued on st.  Returns when the caller's array holds the data.
int host_download(srps_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return SRPS_OK;
    SRPS_REQUIRE(h_dst && d_src, SRPS_ERR_INVALID, "host_download: null pointer");
    std::lock_guard<std::mutex> lk(g_bounce.m);
    char* base = nullptr;
    SRPS_TRY(bounce_base(&base));
    SlotEvents se;
    const size_t nch = (bytes + kSlot - 1) / kSlot;
    constexpr int kRing = 4;                   // copies in flight ahead of the host's memcpy
    int rc = SRPS_OK;
    for (size_t i = 0; i < nch + kRing && rc == SRPS_OK; ++i) {
        if (i >= (size_t)kRing) {              // chunk i - kRing has arrived: hand it to the caller
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
    (void)ctx;
    if (rc != SRPS_OK) { set_error("host_download: a HIP call failed (%s)", hipGetErrorString(hipGetLastError())); return rc; }
    return SRPS_OK;
}

}  // namespace srps
