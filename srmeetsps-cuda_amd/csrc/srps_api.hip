// srps_api.hip -- the C ABI of include/srps.h: context, grid geometry (the host part of
// SRPS::execute, SRPS.cu:100-270), the phase operators and the alternating loop (SRPS.cu:272-335).
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <functional>
#include <thread>
#include <mutex>
#include <dlfcn.h>
#include "srps_internal.h"

namespace srps {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return SRPS_ERR_HIP;
}
int ensure(DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes) return SRPS_OK;
    if (b.p) SRPS_HIP(hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
    SRPS_HIP(hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return SRPS_OK;
}

// ---- tracing: roctx resolved at run time (no link dependency), HIP events per phase --------------------------------------
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
static roctx_push_t g_roctx_push = nullptr;
static roctx_pop_t g_roctx_pop = nullptr;
static bool g_roctx_ok = false;
static std::once_flag g_roctx_once;
// resolved once, whichever thread comes first (srps --gpus N runs one host thread per context; round-3 advisor finding: an
// unsynchronised state flag let a second thread skip a push whose pop it then issued)
static bool roctx_resolve() {
    std::call_once(g_roctx_once, [] {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            roctx_push_t push = (roctx_push_t)dlsym(h, "roctxRangePushA");
            roctx_pop_t pop = (roctx_pop_t)dlsym(h, "roctxRangePop");
            if (push && pop) { g_roctx_push = push; g_roctx_pop = pop; g_roctx_ok = true; break; }
        }
    });
    return g_roctx_ok;
}
static const char* const kPhaseNames[SRPS_N_PHASES] = {"srps:lighting", "srps:albedo_sweep", "srps:albedo_solve", "srps:depth_assembly",
                                                       "srps:depth_solve", "srps:energy", "srps:normals"};
PhaseSpan::PhaseSpan(srps_ctx* ctx, int ph) : c(ctx), phase(ph) {
    if (c->roctx && roctx_resolve()) { g_roctx_push(kPhaseNames[phase]); pushed = true; }
    if (c->phase_timing) {
        if (!c->ev_created) {
            for (int i = 0; i < SRPS_N_PHASES; ++i) { (void)hipEventCreate(&c->ev_begin[i]); (void)hipEventCreate(&c->ev_end[i]); }
            c->ev_created = true;
        }
        // a phase that runs twice between two srps_get_timings calls keeps its first begin and its last end
        if (!(c->ev_mask & (1u << phase))) (void)hipEventRecord(c->ev_begin[phase], c->stream);
    }
}
PhaseSpan::~PhaseSpan() {
    if (c->phase_timing && c->ev_created) { (void)hipEventRecord(c->ev_end[phase], c->stream); c->ev_mask |= 1u << phase; }
    if (pushed) g_roctx_pop();                        // only the range this span opened itself
}

template <typename T>
static int dalloc(T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    SRPS_HIP(hipMalloc((void**)p, n * sizeof(T)));
    return SRPS_OK;
}
template <typename T>
static void dfree(T*& p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

// The grid's arrays are carved out of one arena (Grid::arena), which outlives a release: a later bind that needs no more than it
// holds allocates nothing.  grid_free gives the arena back (srps_destroy).
static void grid_release(Grid& G) {
    G.d_gofp = nullptr; G.d_imask = nullptr; G.d_imasks = nullptr; G.d_flags = nullptr; G.d_lr_index = nullptr;
    G.d_tile_cls[0] = G.d_tile_cls[1] = G.d_tile_cls[2] = nullptr;
    G.d_tile_list[0] = G.d_tile_list[1] = G.d_tile_list[2] = nullptr;
    dfree(G.d_M);                                         // the stored tensor is allocated on demand (grid_need_M), outside the arena
    G.d_q = nullptr; G.d_G = nullptr; G.d_tconsts = nullptr; G.G_planes = 0; G.tensor_channels = 0;
    G.d_x = G.d_x2 = G.d_r = G.d_r2 = G.d_p = G.d_w = G.d_w2 = G.d_save = nullptr; G.d_part4 = nullptr;
    G.d_pw_part = G.d_rr_part = G.d_misc_part = nullptr; G.d_scal = nullptr;      // d_scal lives in the context's report record
    G.arena_used = 0;
    G.M_valid = false;
    G.bound = false;
}
static void grid_free(Grid& G) {
    grid_release(G);
    if (G.arena) (void)hipFree(G.arena);
    G.arena = nullptr; G.arena_bytes = 0;
}

static void state_release(srps_ctx* c) {
    c->i8_state = 0;                       // I8 (its own allocation, made when the images turn out to be bytes) is kept for the next set-up
    c->s = c->rho = c->z = c->Nrm = c->Nrm2 = c->dz = c->dz2 = c->zx = c->zy = c->xx = c->yy = c->z0s = c->I = c->albedo_ex = c->q_ex = nullptr;      // carved out of state_arena
    c->normals_pending = false;
    c->nd_ptr_out = false;                 // every pointer handed out is void with the arrays
    c->q_in_exchange = false; c->energy_ex = nullptr;      // energy_ex lives in the report record
    c->have_state = false;
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// The stored 6-plane tensor: only the operator kernels that stream it need it (the simple kernel, the marching kernel without
// tensor recompute, channel counts other than 1 and 3) -- allocated and zeroed when the first such assembly asks for it.
int grid_need_M(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    if (G.d_M) return SRPS_OK;
    SRPS_TRY(dalloc(&G.d_M, 6 * G.plane));
    SRPS_HIP(hipMemsetAsync(G.d_M, 0, 6 * G.plane * sizeof(float), ctx->stream));      // zero outside the mask
    return SRPS_OK;
}

// Construction of the grid structure: what SRPS.cu:151-203 expresses as index lists and the COO matrices KT, Dx, Dy becomes a
// bounding box, a compact->grid index map and one byte per pixel -- built on the device (kernels_structure.hip) on the context's
// auxiliary stream.  after_release (may be null) runs first, once the old grid has been released and the arguments have been
// checked: srps_setup starts its image uploads there, so that the DMA runs while the structure is built.
// what the arrays carved out of the grid arena take together (build_grid), from the grid's dimensions, tile counts and plan
static size_t grid_arena_bytes(const Grid& G) {
    const size_t n_pw = (size_t)std::max(4096, march_blocks(G) + 8);
    const size_t pl = al256(G.plane * sizeof(float));
    size_t need = 2 * al256((size_t)G.P * sizeof(int)) + al256((size_t)std::max(G.Ps, 1) * sizeof(int)) + al256(G.plane) + al256((size_t)G.Hl * G.Wl * sizeof(int) + 4);
    for (int shape = 0; shape < 3; ++shape) need += al256((size_t)G.n_tiles[shape]) + al256((size_t)G.n_tiles[shape] * sizeof(int));
    need += (3 + 3 + 9) * pl + 256;                       // q [3], g [3], x, x2, r, r2, p [2], w, w2, save
    need += al256(128 * sizeof(float)) + al256(2 * 4 * (size_t)G.n_part4 * sizeof(float)) + al256(n_pw * sizeof(float)) +
            al256(2 * (size_t)G.nb_update * sizeof(float)) + al256(4096 * sizeof(float));
    return need;
}

static int build_grid(srps_ctx* ctx, int h, int w, int sf, const float* mask, const std::function<int()>* after_release = nullptr) {
    Grid& G = ctx->grid;
    grid_release(G);
    SRPS_REQUIRE(h > 0 && w > 0 && sf >= 1, SRPS_ERR_INVALID, "bind_grid: bad dimensions h=%d w=%d sf=%d", h, w, sf);
    SRPS_REQUIRE(h % sf == 0 && w % sf == 0, SRPS_ERR_INVALID, "bind_grid: h=%d, w=%d must be multiples of sf=%d", h, w, sf);
    SRPS_REQUIRE(mask != nullptr, SRPS_ERR_INVALID, "bind_grid: mask is NULL");
    const size_t hw = (size_t)h * w;
    SRPS_REQUIRE(hw < (size_t)1 << 31, SRPS_ERR_UNSUPPORTED, "bind_grid: h*w must fit int32");
    hipStream_t ax = ctx->aux_stream;
    // scratch first (growing it frees the old one, and hipFree waits for the device: not while the images are in flight)
    const size_t mask_bytes = al256(hw * sizeof(float));
    SRPS_TRY(ensure(ctx->ws_struct, mask_bytes + struct_scratch_bytes(h, w, sf) + 256));
    // The grid arena's exact size is known only after the structure kernels have run -- while the images are in flight -- and growing
    // it means hipFree, which waits for the device, i.e. for the whole DMA (round-3 advisor finding).  So an arena that exists and is
    // smaller than what a FULL-frame mask of this h x w would need is replaced now, before the copies are queued; the exact size
    // below can then only be smaller (a plan with more partial sums than the full frame's is the one exception and still grows it).
    // (also for the context's FIRST grid: an arena made to the exact size was replaced by the bound in the second set-up -- a hipFree and
    // a hipMalloc of 300 MB, 5 ms, in exactly the set-up bench.py times)
    {
        Grid ub;
        ub.h = h; ub.w = w; ub.sf = sf; ub.Hg = h; ub.Wg = w; ub.P = (int)hw; ub.Ps = (int)(hw / ((size_t)sf * sf));
        ub.Hs = ((ub.Hg + 2 * PAD + 31) / 32) * 32; ub.Ws = ub.Wg + 512 + 2 * PAD; ub.plane = (size_t)ub.Hs * ub.Ws;
        ub.Hl = h / sf; ub.Wl = w / sf;
        if (ub.plane < ((size_t)1 << 31) - 8 * (size_t)ub.Hs) {
            ub.used = (size_t)ub.Hs * (ub.Wg + 2 * PAD);
            ub.nb_update = std::max(1, std::min(cdiv((long long)ub.used / 4, 256 * 4), 1024));
            march_plan(ub, ctx->march_tj, ctx->num_cus);
            ub.n_part4 = std::max(4096, march_blocks(ub) + 8);
            for (int shape = 0; shape < 3; ++shape) ub.n_tiles[shape] = cdiv(ub.Hg, 256) * cdiv(ub.Wg, shape == 0 ? 32 : shape == 1 ? 64 : 16);
            const size_t need_ub = grid_arena_bytes(ub);
            if (G.arena_bytes < need_ub) {
                if (G.arena) SRPS_HIP(hipFree(G.arena));
                G.arena = nullptr; G.arena_bytes = 0;
                SRPS_HIP(hipMalloc(&G.arena, need_ub));
                G.arena_bytes = need_ub;
            }
        }
    }
    float* d_mask = (float*)ctx->ws_struct.p;
    const StructScratch sc = struct_scratch((char*)ctx->ws_struct.p + mask_bytes, h, w, sf);
    // the images first (a thread of their own, a transfer buffer of their own): the mask crosses beside them instead of ahead of them
    if (after_release) SRPS_TRY((*after_release)());
    SRPS_TRY(host_upload(ctx, d_mask, mask, hw * sizeof(float), ax));
    SRPS_TRY(struct_phase1(ax, d_mask, h, w, sf, sc));
    int* hdr = (int*)(ctx->h_pinned + 128);                // behind everything a pass reads back
    SRPS_HIP(hipMemcpyAsync(hdr, sc.header, 8 * sizeof(int), hipMemcpyDeviceToHost, ax));
    SRPS_HIP(hipStreamSynchronize(ax));
    const int P = hdr[0], Ps = hdr[1], imin = hdr[2], imax = hdr[3], jmin = hdr[4], jmax = hdr[5], bad = hdr[6];
    if (bad != INT_MAX) {
        // the reference indexes with mask != 0 (SRPS.cu:158) but compacts with mask == 1
        // (devicecalls.cuh:19-24): anything but {0,1} silently corrupts it; we refuse.
        const int bi = bad % h, bj = bad / h;
        SRPS_REQUIRE(false, SRPS_ERR_INVALID, "bind_grid: mask must be {0,1}, found %g at (%d,%d)", (double)mask[bad], bi, bj);
    }
    G.h = h; G.w = w; G.sf = sf;
    G.P = P;
    SRPS_REQUIRE(G.P > 0, SRPS_ERR_INVALID, "bind_grid: empty mask");
    G.i_lo = (imin / sf) * sf; G.j_lo = (jmin / sf) * sf;
    const int i_hi = ((imax + sf) / sf) * sf, j_hi = ((jmax + sf) / sf) * sf;
    G.Hg = i_hi - G.i_lo; G.Wg = j_hi - G.j_lo;
    G.Hs = ((G.Hg + 2 * PAD + 31) / 32) * 32;
    G.Ws = G.Wg + 512 + 2 * PAD;                  // strips of the marching kernel (<= 512 columns) stay in bounds
    G.plane = (size_t)G.Hs * G.Ws;
    SRPS_REQUIRE(G.plane < ((size_t)1 << 31) - 8 * (size_t)G.Hs, SRPS_ERR_UNSUPPORTED, "bind_grid: grid plane must fit int32 offsets");
    G.Hl = G.Hg / sf; G.Wl = G.Wg / sf;
    G.Ps = Ps;
    const int nti = cdiv(G.Hg, 64), ntj = cdiv(G.Wg, 4);
    G.nb_apply = std::max(1, std::min(nti * ntj, 1024));
    G.used = (size_t)G.Hs * (G.Wg + 2 * PAD);      // the CG vectors are zero (and stay zero) beyond the used columns
    G.nb_update = std::max(1, std::min(cdiv((long long)G.used / 4, 256 * 4), 1024));
    march_plan(G, ctx->march_tj, ctx->num_cus);
    G.n_part4 = std::max(4096, march_blocks(G) + 8);      // any strip width the options allow stays below this (see march_strip)
    const size_t n_pw = (size_t)std::max(4096, march_blocks(G) + 8);
    for (int shape = 0; shape < 3; ++shape) G.n_tiles[shape] = cdiv(G.Hg, 256) * cdiv(G.Wg, shape == 0 ? 32 : shape == 1 ? 64 : 16);
    // ---- one arena for every array of the grid ----
    const size_t need = grid_arena_bytes(G);
    if (G.arena_bytes < need) {
        if (G.arena) SRPS_HIP(hipFree(G.arena));
        G.arena = nullptr; G.arena_bytes = 0;
        SRPS_HIP(hipMalloc(&G.arena, need));
        G.arena_bytes = need;
    }
    G.arena_used = 0;
    auto carve = [&](size_t bytes) -> void* { void* p = (char*)G.arena + G.arena_used; G.arena_used += al256(bytes); return p; };
    G.d_gofp = (int*)carve((size_t)G.P * sizeof(int)); G.d_imask = (int*)carve((size_t)G.P * sizeof(int));
    G.d_imasks = (int*)carve((size_t)std::max(G.Ps, 1) * sizeof(int));
    G.d_flags = (uint8_t*)carve(G.plane);
    G.d_lr_index = (int*)carve((size_t)G.Hl * G.Wl * sizeof(int) + 4);
    for (int shape = 0; shape < 3; ++shape) G.d_tile_cls[shape] = (uint8_t*)carve((size_t)G.n_tiles[shape]);
    for (int shape = 0; shape < 3; ++shape) G.d_tile_list[shape] = (int*)carve((size_t)G.n_tiles[shape] * sizeof(int));
    const size_t pb = G.plane * sizeof(float);             // multi-plane arrays are [k][plane], contiguous
    G.d_q = (float*)carve(3 * pb);
    G.d_G = (float*)carve(3 * pb); G.G_planes = 3;
    G.d_x = (float*)carve(pb); G.d_x2 = (float*)carve(pb); G.d_r = (float*)carve(pb); G.d_r2 = (float*)carve(pb);
    G.d_p = (float*)carve(2 * pb); G.d_w = (float*)carve(pb); G.d_w2 = (float*)carve(pb); G.d_save = (float*)carve(pb);
    G.d_tconsts = (float*)carve(128 * sizeof(float));      // [8][8] tensor constants + [8][4] right-hand-side constants
    G.d_part4 = (float*)carve(2 * 4 * (size_t)G.n_part4 * sizeof(float));
    G.d_pw_part = (float*)carve(n_pw * sizeof(float)); G.d_rr_part = (float*)carve(2 * (size_t)G.nb_update * sizeof(float));
    G.d_misc_part = (float*)carve(4096 * sizeof(float));
    SRPS_REQUIRE(G.arena_used <= G.arena_bytes, SRPS_ERR_NOMEM, "bind_grid: arena accounting");
    G.d_scal = (CgScalars*)(ctx->d_report + 64);
    // every plane is zero outside the mask (and the vectors start at zero): one memset for the lot
    SRPS_HIP(hipMemsetAsync(G.arena, 0, G.arena_used, ax));
    SRPS_HIP(hipMemsetAsync(G.d_scal, 0, sizeof(CgScalars), ax));
    SRPS_TRY(struct_phase2(ax, G, sc, G.d_imasks));
    int* hrect = hdr + 8;
    SRPS_HIP(hipMemcpyAsync(hrect, sc.n_rect, 3 * sizeof(int), hipMemcpyDeviceToHost, ax));
    // whatever the caller enqueues next on the context's stream sees the finished structure; the tile counts are read by the host
    SRPS_HIP(hipEventRecord(ctx->aux_event, ax));
    SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
    // the occupied tiles of each tiling, in ascending order: what the resident CG launches its blocks for (a few hundred bytes
    // back, the lists up)
    std::vector<uint8_t> h_cls[3];
    std::vector<int> h_list[3];
    for (int shape = 0; shape < 3; ++shape) {
        h_cls[shape].resize((size_t)G.n_tiles[shape]);
        // (library-side vectors are host arrays like a caller's: through the pinned path, which returns when they hold the data)
        SRPS_TRY(host_download(ctx, h_cls[shape].data(), G.d_tile_cls[shape], (size_t)G.n_tiles[shape], ax));
    }
    for (int shape = 0; shape < 3; ++shape) {
        G.n_rect_tiles[shape] = hrect[shape];
        for (int t = 0; t < G.n_tiles[shape]; ++t)
            if (h_cls[shape][t] & TILE_OCCUPIED) h_list[shape].push_back(t);
        G.n_occ[shape] = (int)h_list[shape].size();
        G.h_tile_list[shape] = h_list[shape];                 // the resident strips cut these lists into ranges of tile columns
        if (G.n_occ[shape]) SRPS_TRY(host_upload(ctx, G.d_tile_list[shape], h_list[shape].data(), (size_t)G.n_occ[shape] * sizeof(int), ax));
    }
    SRPS_HIP(hipEventRecord(ctx->aux_event, ax));
    SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
    SRPS_HIP(hipStreamSynchronize(ax));
    ctx->x_swapped = false;
    G.bound = true;
    ctx->tensor_valid = false;
    return SRPS_OK;
}

static int depth_solve_impl(srps_ctx* ctx, const float* d_z0s, float* d_z, float* d_zx, float* d_zy, bool plane_current = false) {
    Grid& G = ctx->grid;
    if (!plane_current) SRPS_TRY(grid_scatter(ctx, d_z, G.d_x));      // the last solve left z on the grid plane
    SRPS_TRY(grid_rhs(ctx, d_z0s));                               // dc.cu:743-745
    SRPS_TRY(grid_cg(ctx, ctx->cg_max_iter + 1, false));          // dc.cu:758-759 (residual; k <= max_iter => 101 steps)
    SRPS_TRY(grid_gradient(ctx, G.d_x, d_zx, d_zy, d_z));         // the new z in the compact layout, and Dx z, Dy z (energy + normals)
    ctx->report_pending = true;          // the CG scalars are fetched with the rest of the report record
    return SRPS_OK;
}

// one copy for all the scalars of a pass (energy terms, iteration counts); the caller synchronises the stream
static int report_fetch(srps_ctx* ctx) {
    SRPS_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_report, 80 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    ctx->report_pending = false;
    return SRPS_OK;
}

// After the report record has been fetched and the stream waited for: did a persistent launch give up a wait
// (device_utils.h SpinState)?  Returns the ABORT_* bits, switches the kernels concerned off for this context (the phases
// fall back to the streaming kernels), clears the device flags and leaves the reason in srps_last_error().
// `others`: ABORT_* bits that OTHER ranks of a sharded job reported (srps_execute_sharded: the ranks switch kernels together).
static int persistent_aborts(srps_ctx* ctx, int* flags_out, int others = 0) {
    *flags_out = 0;
    const CgScalars* hs = (const CgScalars*)(ctx->h_pinned + 64);
    const int flags = hs->abort_flags | others | ctx->debug_inject_abort;
    ctx->debug_inject_abort = 0;
    ctx->persistent_inflight = 0;
    const bool swapped = ctx->x_swapped;
    ctx->x_swapped = false;
    if (!flags) return SRPS_OK;
    // The resident depth CG stored into the other plane: whatever it left there (nothing, the result of some blocks only when a
    // wait gave up during the last step, or a solve on the albedo of an aborted albedo launch), the plane it started from is
    // intact -- make it the current one again; the caller repeats the phase from it.
    if (swapped) std::swap(ctx->grid.d_x, ctx->grid.d_x2);
    ctx->plane_restored = swapped;       // else no depth solve ran since the last look: the plane holds whatever it held before
    ++ctx->persistent_fallbacks;
    if (flags & ABORT_DEPTH) ctx->cg_resident = 0;
    if (flags & ABORT_ALBEDO) ctx->albedo_persistent = 0;
    if (hs->abort_flags)
        set_error("persistent %s%s%s kernel gave up a grid-wide wait after %d ms (wait %d: %d blocks had arrived) -- the device is shared or "
                  "admits fewer resident blocks than the occupancy query reports; this context now uses the streaming kernels",
                  (flags & ABORT_DEPTH) ? "depth-CG" : "", (flags & ABORT_DEPTH) && (flags & ABORT_ALBEDO) ? " and " : "",
                  (flags & ABORT_ALBEDO) ? "albedo-CG" : "", ctx->spin_budget_ms, hs->abort_gen, hs->abort_arrived);
    else
        set_error("a persistent %s%s%s kernel gave up a grid-wide wait on another rank of the job; all ranks now use the streaming kernels",
                  (flags & ABORT_DEPTH) ? "depth-CG" : "", (flags & ABORT_DEPTH) && (flags & ABORT_ALBEDO) ? " and " : "", (flags & ABORT_ALBEDO) ? "albedo-CG" : "");
    SRPS_HIP(hipMemsetAsync(&((CgScalars*)(ctx->d_report + 64))->abort_flags, 0, 3 * sizeof(int), ctx->stream));
    ((CgScalars*)(ctx->h_pinned + 64))->abort_flags = 0;
    *flags_out = flags;
    return SRPS_OK;
}
// the same with its own fetch + wait, for the calls that cannot defer the check to the end of the pass
static int persistent_sync_check(srps_ctx* ctx, int* flags_out) {
    *flags_out = 0;
    if (!ctx->persistent_inflight) return SRPS_OK;
    SRPS_TRY(report_fetch(ctx));
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    albedo_iters_collect(ctx);
    return persistent_aborts(ctx, flags_out);
}

}  // namespace srps

using namespace srps;

#include <atomic>
static std::atomic<int> g_live_contexts[64];
namespace srps {
int contexts_on_device(int device) { return (device >= 0 && device < 64) ? g_live_contexts[device].load() : 2; }
}

#define CTX_CHECK(ctx)                                                        \
    do {                                                                      \
        SRPS_REQUIRE((ctx) != nullptr, SRPS_ERR_INVALID, "null context");    \
        SRPS_HIP(hipSetDevice((ctx)->device));                                \
    } while (0)
#define GRID_CHECK(ctx) SRPS_REQUIRE((ctx)->grid.bound, SRPS_ERR_STATE, "%s: no grid bound (call srps_bind_grid or srps_setup first)", __func__)
#define STATE_CHECK(ctx) SRPS_REQUIRE((ctx)->have_state, SRPS_ERR_STATE, "%s: srps_setup has not been called", __func__)

extern "C" {

const char* srps_last_error(void) { return g_err.c_str(); }
const char* srps_version(void) { return "srps-hip 0.1 (gfx950)"; }

int srps_device_count(int* n) {
    SRPS_REQUIRE(n != nullptr, SRPS_ERR_INVALID, "device_count: n is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
    *n = ndev;
    return SRPS_OK;
}

int srps_transfer_buffers(int* n) {
    SRPS_REQUIRE(n != nullptr, SRPS_ERR_INVALID, "transfer_buffers: n is NULL");
    *n = xfer_buffers_made();
    return SRPS_OK;
}

int srps_create(int device_id, int block_x, int block_y, srps_ctx** out) {
    SRPS_REQUIRE(out != nullptr, SRPS_ERR_INVALID, "srps_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    SRPS_HIP(hipGetDeviceCount(&ndev));
    SRPS_REQUIRE(ndev > 0, SRPS_ERR_HIP, "srps_create: no HIP device visible");
    SRPS_REQUIRE(device_id >= 0 && device_id < ndev, SRPS_ERR_INVALID, "srps_create: device %d out of range (0..%d)", device_id, ndev - 1);
    SRPS_HIP(hipSetDevice(device_id));
    srps_ctx* c = new srps_ctx();
    c->device = device_id;
    if (block_x > 0) c->block_x = block_x;
    if (block_y > 0) c->block_y = block_y;
    {
        int cus = 0, coop = 0;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id);
        (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device_id);
        c->num_cus = cus > 0 ? cus : 1;
        if (!coop) { c->albedo_persistent = 0; c->cg_resident = 0; }
    }
    // every handle made so far is released by ONE helper on each failure path (round-3 advisor finding: aux / gather streams and
    // the event used to leak)
    auto fail = [c](hipError_t err, const char* what, int line) {
        if (c->d_report) (void)hipFree(c->d_report);
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        if (c->aux_event) (void)hipEventDestroy(c->aux_event);
        if (c->gather_stream) (void)hipStreamDestroy(c->gather_stream);
        if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        delete c;
        return hip_fail(err, what, __FILE__, line);
    };
    hipError_t e = hipStreamCreate(&c->own_stream);
    if (e != hipSuccess) return fail(e, "hipStreamCreate", __LINE__);
    c->stream = c->own_stream;
    e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->gather_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return fail(e, "hipStreamCreateWithFlags", __LINE__);
    e = hipEventCreateWithFlags(&c->aux_event, hipEventDisableTiming);
    if (e != hipSuccess) return fail(e, "hipEventCreateWithFlags", __LINE__);
    e = hipHostMalloc((void**)&c->h_pinned, 256 * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent);      // a kernel writes the pass's report into it (ReportFinish)
    if (e != hipSuccess) return fail(e, "hipHostMalloc", __LINE__);
    memset(c->h_pinned, 0, 256 * sizeof(float));
    if (hipHostGetDevicePointer((void**)&c->h_pinned_dev, c->h_pinned, 0) != hipSuccess) { (void)hipGetLastError(); c->h_pinned_dev = nullptr; }
    e = hipMalloc((void**)&c->d_report, 256 * sizeof(float));
    if (e == hipSuccess) e = hipMemset(c->d_report, 0, 256 * sizeof(float));
    if (e != hipSuccess) return fail(e, "hipMalloc", __LINE__);
    if (device_id < 64) g_live_contexts[device_id].fetch_add(1);
    if (const char* e = getenv("SRPS_ROCTX")) c->roctx = atoi(e) != 0;
    *out = c;
    return SRPS_OK;
}

int srps_destroy(srps_ctx* ctx) {
    if (!ctx) return SRPS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    if (ctx->gather_stream) (void)hipStreamSynchronize(ctx->gather_stream);
    if (ctx->device >= 0 && ctx->device < 64) g_live_contexts[ctx->device].fetch_sub(1);
    comm_release(ctx);
    state_release(ctx);
    grid_free(ctx->grid);
    dfree(ctx->I8);
    resident_rank_release(ctx);
    dfree(ctx->d_strip_tot);
    if (ctx->state_arena.p) (void)hipFree(ctx->state_arena.p);
    if (ctx->ws_struct.p) (void)hipFree(ctx->ws_struct.p);
    for (hipEvent_t e : ctx->ev_copied) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_gathered) (void)hipEventDestroy(e);
    if (ctx->aux_event) (void)hipEventDestroy(ctx->aux_event);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->gather_stream) (void)hipStreamDestroy(ctx->gather_stream);
    if (ctx->ws_light.p) (void)hipFree(ctx->ws_light.p);
    if (ctx->ws_resident.p) (void)hipFree(ctx->ws_resident.p);
    if (ctx->ws_ssum.p) (void)hipFree(ctx->ws_ssum.p);
    if (ctx->ws_albedo.p) (void)hipFree(ctx->ws_albedo.p);
    if (ctx->ws_stage.p) (void)hipFree(ctx->ws_stage.p);
    if (ctx->ws_images.p) (void)hipFree(ctx->ws_images.p);
    if (ctx->ws_misc.p) (void)hipFree(ctx->ws_misc.p);
    if (ctx->ev_created)
        for (int i = 0; i < SRPS_N_PHASES; ++i) { (void)hipEventDestroy(ctx->ev_begin[i]); (void)hipEventDestroy(ctx->ev_end[i]); }
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->d_report) (void)hipFree(ctx->d_report);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SRPS_OK;
}

int srps_set_stream(srps_ctx* ctx, void* hip_stream) {
    CTX_CHECK(ctx);
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SRPS_OK;
}

int srps_synchronize(srps_ctx* ctx) {
    CTX_CHECK(ctx);
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    return SRPS_OK;
}

int srps_set_option(srps_ctx* ctx, const char* name, int value) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(name != nullptr, SRPS_ERR_INVALID, "set_option: name is NULL");
    if (!strcmp(name, "albedo_mode")) {
        SRPS_REQUIRE(value == SRPS_ALBEDO_CG || value == SRPS_ALBEDO_CLOSED_FORM || value == SRPS_ALBEDO_FUSED || value == SRPS_ALBEDO_AUTO, SRPS_ERR_INVALID, "albedo_mode: bad value %d", value);
        ctx->albedo_mode = value;
    } else if (!strcmp(name, "apply_mode")) {
        SRPS_REQUIRE(value >= SRPS_APPLY_AUTO && value <= SRPS_APPLY_MARCH, SRPS_ERR_INVALID, "apply_mode: bad value %d", value);
        ctx->apply_mode = value;
        ctx->depth_assembled = false;
    } else if (!strcmp(name, "tensor_recompute")) {
        ctx->tensor_recompute = value ? 1 : 0;
        ctx->depth_assembled = false;        // a depth system the albedo sweep left (SRPS_ALBEDO_FUSED / AUTO) is in the other form
    } else if (!strcmp(name, "march_x2")) {
        SRPS_REQUIRE(value >= 0 && value <= 2, SRPS_ERR_INVALID, "march_x2: 0 (never), 1 (always) or 2 (when the step's planes exceed the Infinity Cache), got %d", value);
        ctx->march_x2 = value;
    } else if (!strcmp(name, "march_snake")) {
        ctx->march_snake = value == 2 ? 2 : (value ? 1 : 0);
    } else if (!strcmp(name, "fuse_energy_lighting")) {
        ctx->fuse_energy_lighting = value ? 1 : 0;
        ctx->light_cache_valid = false;
    } else if (!strcmp(name, "coop_launch")) {
        SRPS_REQUIRE(value >= 0 && value <= 2, SRPS_ERR_INVALID, "coop_launch: 0 (plain), 1 (cooperative) or 2 (cooperative when the device is shared)");
        ctx->coop_launch = value;
    } else if (!strcmp(name, "exclusive_device")) {
        // the caller states that no other process or context uses the device: plain launches of the persistent kernels
        ctx->coop_launch = value ? 0 : 1;
    } else if (!strcmp(name, "host_wait_spin")) {
        ctx->host_wait_spin = value ? 1 : 0;
    } else if (!strcmp(name, "report_zero_copy")) {
        ctx->report_zero_copy = value ? 1 : 0;
    } else if (!strcmp(name, "spin_budget_ms")) {
        SRPS_REQUIRE(value >= 1 && value <= 600000, SRPS_ERR_INVALID, "spin_budget_ms: 1 .. 600000");
        ctx->spin_budget_ms = value;
    } else if (!strcmp(name, "cg_one_sync")) {
        ctx->cg_one_sync = value ? 1 : 0;
    } else if (!strcmp(name, "cg_fused_step")) {
        ctx->cg_fused_step = value ? 1 : 0;
    } else if (!strcmp(name, "pin_uploads")) {
        ctx->pin_uploads = value ? 1 : 0;
    } else if (!strcmp(name, "image_store")) {
        ctx->image_store = value ? 1 : 0;
        if (ctx->i8_state == 2 && value) ctx->i8_state = 0;      // look (again) at the next sweep
        ctx->light_cache_valid = false;
    } else if (!strcmp(name, "phase_timing")) {
        ctx->phase_timing = value ? 1 : 0;
        ctx->ev_mask = 0;
    } else if (!strcmp(name, "roctx")) {
        ctx->roctx = value ? 1 : 0;
    } else if (!strcmp(name, "cg_resident")) {
        ctx->cg_resident = value ? 1 : 0;
    } else if (!strcmp(name, "cg_resident_rect")) {
        ctx->cg_resident_rect = value ? 1 : 0;
    } else if (!strcmp(name, "cg_resident_debug")) {
        ctx->cg_resident_debug = value;
    } else if (!strcmp(name, "shard_range_check")) {
        ctx->shard_range_check = value ? 1 : 0;
    } else if (!strcmp(name, "debug_foreign_pid_twin")) {
        ctx->debug_foreign_pid_twin = value ? 1 : 0;       // tests: see resident_rank_open (same pid, another process number: never the pointer route)
    } else if (!strcmp(name, "debug_ipc_same_process")) {
        ctx->debug_ipc_same_process = value ? 1 : 0;       // tests: see resident_rank_open
    } else if (!strcmp(name, "debug_inject_abort")) {
        // test hook: the next look at the abort flags finds these bits (1 depth, 2 albedo) as if ANOTHER rank had reported them
        SRPS_REQUIRE(value >= 0 && value <= 3, SRPS_ERR_INVALID, "debug_inject_abort: 0..3, got %d", value);
        ctx->debug_inject_abort = value;
    } else if (!strcmp(name, "light_bytes")) {
        ctx->light_bytes = value ? 1 : 0;
        ctx->light_cache_valid = false;
    } else if (!strcmp(name, "light_run")) {
        SRPS_REQUIRE(value == 1 || value == 3, SRPS_ERR_INVALID, "light_run: 1 (vector form) or 3 (matrix pipe, the default), got %d", value);
        ctx->light_run = value;
    } else if (!strcmp(name, "light_tiled")) {
        ctx->light_tiled = value ? 1 : 0;
        ctx->light_cache_valid = false;
    } else if (!strcmp(name, "light_grouped")) {
        ctx->light_grouped = value ? 1 : 0;
        ctx->light_cache_valid = false;
    } else if (!strcmp(name, "fuse_normals")) {
        ctx->fuse_normals = value ? 1 : 0;
        ctx->normals_pending = false;
    } else if (!strcmp(name, "assemble_from_sums")) {
        ctx->assemble_from_sums = value ? 1 : 0;
        ctx->ssum_valid = false;
    } else if (!strcmp(name, "cg_resident_tile")) {
        SRPS_REQUIRE(value == 0 || value == 2 || value == 16 || value == 32 || value == 256 || value == 512, SRPS_ERR_INVALID, "cg_resident_tile: 0, 2, 16, 32, 256 or 512");
        ctx->cg_resident_tile = value;
    } else if (!strcmp(name, "albedo_channels_together")) {
        ctx->albedo_channels_together = value ? 1 : 0;
    } else if (!strcmp(name, "albedo_one_sync")) {
        ctx->albedo_one_sync = value ? 1 : 0;
    } else if (!strcmp(name, "albedo_persistent")) {
        ctx->albedo_persistent = value ? 1 : 0;
    } else if (!strcmp(name, "keep_stored_tensor")) {
        ctx->keep_stored_tensor = value ? 1 : 0;
        ctx->depth_assembled = false;
    } else if (!strcmp(name, "march_strip")) {
        SRPS_REQUIRE(value == 0 || (value >= 4 && value <= 512 && value % 4 == 0), SRPS_ERR_INVALID, "march_strip: 0 (automatic) or a multiple of 4 in [4, 512]");
        if (ctx->grid.bound) {
            // the partial-sum buffers were sized at bind time (build_grid): refuse a strip width whose block count exceeds them
            srps::Grid trial;                     // only the extent enters the plan (no copy of the bound grid's index vectors)
            trial.Hg = ctx->grid.Hg; trial.Wg = ctx->grid.Wg;
            march_plan(trial, value, ctx->num_cus);
            SRPS_REQUIRE(march_blocks(trial) + 8 <= std::max(4096, ctx->grid.n_part4), SRPS_ERR_INVALID,
                         "march_strip: %d-column strips need %d blocks, more than the %d partial sums allocated for the bound grid (set the option before srps_bind_grid / srps_setup)",
                         value, march_blocks(trial), ctx->grid.n_part4);
            march_plan(ctx->grid, value, ctx->num_cus);
        }
        ctx->march_tj = value;
    } else if (!strcmp(name, "overlap_exchange")) {
        ctx->overlap_exchange = value ? 1 : 0;
    } else if (!strcmp(name, "cg_partition")) {
        SRPS_REQUIRE(value >= 0 && value <= 2, SRPS_ERR_INVALID, "cg_partition: 0 (every rank runs the whole depth CG), 1 (column strips over the communicator's ranks, streaming step) or 2 (the resident kernel on strips of tile columns)");
        if (value != 2) ctx->xg_failed = 0;
        ctx->cg_strips = value;
    } else if (!strcmp(name, "cg_max_iter")) {
        SRPS_REQUIRE(value >= 0, SRPS_ERR_INVALID, "cg_max_iter: bad value %d", value);
        ctx->cg_max_iter = value;
    } else {
        SRPS_REQUIRE(false, SRPS_ERR_INVALID, "set_option: unknown option '%s'", name);
    }
    return SRPS_OK;
}

int srps_get_option(srps_ctx* ctx, const char* name, int* value) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(name != nullptr && value != nullptr, SRPS_ERR_INVALID, "get_option: null argument");
    if (!strcmp(name, "albedo_mode")) *value = ctx->albedo_mode;
    else if (!strcmp(name, "apply_mode")) *value = ctx->apply_mode;
    else if (!strcmp(name, "tensor_recompute")) *value = ctx->tensor_recompute;
    else if (!strcmp(name, "march_snake")) *value = ctx->march_snake;
    else if (!strcmp(name, "march_x2")) *value = ctx->march_x2;
    else if (!strcmp(name, "march_x2_active")) *value = (ctx->grid.bound && march_x2_on(ctx)) ? 1 : 0;
    else if (!strcmp(name, "march_strip")) *value = ctx->grid.bound ? ctx->grid.strip_cols : ctx->march_tj;
    else if (!strcmp(name, "keep_stored_tensor")) *value = ctx->keep_stored_tensor;
    else if (!strcmp(name, "fuse_energy_lighting")) *value = ctx->fuse_energy_lighting;
    else if (!strcmp(name, "fuse_normals")) *value = ctx->fuse_normals;
    else if (!strcmp(name, "light_tiled")) *value = ctx->light_tiled;
    else if (!strcmp(name, "light_run")) *value = ctx->light_run;
    else if (!strcmp(name, "exchange_buffer_fine")) *value = (ctx->xg_buf && ctx->xg_fine) ? 1 : 0;      // the resident strips' exchange buffer is fine-grained memory
    else if (!strcmp(name, "light_bytes")) *value = ctx->light_bytes;
    else if (!strcmp(name, "albedo_persistent")) *value = ctx->albedo_persistent;
    else if (!strcmp(name, "cg_resident")) *value = ctx->cg_resident;
    else if (!strcmp(name, "cg_fused_step")) *value = ctx->cg_fused_step;
    else if (!strcmp(name, "phase_timing")) *value = ctx->phase_timing;
    else if (!strcmp(name, "pin_uploads")) *value = ctx->pin_uploads;
    else if (!strcmp(name, "image_store")) *value = ctx->image_store;
    else if (!strcmp(name, "image_store_bytes_active")) *value = (ctx->image_store && ctx->have_state && ctx->i8_state == 1) ? 1 : 0;
    else if (!strcmp(name, "roctx")) *value = ctx->roctx;
    else if (!strcmp(name, "cg_one_sync")) *value = ctx->cg_one_sync;
    else if (!strcmp(name, "cg_resident_rect")) *value = ctx->cg_resident_rect;
    else if (!strcmp(name, "cg_resident_rect_active")) *value = (ctx->grid.bound && resident_rect_active(ctx)) ? 1 : 0;
    else if (!strcmp(name, "cg_resident_rect_tiles_256")) *value = ctx->grid.bound ? ctx->grid.n_rect_tiles[0] : 0;      // of the 256 x 32 tiling
    else if (!strcmp(name, "cg_resident_rect_tiles_512")) *value = ctx->grid.bound ? ctx->grid.n_rect_tiles[1] : 0;      // of the 256 x 64 tiling
    else if (!strcmp(name, "cg_resident_tiles_occupied_256")) *value = ctx->grid.bound ? ctx->grid.n_occ[0] : 0;         // of the 256 x 32 tiling: the blocks it would launch
    else if (!strcmp(name, "cg_resident_tiles_occupied_512")) *value = ctx->grid.bound ? ctx->grid.n_occ[1] : 0;
    else if (!strcmp(name, "cg_resident_tiles_occupied_16")) *value = ctx->grid.bound ? ctx->grid.n_occ[2] : 0;
    else if (!strcmp(name, "cg_resident_tiles_256")) *value = ctx->grid.bound ? ctx->grid.n_tiles[0] : 0;               // all tiles of the bounding box
    else if (!strcmp(name, "cg_resident_tiles_512")) *value = ctx->grid.bound ? ctx->grid.n_tiles[1] : 0;
    else if (!strcmp(name, "cg_resident_tiles_16")) *value = ctx->grid.bound ? ctx->grid.n_tiles[2] : 0;
    else if (!strcmp(name, "cg_resident_rect_tiles_16")) *value = ctx->grid.bound ? ctx->grid.n_rect_tiles[2] : 0;       // of the 256 x 16 tiling
    else if (!strcmp(name, "cg_max_iter")) *value = ctx->cg_max_iter;
    else if (!strcmp(name, "cg_partition")) *value = ctx->cg_strips;
    else if (!strcmp(name, "overlap_exchange")) *value = ctx->overlap_exchange;
    else if (!strcmp(name, "cg_partition_active")) *value = strips_active(ctx) ? 1 : 0;
    else if (!strcmp(name, "cg_partition_resident_active")) *value = (ctx->cg_strips == 2 && ctx->xg_world > 1 && !ctx->xg_failed && ctx->cg_resident) ? 1 : 0;      // the last solve ran as resident strips
    else if (!strcmp(name, "cg_resident_tile")) *value = ctx->cg_resident_tile;
    else if (!strcmp(name, "albedo_channels_together")) *value = ctx->albedo_channels_together;
    else if (!strcmp(name, "albedo_one_sync")) *value = ctx->albedo_one_sync;
    else if (!strcmp(name, "num_cus")) *value = ctx->num_cus;
    else if (!strcmp(name, "coop_launch")) *value = ctx->coop_launch;
    else if (!strcmp(name, "exclusive_device")) *value = ctx->coop_launch == 0 ? 1 : 0;
    else if (!strcmp(name, "host_wait_spin")) *value = ctx->host_wait_spin;
    else if (!strcmp(name, "report_zero_copy")) *value = ctx->report_zero_copy;
    else if (!strcmp(name, "spin_budget_ms")) *value = ctx->spin_budget_ms;
    else if (!strcmp(name, "persistent_fallbacks")) *value = ctx->persistent_fallbacks;
    else if (!strcmp(name, "cg_resident_active")) *value = (ctx->grid.bound && resident_supported(ctx)) ? 1 : 0;
    else SRPS_REQUIRE(false, SRPS_ERR_INVALID, "get_option: unknown option '%s'", name);
    return SRPS_OK;
}

// ---- init kernels ---------------------------------------------------------------------------
int srps_mean_across_channels(srps_ctx* ctx, const float* h_data, int h, int w, int nc, float* d_mean, uint8_t* d_inpaint) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(h_data && d_mean && d_inpaint && h > 0 && w > 0 && nc > 0, SRPS_ERR_INVALID, "mean_across_channels: bad arguments");
    const size_t n = (size_t)h * w * nc;
    SRPS_TRY(ensure(ctx->ws_stage, n * sizeof(float)));
    SRPS_TRY(host_upload(ctx, ctx->ws_stage.p, h_data, n * sizeof(float), ctx->stream));
    return launch_mean_channels(ctx->stream, (const float*)ctx->ws_stage.p, h, w, nc, d_mean, d_inpaint);
}
int srps_rho_init(srps_ctx* ctx, float* d_rho, int npix, int nc) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(d_rho && npix > 0 && nc > 0, SRPS_ERR_INVALID, "rho_init: bad arguments");
    return launch_fill(ctx->stream, d_rho, (size_t)npix * nc, 0.5f);               // dc.cu:137
}
int srps_meshgrid_create(srps_ctx* ctx, int w, int h, float K02, float K12, float* d_xx, float* d_yy) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(d_xx && d_yy && w > 0 && h > 0, SRPS_ERR_INVALID, "meshgrid_create: bad arguments");
    return launch_meshgrid_full(ctx->stream, w, h, K02, K12, d_xx, d_yy);
}

// ---- phase operators on caller-owned arrays ---------------------------------------------------
int srps_normal_init(srps_ctx* ctx, const float* d_z, const float* d_zx, const float* d_zy, const float* d_xx,
                     const float* d_yy, int npix, float K00, float K11, float* d_N, float* d_dz) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(d_z && d_zx && d_zy && d_xx && d_yy && d_N && d_dz && npix > 0, SRPS_ERR_INVALID, "normal_init: bad arguments");
    return launch_normals(ctx->stream, d_z, d_zx, d_zy, d_xx, d_yy, npix, K00, K11, d_N, d_dz);
}

int srps_lightning_estimation(srps_ctx* ctx, float* d_s, const float* d_rho, const float* d_N, const float* d_I,
                              int npix, int nimages, int nchannels) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(d_s && d_rho && d_N && d_I && npix > 0 && nimages > 0 && nchannels > 0, SRPS_ERR_INVALID, "lightning_estimation: bad arguments");
    SRPS_TRY(lighting(ctx, d_s, d_rho, d_N, d_I, npix, nimages, nchannels, nimages, 0, false));
    return SRPS_OK;
}

int srps_albedo_estimation(srps_ctx* ctx, const float* d_s, float* d_rho, const float* d_N, const float* d_I,
                           int npix, int nimages, int nchannels) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(d_s && d_rho && d_N && d_I && npix > 0 && nimages > 0 && nchannels > 0, SRPS_ERR_INVALID, "albedo_estimation: bad arguments");
    SRPS_TRY(ensure(ctx->ws_misc, 2 * (size_t)nchannels * npix * sizeof(float) + 64));
    float* numden = (float*)ctx->ws_misc.p;
    SRPS_TRY(albedo_numden(ctx, d_s, d_N, d_I, npix, nimages, nchannels, 0, numden));
    SRPS_TRY(albedo_finish(ctx, d_rho, numden, npix, nchannels));
    int aborted = 0;
    SRPS_TRY(persistent_sync_check(ctx, &aborted));           // waits only when the persistent kernel was launched
    if (aborted & ABORT_ALBEDO) SRPS_TRY(albedo_finish(ctx, d_rho, numden, npix, nchannels));      // streaming form now
    return SRPS_OK;
}

int srps_bind_grid(srps_ctx* ctx, int h, int w, int sf, const float* mask) {
    CTX_CHECK(ctx);
    ctx->op_pp_set = false;
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    return build_grid(ctx, h, w, sf, mask);
}

int srps_set_principal_point(srps_ctx* ctx, float K02, float K12) {
    CTX_CHECK(ctx);
    ctx->op_cx = K02; ctx->op_cy = K12; ctx->op_pp_set = true;
    return SRPS_OK;
}

int srps_gradient(srps_ctx* ctx, const float* d_z, int npix, float* d_zx, float* d_zy) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    SRPS_REQUIRE(d_z && d_zx && d_zy && npix == ctx->grid.P, SRPS_ERR_INVALID, "gradient: npix=%d does not match the bound mask (%d)", npix, ctx->grid.P);
    SRPS_TRY(grid_scatter(ctx, d_z, ctx->grid.d_x));
    return grid_gradient(ctx, ctx->grid.d_x, d_zx, d_zy);
}

int srps_depth_estimation(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_N, const float* d_I,
                          const float* d_xx, const float* d_yy, const float* d_dz, const float* d_z0s, float* d_z,
                          float K00, float K11, int npix, int nimages, int nchannels, float* energy) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    (void)d_N;   // the linearised system uses dz, not N (N3 == 1 enters B through s3, dc.cu:573)
    SRPS_REQUIRE(d_s && d_rho && d_I && d_xx && d_yy && d_dz && d_z && energy, SRPS_ERR_INVALID, "depth_estimation: null argument");
    SRPS_REQUIRE(d_z0s || ctx->grid.Ps == 0, SRPS_ERR_INVALID, "depth_estimation: d_z0s is NULL but the mask has %d complete blocks", ctx->grid.Ps);
    SRPS_REQUIRE(npix == ctx->grid.P, SRPS_ERR_INVALID, "depth_estimation: npix=%d does not match the bound mask (%d)", npix, ctx->grid.P);
    Grid& G = ctx->grid;
    SRPS_TRY(ensure(ctx->ws_misc, (2 * (size_t)npix + 16) * sizeof(float)));
    float* zx = (float*)ctx->ws_misc.p;
    float* zy = zx + npix;
    float* e2 = zy + npix;
    // the reference's signature carries xx, yy as arrays and no principal point: stream the stored tensor
    // unless the caller announced it (srps_set_principal_point)
    SRPS_TRY(depth_assemble(ctx, d_s, d_rho, d_I, d_xx, d_yy, d_dz, K00, K11, npix, nimages, nchannels, nimages, 0,
                            ctx->op_pp_set ? ctx->op_cx : NAN, ctx->op_pp_set ? ctx->op_cy : NAN));
    ctx->plane_holds_z = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        SRPS_TRY(depth_solve_impl(ctx, d_z0s, d_z, zx, zy, /*plane_current=*/attempt > 0));      // second attempt: the plane the first started from (d_z holds what the aborted launch left)
        SRPS_TRY(grid_energy_t1(ctx, d_z0s, e2));
        SRPS_TRY(energy_photometric_partial(ctx, d_s, d_rho, d_I, d_xx, d_yy, d_dz, d_z, zx, zy, K00, K11, npix, nimages, nchannels, 0, e2 + 1));
        SRPS_TRY(report_fetch(ctx));          // first: the two energy terms of this call overwrite the record's
        SRPS_HIP(hipMemcpyAsync(ctx->h_pinned, e2, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        SRPS_HIP(hipStreamSynchronize(ctx->stream));
        albedo_iters_collect(ctx);
        int aborted = 0;
        SRPS_TRY(persistent_aborts(ctx, &aborted));
        if (!(aborted & ABORT_DEPTH)) break;          // else: solve again, by streaming, from the iterate the aborted launch started from
    }
    *energy = ctx->h_pinned[0] + ctx->lambda * ctx->h_pinned[1];               // dc.cu:785
    ctx->last_depth_iters = ((CgScalars*)(ctx->h_pinned + 64))->iters;
    (void)G;
    return SRPS_OK;
}

int srps_depth_estimation_csr(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_N, const float* d_I,
                              const float* d_xx, const float* d_yy, const float* d_dz,
                              const int* d_Dx_row_ptr, const int* d_Dx_col_ind, const float* d_Dx_val, int n_rows_Dx, int n_cols_Dx, int nnz_Dx,
                              const int* d_Dy_row_ptr, const int* d_Dy_col_ind, const float* d_Dy_val, int n_rows_Dy, int n_cols_Dy, int nnz_Dy,
                              const int* d_KT_row_ptr, const int* d_KT_col_ind, const float* d_KT_val, int n_rows_KT, int n_cols_KT, int nnz_KT,
                              const float* d_z0s, float* d_z, float K00, float K11, int npix, int nimages, int nchannels, float* energy) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    const Grid& G = ctx->grid;
    SRPS_REQUIRE(d_Dx_row_ptr && d_Dx_col_ind && d_Dx_val && d_Dy_row_ptr && d_Dy_col_ind && d_Dy_val, SRPS_ERR_INVALID, "depth_estimation_csr: null gradient matrix");
    SRPS_REQUIRE(G.Ps == 0 || (d_KT_row_ptr && d_KT_col_ind && d_KT_val), SRPS_ERR_INVALID, "depth_estimation_csr: null KT matrix");
    SRPS_REQUIRE(npix == G.P && n_rows_Dx == G.P && n_cols_Dx == G.P && n_rows_Dy == G.P && n_cols_Dy == G.P, SRPS_ERR_INVALID,
                 "depth_estimation_csr: Dx / Dy must be %d x %d (the bound mask), got %d x %d and %d x %d", G.P, G.P, n_rows_Dx, n_cols_Dx, n_rows_Dy, n_cols_Dy);
    SRPS_REQUIRE(n_rows_KT == G.Ps && n_cols_KT == G.P && nnz_KT == G.Ps * G.sf * G.sf, SRPS_ERR_INVALID,
                 "depth_estimation_csr: KT must be %d x %d with %d entries (complete %d x %d blocks of the bound mask), got %d x %d with %d", G.Ps, G.P,
                 G.Ps * G.sf * G.sf, G.sf, G.sf, n_rows_KT, n_cols_KT, nnz_KT);
    SRPS_REQUIRE(nnz_Dx >= 0 && nnz_Dy >= 0, SRPS_ERR_INVALID, "depth_estimation_csr: negative nnz");
    int err = 0;
    SRPS_TRY(csr_matches_grid(ctx, d_Dx_row_ptr, d_Dx_col_ind, d_Dx_val, d_Dy_row_ptr, d_Dy_col_ind, d_Dy_val, d_KT_row_ptr, d_KT_col_ind, d_KT_val, &err));
    SRPS_REQUIRE(err == 0, SRPS_ERR_INVALID, "depth_estimation_csr: %s%s%s not the matrix make_gradient / the KT filter build from the bound mask (SRPS.cu:23-71, 170-193)",
                 (err & 1) ? "Dx " : "", (err & 2) ? "Dy " : "", (err & 4) ? "KT " : "");
    return srps_depth_estimation(ctx, d_s, d_rho, d_N, d_I, d_xx, d_yy, d_dz, d_z0s, d_z, K00, K11, npix, nimages, nchannels, energy);
}

int srps_depth_operator_apply(srps_ctx* ctx, const float* d_x, int npix, float* d_y) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    SRPS_REQUIRE(ctx->tensor_valid, SRPS_ERR_STATE, "depth_operator_apply: no tensor assembled yet");
    SRPS_REQUIRE(d_x && d_y && npix == ctx->grid.P, SRPS_ERR_INVALID, "depth_operator_apply: bad arguments");
    Grid& G = ctx->grid;
    // uses p[0] as the input plane and w as the output plane; both are rewritten by the next solve
    SRPS_TRY(grid_scatter(ctx, d_x, G.d_p));
    SRPS_TRY(grid_apply_plain(ctx, G.d_p, G.d_w));
    return grid_gather(ctx, G.d_w, d_y);
}

// ---- the 8-bit image store --------------------------------------------------------------------
// Looks at the context's images once after they changed (one pass over them and one word read back): when every sample is a
// byte over 255 -- the reference's image-folder input, Utilities.cpp:343 -- they are also kept as bytes, and the albedo sweep and
// the fused energy + lighting sweep of every pass read those.  Same floats, same sums, a quarter of the bytes.
static int image_store_prepare(srps_ctx* ctx) {
    if (ctx->i8_state != 0) return SRPS_OK;
    ctx->i8_state = 2;
    const size_t n = (size_t)ctx->N_local * ctx->C * ctx->grid.P;
    if (!ctx->image_store || n == 0 || ctx->grid.P % 4 != 0) return SRPS_OK;
    int* flag = (int*)ctx->d_report + 200;            // a word of the report record behind everything a pass reads back
    int inexact = 0;
    SRPS_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    SRPS_TRY(launch_pack_bytes(ctx->stream, ctx->I, n, nullptr, flag));
    SRPS_TRY(host_download(ctx, &inexact, flag, sizeof(int), ctx->stream));
    if (inexact) return SRPS_OK;
    if (!ctx->I8_cap_ok(n)) {
        if (ctx->I8) { SRPS_HIP(hipFree(ctx->I8)); ctx->I8 = nullptr; }
        SRPS_TRY(dalloc(&ctx->I8, n));
        ctx->I8_cap = n;
    }
    SRPS_TRY(launch_pack_bytes(ctx->stream, ctx->I, n, ctx->I8, flag));
    ctx->i8_state = 1;
    return SRPS_OK;
}
extern "C++" {
namespace srps {
// the byte copy of d_I when d_I is the context's image array and the copy is current, else null (the caller reads the floats)
const unsigned char* image_store_bytes(srps_ctx* ctx, const float* d_I) {
    if (!ctx->image_store || !ctx->have_state || d_I != ctx->I) return nullptr;
    if (ctx->i8_state == 0 && image_store_prepare(ctx) != SRPS_OK) { (void)hipGetLastError(); ctx->i8_state = 2; }
    return ctx->i8_state == 1 ? ctx->I8 : nullptr;
}
}  // namespace srps
}  // extern "C++"

// ---- pipeline -------------------------------------------------------------------------------
// s = (0, 0, -1, 0) for every image and channel (SRPS.cu:209-217)
__global__ void k_init_s(float* __restrict__ s, int n) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) s[t] = ((t & 3) == 2) ? -1.f : 0.f;
}

// events of the upload pipeline, made once per context
static int setup_events(srps_ctx* ctx, int slots) {
    while ((int)ctx->ev_copied.size() < slots) {
        hipEvent_t a = nullptr, b = nullptr;
        SRPS_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        ctx->ev_copied.push_back(a);
        SRPS_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        ctx->ev_gathered.push_back(b);
    }
    return SRPS_OK;
}

// SRPS.cu:100-270.  Where the time of a set-up goes, and what overlaps with what:
//   * the images are the one large transfer (1.0 GB of floats at the metric's configuration = 18.7 ms at the PCIe rate; 0.25 GB
//     when the caller hands over the bytes its image loader read, srps_problem.I_u8).  The caller's array is pinned in place
//     (hipHostRegister: plain DMA instead of the runtime's staging) and the copies are queued on the context's stream FIRST;
//   * the grid structure is built on the device, on the auxiliary stream, from the mask (kernels_structure.hip: 16.8 MB up,
//     a few kernels, 32 bytes back) while the images are in flight -- round 2's host loop over h*w took as long as the DMA;
//   * every image is compacted (copy_if with the mask, SRPS.cu:223-234) on a third stream as soon as its copy has landed, so the
//     compaction overlaps with the next copies and only the last image's gather is left when the DMA ends;
//   * the arrays come out of two arenas that survive a re-setup of the same size: no hipMalloc / hipFree per solve.
static int setup_impl(srps_ctx* ctx, const srps_problem* pr) {
    ctx->light_cache_valid = false;
    ctx->ssum_valid = false;
    ctx->grad_current = false;
    ctx->depth_assembled = false;
    SRPS_REQUIRE(pr->mask && pr->K && pr->zs_lr && pr->z_full, SRPS_ERR_INVALID, "setup: mask, K, zs_lr and z_full are required");
    SRPS_REQUIRE(pr->n_channels > 0 && pr->n_channels <= 8 && pr->n_images >= 0 && pr->n_images_total > 0, SRPS_ERR_INVALID, "setup: bad image counts");
    SRPS_REQUIRE(pr->image_offset >= 0 && pr->image_offset + pr->n_images <= pr->n_images_total, SRPS_ERR_INVALID, "setup: shard [%d,%d) outside [0,%d)", pr->image_offset, pr->image_offset + pr->n_images, pr->n_images_total);
    SRPS_REQUIRE(!(pr->I && pr->I_u8), SRPS_ERR_INVALID, "setup: give the images as floats (I) or as bytes (I_u8), not both");
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    // SRPS_SETUP_TIMING=1: where the host's set-up time goes (stderr)
    const bool tm = getenv("SRPS_SETUP_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };
    auto t_all = now(), t0 = now();
    state_release(ctx);
    const int C_ = pr->n_channels, NL_ = pr->n_images;
    const bool bytes_in = pr->I_u8 != nullptr;
    const size_t esz = bytes_in ? 1 : sizeof(float);
    const size_t per = (size_t)C_ * pr->h * pr->w;                   // samples of one image
    const char* host_I = bytes_in ? (const char*)pr->I_u8 : (const char*)pr->I;
    struct Pin {
        const void* p = nullptr;
        srps_ctx* c = nullptr;
        ~Pin() {      // also on the error paths: no copy in flight when the caller's array is unpinned
            if (!p) return;
            (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->gather_stream);
            const hipError_t e = hipHostUnregister(const_cast<void*>(p));
            if (e != hipSuccess) {      // the array stays registered: say so (a later array at the same address would be taken for it)
                fprintf(stderr, "srps_setup: hipHostUnregister(%p) failed: %s\n", p, hipGetErrorString(e));
                (void)hipGetLastError();
            }
        }
    } pin;
    pin.c = ctx;
    // The images reach the device in batches of `slots` images (all of them at once while they fit 2 GB of staging memory), each batch
    // one transfer: through the library's pinned buffer (host_upload; on a thread of its own for the first batch, so that the copies
    // run while this thread builds the grid structure), or -- "pin_uploads", off by default -- straight out of the caller's array,
    // registered in place for the duration.
    int slots = 0;
    struct Uploader {
        std::thread t;
        int rc = SRPS_OK;
        std::string msg;                                 // the thread's own error text (srps_last_error is per thread), handed to the caller's
    } first_batch;
    auto copy_batch = [&](int n0, int cnt) -> int {     // images n0 .. n0 + cnt - 1 into the staging slots 0 .. cnt - 1, on the context's stream
        const char* src = host_I + (size_t)n0 * per * esz;
        if (pin.p) { SRPS_HIP(hipMemcpyAsync(ctx->ws_images.p, src, (size_t)cnt * per * esz, hipMemcpyHostToDevice, ctx->stream)); return SRPS_OK; }
        return host_upload(ctx, ctx->ws_images.p, src, (size_t)cnt * per * esz, ctx->stream);
    };
    std::function<int()> start_uploads = [&]() -> int {
        if (!host_I || NL_ <= 0) return SRPS_OK;
        const size_t bytes = (size_t)NL_ * per * esz;
        slots = (int)std::min<size_t>((size_t)NL_, std::max<size_t>(2, ((size_t)2 << 30) / (per * esz)));
        SRPS_TRY(ensure(ctx->ws_images, (size_t)slots * per * esz));      // kept across set-ups (grow-only)
        SRPS_TRY(setup_events(ctx, 1));
        if (ctx->pin_uploads && bytes >= ((size_t)8 << 20)) {
            if (hipHostRegister((void*)host_I, bytes, hipHostRegisterDefault) == hipSuccess) pin.p = host_I;
            else (void)hipGetLastError();
        }
        if (pin.p) return copy_batch(0, slots);                          // queued; the device reads the caller's pages
        try {
            first_batch.t = std::thread([&, cnt = slots]() {
                first_batch.rc = hipSetDevice(ctx->device) == hipSuccess ? copy_batch(0, cnt) : SRPS_ERR_HIP;
                if (first_batch.rc != SRPS_OK) first_batch.msg = srps_last_error();
            });
        } catch (...) {                                                 // no thread to be had: the copies run here, before the structure
            return copy_batch(0, slots);
        }
        return SRPS_OK;
    };
    // declared AFTER the two functions the thread calls, so that on every return path the thread has ended before they are destroyed
    // (round-4 advisor finding); also on the error paths: nothing reads the caller's array after srps_setup
    struct Joiner {
        Uploader& u;
        ~Joiner() { if (u.t.joinable()) u.t.join(); }
    } join_first_batch{first_batch};
    if (getenv("SRPS_SETUP_TRACE"))      // development aid: the caller's arrays (to place a fault address, should the device ever touch one)
        fprintf(stderr, "srps_setup trace: mask %p + %zu, z_full %p + %zu, zs_lr %p + %zu, images %p + %zu\n", (const void*)pr->mask, (size_t)pr->h * pr->w * 4,
                (const void*)pr->z_full, (size_t)pr->h * pr->w * 4, (const void*)pr->zs_lr, (size_t)pr->h * pr->w * 4 / ((size_t)pr->sf * pr->sf), (const void*)host_I,
                (size_t)NL_ * per * esz);
    SRPS_TRY(build_grid(ctx, pr->h, pr->w, pr->sf, pr->mask, &start_uploads));
    if (tm) { fprintf(stderr, "srps_setup: uploads started (%s), grid structure on the device, grid arena: %.2f ms\n", pin.p ? "caller's array registered in place" : "through the pinned transfer buffer", ms_since(t0)); t0 = now(); }
    Grid& G = ctx->grid;
    hipStream_t ax = ctx->aux_stream;
    const int P = G.P, C = pr->n_channels, NL = pr->n_images, NT = pr->n_images_total;
    ctx->C = C; ctx->N_local = NL; ctx->N_total = NT; ctx->img_offset = pr->image_offset;
    ctx->fx = pr->K[0]; ctx->fy = pr->K[4]; ctx->cx = pr->K[6]; ctx->cy = pr->K[7];       // SRPS.cu:256, 269
    // A full-frame mask makes the compaction (copy_if, SRPS.cu:223-234) the identity: the float images as they arrive, [n][c][h w], ARE
    // I[n][c][P] -- when they all fit the transfer area at once the context's I is that area, no compaction pass runs (1 GB read and
    // written: 0.5 ms at the metric's configuration, and a second gigabyte of device memory)
    const bool alias_I = !bytes_in && host_I && NL > 0 && slots == NL && (size_t)P == (size_t)G.h * G.w;
    ctx->I_in_ws_images = alias_I;
    // ---- state arena ----
    {
        const size_t fP = al256((size_t)P * sizeof(float));
        size_t need = al256((size_t)NT * C * 4 * sizeof(float)) + al256((size_t)C * P * sizeof(float)) + 2 * al256(4 * (size_t)P * sizeof(float)) + 8 * fP +
                      al256((size_t)std::max(G.Ps, 1) * sizeof(float)) + (alias_I ? 0 : al256((size_t)std::max(NL, 1) * C * P * sizeof(float))) + al256(2 * (size_t)C * P * sizeof(float)) +
                      (NL != NT ? al256(3 * (size_t)P * sizeof(float)) : 0) + 256;
        SRPS_TRY(ensure(ctx->state_arena, need));                    // grows only when this problem is larger than every earlier one
        size_t used = 0;
        auto carve = [&](size_t bytes) -> float* { float* p = (float*)((char*)ctx->state_arena.p + used); used += al256(bytes); return p; };
        ctx->s = carve((size_t)NT * C * 4 * sizeof(float)); ctx->rho = carve((size_t)C * P * sizeof(float)); ctx->z = carve((size_t)P * sizeof(float));
        ctx->Nrm = carve(4 * (size_t)P * sizeof(float)); ctx->Nrm2 = carve(4 * (size_t)P * sizeof(float));
        ctx->dz = carve((size_t)P * sizeof(float)); ctx->dz2 = carve((size_t)P * sizeof(float));
        ctx->zx = carve((size_t)P * sizeof(float));
        ctx->zy = carve((size_t)P * sizeof(float)); ctx->xx = carve((size_t)P * sizeof(float)); ctx->yy = carve((size_t)P * sizeof(float));
        ctx->z0s = carve((size_t)std::max(G.Ps, 1) * sizeof(float));
        ctx->I = alias_I ? (float*)ctx->ws_images.p : carve((size_t)std::max(NL, 1) * C * P * sizeof(float));
        ctx->albedo_ex = carve(2 * (size_t)C * P * sizeof(float)); ctx->energy_ex = ctx->d_report;
        if (NL != NT) ctx->q_ex = carve(3 * (size_t)P * sizeof(float));      // a shard exchanges q compactly (3 P floats, not 3 padded planes)
    }
    ctx->have_state = true;
    // initial values on the auxiliary stream (the context's stream is busy with the images)
    hipLaunchKernelGGL(k_init_s, dim3(std::max(1, std::min(cdiv((long long)NT * C * 4, 256), 1024))), dim3(256), 0, ax, ctx->s, NT * C * 4);      // SRPS.cu:209-217
    SRPS_LAUNCH_CHECK();
    SRPS_TRY(launch_fill(ax, ctx->rho, (size_t)C * P, 0.5f));                               // SRPS.cu:220
    SRPS_TRY(launch_fill(ax, ctx->Nrm2 + 3 * (size_t)P, (size_t)P, 1.f));                   // N3 == 1 (dc.cu:175) in the second set of normals too
    ctx->n3_one = true;                                                                     // ... and in the first: the set-up's normals kernel writes it
    // masked LR depth and initial HR depth (copy_if SRPS.cu:237-246): uploaded whole into the scratch the mask came through,
    // compacted with the index lists
    {
        float* d_tmp = (float*)ctx->ws_struct.p;
        const size_t hw = (size_t)G.h * G.w, hws = hw / ((size_t)G.sf * G.sf);
        SRPS_TRY(host_upload(ctx, d_tmp, pr->z_full, hw * sizeof(float), ax));
        SRPS_TRY(launch_gather_index(ax, d_tmp, G.d_imask, P, ctx->z));
        // (stream order: the second copy into the scratch follows the gather that read the first)
        SRPS_TRY(host_upload(ctx, d_tmp, pr->zs_lr, hws * sizeof(float), ax));
        SRPS_TRY(launch_gather_index(ax, d_tmp, G.d_imasks, G.Ps, ctx->z0s));
    }
    SRPS_TRY(launch_meshgrid_compact(ax, G.d_imask, P, G.h, ctx->cx, ctx->cy, ctx->xx, ctx->yy));
    SRPS_HIP(hipEventRecord(ctx->aux_event, ax));
    if (tm) { fprintf(stderr, "srps_setup: state arena + initial values queued: %.2f ms\n", ms_since(t0)); t0 = now(); }
    // ---- compaction of the images, pipelined with their copies ----
    const bool want_bytes = bytes_in && ctx->image_store && P % 4 == 0;
    if (want_bytes && !ctx->I8_cap_ok((size_t)NL * C * P)) {
        if (ctx->I8) { (void)hipFree(ctx->I8); ctx->I8 = nullptr; }
        SRPS_TRY(dalloc(&ctx->I8, (size_t)NL * C * P));
        ctx->I8_cap = (size_t)NL * C * P;
    }
    if (host_I && NL > 0) {
        const size_t hwp = (size_t)G.h * G.w;
        hipStream_t gs = ctx->gather_stream;
        SRPS_HIP(hipStreamWaitEvent(gs, ctx->aux_event, 0));      // the index lists
        for (int n0 = 0; n0 < NL; n0 += slots) {
            const int cnt = std::min(slots, NL - n0);
            if (n0 == 0) {
                if (first_batch.t.joinable()) first_batch.t.join();
                if (first_batch.rc != SRPS_OK) { set_error("srps_setup: the image upload failed: %s", first_batch.msg.c_str()); return first_batch.rc; }
            } else {                                               // the staging slots in their next use: the copies wait, on the device, for the gathers that read them
                SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[0], 0));
                SRPS_TRY(copy_batch(n0, cnt));
            }
            SRPS_HIP(hipEventRecord(ctx->ev_copied[0], ctx->stream));
            SRPS_HIP(hipStreamWaitEvent(gs, ctx->ev_copied[0], 0));
            for (int n = n0; n < n0 + cnt && !alias_I; ++n) {
                const char* stage = (const char*)ctx->ws_images.p + (size_t)(n - n0) * per * esz;
                if (bytes_in)
                    SRPS_TRY(launch_gather_images_u8(gs, (const unsigned char*)stage, G.d_imask, P, C, hwp, 1, ctx->I + (size_t)n * C * P,
                                                     want_bytes ? ctx->I8 + (size_t)n * C * P : nullptr));
                else
                    SRPS_TRY(launch_gather_images(gs, (const float*)stage, G.d_imask, P, C, hwp, 1, ctx->I + (size_t)n * C * P));
            }
            SRPS_HIP(hipEventRecord(ctx->ev_gathered[0], gs));    // the batch is compacted
        }
        SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[0], 0));
    }
    SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_event, 0));
    if (tm) { fprintf(stderr, "srps_setup: %d image copies + compactions queued: %.2f ms\n", NL, ms_since(t0)); t0 = now(); }
    if (want_bytes) ctx->i8_state = 1;                     // the bytes ARE the images: nothing to look at
    else { ctx->i8_state = 0; SRPS_TRY(image_store_prepare(ctx)); }
    if (tm) { fprintf(stderr, "srps_setup: %.2f GB of images on the device and compacted, image store (%s): %.2f ms (the host waited here)\n",
                      (double)NL * per * esz * 1e-9, ctx->i8_state == 1 ? "bytes" : "floats", ms_since(t0)); t0 = now(); }
    SRPS_TRY(srps_normals(ctx));                                                            // SRPS.cu:264-270
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    SRPS_HIP(hipStreamSynchronize(ctx->gather_stream));
    if (tm) fprintf(stderr, "srps_setup: total %.2f ms\n", ms_since(t_all));
    return SRPS_OK;
}

int srps_setup(srps_ctx* ctx, const srps_problem* pr) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(pr != nullptr, SRPS_ERR_INVALID, "setup: problem is NULL");
    return setup_impl(ctx, pr);
}

static int upload_image_impl(srps_ctx* ctx, int li, const void* host_image, bool bytes_in) {
    SRPS_REQUIRE(host_image && li >= 0 && li < ctx->N_local, SRPS_ERR_INVALID, "upload_image: bad arguments");
    ctx->light_cache_valid = false;
    ctx->ssum_valid = false;
    Grid& G = ctx->grid;
    const size_t hw = (size_t)G.h * G.w, n = hw * ctx->C, esz = bytes_in ? 1 : sizeof(float);
    SRPS_HIP(hipStreamSynchronize(ctx->stream));     // staging buffer reuse
    SRPS_TRY(ensure(ctx->ws_stage, n * esz));
    SRPS_TRY(host_upload(ctx, ctx->ws_stage.p, host_image, n * esz, ctx->stream));
    float* out = ctx->I + (size_t)li * ctx->C * G.P;
    if (!bytes_in) {
        ctx->i8_state = 0;                   // looked at again at the next sweep
        return launch_gather_images(ctx->stream, (const float*)ctx->ws_stage.p, G.d_imask, G.P, ctx->C, hw, 1, out);
    }
    // bytes: the byte store stays current when it is in use (every other image is a byte image already)
    unsigned char* out8 = (ctx->i8_state == 1 && ctx->I8) ? ctx->I8 + (size_t)li * ctx->C * G.P : nullptr;
    if (!out8) ctx->i8_state = 0;
    return launch_gather_images_u8(ctx->stream, (const unsigned char*)ctx->ws_stage.p, G.d_imask, G.P, ctx->C, hw, 1, out, out8);
}
int srps_upload_image(srps_ctx* ctx, int li, const float* host_image) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    return upload_image_impl(ctx, li, host_image, false);
}
int srps_upload_image_u8(srps_ctx* ctx, int li, const unsigned char* host_image) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    return upload_image_impl(ctx, li, host_image, true);
}

int srps_dims(srps_ctx* ctx, int* npix, int* npixs, int* grid_h, int* grid_w, int* n_images, int* n_channels) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    if (npix) *npix = ctx->grid.P;
    if (npixs) *npixs = ctx->grid.Ps;
    if (grid_h) *grid_h = ctx->grid.Hg;
    if (grid_w) *grid_w = ctx->grid.Wg;
    if (n_images) *n_images = ctx->N_local;
    if (n_channels) *n_channels = ctx->C;
    return SRPS_OK;
}

int srps_lighting_local(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    PhaseSpan span(ctx, SRPS_PHASE_LIGHTING);
    ctx->ssum_valid = false;             // s changes
    ctx->depth_assembled = false;
    return lighting(ctx, ctx->s, ctx->rho, ctx->Nrm, ctx->I, ctx->grid.P, ctx->N_local, ctx->C, ctx->N_total, ctx->img_offset,
                    ctx->N_local != ctx->N_total, /*use_cache=*/true);
}
int srps_lighting(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(ctx->N_local == ctx->N_total, SRPS_ERR_STATE, "srps_lighting: context holds a shard; use srps_lighting_local + all-reduce");
    return srps_lighting_local(ctx);
}

int srps_albedo_partial(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    PhaseSpan span(ctx, SRPS_PHASE_ALBEDO_SWEEP);
    float* ssum = nullptr;
    ctx->ssum_valid = false;
    ctx->depth_assembled = false;
    if ((ctx->albedo_mode == SRPS_ALBEDO_FUSED || ctx->albedo_mode == SRPS_ALBEDO_AUTO) && ctx->N_local == ctx->N_total && ctx->N_local > 0) {
        // one sweep over the images: the albedo's fixed point, g = (rho / dz)^2 and q straight onto the grid
        bool ok = false;
        SRPS_TRY(depth_fused_prepare(ctx, ctx->s, ctx->fx, ctx->fy, ctx->C, ctx->N_total, ctx->N_local, ctx->img_offset, ctx->cx, ctx->cy, &ok));
        if (ok) {
            ctx->light_cache_valid = false;      // rho changes
            SRPS_TRY(albedo_fused(ctx, ctx->s, ctx->Nrm, ctx->I, ctx->grid.P, ctx->N_local, ctx->C, ctx->rho, ctx->grid.d_tconsts + 64, ctx->xx, ctx->yy, ctx->dz,
                                  ctx->fx, ctx->fy));
            ctx->tensor_valid = true;
            ctx->depth_assembled = true;
            return SRPS_OK;
        }
    }
    if (ctx->assemble_from_sums && ctx->N_local > 0) {      // this sweep over I also leaves the image sums of the depth right-hand side
        SRPS_TRY(ensure(ctx->ws_ssum, (size_t)3 * ctx->C * ctx->grid.P * sizeof(float)));
        ssum = (float*)ctx->ws_ssum.p;
    }
    // a shard leaves its part of num and the COMPLETE den (which does not involve the images): only num is exchanged
    SRPS_TRY(albedo_numden(ctx, ctx->s, ctx->Nrm, ctx->I, ctx->grid.P, ctx->N_local, ctx->C, ctx->img_offset, ctx->albedo_ex,
                           ctx->fx, ctx->fy, ssum, ctx->N_total));
    ctx->ssum_valid = ssum != nullptr;
    return SRPS_OK;
}
int srps_albedo_finish(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    if (ctx->depth_assembled) return SRPS_OK;      // SRPS_ALBEDO_FUSED: the sweep formed the albedo itself
    PhaseSpan span(ctx, SRPS_PHASE_ALBEDO_SOLVE);
    ctx->light_cache_valid = false;      // rho changes
    SRPS_TRY(albedo_finish(ctx, ctx->rho, ctx->albedo_ex, ctx->grid.P, ctx->C, /*pipeline=*/true));
    if (ctx->N_local != ctx->N_total && !ctx->defer_shard_checks) {  // a shard driven phase by phase cannot repeat the pass on its own later (srps_energy_finish): look now
        int aborted = 0;
        SRPS_TRY(persistent_sync_check(ctx, &aborted));
        if (aborted & ABORT_ALBEDO) SRPS_TRY(albedo_finish(ctx, ctx->rho, ctx->albedo_ex, ctx->grid.P, ctx->C, /*pipeline=*/true));
    }
    return SRPS_OK;
}
int srps_albedo(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(ctx->N_local == ctx->N_total, SRPS_ERR_STATE, "srps_albedo: context holds a shard; use the *_partial/_finish pair");
    SRPS_TRY(srps_albedo_partial(ctx));
    return srps_albedo_finish(ctx);
}

int srps_depth_partial(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    if (ctx->depth_assembled) { ctx->depth_assembled = false; ctx->q_in_exchange = false; return SRPS_OK; }      // SRPS_ALBEDO_FUSED: g and q are on the grid
    PhaseSpan span(ctx, SRPS_PHASE_DEPTH_ASSEMBLY);
    const float* ssum = (ctx->assemble_from_sums && ctx->ssum_valid) ? (const float*)ctx->ws_ssum.p : nullptr;
    ctx->q_in_exchange = ctx->q_ex != nullptr;
    return depth_assemble(ctx, ctx->s, ctx->rho, ctx->I, ctx->xx, ctx->yy, ctx->dz, ctx->fx, ctx->fy, ctx->grid.P, ctx->N_local, ctx->C,
                          ctx->N_total, ctx->img_offset, ctx->cx, ctx->cy, ssum, ctx->q_ex);
}
extern "C++" {
namespace srps {
// the two halves of srps_depth_solve around the CG (the strip-partitioned group solve runs the CG of several contexts in lockstep)
int depth_solve_prepare(srps_ctx* ctx) {
    SRPS_REQUIRE(ctx->have_state && ctx->tensor_valid, SRPS_ERR_STATE, "depth_solve: call srps_depth_partial first");
    if (ctx->q_in_exchange) { SRPS_TRY(depth_q_scatter(ctx, ctx->q_ex)); ctx->q_in_exchange = false; }      // the all-reduced q of a shard
    ctx->light_cache_valid = false;      // z changes
    ctx->normals_pending = false;
    const bool plane_current = ctx->grad_current && ctx->plane_holds_z;      // nothing wrote z or the plane since the last solve
    ctx->grad_current = false;
    ctx->plane_holds_z = false;
    if (!plane_current) SRPS_TRY(grid_scatter(ctx, ctx->z, ctx->grid.d_x));      // else the last solve left z on the grid plane
    return grid_rhs(ctx, ctx->z0s);                                                // dc.cu:743-745
}
int depth_solve_finish(srps_ctx* ctx) {
    SRPS_TRY(grid_gradient(ctx, ctx->grid.d_x, ctx->zx, ctx->zy, ctx->z));         // the new z in the compact layout, and Dx z, Dy z (energy + normals)
    ctx->report_pending = true;          // the CG scalars are fetched with the rest of the report record
    ctx->grad_current = true;
    ctx->plane_holds_z = true;
    return SRPS_OK;
}
}  // namespace srps
}  // extern "C++"

int srps_depth_solve(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    PhaseSpan span(ctx, SRPS_PHASE_DEPTH_SOLVE);
    SRPS_TRY(depth_solve_prepare(ctx));
    SRPS_TRY(grid_cg(ctx, ctx->cg_max_iter + 1, false));          // dc.cu:758-759 (residual; k <= max_iter => 101 steps)
    if (ctx->N_local != ctx->N_total && !ctx->defer_shard_checks) {  // see srps_albedo_finish
        int aborted = 0;
        SRPS_TRY(persistent_sync_check(ctx, &aborted));
        // the plane the launch started from is current again (persistent_aborts)
        if (aborted & ABORT_DEPTH) { SRPS_TRY(grid_rhs(ctx, ctx->z0s)); SRPS_TRY(grid_cg(ctx, ctx->cg_max_iter + 1, false)); }
    }
    return depth_solve_finish(ctx);
}
int srps_energy_partial(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    PhaseSpan span(ctx, SRPS_PHASE_ENERGY);
    ctx->report_seq_armed = 0;
    // One GPU, fused sweep: the sweep's last block adds both energy terms and writes the pass's report into the host's pinned record
    // itself (ReportFinish) -- no k_sum_to, no k_final_sum, no copy; srps_energy_finish waits for the record's sequence number.
    // (A shard's energy term goes through an all-reduce first: the old way.)
    const bool in_sweep = ctx->report_zero_copy && ctx->h_pinned_dev && ctx->fuse_energy_lighting && ctx->N_local > 0 && ctx->N_local == ctx->N_total &&
                          !ctx->defer_shard_checks && ctx->energy_ex == ctx->d_report;
    if (in_sweep) {
        ReportFinish fin;
        fin.ticket = reinterpret_cast<unsigned*>(ctx->d_report + REPORT_TICKET_AT);
        SRPS_TRY(grid_energy_t1(ctx, ctx->z0s, ctx->energy_ex, &fin.t1_part, &fin.n_t1));
        fin.report = ctx->d_report; fin.host_report = ctx->h_pinned_dev;
        fin.seq = ++ctx->report_seq;
        if (fin.seq == 0) fin.seq = ++ctx->report_seq;
        bool armed = false;
        SRPS_TRY(energy_light_fused(ctx, ctx->s, ctx->rho, ctx->I, ctx->xx, ctx->yy, ctx->dz, ctx->z, ctx->zx, ctx->zy, ctx->fx, ctx->fy,
                                    ctx->grid.P, ctx->N_local, ctx->C, ctx->img_offset, ctx->energy_ex + 1, &fin, &armed));
        if (armed) ctx->report_seq_armed = fin.seq;
        else SRPS_TRY(launch_final_sum(ctx->stream, fin.t1_part, fin.n_t1, ctx->energy_ex));      // another sweep kernel ran: term 1 the old way (term 2 is done)
        return SRPS_OK;
    }
    SRPS_TRY(grid_energy_t1(ctx, ctx->z0s, ctx->energy_ex));
    // the sweep over I that evaluates the energy also leaves the lighting sums of the next outer iteration
    if (ctx->fuse_energy_lighting && ctx->N_local > 0)
        return energy_light_fused(ctx, ctx->s, ctx->rho, ctx->I, ctx->xx, ctx->yy, ctx->dz, ctx->z, ctx->zx, ctx->zy, ctx->fx, ctx->fy,
                                  ctx->grid.P, ctx->N_local, ctx->C, ctx->img_offset, ctx->energy_ex + 1);
    return energy_photometric_partial(ctx, ctx->s, ctx->rho, ctx->I, ctx->xx, ctx->yy, ctx->dz, ctx->z, ctx->zx, ctx->zy, ctx->fx, ctx->fy,
                                      ctx->grid.P, ctx->N_local, ctx->C, ctx->img_offset, ctx->energy_ex + 1);
}
// the abort flags of this rank's persistent kernels as two floats behind the energy terms (report record [2], [3]): they travel
// with the energy term of a sharded pass
__global__ void k_abort_flags_to_float(const CgScalars* __restrict__ scal, float* __restrict__ out) {
    if (threadIdx.x == 0) { const int f = scal->abort_flags; out[0] = (f & ABORT_ALBEDO) ? 1.f : 0.f; out[1] = (f & ABORT_DEPTH) ? 1.f : 0.f; }
}
static int sharded_energy_exchange(srps_ctx* ctx) {
    hipLaunchKernelGGL(k_abort_flags_to_float, dim3(1), dim3(64), 0, ctx->stream, (const CgScalars*)(ctx->d_report + 64), ctx->d_report + 2);
    SRPS_LAUNCH_CHECK();
    return comm_all_reduce_sum(ctx, ctx->energy_ex + 1, 3);      // t2 of the local images, albedo aborts, depth aborts
}

// A persistent kernel of this pass gave up a wait (on this rank or, in a sharded job, on any rank): the pass's tail is repeated
// with the streaming kernels, from the state the pass had before the aborted launch.  persistent_aborts has made the depth
// plane of the pass's start current again; the gradient, normals and dz the aborted pass took from the discarded depth are
// formed again from it.  After an albedo abort the streaming CG starts from whatever the aborted launch left in rho (channels
// it had finished are stored, devicecalls.cu:540 converges to the same fixed point from any start: 1e-7); in a sharded job
// rank 0's albedo is then broadcast, so that the replicas stay bit-identical.
static int redo_pass_tail(srps_ctx* ctx, int aborted) {
    Grid& G = ctx->grid;
    const bool sharded = ctx->N_local != ctx->N_total || (ctx->defer_shard_checks && comm_bound(ctx));
    ctx->normals_pending = false;
    if (!ctx->plane_restored) SRPS_TRY(grid_scatter(ctx, ctx->z, G.d_x));      // no solve touched z or the plane in this pass: z itself is the pass's depth
    ctx->plane_restored = false;
    SRPS_TRY(grid_gradient(ctx, G.d_x, ctx->zx, ctx->zy, ctx->z));
    SRPS_TRY(launch_normals(ctx->stream, ctx->z, ctx->zx, ctx->zy, ctx->xx, ctx->yy, G.P, ctx->fx, ctx->fy, ctx->Nrm, ctx->dz));
    ctx->grad_current = true; ctx->plane_holds_z = true;
    if (aborted & ABORT_ALBEDO) {
        SRPS_TRY(srps_albedo_finish(ctx));
        if (sharded && comm_bound(ctx) && ctx->comm_world > 1) SRPS_TRY(comm_broadcast(ctx, ctx->rho, (size_t)ctx->C * G.P, 0));
        SRPS_TRY(srps_depth_partial(ctx));
        if (sharded && ctx->q_ex) SRPS_TRY(comm_all_reduce_sum(ctx, ctx->q_ex, 3 * (size_t)G.P));
    }
    SRPS_TRY(srps_depth_solve(ctx));
    SRPS_TRY(srps_energy_partial(ctx));
    if (sharded) SRPS_TRY(sharded_energy_exchange(ctx));
    SRPS_TRY(srps_normals(ctx));
    return SRPS_OK;
}

// The end of a solve: the caller is waiting for this and nothing else, so the thread polls (option "host_wait_spin") -- a sleeping
// hipStreamSynchronize wakes 20 - 40 us after the stream has drained, and with the report written by the sweep's last block the
// stream is still finishing that sweep when the loop ends.
static int wait_for_stream(srps_ctx* ctx) {
    if (ctx->host_wait_spin) {
        hipError_t q;
        while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady) { }
        if (q != hipSuccess) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
        return SRPS_OK;
    }
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    return SRPS_OK;
}

// `drain`: return with the context's stream idle -- what the entry points promise a caller who reads the results through other streams.
// The library's own pass loops go on as soon as the record has arrived (the sweep that wrote it is in its last microseconds; everything
// they queue follows it on the same stream) and drain once, at their end.
static int energy_finish_impl(srps_ctx* ctx, float* energy, bool drain);
int srps_energy_finish(srps_ctx* ctx, float* energy) { return energy_finish_impl(ctx, energy, true); }
static int energy_finish_impl(srps_ctx* ctx, float* energy, bool drain) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(energy != nullptr, SRPS_ERR_INVALID, "energy_finish: energy is NULL");
    bool record_arrived = false;
    if (ctx->report_seq_armed) {
        // the sweep's last block writes the record into h_pinned and the sequence number behind it: the host looks at that word (and,
        // every thousand looks, at the stream: an error ends the wait, and a stream that has drained without the word having arrived --
        // it cannot, short of host memory the device's stores do not reach -- switches this way of reporting off for the context)
        const unsigned want = ctx->report_seq_armed;
        ctx->report_seq_armed = 0;
        const unsigned* word = reinterpret_cast<const unsigned*>(ctx->h_pinned + REPORT_SEQ_AT);
        // the sequence word says "complete", the check word proves it: seq ^ (xor of the record's words) as the device formed it; a
        // record whose words are not all in yet does not give it, and is looked at again
        auto complete = [&]() -> bool {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) return false;
            const unsigned* rec = reinterpret_cast<const unsigned*>(ctx->h_pinned);
            unsigned x = want;
            for (int i = 0; i < REPORT_FLOATS; ++i) x ^= __atomic_load_n(rec + i, __ATOMIC_RELAXED);
            return x == __atomic_load_n(rec + REPORT_CHECK_AT, __ATOMIC_RELAXED);
        };
        for (unsigned looks = 1;; ++looks) {
            if (complete()) { record_arrived = true; break; }
            if ((looks & 1023u) == 0u) {
                const hipError_t q = hipStreamQuery(ctx->stream);
                if (q == hipSuccess) {
                    if (complete()) record_arrived = true;
                    else { fprintf(stderr, "srps: the report record written by the energy sweep did not reach the host whole; fetching it by copy from now on\n"); ctx->report_zero_copy = 0; }
                    break;
                }
                if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
            }
        }
        if (record_arrived) ctx->report_pending = false;
    }
    if (!record_arrived) SRPS_TRY(report_fetch(ctx));
    // The host's one wait per pass -- the stop rule needs the energy (SRPS.cu:318-331) -- and the only stretch of a pass in which
    // the device has nothing queued: hipStreamSynchronize puts the thread to sleep on the completion signal and wakes it ~20 us
    // late; polling the stream picks the result up within a few microseconds (option "host_wait_spin", on by default: the caller is
    // waiting for this result and nothing else).
    if (record_arrived && !drain) {
        // nothing to wait for: the record is complete (every kernel of the pass ran before the sweep's last block wrote it)
    } else if (ctx->host_wait_spin) {
        hipError_t q;
        while ((q = hipStreamQuery(ctx->stream)) == hipErrorNotReady) { }
        if (q != hipSuccess) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
    } else
        SRPS_HIP(hipStreamSynchronize(ctx->stream));
    albedo_iters_collect(ctx);
    // The one place a pass waits for the device: were the persistent kernels of this pass served?  If not (nothing was stored by
    // them), the part of the pass that followed is repeated with the streaming kernels -- on one GPU only: a shard has looked
    // already (srps_albedo_finish, srps_depth_solve), its energy term went through the all-reduce.
    int aborted = 0;
    // a sharded pass run by the library (srps_execute_sharded) carries every rank's abort flags in its energy all-reduce:
    // [2] albedo, [3] depth, as counts of the ranks concerned -- the same on all ranks, which therefore decide the same
    const int others = ctx->defer_shard_checks ? ((ctx->h_pinned[2] > 0.f ? ABORT_ALBEDO : 0) | (ctx->h_pinned[3] > 0.f ? ABORT_DEPTH : 0)) : 0;
    SRPS_TRY(persistent_aborts(ctx, &aborted, others));
    if (aborted && (ctx->N_local == ctx->N_total || ctx->defer_shard_checks)) {
        SRPS_TRY(redo_pass_tail(ctx, aborted));
        SRPS_TRY(report_fetch(ctx));
        SRPS_HIP(hipStreamSynchronize(ctx->stream));
        albedo_iters_collect(ctx);
    }
    *energy = ctx->h_pinned[0] + ctx->lambda * ctx->h_pinned[1];
    ctx->last_depth_iters = ((CgScalars*)(ctx->h_pinned + 64))->iters;
    ctx->last_light_iters = *(int*)(ctx->h_pinned + 8);
    return SRPS_OK;
}
int srps_depth(srps_ctx* ctx, float* energy) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(ctx->N_local == ctx->N_total, SRPS_ERR_STATE, "srps_depth: context holds a shard; use the sharded phases");
    SRPS_TRY(srps_depth_partial(ctx));
    SRPS_TRY(srps_depth_solve(ctx));
    SRPS_TRY(srps_energy_partial(ctx));
    return srps_energy_finish(ctx, energy);
}

int srps_normals(srps_ctx* ctx) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    PhaseSpan span(ctx, SRPS_PHASE_NORMALS);
    Grid& G = ctx->grid;
    if (ctx->normals_pending && ctx->grad_current) {
        // the fused energy + lighting sweep of this pass has stored the normals of the new depth and its dz in the SECOND set of
        // arrays (option "fuse_normals"; until this call every reader sees the normals and dz of the previous depth, as with the
        // kernel): the two sets swap roles, nothing is launched
        std::swap(ctx->Nrm, ctx->Nrm2);
        std::swap(ctx->dz, ctx->dz2);
        ctx->normals_pending = false;
        ctx->light_cache_normals = true;
        return SRPS_OK;
    }
    ctx->normals_pending = false;
    if (!ctx->grad_current) {
        SRPS_TRY(grid_scatter(ctx, ctx->z, G.d_x));
        SRPS_TRY(grid_gradient(ctx, G.d_x, ctx->zx, ctx->zy));                           // SRPS.cu:310-311
        ctx->grad_current = true;
    }
    ctx->light_cache_normals = true;
    return launch_normals(ctx->stream, ctx->z, ctx->zx, ctx->zy, ctx->xx, ctx->yy, G.P, ctx->fx, ctx->fy, ctx->Nrm, ctx->dz);  // SRPS.cu:315
}

int srps_exchange(srps_ctx* ctx, const char* which, void** d_ptr, size_t* n_floats) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(which && d_ptr && n_floats, SRPS_ERR_INVALID, "exchange: null argument");
    Grid& G = ctx->grid;
    if (!strcmp(which, "s")) { *d_ptr = ctx->s; *n_floats = (size_t)ctx->N_total * ctx->C * 4; ctx->ssum_valid = false; }
    else if (!strcmp(which, "albedo")) { *d_ptr = ctx->albedo_ex; *n_floats = (size_t)ctx->C * G.P; }      // num [C][P]; den behind it is complete on every rank
    else if (!strcmp(which, "depth")) {
        if (ctx->q_ex) { *d_ptr = ctx->q_ex; *n_floats = 3 * (size_t)G.P; }       // compact: 3 P floats
        else { *d_ptr = G.d_q; *n_floats = 3 * G.plane; }                          // one GPU: nothing to exchange, the grid planes themselves
    }
    else if (!strcmp(which, "energy")) { *d_ptr = ctx->energy_ex + 1; *n_floats = 1; }
    else SRPS_REQUIRE(false, SRPS_ERR_INVALID, "exchange: unknown buffer '%s'", which);
    return SRPS_OK;
}

// SRPS.cu:272-335 on one GPU
int srps_execute(srps_ctx* ctx, int max_outer, float* energies, int* n_outer) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    const float TOLERANCE = 5e-3f;         // SRPS.cu:85
    const int MAX_ITERATIONS = 10;         // SRPS.cu:86
    float last_error = NAN;                // SRPS.cu:273
    int iteration = 1, done = 0;
    bool stop = false;
    do {
        float error = 0.f;
        SRPS_TRY(srps_lighting(ctx));      // SRPS.cu:281
        SRPS_TRY(srps_albedo(ctx));        // SRPS.cu:287
        SRPS_TRY(srps_depth_partial(ctx)); // SRPS.cu:293
        SRPS_TRY(srps_depth_solve(ctx));
        SRPS_TRY(srps_energy_partial(ctx));
        SRPS_TRY(srps_normals(ctx));       // SRPS.cu:310-315, enqueued before the host waits for the energy
        SRPS_TRY(energy_finish_impl(ctx, &error, false));
        const float rel_err = fabsf(last_error - error) / fabsf(error);          // SRPS.cu:298
        if (error > last_error || rel_err < TOLERANCE || iteration > MAX_ITERATIONS) stop = true;   // SRPS.cu:299
        last_error = error;
        if (energies && done < (max_outer > 0 ? max_outer : 12)) energies[done] = error;
        ++iteration; ++done;
        if (max_outer > 0 && done >= max_outer) stop = true;
    } while (!stop);
    if (n_outer) *n_outer = done;
    return wait_for_stream(ctx);
}

// Option "overlap_exchange": the sweep that feeds an all-reduce (the albedo sweep over the images -> num; the depth assembly from
// the image sums -> q) is cut into pixel ranges, and the all-reduce of a range runs on a second stream while the next range is
// computed.  Same kernels, same sums, same bits; what changes is when the bytes travel.
static int sharded_albedo_partial_overlapped(srps_ctx* ctx) {
    PhaseSpan span(ctx, SRPS_PHASE_ALBEDO_SWEEP);
    Grid& G = ctx->grid;
    const int P = G.P, C = ctx->C, K = 4;
    float* ssum = nullptr;
    ctx->ssum_valid = false;
    if (ctx->assemble_from_sums && ctx->N_local > 0) {
        SRPS_TRY(ensure(ctx->ws_ssum, (size_t)3 * C * P * sizeof(float)));
        ssum = (float*)ctx->ws_ssum.p;
    }
    SRPS_TRY(setup_events(ctx, K + 1));
    hipStream_t cs = ctx->gather_stream;
    const int step = std::max(1024, cdiv(cdiv(P, K), 1024) * 1024);
    for (int q0 = 0, k = 0; q0 < P; q0 += step, ++k) {
        const int q1 = std::min(P, q0 + step);
        SRPS_TRY(albedo_numden(ctx, ctx->s, ctx->Nrm, ctx->I, P, ctx->N_local, C, ctx->img_offset, ctx->albedo_ex, ctx->fx, ctx->fy, ssum, ctx->N_total, q0, q1));
        SRPS_HIP(hipEventRecord(ctx->ev_copied[k], ctx->stream));
        SRPS_HIP(hipStreamWaitEvent(cs, ctx->ev_copied[k], 0));
        float* piece[8]; size_t n[8];
        for (int c = 0; c < C; ++c) { piece[c] = ctx->albedo_ex + (size_t)c * P + q0; n[c] = (size_t)(q1 - q0); }
        SRPS_TRY(comm_all_reduce_pieces_on(ctx, cs, piece, n, C));
    }
    SRPS_HIP(hipEventRecord(ctx->ev_gathered[0], cs));
    SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[0], 0));
    ctx->ssum_valid = ssum != nullptr;
    return SRPS_OK;
}
static int sharded_depth_partial_overlapped(srps_ctx* ctx) {
    PhaseSpan span(ctx, SRPS_PHASE_DEPTH_ASSEMBLY);
    Grid& G = ctx->grid;
    const int P = G.P, K = 4;
    const float* ssum = (ctx->assemble_from_sums && ctx->ssum_valid) ? (const float*)ctx->ws_ssum.p : nullptr;
    if (!ssum || !ctx->q_ex) {                             // nothing to cut into ranges (the assembly streams the images, or q stays on the grid)
        SRPS_TRY(srps_depth_partial(ctx));
        if (ctx->q_ex) SRPS_TRY(comm_all_reduce_sum(ctx, ctx->q_ex, 3 * (size_t)P));
        return SRPS_OK;
    }
    ctx->q_in_exchange = true;
    SRPS_TRY(setup_events(ctx, K + 1));
    hipStream_t cs = ctx->gather_stream;
    const int step = std::max(1024, cdiv(cdiv(P, K), 1024) * 1024);
    for (int q0 = 0, k = 0; q0 < P; q0 += step, ++k) {
        const int q1 = std::min(P, q0 + step);
        SRPS_TRY(depth_assemble(ctx, ctx->s, ctx->rho, ctx->I, ctx->xx, ctx->yy, ctx->dz, ctx->fx, ctx->fy, P, ctx->N_local, ctx->C, ctx->N_total, ctx->img_offset,
                                ctx->cx, ctx->cy, ssum, ctx->q_ex, q0, q1));
        SRPS_HIP(hipEventRecord(ctx->ev_copied[k], ctx->stream));
        SRPS_HIP(hipStreamWaitEvent(cs, ctx->ev_copied[k], 0));
        float* piece[3]; size_t n[3];
        for (int t = 0; t < 3; ++t) { piece[t] = ctx->q_ex + (size_t)t * P + q0; n[t] = (size_t)(q1 - q0); }
        SRPS_TRY(comm_all_reduce_pieces_on(ctx, cs, piece, n, 3));
    }
    SRPS_HIP(hipEventRecord(ctx->ev_gathered[0], cs));
    SRPS_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_gathered[0], 0));
    return SRPS_OK;
}

// SRPS.cu:272-335 on a context that holds a shard of the images and an RCCL communicator
// Do the ranks' image ranges [img_offset, img_offset + N_local) tile [0, N_total) -- every image on exactly one rank?  Two ranks that
// both hold all images would silently double s, num and q (round-3 advisor finding).  Every rank marks its images with 1 in a
// vector of N_total floats, the vectors are all-reduced, and every entry must come back as exactly 1; the first entry that does not
// names the image.  One tiny collective per solve, before the first pass.
__global__ void k_mark_images(float* __restrict__ cover, int n_total, int lo, int n_local) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += gridDim.x * blockDim.x) cover[i] = (i >= lo && i < lo + n_local) ? 1.f : 0.f;
}
static int sharded_ranges_tile(srps_ctx* ctx) {
    const int n = ctx->N_total;
    SRPS_TRY(ensure(ctx->ws_misc, (size_t)n * sizeof(float)));
    float* cover = (float*)ctx->ws_misc.p;
    hipLaunchKernelGGL(k_mark_images, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, cover, n, ctx->img_offset, ctx->N_local);
    SRPS_LAUNCH_CHECK();
    SRPS_TRY(comm_all_reduce_sum(ctx, cover, (size_t)n));
    std::vector<float> h((size_t)n);
    SRPS_TRY(host_download(ctx, h.data(), cover, (size_t)n * sizeof(float), ctx->stream));
    for (int i = 0; i < n; ++i)
        SRPS_REQUIRE(h[i] == 1.f, SRPS_ERR_INVALID,
                     "execute_sharded: image %d of %d is held by %d ranks (this rank %d of %d holds [%d, %d)): the ranks' image ranges must tile the image set",
                     i, n, (int)h[i], ctx->comm_rank, ctx->comm_world, ctx->img_offset, ctx->img_offset + ctx->N_local);
    return SRPS_OK;
}

int srps_execute_sharded(srps_ctx* ctx, int max_outer, float* energies, int* n_outer) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(comm_bound(ctx), SRPS_ERR_STATE, "execute_sharded: no communicator bound to the context (srps_comm_init_rank / srps_comm_init_all / srps_set_comm)");
    if (ctx->shard_range_check) SRPS_TRY(sharded_ranges_tile(ctx));
    if (ctx->comm != nullptr && ctx->comm_world > 1) {
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true) && !getenv("SRPS_QUIET"))
            fprintf(stderr, "srps: note: the RCCL paths with more than one rank (all-reduce of s / num / q, strip exchange, srps --gpus N) were developed on "
                            "one-GPU boxes; tests/test_multi_gpu.py exercises them wherever two devices exist -- run it on this node before relying on them "
                            "(SRPS_QUIET=1 silences this note)\n");
    }
    const float TOLERANCE = 5e-3f;         // SRPS.cu:85
    const int MAX_ITERATIONS = 10;         // SRPS.cu:86
    float last_error = NAN;                // SRPS.cu:273
    int iteration = 1, done = 0;
    bool stop = false;
    Grid& G = ctx->grid;
    struct Defer {                         // the phases leave the abort check to the end of the pass, where the ranks decide together
        srps_ctx* c;
        ~Defer() { c->defer_shard_checks = false; }
    } defer{ctx};
    ctx->defer_shard_checks = true;
    do {
        float error = 0.f;
        SRPS_TRY(srps_lighting_local(ctx));                                                     // SRPS.cu:281
        SRPS_TRY(comm_all_reduce_sum(ctx, ctx->s, (size_t)ctx->N_total * ctx->C * 4));
        if (ctx->overlap_exchange) {
            SRPS_TRY(sharded_albedo_partial_overlapped(ctx));                                   // SRPS.cu:287, the all-reduce of num under the sweep
            SRPS_TRY(srps_albedo_finish(ctx));
            SRPS_TRY(sharded_depth_partial_overlapped(ctx));                                    // SRPS.cu:293, the all-reduce of q under the assembly
        } else {
            SRPS_TRY(srps_albedo_partial(ctx));                                                 // SRPS.cu:287
            SRPS_TRY(comm_all_reduce_sum(ctx, ctx->albedo_ex, (size_t)ctx->C * G.P));
            SRPS_TRY(srps_albedo_finish(ctx));
            SRPS_TRY(srps_depth_partial(ctx));                                                  // SRPS.cu:293
            if (ctx->q_ex) SRPS_TRY(comm_all_reduce_sum(ctx, ctx->q_ex, 3 * (size_t)G.P));    // a context that holds all images has q on the grid already
        }
        SRPS_TRY(srps_depth_solve(ctx));
        SRPS_TRY(srps_energy_partial(ctx));
        SRPS_TRY(sharded_energy_exchange(ctx));
        SRPS_TRY(srps_normals(ctx));                                                            // SRPS.cu:310-315
        SRPS_TRY(energy_finish_impl(ctx, &error, false));
        const float rel_err = fabsf(last_error - error) / fabsf(error);                         // SRPS.cu:298
        if (error > last_error || rel_err < TOLERANCE || iteration > MAX_ITERATIONS) stop = true;   // SRPS.cu:299
        last_error = error;
        if (energies && done < (max_outer > 0 ? max_outer : 12)) energies[done] = error;
        ++iteration; ++done;
        if (max_outer > 0 && done >= max_outer) stop = true;
    } while (!stop);
    if (n_outer) *n_outer = done;
    return wait_for_stream(ctx);
}

static int lookup(srps_ctx* ctx, const char* name, float** p, size_t* n) {
    Grid& G = ctx->grid;
    const size_t P = G.P;
    if (!strcmp(name, "z")) { *p = ctx->z; *n = P; }
    else if (!strcmp(name, "rho")) { *p = ctx->rho; *n = P * ctx->C; }
    else if (!strcmp(name, "s")) { *p = ctx->s; *n = (size_t)ctx->N_total * ctx->C * 4; }
    else if (!strcmp(name, "N")) { *p = ctx->Nrm; *n = 4 * P; }
    else if (!strcmp(name, "dz")) { *p = ctx->dz; *n = P; }
    else if (!strcmp(name, "zx")) { *p = ctx->zx; *n = P; }
    else if (!strcmp(name, "zy")) { *p = ctx->zy; *n = P; }
    else if (!strcmp(name, "xx")) { *p = ctx->xx; *n = P; }
    else if (!strcmp(name, "yy")) { *p = ctx->yy; *n = P; }
    else if (!strcmp(name, "z0s")) { *p = ctx->z0s; *n = G.Ps; }
    else if (!strcmp(name, "I")) { *p = ctx->I; *n = (size_t)ctx->N_local * ctx->C * P; }
    else SRPS_REQUIRE(false, SRPS_ERR_INVALID, "unknown state array '%s'", name);
    return SRPS_OK;
}

int srps_get(srps_ctx* ctx, const char* name, float* host, size_t n) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(name && host, SRPS_ERR_INVALID, "get: null argument");
    float* p; size_t len;
    SRPS_TRY(lookup(ctx, name, &p, &len));
    SRPS_REQUIRE(n == len, SRPS_ERR_INVALID, "get('%s'): buffer holds %zu floats, array has %zu", name, n, len);
    SRPS_TRY(host_download(ctx, host, p, len * sizeof(float), ctx->stream));
    return SRPS_OK;
}
int srps_set(srps_ctx* ctx, const char* name, const float* host, size_t n) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(name && host, SRPS_ERR_INVALID, "set: null argument");
    float* p; size_t len;
    SRPS_TRY(lookup(ctx, name, &p, &len));
    SRPS_REQUIRE(n == len, SRPS_ERR_INVALID, "set('%s'): buffer holds %zu floats, array has %zu", name, n, len);
    ctx->light_cache_valid = false;
    ctx->ssum_valid = false;
    ctx->grad_current = false;
    ctx->depth_assembled = false;
    ctx->normals_pending = false;
    if (p == ctx->I) ctx->i8_state = 0;
    if (p == ctx->Nrm) ctx->n3_one = false;             // the caller's N3 is taken as it is from here on
    SRPS_TRY(host_upload(ctx, p, host, len * sizeof(float), ctx->stream));
    return SRPS_OK;
}
int srps_array_size(srps_ctx* ctx, const char* name, size_t* n_floats) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(name && n_floats, SRPS_ERR_INVALID, "array_size: null argument");
    float* p;
    return lookup(ctx, name, &p, n_floats);
}
int srps_get_device_ptr(srps_ctx* ctx, const char* name, void** d_ptr, size_t* n_floats) {
    CTX_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(name && d_ptr && n_floats, SRPS_ERR_INVALID, "get_device_ptr: null argument");
    float* p; size_t len;
    SRPS_TRY(lookup(ctx, name, &p, &len));
    *d_ptr = p; *n_floats = len;
    if (p == ctx->I && ctx->i8_state != 2) {
        // the caller may write the images through the pointer at any later time: the byte copy cannot follow that
        ctx->i8_state = 2;
    }
    // "N" / "dz": with fuse_normals the energy sweep stores the next normals into a SECOND set of arrays and srps_normals swaps the
    // sets -- a pointer handed out would point at the non-current set from the next pass on (round-3 advisor finding).  From here on
    // this context keeps ONE set (the normals kernel writes it in place, as before round 3) until the next srps_setup.
    if (p == ctx->Nrm || p == ctx->dz) ctx->nd_ptr_out = true;
    if (p == ctx->Nrm) ctx->n3_one = false;
    ctx->light_cache_valid = false;      // the caller may write through the pointer
    ctx->depth_assembled = false;
    ctx->normals_pending = false;
    ctx->ssum_valid = false;
    ctx->grad_current = false;
    return SRPS_OK;
}
int srps_last_cg_iterations(srps_ctx* ctx, int* depth_iters, int* albedo_iters, int* lighting_iters_max) {
    CTX_CHECK(ctx);
    if (ctx->albedo_iters_pending > 0 || ctx->report_pending) {
        SRPS_TRY(report_fetch(ctx));
        SRPS_HIP(hipStreamSynchronize(ctx->stream));
        albedo_iters_collect(ctx);
        ctx->last_depth_iters = ((CgScalars*)(ctx->h_pinned + 64))->iters;
        ctx->last_light_iters = *(int*)(ctx->h_pinned + 8);
    }
    if (depth_iters) *depth_iters = ctx->last_depth_iters;
    if (albedo_iters) for (int c = 0; c < 8; ++c) albedo_iters[c] = ctx->last_albedo_iters[c];
    if (lighting_iters_max) *lighting_iters_max = ctx->last_light_iters;
    return SRPS_OK;
}

// ---- tracing --------------------------------------------------------------------------------
const char* srps_phase_name(int phase) { return (phase >= 0 && phase < SRPS_N_PHASES) ? kPhaseNames[phase] + 5 : ""; }

int srps_get_timings(srps_ctx* ctx, float* ms) {
    CTX_CHECK(ctx);
    SRPS_REQUIRE(ms != nullptr, SRPS_ERR_INVALID, "get_timings: ms is NULL");
    SRPS_REQUIRE(ctx->phase_timing, SRPS_ERR_STATE, "get_timings: option phase_timing is off");
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < SRPS_N_PHASES; ++i) {
        ms[i] = -1.f;
        if (ctx->ev_mask & (1u << i)) SRPS_HIP(hipEventElapsedTime(&ms[i], ctx->ev_begin[i], ctx->ev_end[i]));
    }
    ctx->ev_mask = 0;
    return SRPS_OK;
}

// ---- measurement ----------------------------------------------------------------------------
int srps_cg_bytes(srps_ctx* ctx, double* apply_bytes, double* update_bytes) {
    CTX_CHECK(ctx); GRID_CHECK(ctx);
    // algorithmic bytes per masked unknown: operator kernel reads M (24),
    // r (4), p (4), structure byte (1), writes p (4), w (4); update kernel reads x,r,p,w (16), writes x,r (8)
    const double P = (double)ctx->grid.P;
    const int nc = use_march(ctx) ? march_recompute_channels(ctx) : 0;
    // tensor-recompute form: nc planes of (rho_c/dz)^2 replace the 6 planes of M
    // marching operator also carries the deferred x update (x read + write, +8); its update kernel then
    // reads r, w and writes r (12) instead of reading x, r, p, w and writing x, r (24)
    const bool fused = use_march(ctx);
    // one-launch step: the operator launch also reads omega of the previous step and reads + writes r (+8), no update kernel
    const bool one = cg_fused_step(ctx);
    // ... and with the two-step x update (march_x2_on) x is read and written every second launch only and p of two steps ago read once: -2 on average
    const double x2 = march_x2_on(ctx) ? -2.0 : 0.0;
    if (apply_bytes) *apply_bytes = ((nc > 0 ? 17.0 + 4.0 * nc : 41.0) + (fused ? 8.0 : 0.0) + (one ? 8.0 : 0.0) + x2) * P;
    if (update_bytes) *update_bytes = one ? 0.0 : (fused ? 12.0 : 24.0) * P;
    return SRPS_OK;
}

int srps_bench_cg(srps_ctx* ctx, int solves, int iters, double* seconds, double* apply_us, double* update_us) {
    CTX_CHECK(ctx); GRID_CHECK(ctx); STATE_CHECK(ctx);
    SRPS_REQUIRE(ctx->tensor_valid, SRPS_ERR_STATE, "bench_cg: run a depth phase first (no system assembled)");
    SRPS_REQUIRE(solves > 0 && iters > 0, SRPS_ERR_INVALID, "bench_cg: solves and iters must be positive");
    Grid& G = ctx->grid;
    hipStream_t st = ctx->stream;
    SRPS_TRY(grid_scatter(ctx, ctx->z, G.d_x));
    SRPS_HIP(hipMemcpyAsync(G.d_save, G.d_x, G.plane * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipEvent_t e0, e1;
    SRPS_HIP(hipEventCreate(&e0)); SRPS_HIP(hipEventCreate(&e1));
    double total_ms = 0.0;
    for (int sidx = 0; sidx < solves; ++sidx) {
        SRPS_HIP(hipMemcpyAsync(G.d_x, G.d_save, G.plane * sizeof(float), hipMemcpyDeviceToDevice, st));
        SRPS_TRY(grid_rhs(ctx, ctx->z0s));
        SRPS_HIP(hipEventRecord(e0, st));
        SRPS_TRY(grid_cg(ctx, iters, true));              // residual + `iters` steps
        SRPS_HIP(hipEventRecord(e1, st));
        SRPS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SRPS_HIP(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
    }
    if (seconds) *seconds = total_ms * 1e-3;
    // per-kernel durations: one more solve with an event after every launch
    if (apply_us || update_us) {
        std::vector<hipEvent_t> ev(2 * (size_t)iters + 1);
        for (auto& e : ev) SRPS_HIP(hipEventCreate(&e));
        SRPS_HIP(hipMemcpyAsync(G.d_x, G.d_save, G.plane * sizeof(float), hipMemcpyDeviceToDevice, st));
        SRPS_TRY(grid_rhs(ctx, ctx->z0s));
        SRPS_TRY(grid_residual(ctx));
        ctx->cg_fixed = true;
        SRPS_HIP(hipEventRecord(ev[0], st));
        for (int k = 1; k <= iters; ++k) {
            cg_launch_apply(ctx, k);
            SRPS_HIP(hipEventRecord(ev[2 * k - 1], st));
            cg_launch_update(ctx, k);
            SRPS_HIP(hipEventRecord(ev[2 * k], st));
        }
        ctx->cg_fixed = false;
        SRPS_TRY(cg_flush_x(ctx));
        SRPS_HIP(hipEventSynchronize(ev[2 * (size_t)iters]));
        double a = 0, u = 0;
        for (int k = 1; k <= iters; ++k) {
            float ms = 0.f;
            SRPS_HIP(hipEventElapsedTime(&ms, ev[2 * k - 2], ev[2 * k - 1])); a += ms;
            SRPS_HIP(hipEventElapsedTime(&ms, ev[2 * k - 1], ev[2 * k])); u += ms;
        }
        if (apply_us) *apply_us = a * 1e3 / iters;
        if (update_us) *update_us = u * 1e3 / iters;
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    SRPS_HIP(hipMemcpyAsync(G.d_x, G.d_save, G.plane * sizeof(float), hipMemcpyDeviceToDevice, st));
    SRPS_HIP(hipStreamSynchronize(st));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    int aborted = 0;
    ctx->persistent_inflight |= ABORT_DEPTH;
    SRPS_TRY(persistent_sync_check(ctx, &aborted));
    SRPS_REQUIRE(!aborted, SRPS_ERR_HIP, "bench_cg: %s", srps_last_error());      // a timing of an aborted launch is worthless
    return SRPS_OK;
}

}  // extern "C"
