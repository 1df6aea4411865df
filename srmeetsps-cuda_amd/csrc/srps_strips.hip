// srps_strips.hip -- the depth CG partitioned into column strips over several ranks: the one way CG iterations per second can
// grow with the number of GPUs (the image-sharded pass replicates the CG).  Recurrence kept: devicecalls.cu:252-275, in the
// one-launch-per-step form of kernels_march.hip (MODE 3).
//
// Every rank holds the whole grid (planes, structure, the assembled g_c and q: 1 GB at 4096 x 4096 -- nothing on a 288 GB device)
// and OWNS the bounding-box columns [c0, c1), cut at multiples of sf so that no sf x sf block of KT straddles two ranks
// (tests/_strip_protocol.py: strip_ranges; tests/test_strip_partition.py asserts on the assembled matrix that a row of A_ at
// an owned pixel references nothing further away than ONE column).  The marching kernel runs on a VIEW of the grid -- the
// planes' base pointers moved by c0 columns, width c1 - c0 -- so its halo columns are the neighbours' edge columns.  Per step:
//
//   launch k on the strip            p_k = beta p_{k-1} + r_{k-1} (halo columns included), omega_k = A_ p_k, x, r updates of step k-1
//   four sums -> 4 doubles           p.omega, r.omega, omega.omega, r.r of the strip (fixed order)
//   all-reduce of the 4 doubles      RCCL, 32 bytes: the same bits on every rank -> the same alpha, beta, stop decision
//   halo exchange                    the edge columns of p_k, r_{k-1}, omega_k to / from both neighbours: 3 x Hs floats each way
//                                    (48 KB at 4096 rows), ncclSend / ncclRecv in one group, xGMI point to point
//
// and after the last step the strips of x are gathered on every rank (one broadcast per rank: 67 MB in total at 4096 x 4096).
// Comm budget per step: one 32-byte all-reduce + one neighbour exchange of 48 KB each way -- latency, not bandwidth: ~2 x 10 - 20 us
// over xGMI against 170 us / N of streaming per strip.
//
// Without multi-GPU hardware the same driver runs on SEVERAL CONTEXTS OF ONE PROCESS that share a device and a stream
// (srps_strip_group_solve): the collectives become device copies and a summing kernel, everything else -- views, halo columns,
// totals, the kernels -- is identical.  tests/test_gpu_strips.py holds 2, 3 and 4 such ranks against the single-grid CG.
#include "srps_internal.h"

namespace srps {

namespace {

struct Range { int c0, w; };
// tests/_strip_protocol.py: strip_ranges -- multiples of sf, sizes differ by at most one block column
Range strip_range(int Wg, int sf, int world, int rank) {
    const int blocks = Wg / sf, base = blocks / world, rem = blocks % world;
    const int b = rank * base + std::min(rank, rem), e = b + base + (rank < rem ? 1 : 0);
    return {b * sf, (e - b) * sf};
}

struct PtrTable { const double* local[16]; double* total[16]; };
__global__ void k_sum_totals_tab(int n, PtrTable t) {
    const int v = threadIdx.x;
    if (v >= 4) return;
    double s = 0.0;
    for (int q = 0; q < n; ++q) s += t.local[q][v];
    for (int r = 0; r < n; ++r) t.total[r][v] = s;
}

// the collectives of a group of ranks, two ways
struct Collectives {
    std::vector<srps_ctx*> L;          // the ranks this caller drives (loopback: all of them; RCCL / caller's transport: its own)
    bool loopback = false;
    int world = 1;
    // the caller's own transport (srps_set_strip_transport): host functions on device pointers, called with the stream drained
    bool hosted() const { return !loopback && L[0]->strip_allreduce != nullptr; }
    int rank_of(size_t li) const { return loopback ? (int)li : (hosted() ? L[0]->strip_rank : L[li]->comm_rank); }
    static int drained(srps_ctx* c) { SRPS_HIP(hipStreamSynchronize(c->stream)); return SRPS_OK; }

    // the sums a launch of parity `par` left on every rank ([0..3] of d_strip_tot) -> their total over the ranks, slot `par`
    // ([4 + 4 par ..]): launch k + 1 reads slot k & 1, and so does the final x update when step k was the last one executed
    int all_reduce_totals(int par) {
        if (loopback) {
            PtrTable t;
            for (int q = 0; q < world; ++q) { t.local[q] = L[q]->d_strip_tot; t.total[q] = L[q]->d_strip_tot + 4 + 4 * (par & 1); }
            hipLaunchKernelGGL(k_sum_totals_tab, dim3(1), dim3(64), 0, L[0]->stream, world, t);
            SRPS_LAUNCH_CHECK();
            return SRPS_OK;
        }
        if (hosted()) {
            SRPS_TRY(drained(L[0]));
            const int rc = L[0]->strip_allreduce(L[0]->strip_user, L[0]->d_strip_tot, L[0]->d_strip_tot + 4 + 4 * (par & 1));
            SRPS_REQUIRE(rc == 0, SRPS_ERR_HIP, "strip transport: the caller's all-reduce returned %d", rc);
            return SRPS_OK;
        }
        return comm_all_reduce_sum_f64(L[0], L[0]->d_strip_tot, L[0]->d_strip_tot + 4 + 4 * (par & 1), 4);
    }
    // the edge columns of the planes planes[.][b] (b < nbuf), both directions
    int exchange(int nbuf, float* const* const* planes /* [rank in L][nbuf] */) {
        for (size_t li = 0; li < L.size(); ++li) {
            srps_ctx* c = L[li];
            const Grid& G = c->grid;
            const int rank = rank_of(li);
            const size_t first = (size_t)(G.view_c0 + PAD) * G.Hs, last = (size_t)(G.view_c0 + G.view_w - 1 + PAD) * G.Hs, n = (size_t)G.Hs;
            if (loopback) {
                // my last column -> the right neighbour's plane (its left halo), its first column -> my right halo
                if (rank + 1 < world) {
                    srps_ctx* nb = L[li + 1];
                    for (int b = 0; b < nbuf; ++b) {
                        SRPS_HIP(hipMemcpyAsync(planes[li + 1][b] + last, planes[li][b] + last, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                        SRPS_HIP(hipMemcpyAsync(planes[li][b] + last + G.Hs, planes[li + 1][b] + last + G.Hs, n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                    }
                    (void)nb;
                }
                continue;
            }
            const float* sl[4]; float* rl[4]; const float* sr[4]; float* rr[4];
            for (int b = 0; b < nbuf; ++b) {
                sl[b] = planes[li][b] + first; rl[b] = planes[li][b] + first - G.Hs;      // my first column out, the column left of it in
                sr[b] = planes[li][b] + last; rr[b] = planes[li][b] + last + G.Hs;
            }
            if (hosted()) {
                SRPS_TRY(drained(c));
                const int rc = c->strip_exchange(c->strip_user, nbuf, rank > 0 ? sl : nullptr, rank > 0 ? rl : nullptr, rank + 1 < world ? sr : nullptr,
                                                 rank + 1 < world ? rr : nullptr, n);
                SRPS_REQUIRE(rc == 0, SRPS_ERR_HIP, "strip transport: the caller's exchange returned %d", rc);
                continue;
            }
            SRPS_TRY(comm_exchange(c, nbuf, sl, rl, rank > 0 ? rank - 1 : -1, sr, rr, rank + 1 < world ? rank + 1 : -1, n));
        }
        return SRPS_OK;
    }
    int all_gather_x() {
        if (loopback) {
            for (int q = 0; q < world; ++q) {
                const Grid& Gq = L[q]->grid;
                const size_t off = (size_t)(Gq.view_c0 + PAD) * Gq.Hs, n = (size_t)Gq.view_w * Gq.Hs;
                for (int r = 0; r < world; ++r)
                    if (r != q) SRPS_HIP(hipMemcpyAsync(L[r]->grid.d_x + off, Gq.d_x + off, n * sizeof(float), hipMemcpyDeviceToDevice, L[0]->stream));
            }
            return SRPS_OK;
        }
        srps_ctx* c = L[0];
        const Grid& G = c->grid;
        std::vector<size_t> off(world), cnt(world);
        for (int q = 0; q < world; ++q) {
            const Range rg = strip_range(G.Wg, G.sf, world, q);
            off[q] = (size_t)(rg.c0 + PAD) * G.Hs; cnt[q] = (size_t)rg.w * G.Hs;
        }
        if (hosted()) {
            SRPS_TRY(drained(c));
            const int rc = c->strip_allgather(c->strip_user, G.d_x, off.data(), cnt.data());
            SRPS_REQUIRE(rc == 0, SRPS_ERR_HIP, "strip transport: the caller's all-gather returned %d", rc);
            return SRPS_OK;
        }
        return comm_all_gather_pieces(c, G.d_x, off.data(), cnt.data());
    }
};

// the CG on the ranks of `co`: every rank's plane d_x holds the whole current iterate, d_r the whole right-hand side
int strips_run(Collectives& co, int max_steps, bool fixed_steps) {
    std::vector<srps_ctx*>& L = co.L;
    const int nl = (int)L.size();
    std::vector<float*> tab((size_t)nl * 3);
    std::vector<float* const*> rows(nl);
    auto planes = [&](int nbuf, auto pick) -> float* const* const* {
        for (int i = 0; i < nl; ++i) {
            for (int b = 0; b < nbuf; ++b) tab[(size_t)i * 3 + b] = pick(L[i]->grid, b);
            rows[i] = &tab[(size_t)i * 3];
        }
        return rows.data();
    };
    for (srps_ctx* c : L) {
        c->grid.d_totals4 = nullptr;
        SRPS_TRY(grid_residual(c));                                      // r = b - A_ x on the strip (devicecalls.cu:758); r.r partials in slot 3 of the sums [0]
        SRPS_TRY(march_part4_totals(c, 0, c->d_strip_tot));
    }
    SRPS_TRY(co.all_reduce_totals(0));
    SRPS_TRY(co.exchange(1, planes(1, [](Grid& G, int) { return G.d_r; })));      // step 1 reads p = r on its halo columns
    for (srps_ctx* c : L) { c->grid.d_totals4 = c->d_strip_tot + 4; c->cg_fixed = fixed_steps; }
    int rc = SRPS_OK;
    for (int k = 1; k <= max_steps && rc == SRPS_OK; ++k) {
        for (srps_ctx* c : L) {
            if ((rc = march_cg_step(c, k)) != SRPS_OK) break;
            if ((rc = march_part4_totals(c, k & 1, c->d_strip_tot)) != SRPS_OK) break;
        }
        if (rc == SRPS_OK) rc = co.all_reduce_totals(k);
        // what launch k + 1 reads on its halo columns: p_k, r_{k-1}, omega_k -- the planes launch k wrote
        if (rc == SRPS_OK)
            rc = co.exchange(3, planes(3, [k](Grid& G, int b) { return b == 0 ? G.d_p + (size_t)(k & 1) * G.plane : b == 1 ? ((k & 1) ? G.d_r2 : G.d_r) : ((k & 1) ? G.d_w2 : G.d_w); }));
    }
    for (srps_ctx* c : L) c->cg_fixed = false;
    SRPS_TRY(rc);
    for (srps_ctx* c : L) SRPS_TRY(cg_flush_x(c));                       // the x update still pending, on the strip
    SRPS_TRY(co.all_gather_x());
    for (srps_ctx* c : L) c->grid.d_totals4 = nullptr;
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

int ensure_strip_buffers(srps_ctx* ctx) {
    if (ctx->d_strip_tot) return SRPS_OK;
    SRPS_HIP(hipMalloc((void**)&ctx->d_strip_tot, 12 * sizeof(double)));
    SRPS_HIP(hipMemset(ctx->d_strip_tot, 0, 12 * sizeof(double)));
    return SRPS_OK;
}

}  // namespace

bool strips_active(const srps_ctx* ctx) {
    const bool transport = (ctx->comm != nullptr && ctx->comm_world > 1) || (ctx->strip_allreduce != nullptr && ctx->strip_world > 1);
    return ctx->cg_strips && transport && ctx->grid.bound && cg_fused_step(ctx) && !forced_failure("strips");
}

void strips_clear_view(srps_ctx* ctx) {
    Grid& G = ctx->grid;
    if (G.view_w == 0) return;
    G.view_c0 = 0; G.view_w = 0; G.d_totals4 = nullptr;
    if (G.bound) march_plan(G, ctx->march_tj, ctx->num_cus);
}

int strips_bind_view(srps_ctx* ctx, int rank, int world) {
    Grid& G = ctx->grid;
    SRPS_REQUIRE(G.bound, SRPS_ERR_STATE, "strip partition: no grid bound");
    SRPS_REQUIRE(world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "strip partition: rank %d of %d", rank, world);
    SRPS_REQUIRE(world <= 16, SRPS_ERR_UNSUPPORTED, "strip partition: at most 16 ranks");
    SRPS_REQUIRE(G.Wg / G.sf >= world, SRPS_ERR_INVALID, "strip partition: %d ranks for a grid of %d block columns", world, G.Wg / G.sf);
    SRPS_REQUIRE(march_supported(ctx) && ctx->cg_fused_step, SRPS_ERR_UNSUPPORTED, "strip partition: needs the one-launch streaming CG step (sf 1, 2 or 4, option cg_fused_step)");
    SRPS_TRY(ensure_strip_buffers(ctx));
    const Range rg = strip_range(G.Wg, G.sf, world, rank);
    G.view_c0 = rg.c0; G.view_w = rg.w;
    march_plan(G, ctx->march_tj, ctx->num_cus);
    SRPS_REQUIRE(march_blocks(G) + 8 <= G.n_part4, SRPS_ERR_UNSUPPORTED, "strip partition: partial-sum buffers too small for the strip's plan");
    return SRPS_OK;
}

int strips_cg(srps_ctx* ctx, int max_steps, bool fixed_steps) {
    Collectives co;
    const bool hosted = ctx->strip_allreduce != nullptr;
    co.L = {ctx}; co.loopback = false; co.world = hosted ? ctx->strip_world : ctx->comm_world;
    // the view lasts for the solve only: every other user of the marching kernels (srps_depth_operator_apply, ...) sees the whole grid
    SRPS_TRY(strips_bind_view(ctx, hosted ? ctx->strip_rank : ctx->comm_rank, co.world));
    const int rc = strips_run(co, max_steps, fixed_steps);
    strips_clear_view(ctx);
    return rc;
}

}  // namespace srps

using namespace srps;

extern "C" {

// n contexts of THIS process that share one device and one stream as the ranks of a strip-partitioned depth solve: every
// context must have gone through srps_depth_partial on the same problem (replicated state); afterwards each holds the new depth
// like after srps_depth_solve.  The collectives are device copies -- the arithmetic is the multi-GPU path's.
int srps_strip_group_solve(srps_ctx* const* ctxs, int n) {
    SRPS_REQUIRE(ctxs != nullptr && n >= 1 && n <= 16, SRPS_ERR_INVALID, "strip_group_solve: 1 .. 16 contexts");
    for (int i = 0; i < n; ++i) {
        SRPS_REQUIRE(ctxs[i] != nullptr && ctxs[i]->have_state && ctxs[i]->tensor_valid, SRPS_ERR_STATE, "strip_group_solve: context %d has no assembled depth system (srps_depth_partial)", i);
        SRPS_REQUIRE(ctxs[i]->device == ctxs[0]->device && ctxs[i]->stream == ctxs[0]->stream, SRPS_ERR_INVALID,
                     "strip_group_solve: the contexts must share one device and one stream (srps_set_stream)");
        SRPS_REQUIRE(ctxs[i]->grid.Hg == ctxs[0]->grid.Hg && ctxs[i]->grid.Wg == ctxs[0]->grid.Wg && ctxs[i]->grid.sf == ctxs[0]->grid.sf, SRPS_ERR_INVALID,
                     "strip_group_solve: the contexts hold different grids");
    }
    SRPS_HIP(hipSetDevice(ctxs[0]->device));
    Collectives co;
    co.loopback = true; co.world = n;
    for (int i = 0; i < n; ++i) {
        SRPS_TRY(strips_bind_view(ctxs[i], i, n));
        co.L.push_back(ctxs[i]);
    }
    int rc = SRPS_OK;
    for (int i = 0; i < n && rc == SRPS_OK; ++i) rc = depth_solve_prepare(ctxs[i]);
    if (rc == SRPS_OK) rc = strips_run(co, ctxs[0]->cg_max_iter + 1, false);
    for (int i = 0; i < n && rc == SRPS_OK; ++i) rc = depth_solve_finish(ctxs[i]);
    for (int i = 0; i < n; ++i) strips_clear_view(ctxs[i]);
    return rc;
}


// The same group with the RESIDENT kernel on every strip (kernels_resident.hip: resident_cg_group): the strips are ranges of
// 256 x 64 tile columns, every context launches the persistent CG kernel on its own tiles, on its OWN stream, and the launches
// exchange their sums and border edges through each other's memory while they run -- no host step, no collective between the
// 101 CG steps.  The contexts may share a device (all tiles together must then fit its CUs: the one-GPU test bed) or sit on peer
// devices of one process.  Results: those of the single resident launch, bit for bit (tests/test_gpu_strips.py).
// SRPS_ERR_UNSUPPORTED: the grid does not fit this form, or the launches could not be resident together (nothing stored: call
// srps_strip_group_solve, the streaming strips, instead).
int srps_strip_group_solve_resident(srps_ctx* const* ctxs, int n) {
    SRPS_REQUIRE(ctxs != nullptr && n >= 1 && n <= 8, SRPS_ERR_INVALID, "strip_group_solve_resident: 1 .. 8 contexts");
    for (int i = 0; i < n; ++i) {
        SRPS_REQUIRE(ctxs[i] != nullptr && ctxs[i]->have_state && ctxs[i]->tensor_valid, SRPS_ERR_STATE, "strip_group_solve_resident: context %d has no assembled depth system (srps_depth_partial)", i);
        strips_clear_view(ctxs[i]);
    }
    int rc = SRPS_OK;
    for (int i = 0; i < n && rc == SRPS_OK; ++i) { (void)hipSetDevice(ctxs[i]->device); rc = depth_solve_prepare(ctxs[i]); }
    if (rc == SRPS_OK) rc = resident_cg_group(ctxs, n, ctxs[0]->cg_max_iter + 1, false);
    for (int i = 0; i < n && rc == SRPS_OK; ++i) { (void)hipSetDevice(ctxs[i]->device); rc = depth_solve_finish(ctxs[i]); }
    return rc;
}

// The partitions themselves, as pure functions (no device): the columns [*c0, *c0 + *width) of a grid of `grid_cols` columns that rank
// `rank` of `world` owns (multiples of sf, sizes differ by at most one block column), and the images [*begin, *begin + *count)
// of a shard (contiguous, sizes differ by at most one) -- what srps_depth_solve / the C++ host / api.shard_range use.
int srps_strip_range(int grid_cols, int sf, int world, int rank, int* c0, int* width) {
    SRPS_REQUIRE(c0 && width && sf >= 1 && grid_cols > 0 && grid_cols % sf == 0 && world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "strip_range: bad arguments");
    const Range rg = strip_range(grid_cols, sf, world, rank);
    *c0 = rg.c0; *width = rg.w;
    return SRPS_OK;
}
int srps_shard_range(int n_images, int world, int rank, int* begin, int* count) {
    SRPS_REQUIRE(begin && count && n_images >= 0 && world >= 1 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "shard_range: bad arguments");
    const int base = n_images / world, rem = n_images % world;
    *begin = rank * base + std::min(rank, rem);
    *count = base + (rank < rem ? 1 : 0);
    return SRPS_OK;
}

// The caller's own transport for the strip-partitioned CG (instead of RCCL): three host functions on DEVICE pointers, called
// from srps_depth_solve with the context's stream drained; each must have completed its reads and writes when it returns.
int srps_set_strip_transport(srps_ctx* ctx, int rank, int world, srps_strip_allreduce_fn allreduce, srps_strip_exchange_fn exchange,
                             srps_strip_allgather_fn allgather, void* user) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "set_strip_transport: null context");
    if (!allreduce && !exchange && !allgather) { ctx->strip_allreduce = nullptr; ctx->strip_exchange = nullptr; ctx->strip_allgather = nullptr; ctx->strip_world = 1; ctx->strip_rank = 0; return SRPS_OK; }
    SRPS_REQUIRE(allreduce && exchange && allgather, SRPS_ERR_INVALID, "set_strip_transport: all three functions, or none");
    SRPS_REQUIRE(world >= 1 && world <= 16 && rank >= 0 && rank < world, SRPS_ERR_INVALID, "set_strip_transport: rank %d of %d", rank, world);
    ctx->strip_allreduce = allreduce; ctx->strip_exchange = exchange; ctx->strip_allgather = allgather; ctx->strip_user = user;
    ctx->strip_rank = rank; ctx->strip_world = world;
    return SRPS_OK;
}

}  // extern "C"
