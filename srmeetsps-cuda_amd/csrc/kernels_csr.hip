// kernels_csr.hip -- generic CSR operators of the reference's operator API (devicecalls.cuh:26,37
// and devicecalls.cu:229-279) for callers that bring their own sparse matrices: COO->CSR upload,
// y = A x / y = A^T x, and the reference's CG on an arbitrary CSR matrix.  The SRPS pipeline itself
// never assembles a matrix (see kernels_cg.hip); these exist so that every cuda_based_* entry of
// the reference has a counterpart, and as an independent cross-check of the matrix-free path.
#include <algorithm>
#include <numeric>
#include "srps_internal.h"
#include "device_utils.h"

namespace srps {

// rows here are short (2..20 entries for Dx, Dy, KT, A_), so one thread per row
__global__ void k_csr_spmv(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ v,
                           int n_rows, const float* __restrict__ x, float* __restrict__ y) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int t = rp[r]; t < rp[r + 1]; ++t) acc = fmaf(v[t], x[ci[t]], acc);
        y[r] = acc;
    }
}
// y += A^T x by scatter; float atomics => summation order (last bits) may vary run to run,
// like cusparseScsrmv(TRANSPOSE) in the reference
__global__ void k_csr_spmv_t(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ v,
                             int n_rows, const float* __restrict__ x, float* __restrict__ y) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        const float xr = x[r];
        for (int t = rp[r]; t < rp[r + 1]; ++t) atomicAdd(&y[ci[t]], v[t] * xr);
    }
}

int csr_spmv(srps_ctx* ctx, const int* rp, const int* ci, const float* v, int n_rows, int n_cols, int nnz,
             const float* x, int transpose, float* y) {
    (void)nnz;
    const int nb = std::max(1, std::min(cdiv(n_rows, 256), 4096));
    if (!transpose) {
        hipLaunchKernelGGL(k_csr_spmv, dim3(nb), dim3(256), 0, ctx->stream, rp, ci, v, n_rows, x, y);
    } else {
        SRPS_HIP(hipMemsetAsync(y, 0, (size_t)n_cols * sizeof(float), ctx->stream));
        hipLaunchKernelGGL(k_csr_spmv_t, dim3(nb), dim3(256), 0, ctx->stream, rp, ci, v, n_rows, x, y);
    }
    SRPS_LAUNCH_CHECK();
    return SRPS_OK;
}

// ---- do caller-built Dx / Dy / KT (the CSR arguments of cuda_based_depth_estimation, devicecalls.cuh:36) describe the bound mask? ----
// One thread per row.  A gradient row of pixel p must be empty (isolated pixel), or hold the pair (+1 at the forward
// neighbour, -1 at p) / (+1 at p, -1 at the backward neighbour) that make_gradient writes (SRPS.cu:29-47), in either
// order; a KT row must hold the sf*sf pixels of its block with value 1/sf^2 (SRPS.cu:176-190).  err: bit 0 Dx, 1 Dy, 2 KT.
__global__ void k_check_gradient_csr(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ v, int P,
                                     const int* __restrict__ gofp, const uint8_t* __restrict__ flags, int step /* Hs for Dx, 1 for Dy */,
                                     unsigned fwd_bit, unsigned bwd_bit, int err_bit, int* __restrict__ err) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int g = gofp[p];
        const unsigned f = flags[g];
        const int t0 = rp[p], n = rp[p + 1] - t0;
        bool ok;
        if (!(f & (fwd_bit | bwd_bit))) ok = (n == 0);
        else {
            const bool fwd = (f & fwd_bit) != 0u;
            const int g_other = fwd ? g + step : g - step;
            const float v_self = fwd ? -1.f : 1.f;
            ok = (n == 2);
            bool seen_self = false, seen_other = false;
            for (int t = t0; ok && t < t0 + 2; ++t) {
                const int c = ci[t];
                if (c < 0 || c >= P) { ok = false; break; }
                if (c == p) { ok = (v[t] == v_self) && !seen_self; seen_self = true; }
                else { ok = (gofp[c] == g_other) && (v[t] == -v_self) && !seen_other; seen_other = true; }
            }
            ok = ok && seen_self && seen_other;
        }
        if (!ok) atomicOr(err, err_bit);
    }
}
__global__ void k_check_kt_csr(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ v, int Ps, int P,
                               const int* __restrict__ gofp, const int* __restrict__ lr_index, int Hs, int Hl, int sf, int* __restrict__ err) {
    const float val = 1.0f / (float)(sf * sf);                 // SRPS.cu:188
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < Ps; b += gridDim.x * blockDim.x) {
        const int t0 = rp[b], n = rp[b + 1] - t0;
        bool ok = (n == sf * sf);
        unsigned long long seen = 0ull;                        // one bit per pixel of the block (sf <= 8)
        for (int t = t0; ok && t < t0 + n; ++t) {
            const int c = ci[t];
            if (c < 0 || c >= P || v[t] != val) { ok = false; break; }
            const int g = gofp[c];
            const int gj = g / Hs - PAD, gi = g - (gj + PAD) * Hs - PAD;
            if (lr_index[(gj / sf) * Hl + gi / sf] != b) { ok = false; break; }
            const unsigned long long bit = 1ull << ((gj % sf) * sf + gi % sf);
            ok = !(seen & bit);
            seen |= bit;
        }
        if (!ok) atomicOr(err, 4);
    }
}

// returns the error bits through *h_err (synchronises)
int csr_matches_grid(srps_ctx* ctx, const int* dx_rp, const int* dx_ci, const float* dx_v, const int* dy_rp, const int* dy_ci, const float* dy_v,
                     const int* kt_rp, const int* kt_ci, const float* kt_v, int* h_err) {
    Grid& G = ctx->grid;
    SRPS_TRY(ensure(ctx->ws_misc, 64));
    int* d_err = (int*)ctx->ws_misc.p;
    SRPS_HIP(hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream));
    const int nb = std::max(1, std::min(cdiv(G.P, 256), 4096));
    hipLaunchKernelGGL(k_check_gradient_csr, dim3(nb), dim3(256), 0, ctx->stream, dx_rp, dx_ci, dx_v, G.P, G.d_gofp, G.d_flags, G.Hs, (unsigned)F_FX, (unsigned)F_BX, 1, d_err);
    hipLaunchKernelGGL(k_check_gradient_csr, dim3(nb), dim3(256), 0, ctx->stream, dy_rp, dy_ci, dy_v, G.P, G.d_gofp, G.d_flags, 1, (unsigned)F_FY, (unsigned)F_BY, 2, d_err);
    if (G.Ps > 0)
        hipLaunchKernelGGL(k_check_kt_csr, dim3(std::max(1, std::min(cdiv(G.Ps, 256), 4096))), dim3(256), 0, ctx->stream, kt_rp, kt_ci, kt_v, G.Ps, G.P,
                           G.d_gofp, G.d_lr_index, G.Hs, G.Hl, G.sf, d_err);
    SRPS_LAUNCH_CHECK();
    SRPS_HIP(hipMemcpyAsync(h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    return SRPS_OK;
}

// ---- CG on a CSR matrix, device-resident scalars, same step structure as kernels_cg.hip ------
struct CsrCgScal {
    float r0;
    int iters;
    int active;
    int pad;
};

__global__ __launch_bounds__(256) void k_ccg_init(const float* __restrict__ b, int n, float* __restrict__ rr_part, int nb, CsrCgScal* scal) {
    __shared__ float sm[16];
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) acc = fmaf(b[i], b[i], acc);
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_part[blockIdx.x] = t;
        if (blockIdx.x == 0) { scal->r0 = 0.f; scal->iters = 0; scal->active = 1; }
    }
}
// p = beta p + b  (dc.cu:256-265)
__global__ __launch_bounds__(256) void k_ccg_p(int k, const float* __restrict__ b, float* __restrict__ p, int n,
                                               const float* __restrict__ rr_old, int nb, const CsrCgScal* scal, float tol2) {
    __shared__ double smd[4];
    const float r1 = (float)sum_partials(rr_old, nb, smd);
    if (!(r1 > tol2)) return;
    const float beta = (k == 1) ? 0.f : r1 / scal->r0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        if (k == 1) p[i] = b[i];
        else { const float t = beta * p[i]; p[i] = t + b[i]; }
    }
}
// omega = A p ; partial p.omega  (dc.cu:267-268)
__global__ __launch_bounds__(256) void k_ccg_spmv(const int* __restrict__ rp, const int* __restrict__ ci, const float* __restrict__ v,
                                                  int n, const float* __restrict__ p, float* __restrict__ w,
                                                  const float* __restrict__ rr_old, int nb, float* __restrict__ pw_part, float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const float r1 = (float)sum_partials(rr_old, nb, smd);
    if (!(r1 > tol2)) return;
    float acc = 0.f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += nb * 256) {
        float a = 0.f;
        for (int t = rp[r]; t < rp[r + 1]; ++t) a = fmaf(v[t], p[ci[t]], a);
        w[r] = a;
        acc = fmaf(p[r], a, acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) pw_part[blockIdx.x] = t;
}
// x += alpha p ; b -= alpha omega ; partial b.b  (dc.cu:269-274)
__global__ __launch_bounds__(256) void k_ccg_update(int k, float* __restrict__ x, float* __restrict__ b, const float* __restrict__ p,
                                                    const float* __restrict__ w, int n, const float* __restrict__ rr_old,
                                                    float* __restrict__ rr_new, const float* __restrict__ pw_part, int nb,
                                                    CsrCgScal* scal, float tol2) {
    __shared__ float sm[16];
    __shared__ double smd[4];
    const float r1 = (float)sum_partials(rr_old, nb, smd);
    if (!(r1 > tol2)) {
        if (threadIdx.x == 0) { rr_new[blockIdx.x] = rr_old[blockIdx.x]; if (blockIdx.x == 0) scal->active = 0; }
        return;
    }
    const float dot = (float)sum_partials(pw_part, nb, smd);
    const float alpha = r1 / dot;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) {
        x[i] = fmaf(alpha, p[i], x[i]);
        const float bv = fmaf(-alpha, w[i], b[i]);
        b[i] = bv;
        acc = fmaf(bv, bv, acc);
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        rr_new[blockIdx.x] = t;
        if (blockIdx.x == 0) { scal->r0 = r1; scal->iters = k; scal->active = 1; }
    }
}

int csr_cg(srps_ctx* ctx, const int* rp, const int* ci, const float* v, int n, int nnz, float* x, float* b, int* iters) {
    (void)nnz;
    const int nb = std::max(1, std::min(cdiv(n, 256), 512));
    const size_t bytes = (2 * (size_t)n + 3 * (size_t)nb) * sizeof(float) + sizeof(CsrCgScal) + 64;
    SRPS_TRY(ensure(ctx->ws_albedo, bytes));
    float* p = (float*)ctx->ws_albedo.p;
    float* w = p + n;
    float* rr = w + n;              // [2][nb]
    float* pw = rr + 2 * (size_t)nb;
    CsrCgScal* scal = (CsrCgScal*)(pw + nb);
    const float tol2 = ctx->cg_tol * ctx->cg_tol;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_ccg_init, dim3(nb), dim3(256), 0, st, b, n, rr, nb, scal);
    CsrCgScal* hs = (CsrCgScal*)(ctx->h_pinned + 48);
    const int kmax = ctx->cg_max_iter + 1;
    for (int k = 1; k <= kmax; ++k) {
        const float* rr_old = rr + (size_t)((k - 1) & 1) * nb;
        float* rr_new = rr + (size_t)(k & 1) * nb;
        hipLaunchKernelGGL(k_ccg_p, dim3(nb), dim3(256), 0, st, k, b, p, n, rr_old, nb, scal, tol2);
        hipLaunchKernelGGL(k_ccg_spmv, dim3(nb), dim3(256), 0, st, rp, ci, v, n, p, w, rr_old, nb, pw, tol2);
        hipLaunchKernelGGL(k_ccg_update, dim3(nb), dim3(256), 0, st, k, x, b, p, w, n, rr_old, rr_new, pw, nb, scal, tol2);
        if ((k % 16) == 0 || k == kmax) {
            SRPS_LAUNCH_CHECK();
            SRPS_HIP(hipMemcpyAsync(hs, scal, sizeof(CsrCgScal), hipMemcpyDeviceToHost, st));
            SRPS_HIP(hipStreamSynchronize(st));
            if (!(hs->active != 0 && hs->iters == k)) break;
        }
    }
    SRPS_HIP(hipMemcpyAsync(hs, scal, sizeof(CsrCgScal), hipMemcpyDeviceToHost, st));
    SRPS_HIP(hipStreamSynchronize(st));
    if (iters) *iters = hs->iters;
    return SRPS_OK;
}

}  // namespace srps

using namespace srps;

extern "C" {

int srps_host_COO_to_device_CSR(srps_ctx* ctx, const int* row, const int* col, const float* val, int n_row, int n_col,
                                int nnz, int* d_row_ptr, int* d_col_ind, float* d_val) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "null context");
    SRPS_HIP(hipSetDevice(ctx->device));
    SRPS_REQUIRE(row && col && val && d_row_ptr && d_col_ind && d_val && n_row > 0 && n_col > 0 && nnz >= 0, SRPS_ERR_INVALID, "host_COO_to_device_CSR: bad arguments");
    // stable counting sort by row (sort_COO + coo2csr, devicecalls.cu:4-21, 60-62), on the host: set-up only
    std::vector<int> rp(n_row + 1, 0);
    for (int t = 0; t < nnz; ++t) {
        SRPS_REQUIRE(row[t] >= 0 && row[t] < n_row && col[t] >= 0 && col[t] < n_col, SRPS_ERR_INVALID, "host_COO_to_device_CSR: entry %d (%d,%d) out of range", t, row[t], col[t]);
        rp[row[t] + 1]++;
    }
    std::partial_sum(rp.begin(), rp.end(), rp.begin());
    std::vector<int> cur(rp.begin(), rp.end() - 1), ci(std::max(nnz, 1));
    std::vector<float> vv(std::max(nnz, 1));
    for (int t = 0; t < nnz; ++t) {
        const int d = cur[row[t]]++;
        ci[d] = col[t]; vv[d] = val[t];
    }
    SRPS_TRY(host_upload(ctx, d_row_ptr, rp.data(), (size_t)(n_row + 1) * sizeof(int), ctx->stream));
    if (nnz > 0) {
        SRPS_TRY(host_upload(ctx, d_col_ind, ci.data(), (size_t)nnz * sizeof(int), ctx->stream));
        SRPS_TRY(host_upload(ctx, d_val, vv.data(), (size_t)nnz * sizeof(float), ctx->stream));
    }
    SRPS_HIP(hipStreamSynchronize(ctx->stream));
    return SRPS_OK;
}

int srps_sparsemat_densevec_mul(srps_ctx* ctx, const int* d_row_ptr, const int* d_col_ind, const float* d_val, int n_rows,
                                int n_cols, int nnz, const float* d_x, int transpose, float* d_y) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "null context");
    SRPS_HIP(hipSetDevice(ctx->device));
    SRPS_REQUIRE(d_row_ptr && d_col_ind && d_val && d_x && d_y && n_rows > 0 && n_cols > 0, SRPS_ERR_INVALID, "sparsemat_densevec_mul: bad arguments");
    return csr_spmv(ctx, d_row_ptr, d_col_ind, d_val, n_rows, n_cols, nnz, d_x, transpose, d_y);
}

int srps_conjugate_gradient(srps_ctx* ctx, const int* d_row_ptr, const int* d_col_ind, const float* d_val, int n, int nnz,
                            float* d_x, float* d_b, int* iters) {
    SRPS_REQUIRE(ctx != nullptr, SRPS_ERR_INVALID, "null context");
    SRPS_HIP(hipSetDevice(ctx->device));
    SRPS_REQUIRE(d_row_ptr && d_col_ind && d_val && d_x && d_b && n > 0, SRPS_ERR_INVALID, "conjugate_gradient: bad arguments");
    return csr_cg(ctx, d_row_ptr, d_col_ind, d_val, n, nnz, d_x, d_b, iters);
}

}  // extern "C"
