/*
 * srps_oracle.c -- plain-C (+OpenMP) restatement of the SRPS depth step: the per-pixel
 * photometric tensor, the CG system in BOTH formulations (assembled CSR as the reference builds
 * it, and matrix-free), and the reference's CG recurrence.
 *
 * TEST INFRASTRUCTURE ONLY: this is the checker and the CPU baseline ("port") of bench.py.
 * Nothing in the product path links or calls it.  PARITY UNPINNED: the reference ships no golden
 * vectors and cannot be built here (see oracle/srps_oracle.py header); this file is pinned
 * against the numpy restatement (tests/test_c_oracle.py), which follows the reference line by line.
 *
 * Reference lines followed (paths under /root/reference/SRmeetsPS-GPU/):
 *   structure : make_gradient SRPS.cu:23-71, KT SRPS.cu:170-193, D Utilities.cpp:201-220
 *   tensor    : calculate_A_ch_1_2 / _3 devicecalls.cu:583-599, compute_B_for_depth 550-556
 *   system    : A_ = KT'KT + lambda A'A, rhs = KT'z0s + lambda A'B   devicecalls.cu:734-745
 *   CG        : cuda_based_conjugate_gradient devicecalls.cu:229-279
 *   energy    : devicecalls.cu:762-785
 *
 * Layout: compact masked vectors in ascending column-major HR index; I[n][c][p]; s[n][c][4].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- structure ------------------------------------------------------------------------------
 * nb[0..3][P]: compact index of the right (j+1), left (j-1), lower (i+1), upper (i-1) neighbour,
 * -1 when outside the image or the mask.  Dx row p is forward when nb[0][p] >= 0, else backward
 * when nb[1][p] >= 0, else empty (SRPS.cu:39-46); Dy likewise with nb[2], nb[3] (SRPS.cu:31-38).
 * blk[p]: compact LR index of p's sf x sf block when the block is fully masked (a row of KT), else -1.
 * blk_pix[Ps][sf*sf]: the compact HR pixels of every such block. */
int oc_count(int h, int w, int sf, const float* mask, int* P_out, int* Ps_out) {
    int P = 0, Ps = 0;
    for (long t = 0; t < (long)h * w; ++t) P += mask[t] != 0.f;
    for (int bj = 0; bj < w / sf; ++bj)
        for (int bi = 0; bi < h / sf; ++bi) {
            int full = 1;
            for (int dj = 0; dj < sf && full; ++dj)
                for (int di = 0; di < sf; ++di)
                    if (mask[(long)(bj * sf + dj) * h + bi * sf + di] == 0.f) { full = 0; break; }
            Ps += full;
        }
    *P_out = P; *Ps_out = Ps;
    return 0;
}

int oc_structure(int h, int w, int sf, const float* mask, int* imask, int* nb, int* blk, int* blk_pix) {
    long hw = (long)h * w;
    int* idx = (int*)malloc(hw * sizeof(int));
    if (!idx) return 1;
    int P = 0;
    for (long t = 0; t < hw; ++t) { idx[t] = mask[t] != 0.f ? P : -1; if (mask[t] != 0.f) imask[P++] = (int)t; }
    for (int p = 0; p < P; ++p) {
        int lin = imask[p], j = lin / h, i = lin - j * h;
        nb[0 * (long)P + p] = (j + 1 < w) ? idx[lin + h] : -1;
        nb[1 * (long)P + p] = (j - 1 >= 0) ? idx[lin - h] : -1;
        nb[2 * (long)P + p] = (i + 1 < h) ? idx[lin + 1] : -1;
        nb[3 * (long)P + p] = (i - 1 >= 0) ? idx[lin - 1] : -1;
        blk[p] = -1;
    }
    int Ps = 0, per = sf * sf;
    for (int bj = 0; bj < w / sf; ++bj)            /* LR pixels in column-major order: Utilities.cpp:209-216 */
        for (int bi = 0; bi < h / sf; ++bi) {
            int full = 1;
            for (int dj = 0; dj < sf && full; ++dj)
                for (int di = 0; di < sf; ++di)
                    if (idx[(long)(bj * sf + dj) * h + bi * sf + di] < 0) { full = 0; break; }
            if (!full) continue;
            for (int dj = 0; dj < sf; ++dj)          /* column order of D's row: j*h + k, Utilities.cpp:216 */
                for (int di = 0; di < sf; ++di) {
                    int p = idx[(long)(bj * sf + dj) * h + bi * sf + di];
                    blk[p] = Ps;
                    blk_pix[(long)Ps * per + dj * sf + di] = p;
                }
            ++Ps;
        }
    free(idx);
    return 0;
}

/* ---- tensor -------------------------------------------------------------------------------- */
void oc_tensor(int P, int N, int C, const float* s, const float* rho, const float* dz, const float* xx,
               const float* yy, float fx, float fy, const float* I, float* M, float* q) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) {
        float m[6] = {0, 0, 0, 0, 0, 0}, qq[3] = {0, 0, 0};
        for (int c = 0; c < C; ++c) {
            const float r = rho[(long)c * P + p];
            const float g = r / dz[p];
            for (int i = 0; i < N; ++i) {
                const float* sv = s + ((long)i * C + c) * 4;
                const float a1 = g * (fx * sv[0] - xx[p] * sv[2]);     /* devicecalls.cu:588 (launch 616) */
                const float a2 = g * (fy * sv[1] - yy[p] * sv[2]);     /* devicecalls.cu:588 (launch 617) */
                const float a3 = g * sv[2];                             /* devicecalls.cu:597 */
                const float b = I[((long)i * C + c) * P + p] - r * sv[3];   /* devicecalls.cu:554, N3 == 1 */
                const float v0 = a1, v1 = a2, v2 = -a3;
                m[0] += v0 * v0; m[1] += v0 * v1; m[2] += v0 * v2; m[3] += v1 * v1; m[4] += v1 * v2; m[5] += v2 * v2;
                qq[0] += v0 * b; qq[1] += v1 * b; qq[2] += v2 * b;
            }
        }
        for (int t = 0; t < 6; ++t) M[(long)t * P + p] = m[t];
        for (int t = 0; t < 3; ++t) q[(long)t * P + p] = qq[t];
    }
}

static inline void grad_p(const int* nb, long P, int p, const float* x, float* gx, float* gy) {
    const int r = nb[p], l = nb[P + p], d = nb[2 * P + p], u = nb[3 * P + p];
    *gx = r >= 0 ? x[r] - x[p] : (l >= 0 ? x[p] - x[l] : 0.f);
    *gy = d >= 0 ? x[d] - x[p] : (u >= 0 ? x[p] - x[u] : 0.f);
}

void oc_gradient(int P, const int* nb, const float* x, float* gx, float* gy) {
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) grad_p(nb, P, p, x, gx + p, gy + p);
}

/* ---- matrix-free operator: y = KT'(KT x) + lambda (Dx'u + Dy'v + w) ------------------------ */
void oc_mf_apply(int P, int Ps, int sf, const int* nb, const int* blk, const int* blk_pix, const float* M,
                 float lambda, const float* x, float* y, float* work /* 3P + Ps */) {
    float* u = work; float* v = work + P; float* wv = work + 2L * P; float* ks = work + 3L * P;
    const long LP = P;
    const int per = sf * sf;
    const float inv = 1.0f / (float)per;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) {
        float gx, gy;
        grad_p(nb, LP, p, x, &gx, &gy);
        u[p] = M[p] * gx + M[LP + p] * gy + M[2 * LP + p] * x[p];
        v[p] = M[LP + p] * gx + M[3 * LP + p] * gy + M[4 * LP + p] * x[p];
        wv[p] = M[2 * LP + p] * gx + M[4 * LP + p] * gy + M[5 * LP + p] * x[p];
    }
#pragma omp parallel for schedule(static)
    for (int b = 0; b < Ps; ++b) {
        float a = 0.f;
        for (int t = 0; t < per; ++t) a += inv * x[blk_pix[(long)b * per + t]];
        ks[b] = a;
    }
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) {
        const int r = nb[p], l = nb[LP + p], d = nb[2 * LP + p], up = nb[3 * LP + p];
        float acc = wv[p];
        if (r >= 0) acc -= u[p]; else if (l >= 0) acc += u[p];
        if (l >= 0) acc += u[l];                                   /* left neighbour is forward (its right = p) */
        if (r >= 0 && nb[r] < 0) acc -= u[r];                      /* right neighbour is backward */
        if (d >= 0) acc -= v[p]; else if (up >= 0) acc += v[p];
        if (up >= 0) acc += v[up];
        if (d >= 0 && nb[2 * LP + d] < 0) acc -= v[d];
        acc *= lambda;
        if (blk[p] >= 0) acc += inv * ks[blk[p]];
        y[p] = acc;
    }
}

/* rhs = KT' z0s + lambda (Dx'q0 + Dy'q1 + q2) */
void oc_rhs(int P, int sf, const int* nb, const int* blk, const float* q, const float* z0s, float lambda, float* rhs) {
    const long LP = P;
    const float inv = 1.0f / (float)(sf * sf);
    const float *q0 = q, *q1 = q + LP, *q2 = q + 2 * LP;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < P; ++p) {
        const int r = nb[p], l = nb[LP + p], d = nb[2 * LP + p], up = nb[3 * LP + p];
        float acc = q2[p];
        if (r >= 0) acc -= q0[p]; else if (l >= 0) acc += q0[p];
        if (l >= 0) acc += q0[l];
        if (r >= 0 && nb[r] < 0) acc -= q0[r];
        if (d >= 0) acc -= q1[p]; else if (up >= 0) acc += q1[p];
        if (up >= 0) acc += q1[up];
        if (d >= 0 && nb[2 * LP + d] < 0) acc -= q1[d];
        acc *= lambda;
        if (blk[p] >= 0) acc += inv * z0s[blk[p]];
        rhs[p] = acc;
    }
}

/* ---- assembled A_ = KT'KT + lambda A'A as CSR (what devicecalls.cu:734-736 produces) -------
 * Row r gathers, from every pixel p whose stencil rows (dx_p, dy_p, e_p) touch r, the product
 * (S_p[:,r])' M_p S_p ; at most 5 such p, each with at most 5 columns.  Returns nnz (<= cap) or -1. */
typedef struct { int col[5]; float c[3][5]; int n; } stencil_t;   /* S_p: 3 sparse rows over <= 5 columns */

static inline void stencil_of(const int* nb, long P, int p, stencil_t* S) {
    const int r = nb[p], l = nb[P + p], d = nb[2 * P + p], u = nb[3 * P + p];
    S->n = 0;
    memset(S->c, 0, sizeof(S->c));
    S->col[S->n] = p; int ip = S->n++;
    S->c[2][ip] = 1.f;                                               /* e_p */
    if (r >= 0) { S->col[S->n] = r; S->c[0][S->n] = 1.f; S->c[0][ip] = -1.f; S->n++; }
    else if (l >= 0) { S->col[S->n] = l; S->c[0][S->n] = -1.f; S->c[0][ip] = 1.f; S->n++; }
    if (d >= 0) { S->col[S->n] = d; S->c[1][S->n] = 1.f; S->c[1][ip] = -1.f; S->n++; }
    else if (u >= 0) { S->col[S->n] = u; S->c[1][S->n] = -1.f; S->c[1][ip] = 1.f; S->n++; }
}

long oc_assemble(int P, int sf, const int* nb, const int* blk, const int* blk_pix, const float* M, float lambda,
                 int* rowptr, int* col, float* val, long cap) {
    const long LP = P;
    const int per = sf * sf;
    const float inv4 = (1.0f / (float)per) * (1.0f / (float)per);
    const int maxrow = 32 + per;
    /* pass 1: build rows into a padded table, pass 2: compact */
    int* cnt = (int*)calloc(P, sizeof(int));
    int* tcol = (int*)malloc((size_t)P * maxrow * sizeof(int));
    float* tval = (float*)malloc((size_t)P * maxrow * sizeof(float));
    if (!cnt || !tcol || !tval) { free(cnt); free(tcol); free(tval); return -1; }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < P; ++r) {
        int* rc = tcol + (size_t)r * maxrow; float* rv = tval + (size_t)r * maxrow; int n = 0;
        int cand[5] = {r, nb[r], nb[LP + r], nb[2 * LP + r], nb[3 * LP + r]};
        for (int k = 0; k < 5; ++k) {
            const int p = cand[k];
            if (p < 0) continue;
            stencil_t S; stencil_of(nb, LP, p, &S);
            int ir = -1;
            for (int t = 0; t < S.n; ++t) if (S.col[t] == r) ir = t;
            if (ir < 0) continue;
            const float Mp[3][3] = {{M[p], M[LP + p], M[2 * LP + p]}, {M[LP + p], M[3 * LP + p], M[4 * LP + p]}, {M[2 * LP + p], M[4 * LP + p], M[5 * LP + p]}};
            float tv[3];
            for (int b = 0; b < 3; ++b) tv[b] = S.c[0][ir] * Mp[0][b] + S.c[1][ir] * Mp[1][b] + S.c[2][ir] * Mp[2][b];
            for (int t = 0; t < S.n; ++t) {
                const float contrib = lambda * (tv[0] * S.c[0][t] + tv[1] * S.c[1][t] + tv[2] * S.c[2][t]);
                int f = -1;
                for (int e = 0; e < n; ++e) if (rc[e] == S.col[t]) { f = e; break; }
                if (f < 0) { f = n++; rc[f] = S.col[t]; rv[f] = 0.f; }
                rv[f] += contrib;
            }
        }
        if (blk[r] >= 0) {
            for (int t = 0; t < per; ++t) {
                const int c2 = blk_pix[(long)blk[r] * per + t];
                int f = -1;
                for (int e = 0; e < n; ++e) if (rc[e] == c2) { f = e; break; }
                if (f < 0) { f = n++; rc[f] = c2; rv[f] = 0.f; }
                rv[f] += inv4;
            }
        }
        /* sort the row by column (CSR from csrgeam is column-sorted) */
        for (int a = 1; a < n; ++a) { int cc = rc[a]; float vv = rv[a]; int b = a - 1; while (b >= 0 && rc[b] > cc) { rc[b + 1] = rc[b]; rv[b + 1] = rv[b]; --b; } rc[b + 1] = cc; rv[b + 1] = vv; }
        cnt[r] = n;
    }
    long nnz = 0;
    rowptr[0] = 0;
    for (int r = 0; r < P; ++r) { nnz += cnt[r]; rowptr[r + 1] = (int)nnz; }
    if (nnz > cap) { free(cnt); free(tcol); free(tval); return -1; }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < P; ++r) {
        memcpy(col + rowptr[r], tcol + (size_t)r * maxrow, cnt[r] * sizeof(int));
        memcpy(val + rowptr[r], tval + (size_t)r * maxrow, cnt[r] * sizeof(float));
    }
    free(cnt); free(tcol); free(tval);
    return nnz;
}

void oc_csr_spmv(int n, const int* rowptr, const int* col, const float* val, const float* x, float* y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        float a = 0.f;
        for (int t = rowptr[r]; t < rowptr[r + 1]; ++t) a += val[t] * x[col[t]];
        y[r] = a;
    }
}

/* cublasSdot (dc.cu:251, 268, 274).  The library's summation order is not specified; what a GPU BLAS does is a tree over blocks.
 * Restated so that the result does NOT depend on the number of host threads: float sums over fixed blocks of 256 elements (a
 * thread block's share), the block sums added in double in index order.  (One float accumulator per OpenMP thread, as this
 * function had it first, made the truncated 101-step solve -- and the energy after it -- move with the thread count: 38 560 on 8
 * threads, 38 522 on 64, 38 487 on 256 for the 1024 x 1024 scene whose energy is 38 487.29 on the GPU, and from run to run with
 * the order in which OpenMP combined the threads' sums.) */
static float sdot(int n, const float* a, const float* b) {
    enum { BLK = 256 };
    const int nblk = (n + BLK - 1) / BLK;
    double* part = (double*)malloc((size_t)(nblk > 0 ? nblk : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (int q = 0; q < nblk; ++q) {
        const int i0 = q * BLK, i1 = i0 + BLK < n ? i0 + BLK : n;
        float acc = 0.f;
        for (int i = i0; i < i1; ++i) acc += a[i] * b[i];
        part[q] = (double)acc;
    }
    double tot = 0.0;
    for (int q = 0; q < nblk; ++q) tot += part[q];
    free(part);
    return (float)tot;
}

/* cuda_based_conjugate_gradient, devicecalls.cu:229-279, statement by statement (BLAS-1 ops unfused).
 * mode 0: assembled CSR; mode 1: matrix-free.  fixed_iters > 0 runs exactly that many steps (bench).
 * x = warm start in/out, b = residual in / destroyed.  Returns the number of steps. */
typedef struct {
    int mode, n, Ps, sf;
    const int *rowptr, *col; const float* val;                       /* mode 0 */
    const int *nb, *blk, *blk_pix; const float* M; float lambda; float* work;   /* mode 1 */
} oc_op;

static void op_apply(const oc_op* A, const float* x, float* y) {
    if (A->mode == 0) oc_csr_spmv(A->n, A->rowptr, A->col, A->val, x, y);
    else oc_mf_apply(A->n, A->Ps, A->sf, A->nb, A->blk, A->blk_pix, A->M, A->lambda, x, y, A->work);
}

static int cg_run_ws(const oc_op* A, float* x, float* b, float tol, int max_iter, int fixed_iters, float* p, float* om) {
    const int n = A->n;
    float r0 = 0.f, r1 = sdot(n, b, b);                              /* dc.cu:251 */
    int k = 0;
    while (fixed_iters > 0 ? k < fixed_iters : (r1 > tol * tol && k <= max_iter)) {   /* dc.cu:252 */
        ++k;
        if (k == 1) {
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n; ++i) p[i] = b[i];                 /* Scopy dc.cu:258 */
        } else {
            const float beta = r1 / r0;                              /* dc.cu:262 */
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n; ++i) p[i] = beta * p[i];          /* Sscal dc.cu:263 */
#pragma omp parallel for schedule(static)
            for (int i = 0; i < n; ++i) p[i] = p[i] + b[i];          /* Saxpy dc.cu:264 */
        }
        op_apply(A, p, om);                                          /* csrmv dc.cu:267 */
        const float dot = sdot(n, p, om);                            /* dc.cu:268 */
        const float alpha = r1 / dot;                                /* dc.cu:269 */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) x[i] += alpha * p[i];            /* dc.cu:270 */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) b[i] -= alpha * om[i];           /* dc.cu:272 */
        r0 = r1;
        r1 = sdot(n, b, b);                                          /* dc.cu:274 */
    }
    return k;
}

static int cg_run(const oc_op* A, float* x, float* b, float tol, int max_iter, int fixed_iters) {
    const int n = A->n;
    float* p = (float*)malloc((size_t)n * sizeof(float));
    float* om = (float*)malloc((size_t)n * sizeof(float));
    const int k = cg_run_ws(A, x, b, tol, max_iter, fixed_iters, p, om);
    free(p); free(om);
    return k;
}

/* bench.py's cpu_baseline leg (oracle/cpu_baseline_main.py): `reps` timed solves of `iters` steps each of the reference's CG
 * (cg_run_ws above, dc.cu:229-279) on the assembled CSR system, after one untimed warm-up solve, on `threads` OpenMP threads.
 * What makes the figure repeat from run to run on a two-socket host: the matrix and every vector are COPIES made here and first
 * touched by the thread that will stream them (the static schedule of the loops above: a row's non-zeros, its x, b, p and omega
 * element live on the memory node of the thread that owns the row), the workspaces are allocated once (no page faults inside the
 * timed solves), and the caller binds the threads (OMP_PROC_BIND=spread, OMP_PLACES=cores in the environment of the process).
 * seconds[reps] receives the wall time of each solve. */
int oc_bench_cg_csr(int n, const int* rowptr, const int* col, const float* val, const float* b0, int iters, int reps, int threads, double* seconds) {
#ifdef _OPENMP
    const int before = omp_get_max_threads();
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const long nnz = rowptr[n];
    int* rp = (int*)malloc(((size_t)n + 1) * sizeof(int));
    int* ci = (int*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    float* va = (float*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    float *x = (float*)malloc((size_t)n * sizeof(float)), *b = (float*)malloc((size_t)n * sizeof(float));
    float *p = (float*)malloc((size_t)n * sizeof(float)), *om = (float*)malloc((size_t)n * sizeof(float));
    if (!rp || !ci || !va || !x || !b || !p || !om) { free(rp); free(ci); free(va); free(x); free(b); free(p); free(om); return -1; }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        rp[r] = rowptr[r]; if (r == n - 1) rp[n] = rowptr[n];
        for (int t = rowptr[r]; t < rowptr[r + 1]; ++t) { ci[t] = col[t]; va[t] = val[t]; }
        x[r] = 0.f; b[r] = b0[r]; p[r] = 0.f; om[r] = 0.f;
    }
    oc_op A; memset(&A, 0, sizeof(A));
    A.mode = 0; A.n = n; A.rowptr = rp; A.col = ci; A.val = va;
    for (int k = -2; k < reps; ++k) {                                 /* k < 0: two warm-up solves (round 5's first timed solve still ran at 0.6 x) */
#pragma omp parallel for schedule(static)
        for (int r = 0; r < n; ++r) { x[r] = 0.f; b[r] = b0[r]; }
#ifdef _OPENMP
        const double t0 = omp_get_wtime();
#endif
        cg_run_ws(&A, x, b, 0.f, 0, iters, p, om);
#ifdef _OPENMP
        if (k >= 0) seconds[k] = omp_get_wtime() - t0;
#else
        if (k >= 0) seconds[k] = 0.0;
#endif
    }
    free(rp); free(ci); free(va); free(x); free(b); free(p); free(om);
#ifdef _OPENMP
    omp_set_num_threads(before);
#endif
    return 0;
}

/* The same for the MATRIX-FREE form of the system (oc_mf_apply: 6 tensor planes + 4 neighbour indices + the block index per unknown
 * instead of 17.9 non-zeros with their column indices): BASELINE.md section 3's second CPU variant, the stronger baseline -- it moves
 * about a third of the bytes.  Same recurrence (cg_run_ws), same first-touch discipline, two warm-up solves. */
int oc_bench_cg_mf(int P, int Ps, int sf, const int* nb, const int* blk, const int* blk_pix, const float* M, float lambda, const float* b0,
                   int iters, int reps, int threads, double* seconds) {
#ifdef _OPENMP
    const int before = omp_get_max_threads();
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const long LP = P;
    const int per = sf * sf;
    int* nb2 = (int*)malloc((size_t)LP * 4 * sizeof(int));
    int* blk2 = (int*)malloc((size_t)LP * sizeof(int));
    int* bp2 = (int*)malloc((size_t)(Ps > 0 ? Ps : 1) * per * sizeof(int));
    float* M2 = (float*)malloc((size_t)LP * 6 * sizeof(float));
    float* work = (float*)malloc((3 * (size_t)LP + Ps + 8) * sizeof(float));
    float *x = (float*)malloc((size_t)LP * sizeof(float)), *b = (float*)malloc((size_t)LP * sizeof(float));
    float *p = (float*)malloc((size_t)LP * sizeof(float)), *om = (float*)malloc((size_t)LP * sizeof(float));
    if (!nb2 || !blk2 || !bp2 || !M2 || !work || !x || !b || !p || !om) { free(nb2); free(blk2); free(bp2); free(M2); free(work); free(x); free(b); free(p); free(om); return -1; }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < LP; ++r) {
        for (int t = 0; t < 4; ++t) nb2[t * LP + r] = nb[t * LP + r];
        for (int t = 0; t < 6; ++t) M2[t * LP + r] = M[t * LP + r];
        blk2[r] = blk[r];
        work[r] = work[LP + r] = work[2 * LP + r] = 0.f;
        x[r] = 0.f; b[r] = b0[r]; p[r] = 0.f; om[r] = 0.f;
    }
#pragma omp parallel for schedule(static)
    for (long q = 0; q < Ps; ++q) {
        for (int t = 0; t < per; ++t) bp2[q * per + t] = blk_pix[q * per + t];
        work[3 * LP + q] = 0.f;
    }
    oc_op A; memset(&A, 0, sizeof(A));
    A.mode = 1; A.n = P; A.Ps = Ps; A.sf = sf; A.nb = nb2; A.blk = blk2; A.blk_pix = bp2; A.M = M2; A.lambda = lambda; A.work = work;
    for (int k = -2; k < reps; ++k) {
#pragma omp parallel for schedule(static)
        for (long r = 0; r < LP; ++r) { x[r] = 0.f; b[r] = b0[r]; }
#ifdef _OPENMP
        const double t0 = omp_get_wtime();
#endif
        cg_run_ws(&A, x, b, 0.f, 0, iters, p, om);
#ifdef _OPENMP
        if (k >= 0) seconds[k] = omp_get_wtime() - t0;
#else
        if (k >= 0) seconds[k] = 0.0;
#endif
    }
    free(nb2); free(blk2); free(bp2); free(M2); free(work); free(x); free(b); free(p); free(om);
#ifdef _OPENMP
    omp_set_num_threads(before);
#endif
    return 0;
}

int oc_cg_csr(int n, const int* rowptr, const int* col, const float* val, float* x, float* b, float tol, int max_iter, int fixed_iters) {
    oc_op A; memset(&A, 0, sizeof(A));
    A.mode = 0; A.n = n; A.rowptr = rowptr; A.col = col; A.val = val;
    return cg_run(&A, x, b, tol, max_iter, fixed_iters);
}

int oc_cg_mf(int P, int Ps, int sf, const int* nb, const int* blk, const int* blk_pix, const float* M, float lambda,
             float* x, float* b, float tol, int max_iter, int fixed_iters) {
    oc_op A; memset(&A, 0, sizeof(A));
    A.mode = 1; A.n = P; A.Ps = Ps; A.sf = sf; A.nb = nb; A.blk = blk; A.blk_pix = blk_pix; A.M = M; A.lambda = lambda;
    A.work = (float*)malloc((3 * (size_t)P + Ps + 8) * sizeof(float));
    const int k = cg_run(&A, x, b, tol, max_iter, fixed_iters);
    free(A.work);
    return k;
}

/* energy = ||KT z - z0s||^2 + lambda sum_{c,i,p} (a1 zx + a2 zy - a3 z - b)^2, devicecalls.cu:762-785 */
double oc_energy(int P, int Ps, int N, int C, int sf, const int* nb, const int* blk_pix, const float* s,
                 const float* rho, const float* dz, const float* xx, const float* yy, float fx, float fy,
                 const float* I, const float* z0s, const float* z, float lambda) {
    const int per = sf * sf;
    const float inv = 1.0f / (float)per;
    double t1 = 0.0, t2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : t1)
    for (int b = 0; b < Ps; ++b) {
        float a = 0.f;
        for (int t = 0; t < per; ++t) a += inv * z[blk_pix[(long)b * per + t]];
        const float d = a - z0s[b];
        t1 += (double)(d * d);
    }
#pragma omp parallel for schedule(static) reduction(+ : t2)
    for (int p = 0; p < P; ++p) {
        float gx, gy;
        grad_p(nb, P, p, z, &gx, &gy);
        for (int c = 0; c < C; ++c) {
            const float r = rho[(long)c * P + p];
            const float g = r / dz[p];
            for (int i = 0; i < N; ++i) {
                const float* sv = s + ((long)i * C + c) * 4;
                const float a1 = g * (fx * sv[0] - xx[p] * sv[2]);
                const float a2 = g * (fy * sv[1] - yy[p] * sv[2]);
                const float a3 = g * sv[2];
                const float bb = I[((long)i * C + c) * P + p] - r * sv[3];
                const float res = a1 * gx + a2 * gy - a3 * z[p] - bb;
                t2 += (double)(res * res);
            }
        }
    }
    return t1 + (double)lambda * t2;
}
