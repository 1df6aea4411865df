/*
 * srps_solve_mf.c -- the WHOLE alternating pass (lighting -> albedo -> depth -> normals, SRPS.cu:276-315) restated matrix-free in
 * ONE arithmetic type, REAL, chosen at compile time.  The Makefile builds it twice into libsrps_oracle.so:
 *
 *     -DREAL=double -DPFX=oc64   the calibration reference: the reference's loop, recurrences, iteration caps and stop tests with
 *                                every value and every sum in fp64.  What the fp32 restatements (the assembled-CSR oracle of
 *                                srps_oracle.c, this file's own fp32 build, the HIP library) are measured AGAINST when the question
 *                                is "whose rounding is it" (tests/test_gpu_drift_calibration.py, tests/test_oracle_fp64.py);
 *     -DREAL=float  -DPFX=oc32   the same statements in fp32: a matrix-free fp32 solve on the CPU, i.e. the HIP library's
 *                                FORMULATION with another summation order -- separates "assembled versus matrix-free" from
 *                                "CPU versus GPU".
 *
 * TEST INFRASTRUCTURE ONLY (checker; see srps_oracle.py's header).  PARITY UNPINNED like the rest of oracle/: the reference ships
 * no vectors and cannot be built here.  Pinned by tests/test_oracle_fp64.py against an independent numpy fp64 statement of the
 * same lines and by the literal transcriptions of tests/test_oracle_known_answers.py.
 *
 * Reference lines followed (paths under /root/reference/SRmeetsPS-GPU/):
 *   lighting : cuda_based_lightning_estimation devicecalls.cu:408-444 (A = rho_c (.) [N0..N3] :376-383, ATA sgemm :422,
 *              ATb sgemv :423, ATb -= ATA s :424, CG on the 4 x 4 system :437)
 *   albedo   : cuda_based_albedo_estimation devicecalls.cu:513-548 (A = N s_c sgemm :507, the diagonal normal equations
 *              cuda_based_MA_Mb :395-406, CG :540)
 *   depth    : cuda_based_depth_estimation devicecalls.cu:636-786 (coefficients :583-599, b :550-556, system :734-745,
 *              residual + CG :758-759, energy :762-785)
 *   normals  : cuda_based_normal_init devicecalls.cu:171-223
 *   CG       : cuda_based_conjugate_gradient devicecalls.cu:229-279 (tol 1e-9f, squared; k <= 100 => at most 101 steps)
 *
 * Layout: compact masked vectors in ascending column-major HR index; the images stay what the caller has, float I[n][c][P] (1 GB at
 * 2048^2 x 20: not copied), every state array is REAL.  Structure arrays (nb, blk, blk_pix) are oc_structure's (srps_oracle.c).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#error "compile with -DREAL=double -DPFX=oc64 or -DREAL=float -DPFX=oc32"
#endif
#define CAT2(a, b) a##_##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(PFX, name)

typedef REAL real;

#define CG_TOL ((real)1e-9f)      /* devicecalls.cu:230: a float constant */
#define CG_MAX_ITER 100           /* devicecalls.cu:231 */

int FN(sizeof_real)(void) { return (int)sizeof(real); }

/* Two knobs for the tests, both at the reference's values unless a test says otherwise: the CG's step cap (the reference's 100, i.e.
 * 101 steps; a test that compares two fp64 statements of a truncated solve stops before rounding differences have been amplified --
 * tests/test_oracle_fp64.py) and the block length of the dot products (another summation order: how far does an fp64 solve move with
 * it?  That is the reference run's own uncertainty.) */
static int g_max_iter = CG_MAX_ITER;
static int g_dot_block = 256;
void FN(set_max_iter)(int k) { g_max_iter = k; }
void FN(set_dot_block)(int n) { g_dot_block = n > 0 ? n : 256; }

/* cublasSdot: sums over fixed blocks of 256 elements in REAL, block sums added in double in index order -- independent of the host
 * thread count (srps_oracle.c's sdot, same reason) */
static real rdot(long n, const real* a, const real* b) {
    const long BLK = g_dot_block;
    const long nblk = (n + BLK - 1) / BLK;
    double* part = (double*)malloc((size_t)(nblk > 0 ? nblk : 1) * sizeof(double));
#pragma omp parallel for schedule(static)
    for (long q = 0; q < nblk; ++q) {
        const long i0 = q * BLK, i1 = i0 + BLK < n ? i0 + BLK : n;
        real acc = 0;
        for (long i = i0; i < i1; ++i) acc += a[i] * b[i];
        part[q] = (double)acc;
    }
    double tot = 0.0;
    for (long q = 0; q < nblk; ++q) tot += part[q];
    free(part);
    return (real)tot;
}

/* ---- the generic CG of devicecalls.cu:229-279 on an operator given as a callback ---------------------------------------- */
typedef void (*apply_fn)(const void* ctx, const real* x, real* y);

static int cg(long n, apply_fn A, const void* ctx, real* x, real* b, real* p, real* om) {
    real r0 = 0, r1 = rdot(n, b, b);                                  /* :251 */
    int k = 0;
    while (r1 > CG_TOL * CG_TOL && k <= g_max_iter) {                /* :252 */
        ++k;
        if (k == 1) {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < n; ++i) p[i] = b[i];                 /* Scopy :258 */
        } else {
            const real beta = r1 / r0;                                /* :262 */
#pragma omp parallel for schedule(static)
            for (long i = 0; i < n; ++i) p[i] = beta * p[i];          /* Sscal :263 */
#pragma omp parallel for schedule(static)
            for (long i = 0; i < n; ++i) p[i] = p[i] + b[i];          /* Saxpy :264 */
        }
        A(ctx, p, om);                                                /* csrmv :267 */
        const real dot = rdot(n, p, om);                              /* :268 */
        const real alpha = r1 / dot;                                  /* :269 */
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) x[i] += alpha * p[i];            /* :270 */
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) b[i] -= alpha * om[i];           /* :272 */
        r0 = r1;
        r1 = rdot(n, b, b);                                           /* :274 */
    }
    return k;
}

/* ---- lighting, devicecalls.cu:408-444 ------------------------------------------------------------------------------------
 * s[n][c][4] in/out (warm start), iters[n*c] (may be NULL).  The 4 x 4 Gram matrix of a channel is formed once (the reference forms
 * it once per image: the same sgemm on the same data), the right-hand side per image. */
typedef struct { real m[16]; } mat4;
static void apply4(const void* ctx, const real* x, real* y) {
    const mat4* M = (const mat4*)ctx;
    for (int r = 0; r < 4; ++r) { real a = 0; for (int c = 0; c < 4; ++c) a += M->m[r * 4 + c] * x[c]; y[r] = a; }
}

int FN(lighting)(int P, int N, int C, const real* rho, const real* Nrm, const float* I, real* s, int* iters) {
    enum { BLK = 256 };
    const long nblk = ((long)P + BLK - 1) / BLK;
    double* part = (double*)malloc((size_t)nblk * 16 * sizeof(double));
    if (!part) return 1;
    for (int c = 0; c < C; ++c) {
        const real* r = rho + (long)c * P;
        /* ATA[k][l] = sum_p (rho N_k)(rho N_l), :422 */
#pragma omp parallel for schedule(static)
        for (long q = 0; q < nblk; ++q) {
            const long i0 = q * BLK, i1 = i0 + BLK < P ? i0 + BLK : P;
            real acc[16]; for (int t = 0; t < 16; ++t) acc[t] = 0;
            for (long p = i0; p < i1; ++p) {
                real a[4];
                for (int k = 0; k < 4; ++k) a[k] = r[p] * Nrm[(long)k * P + p];        /* A_for_lightning_estimation :381 */
                for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) acc[k * 4 + l] += a[k] * a[l];
            }
            for (int t = 0; t < 16; ++t) part[q * 16 + t] = (double)acc[t];
        }
        mat4 G;
        for (int t = 0; t < 16; ++t) { double tot = 0; for (long q = 0; q < nblk; ++q) tot += part[q * 16 + t]; G.m[t] = (real)tot; }
        for (int i = 0; i < N; ++i) {
            const float* img = I + ((long)i * C + c) * P;
#pragma omp parallel for schedule(static)
            for (long q = 0; q < nblk; ++q) {
                const long i0 = q * BLK, i1 = i0 + BLK < P ? i0 + BLK : P;
                real acc[4] = {0, 0, 0, 0};
                for (long p = i0; p < i1; ++p)
                    for (int k = 0; k < 4; ++k) acc[k] += (r[p] * Nrm[(long)k * P + p]) * (real)img[p];   /* sgemv :423 */
                for (int k = 0; k < 4; ++k) part[q * 16 + k] = (double)acc[k];
            }
            real b[4], x[4], pv[4], om[4];
            for (int k = 0; k < 4; ++k) { double tot = 0; for (long q = 0; q < nblk; ++q) tot += part[q * 16 + k]; b[k] = (real)tot; }
            real* sv = s + ((long)i * C + c) * 4;
            for (int k = 0; k < 4; ++k) { x[k] = sv[k]; }
            apply4(&G, x, om);
            for (int k = 0; k < 4; ++k) b[k] -= om[k];                                    /* sgemv :424 */
            const int it = cg(4, apply4, &G, x, b, pv, om);                                /* :437 */
            for (int k = 0; k < 4; ++k) sv[k] = x[k];
            if (iters) iters[i * C + c] = it;
        }
    }
    free(part);
    return 0;
}

/* ---- albedo, devicecalls.cu:513-548 --------------------------------------------------------------------------------------
 * per channel the stacked system [diag(A_i)] rho = [I_i] (fill_A_expansion :447-454), its normal equations diag(sum_i A_i^2) and
 * sum_i A_i I_i (cuda_based_MA_Mb :395-406), the reference's CG from the warm start (:540) */
typedef struct { const real* d; long n; } diag_t;
static void apply_diag(const void* ctx, const real* x, real* y) {
    const diag_t* D = (const diag_t*)ctx;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < D->n; ++i) y[i] = D->d[i] * x[i];
}

int FN(albedo)(int P, int N, int C, const real* s, const real* Nrm, const float* I, real* rho, int* iters) {
    real* den = (real*)malloc((size_t)P * sizeof(real));
    real* b = (real*)malloc((size_t)P * sizeof(real));
    real* pv = (real*)malloc((size_t)P * sizeof(real));
    real* om = (real*)malloc((size_t)P * sizeof(real));
    if (!den || !b || !pv || !om) { free(den); free(b); free(pv); free(om); return 1; }
    for (int c = 0; c < C; ++c) {
        real* x = rho + (long)c * P;
#pragma omp parallel for schedule(static)
        for (long p = 0; p < P; ++p) {
            real d = 0, nm = 0;
            for (int i = 0; i < N; ++i) {
                const real* sv = s + ((long)i * C + c) * 4;
                real a = 0;
                for (int k = 0; k < 4; ++k) a += Nrm[(long)k * P + p] * sv[k];             /* sgemm :507 */
                d += a * a;                                                                 /* csrgemm :398-401 (diagonal) */
                nm += a * (real)I[((long)i * C + c) * P + p];                               /* csrmv :404 */
            }
            den[p] = d;
            b[p] = nm - d * x[p];                                                           /* csrmv :405 */
        }
        diag_t D = {den, P};
        const int it = cg(P, apply_diag, &D, x, b, pv, om);
        if (iters) iters[c] = it;
    }
    free(den); free(b); free(pv); free(om);
    return 0;
}

/* ---- depth, devicecalls.cu:636-786 --------------------------------------------------------------------------------------- */
static inline void grad_p(const int* nb, long P, long p, const real* x, real* gx, real* gy) {
    const int r = nb[p], l = nb[P + p], d = nb[2 * P + p], u = nb[3 * P + p];
    *gx = r >= 0 ? x[r] - x[p] : (l >= 0 ? x[p] - x[l] : (real)0);       /* make_gradient SRPS.cu:39-46 */
    *gy = d >= 0 ? x[d] - x[p] : (u >= 0 ? x[p] - x[u] : (real)0);       /* SRPS.cu:31-38 */
}

typedef struct {
    long P; int Ps, sf; const int *nb, *blk, *blk_pix; const real* M; real lambda; real* work;   /* work: 3P + Ps */
} mf_t;

/* y = KT'(KT x) + lambda (Dx'u + Dy'v + w), (u, v, w) = M (Dx x, Dy x, x)          (A_ of :734-736 applied, never assembled) */
static void apply_mf(const void* ctx, const real* x, real* y) {
    const mf_t* A = (const mf_t*)ctx;
    const long P = A->P;
    const int per = A->sf * A->sf;
    const real inv = (real)1 / (real)per;
    real *u = A->work, *v = A->work + P, *wv = A->work + 2 * P, *ks = A->work + 3 * P;
    const real* M = A->M;
    const int* nb = A->nb;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        real gx, gy;
        grad_p(nb, P, p, x, &gx, &gy);
        u[p] = M[p] * gx + M[P + p] * gy + M[2 * P + p] * x[p];
        v[p] = M[P + p] * gx + M[3 * P + p] * gy + M[4 * P + p] * x[p];
        wv[p] = M[2 * P + p] * gx + M[4 * P + p] * gy + M[5 * P + p] * x[p];
    }
#pragma omp parallel for schedule(static)
    for (long b = 0; b < A->Ps; ++b) {
        real a = 0;
        for (int t = 0; t < per; ++t) a += inv * x[A->blk_pix[b * per + t]];
        ks[b] = a;
    }
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        const int r = nb[p], l = nb[P + p], d = nb[2 * P + p], up = nb[3 * P + p];
        real acc = wv[p];
        if (r >= 0) acc -= u[p]; else if (l >= 0) acc += u[p];
        if (l >= 0) acc += u[l];
        if (r >= 0 && nb[r] < 0) acc -= u[r];
        if (d >= 0) acc -= v[p]; else if (up >= 0) acc += v[p];
        if (up >= 0) acc += v[up];
        if (d >= 0 && nb[2 * P + d] < 0) acc -= v[d];
        acc *= A->lambda;
        if (A->blk[p] >= 0) acc += inv * ks[A->blk[p]];
        y[p] = acc;
    }
}

/* z in/out, energy and step count out */
int FN(depth)(int P_, int Ps, int N, int C, int sf, const int* nb, const int* blk, const int* blk_pix, const real* s, const real* rho,
              const real* dz, const real* xx, const real* yy, real fx, real fy, const float* I, const real* z0s, real lambda, real* z,
              double* energy_out, int* iters_out) {
    const long P = P_;
    const int per = sf * sf;
    const real inv = (real)1 / (real)per;
    real* M = (real*)malloc((size_t)P * 6 * sizeof(real));
    real* q = (real*)malloc((size_t)P * 3 * sizeof(real));
    real* b = (real*)malloc((size_t)P * sizeof(real));
    real* pv = (real*)malloc((size_t)P * sizeof(real));
    real* om = (real*)malloc((size_t)P * sizeof(real));
    real* work = (real*)malloc(((size_t)P * 3 + Ps + 8) * sizeof(real));
    if (!M || !q || !b || !pv || !om || !work) { free(M); free(q); free(b); free(pv); free(om); free(work); return 1; }
    /* M = sum_{c,i} v v', q = sum v b, v = (a1, a2, -a3)                       (:583-599, :550-556; the rows of A of :676-691) */
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        real m[6] = {0, 0, 0, 0, 0, 0}, qq[3] = {0, 0, 0};
        for (int c = 0; c < C; ++c) {
            const real r = rho[(long)c * P + p];
            const real g = r / dz[p];
            for (int i = 0; i < N; ++i) {
                const real* sv = s + ((long)i * C + c) * 4;
                const real a1 = g * (fx * sv[0] - xx[p] * sv[2]);
                const real a2 = g * (fy * sv[1] - yy[p] * sv[2]);
                const real a3 = g * sv[2];
                const real bb = (real)I[((long)i * C + c) * P + p] - r * sv[3];
                const real v0 = a1, v1 = a2, v2 = -a3;
                m[0] += v0 * v0; m[1] += v0 * v1; m[2] += v0 * v2; m[3] += v1 * v1; m[4] += v1 * v2; m[5] += v2 * v2;
                qq[0] += v0 * bb; qq[1] += v1 * bb; qq[2] += v2 * bb;
            }
        }
        for (int t = 0; t < 6; ++t) M[(long)t * P + p] = m[t];
        for (int t = 0; t < 3; ++t) q[(long)t * P + p] = qq[t];
    }
    /* rhs = KT' z0s + lambda (Dx'q0 + Dy'q1 + q2)                                                                   (:743-745) */
    const real *q0 = q, *q1 = q + P, *q2 = q + 2 * P;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        const int r = nb[p], l = nb[P + p], d = nb[2 * P + p], up = nb[3 * P + p];
        real acc = q2[p];
        if (r >= 0) acc -= q0[p]; else if (l >= 0) acc += q0[p];
        if (l >= 0) acc += q0[l];
        if (r >= 0 && nb[r] < 0) acc -= q0[r];
        if (d >= 0) acc -= q1[p]; else if (up >= 0) acc += q1[p];
        if (up >= 0) acc += q1[up];
        if (d >= 0 && nb[2 * P + d] < 0) acc -= q1[d];
        acc *= lambda;
        if (blk[p] >= 0) acc += inv * z0s[blk[p]];
        b[p] = acc;
    }
    mf_t A = {P, Ps, sf, nb, blk, blk_pix, M, lambda, work};
    apply_mf(&A, z, om);
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) b[p] -= om[p];                                                                      /* :758 */
    *iters_out = cg(P, apply_mf, &A, z, b, pv, om);                                                                  /* :759 */
    /* energy = ||KT z - z0s||^2 + lambda sum (a1 zx + a2 zy - a3 z - b)^2, summed in double                         (:762-785) */
    double t1 = 0.0, t2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : t1)
    for (long bq = 0; bq < Ps; ++bq) {
        real a = 0;
        for (int t = 0; t < per; ++t) a += inv * z[blk_pix[bq * per + t]];
        const real d = a - z0s[bq];
        t1 += (double)(d * d);
    }
#pragma omp parallel for schedule(static) reduction(+ : t2)
    for (long p = 0; p < P; ++p) {
        real gx, gy;
        grad_p(nb, P, p, z, &gx, &gy);
        double acc = 0.0;
        for (int c = 0; c < C; ++c) {
            const real r = rho[(long)c * P + p];
            const real g = r / dz[p];
            for (int i = 0; i < N; ++i) {
                const real* sv = s + ((long)i * C + c) * 4;
                const real a1 = g * (fx * sv[0] - xx[p] * sv[2]);
                const real a2 = g * (fy * sv[1] - yy[p] * sv[2]);
                const real a3 = g * sv[2];
                const real bb = (real)I[((long)i * C + c) * P + p] - r * sv[3];
                const real res = a1 * gx + a2 * gy - a3 * z[p] - bb;
                acc += (double)(res * res);
            }
        }
        t2 += acc;
    }
    *energy_out = t1 + (double)lambda * t2;
    free(M); free(q); free(b); free(pv); free(om); free(work);
    return 0;
}

/* ---- normals, devicecalls.cu:171-223 (zx = Dx z, zy = Dy z: SRPS.cu:264-265, 304-305) ---------------------------------- */
int FN(normals)(int P_, const int* nb, const real* z, const real* xx, const real* yy, real fx, real fy, real* Nrm, real* dz) {
    const long P = P_;
#pragma omp parallel for schedule(static)
    for (long p = 0; p < P; ++p) {
        real zx, zy;
        grad_p(nb, P, p, z, &zx, &zy);
        const real n0 = fx * zx, n1 = fy * zy;                            /* :204, :211 */
        const real n2 = -z[p] - xx[p] * zx - yy[p] * zy;                  /* :174 */
        real d = (real)sqrt((double)(n0 * n0 + n1 * n1 + n2 * n2));
        if (sizeof(real) == 4) d = (real)sqrtf((float)(n0 * n0 + n1 * n1 + n2 * n2));
        if (d < (real)1e-10f) d = (real)1e-10f;                           /* :182 */
        Nrm[p] = n0 / d; Nrm[P + p] = n1 / d; Nrm[2 * P + p] = n2 / d; Nrm[3 * P + p] = 1;   /* :190, :175 */
        dz[p] = d;
    }
    return 0;
}
