/*
 * srps.h -- C ABI of libsrps_hip.so: the MI355X (gfx950) implementation of the SRPS
 * alternating-optimisation hot path of nihalsid/SRmeetsPS-CUDA.
 *
 * This header is the drop-in boundary.  The reference has no FFI layer; its seam is the C++
 * header devicecalls.cuh:26-37 (free functions cuda_based_*) plus SRPS::execute (SRPS.cu:84-370).
 * Every entry point below names the reference interface it replaces ("replaces: file:line").
 * Paths are relative to /root/reference/SRmeetsPS-GPU/.
 *
 * Conventions
 *   - plain pointers and sizes only; float = fp32, int = int32, all device pointers are HIP
 *     device memory on the context's device;
 *   - layouts are the reference's: flat column-major images (linear index i + j*h), masked
 *     ("compact") vectors in ascending linear index, I[n][c][p], s[n][c][4], rho[c][p], N[k][p];
 *   - every function returns an int status (SRPS_OK == 0); srps_last_error() gives the text of
 *     the calling thread's last failure.  The library never calls exit() and never throws
 *     across the boundary (the reference prints + exit(1) on CUDA errors, Utilities.cpp:8-19,
 *     and throws std::runtime_error on library errors, Utilities.cpp:21-31; the C++ facade in
 *     srmeetsps-cuda_amd/host/ re-creates the throwing behaviour on top of these codes);
 *   - ownership: the caller allocates every output buffer (the reference's functions
 *     cudaMalloc their results and make the caller cudaFree them, e.g. devicecalls.cu:24-29,
 *     197-199); the context owns all workspace;
 *   - one context per device, one host thread per context; kernels are enqueued on the
 *     context's stream (srps_set_stream) and calls return without synchronising unless they
 *     hand a host scalar back (energy, iteration counts, srps_get_*).
 */
#ifndef SRPS_H
#define SRPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srps_ctx srps_ctx;

enum {
    SRPS_OK = 0,
    SRPS_ERR_INVALID = 1,      /* bad argument (null pointer, non {0,1} mask, size mismatch) */
    SRPS_ERR_HIP = 2,          /* HIP runtime error (text in srps_last_error)               */
    SRPS_ERR_STATE = 3,        /* call order violated (e.g. depth before srps_bind_grid)    */
    SRPS_ERR_NOMEM = 4,
    SRPS_ERR_UNSUPPORTED = 5
};

/* albedo solver: the reference solves the per-pixel diagonal system with its global CG (devicecalls.cu:513-548); CLOSED_FORM is the
 * fixed point of that CG (num / den per pixel; pixels with a zero denominator keep their value).
 * AUTO (the default since round 4): the PIPELINE phases (srps_albedo, srps_albedo_partial / _finish, srps_execute*) form the fixed
 * point -- on a context that holds all images inside the one albedo sweep (FUSED), on a shard from the all-reduced num and den --, the
 * OPERATOR srps_albedo_estimation, which stands in for cuda_based_albedo_estimation call for call, runs the reference's CG.
 * Why the pipeline may: from identical inputs one albedo step of the CG lands within 4e-7 of the fixed point (its recursive residual
 * falls below 1e-18 after 6 - 17 steps; the diagonal's condition number is 2 - 3 on the reference's Mitten data); after whole solves the
 * two modes differ by depth 3e-7 ... 7e-7 relative RMSE and albedo 2e-4 ... 1e-3 max-abs (4e-5 RMSE) -- which is what two faithful
 * implementations of the CG mode differ by as well (this library against the oracle on Mitten: 1e-3): rounding differences of the
 * depth (1 ulp of z) reach the normals through the finite differences amplified by 2 f / z, and the albedo through the shading.
 * tools/albedo_mode_compare.py, DESIGN.md section 4.  Set SRPS_ALBEDO_CG to have the reference's CG in the pipeline too. */
enum { SRPS_ALBEDO_CG = 0, SRPS_ALBEDO_CLOSED_FORM = 1,
       /* the fixed point AND the depth system (g, q) formed inside the one albedo sweep over the images: no num / den / image-sum
        * planes, no albedo solve, no depth assembly kernel.  The same bits as CLOSED_FORM; pipeline phases on one GPU (elsewhere it
        * behaves as CLOSED_FORM). */
       SRPS_ALBEDO_FUSED = 2,
       SRPS_ALBEDO_AUTO = 3 };
/* operator used by the depth CG: AUTO picks the register-marching kernel when sf is 1, 2 or 4 */
enum { SRPS_APPLY_AUTO = 0, SRPS_APPLY_SIMPLE = 1, SRPS_APPLY_MARCH = 2 };

const char* srps_last_error(void);
const char* srps_version(void);

/* ---- context ------------------------------------------------------------------------------
 * replaces: cudaSetDevice + cusparseCreate + cublasCreate (SRPS.cu:88-98) and their destroy
 * calls (SRPS.cu:340-345).  block_x / block_y are the reference's --blockx/--blocky
 * (Main.cpp:27-28, Preferences Utilities.h:224-230); they are accepted and advisory. */
int srps_create(int device_id, int block_x, int block_y, srps_ctx** out);
int srps_destroy(srps_ctx* ctx);
/* The number of HIP devices this process sees (0 without a device; never an error for "no device").  The reference passes
 * --device straight to cudaSetDevice (Main.cpp:29, SRPS.cu:88) and lets the runtime fail; a host that spreads a job over
 * --gpus N devices asks first. */
int srps_device_count(int* n);
/* Diagnostic: the pinned transfer buffers this process has made so far (csrc/srps_xfer.hip).  Every transfer between a caller's host
 * array and the device holds one of its own for its duration and returns it to a process-wide pool: two transfers that overlap in
 * time -- two contexts on two host threads, or srps_setup's image stream beside its mask and depth uploads -- show up as two buffers. */
int srps_transfer_buffers(int* n);
int srps_set_stream(srps_ctx* ctx, void* hip_stream);      /* NULL = the context's own stream */
int srps_synchronize(srps_ctx* ctx);
/* options: "albedo_mode" (SRPS_ALBEDO_*), "apply_mode" (SRPS_APPLY_*), "cg_max_iter", "march_strip" (0 = automatic,
 * or a multiple of 4 in [4,512]), "march_snake" (0|1|2, below), "tensor_recompute" (0|1), "keep_stored_tensor" (0|1),
 * "fuse_energy_lighting" (0|1: the energy sweep over I also leaves the lighting sums of the next pass),
 * "fuse_normals" (0|1, default 1: that sweep also stores the normals and dz of the depth just solved, which it forms in registers anyway --
 *  srps_normals then launches nothing; same bits as the normals kernel),
 * "albedo_persistent" (0|1: albedo CG in registers, one cooperative launch, when the mask fits),
 * "cg_resident" (0|1: depth CG as one persistent launch with its state in registers and LDS, when the grid fits one
 * 256 x 64 tile per CU; otherwise, and with 0, one operator + one update kernel per step), "light_grouped" (0|1),
 * "coop_launch" (0 plain | 1 hipLaunchCooperativeKernel | 2 cooperative only when the process has several contexts on the device),
 * "assemble_from_sums" (0|1: depth right-hand side from image sums left by the albedo sweep; no second pass over I),
 * "albedo_one_sync" (0|1: persistent albedo CG with one grid-wide wait per step),
 * "albedo_channels_together" (0|1: persistent albedo CG of 3 channels on masks up to 1 M pixels: the channels share the grid-wide waits),
 * "cg_one_sync" (0|1, default 1: resident CG with one grid-wide wait per step, see DESIGN.md section 4.  A DELIBERATE DEPARTURE from
 *  devicecalls.cu:274, where r1 is the direct dot product of the updated residual (cublasSdot): with 1 the kernel takes beta from a
 *  PREDICTED r.r = r.r - 2 alpha r.w + alpha^2 w.w, whose three products are summed together with p.w before alpha is known
 *  (kernels_resident.hip "one wait"; kernels_march.hip MODE 3 does the same for the streaming step); the prediction is re-anchored on a
 *  direct sum every 16th step, whenever r.r has fallen to a quarter of the anchor, and whenever its terms cancel more than two digits.
 *  Measured against 0 (two waits, the reference's direct dot in every step): depth RMSE 3e-7 ... 6e-6 on unit-scale depth, bounded
 *  at 2e-5 by tests/test_gpu_edge_and_scale.py::test_one_wait_per_cg_step_equals_two and tests/test_gpu_full_size.py -- inside the
 *  1e-4 of north_star, but not the reference's arithmetic: set 0 to have the reference's),
 * "cg_resident_tile" (0|2|16|32|256|512: tile shape of the resident CG; 0 = 256 x 16 tiles while the grid has few of them (512
 *  threads and 2 columns per thread for sf 1 and 2, up to 240 tiles; 256 threads and 4 columns per thread for sf 4, up to 96),
 *  else 256 x 32 tiles (512 threads, 4 columns per thread) wherever the device has a CU for each of them, else 256 x 64 (512
 *  threads, 8 columns per thread); 2 | 16 | 32 | 512 force one of these, 256 the 256 x 32 tiles with 256 threads),
 * "cg_resident_debug" (timing experiments only: wrong results),
 * "shard_range_check" (0|1, default 1: srps_execute_sharded all-reduces an image-coverage vector before its first pass and refuses image
 *  ranges that overlap or leave an image out; 0: the caller vouches for the partition -- kernel tests that run ONE shard alone),
 * "debug_inject_abort" (test hook, 0..3: the pass's next look at the persistent kernels' abort flags finds bit 1 (depth CG) / bit 2
 *  (albedo CG) set, as if another rank of a sharded job had reported an abort -- the pass's tail is then repeated by the streaming kernels),
 * "cg_resident_rect" (0|1: tiles wholly inside the mask run the resident CG's body without structure bits),
 * "cg_fused_step" (0|1: streaming depth CG with the whole step in one launch instead of operator + update kernel),
 * "exclusive_device" (0|1: the caller states that nothing else uses the device: plain instead of cooperative launches of the
 *  persistent kernels), "spin_budget_ms" (a persistent launch whose grid-wide waits are not served within this time aborts and the
 *  phase is repeated by the streaming kernels; default 200), "host_wait_spin" (0|1, default 1: the one wait of a pass for the
 *  device -- the energy of the stop rule -- polls the stream instead of sleeping on it: the result is picked up ~10 - 20 us earlier,
 *  the calling thread is busy meanwhile), "phase_timing", "roctx" (see srps_get_timings),
 * "overlap_exchange" (0|1, default 0: srps_execute_sharded cuts the albedo sweep and the depth assembly into four pixel ranges and
 *  all-reduces a range on a second stream while the next is computed -- same bits, the bytes travel under the sweeps; off until a
 *  multi-GPU run has timed it),
 * "cg_partition" (0|1|2: the depth CG over the ranks of the context's communicator -- 1: column strips with the streaming step and a
 *  4-double all-reduce + edge-column exchange per step, see srps_strip_group_solve; 2 (round 4): the RESIDENT kernel on strips of
 *  256 x 64 tile columns, the ranks' kernels side by side for the whole solve, exchanging their sums and border edges through
 *  each other's exchange buffers -- no collective between the 101 steps.  The handshake (one all-reduce of a 96-float record per rank:
 *  hipIpc handle, buffer address, process id, PCI id, memory kind) maps a rank of ANOTHER process with hipIpcOpenMemHandle and reaches
 *  a rank of the SAME process (srps_comm_init_all, a thread per device) through its pointer + hipDeviceEnablePeerAccess -- HIP opens a
 *  handle in other processes only.  Buffers are fine-grained memory (coherent across devices while kernels run); ordinary memory is
 *  accepted only when every rank sits on one device.  Falls back to 1 / to the replicated CG where the grid does not fit or the mapping
 *  fails, on all ranks together; "cg_partition_resident_active" tells.  Exercised on one device with two and three PROCESSES and with
 *  two and three THREADS of one process (tests/test_gpu_strips.py); across devices not yet),
 * "debug_ipc_same_process" (test hook: same-process ranks map each other through the hipIpc handles -- HIP refuses; the ranks must
 *  recognise it together),
 * "debug_foreign_pid_twin" (test hook, round 6: the rank's handshake record carries a process number of its own.  "Same process" is decided
 *  by pid AND a random 64-bit number drawn once per process AND a hash of the host's boot id and name -- pids repeat across PID
 *  namespaces and hosts --, and a same-process peer's address is checked with hipPointerGetAttributes (device memory of the ordinal
 *  it claims) before a kernel stores through it),
 * "light_run" (1|3, default 3: the form of the tiled energy + lighting sweep on three channels.  3: the contraction A'I of dc.cu:408-444 and
 *  the Gram matrices as v_mfma_f32_4x4x1 outer products on the matrix pipe -- exact f32 --, the four waves of a block decoupled, a quarter of
 *  the pixels and all the images (of a round of at most twenty) each; 1: the vector form, a wave takes one image plane's four 1 KiB pieces
 *  of a tile back to back, channel by channel (what one-channel images always run).  0.255 / 0.23 ms same box; results agree to rounding.
 *  Round 6 removed the forms that were neither a default on some input class nor a tested fall-back: docs/HISTORY.md),
 * "light_tiled" (0|1, default 1: 0 runs the generic sweep -- any channel count, four blocks per pixel range -- on 1 and 3 channels too),
 * "march_x2" (0|1|2, default 2: the one-launch streaming CG step reads and writes x every SECOND launch only and applies both pending
 *  updates there in their order -- the same bits, 43 B per unknown and step instead of 45; 2 = where the step's planes exceed the Infinity
 *  Cache; "march_x2_active" tells),
 * "march_snake" (0|1|2, default 2: 1 odd strips of the streaming CG step march right to left; 2 the directions also alternate from step to
 *  step, so that a step starts where the one before it ended -- in the Infinity Cache),
 * "image_store" (0|1, default 1: when every image sample is k / 255.f for a byte k -- what the reference's image loader
 *  produces, Utilities.cpp:343 -- the context also keeps the images as bytes and the two image sweeps of a pass read those:
 *  the same floats, the same results bit for bit, a quarter of the traffic; other images are read as floats) */
int srps_set_option(srps_ctx* ctx, const char* name, int value);
/* reads an option back; also "cg_resident_active" (1 when the bound grid fits one tile per CU and the depth CG
 * therefore runs as the persistent on-chip kernel), "num_cus", "persistent_fallbacks" (persistent launches that gave up a wait
 * so far), "cg_resident_rect_tiles_16" / "_256" / "_512" (tiles of the bound grid that qualify for the body without structure bits),
 * "cg_resident_rect_active" (1 when all of them do and the next depth CG therefore runs the kernel without structure bits),
 * "exchange_buffer_fine" (1 when the context's exchange buffer of the resident strips, "cg_partition" = 2, is fine-grained device memory),
 * "image_store_bytes_active" (1 when the context's images are held as bytes and the sweeps read them) */
int srps_get_option(srps_ctx* ctx, const char* name, int* value);

/* ---- generic sparse operators (device pointers) -------------------------------------------*/
/* replaces: cuda_based_host_COO_to_device_CSR (devicecalls.cuh:37, devicecalls.cu:51-67) incl.
 * sort_COO (devicecalls.cu:4-21).  Host COO in, device CSR out (d_row_ptr[n_row+1],
 * d_col_ind[nnz], d_val[nnz], caller-allocated).  Entries of a row keep their COO order. */
int srps_host_COO_to_device_CSR(srps_ctx* ctx, const int* row, const int* col, const float* val,
                                int n_row, int n_col, int nnz,
                                int* d_row_ptr, int* d_col_ind, float* d_val);
/* replaces: cuda_based_sparsemat_densevec_mul (devicecalls.cuh:26, devicecalls.cu:23-49).
 * y = A x (transpose == 0, y has n_rows) or y = A^T x (transpose != 0, y has n_cols). */
int srps_sparsemat_densevec_mul(srps_ctx* ctx, const int* d_row_ptr, const int* d_col_ind,
                                const float* d_val, int n_rows, int n_cols, int nnz,
                                const float* d_x, int transpose, float* d_y);
/* replaces: cuda_based_conjugate_gradient (devicecalls.cu:229-279): un-preconditioned CG,
 * tol 1e-9 (squared), at most 101 steps, x = warm start (in/out), b = residual (destroyed). */
int srps_conjugate_gradient(srps_ctx* ctx, const int* d_row_ptr, const int* d_col_ind,
                            const float* d_val, int n, int nnz, float* d_x, float* d_b, int* iters);

/* ---- init kernels -------------------------------------------------------------------------*/
/* replaces: cuda_based_mean_across_channels (devicecalls.cuh:28, devicecalls.cu:95-125).
 * h_data is HOST memory [nc][h*w]; d_mean [h*w] and d_inpaint_locations [h*w] are device. */
int srps_mean_across_channels(srps_ctx* ctx, const float* h_data, int h, int w, int nc,
                              float* d_mean, uint8_t* d_inpaint_locations);
/* replaces: cuda_based_rho_init (devicecalls.cuh:31, devicecalls.cu:133-149): rho = 0.5 */
int srps_rho_init(srps_ctx* ctx, float* d_rho, int npix, int nc);
/* replaces: cuda_based_meshgrid_create (devicecalls.cuh:32, devicecalls.cu:151-169):
 * xx[j*h+i] = j - K02, yy[j*h+i] = i - K12 on the full h x w grid (every pixel is written;
 * the reference kernel's landscape indexing bug is not reproduced). */
int srps_meshgrid_create(srps_ctx* ctx, int w, int h, float K02, float K12, float* d_xx, float* d_yy);

/* ---- the four phase operators (device pointers, reference layouts) ------------------------*/
/* replaces: cuda_based_normal_init (devicecalls.cuh:33, devicecalls.cu:171-223).
 * d_N [4][npix] and d_dz [npix] are caller-allocated outputs. */
int srps_normal_init(srps_ctx* ctx, const float* d_z, const float* d_zx, const float* d_zy,
                     const float* d_xx, const float* d_yy, int npix, float K00, float K11,
                     float* d_N, float* d_dz);
/* replaces: cuda_based_lightning_estimation (devicecalls.cuh:34, devicecalls.cu:376-444).
 * d_s [nimages][nchannels][4] is updated in place. */
int srps_lightning_estimation(srps_ctx* ctx, float* d_s, const float* d_rho, const float* d_N,
                              const float* d_I, int npix, int nimages, int nchannels);
/* replaces: cuda_based_albedo_estimation (devicecalls.cuh:35, devicecalls.cu:447-548).
 * d_rho [nchannels][npix] is updated in place. */
int srps_albedo_estimation(srps_ctx* ctx, const float* d_s, float* d_rho, const float* d_N,
                           const float* d_I, int npix, int nimages, int nchannels);
/* Grid geometry for the matrix-free depth operator.  The reference hands Dx, Dy, KT to
 * cuda_based_depth_estimation as CSR matrices it built on the host (make_gradient SRPS.cu:23-71,
 * KT SRPS.cu:170-193); this library rebuilds the same operators from the mask itself and applies
 * them matrix-free, so the mask is bound once instead.  mask is HOST memory, h*w floats in {0,1}. */
int srps_bind_grid(srps_ctx* ctx, int h, int w, int sf, const float* mask);
/* Optional hint for the operator-level depth call: the principal point (K[6], K[7]) the caller used for
 * cuda_based_meshgrid_create (SRPS.cu:256).  With it the library knows that d_xx / d_yy are j - K02 and
 * i - K12 and runs the faster tensor-recompute operator; without it the stored
 * tensor is streamed.  Reset by srps_bind_grid. */
int srps_set_principal_point(srps_ctx* ctx, float K02, float K12);
/* replaces: cuda_based_depth_estimation (devicecalls.cuh:36, devicecalls.cu:550-786).
 * The nine CSR arguments of the reference (Dx, Dy, KT) are implied by srps_bind_grid (srps_depth_estimation_csr below
 * takes them).  d_N is accepted for the reference's argument order and not read (N3 == 1 enters through s3, devicecalls.cu:573).
 * d_z [npix] is updated in place (101 CG steps from the warm start); *energy receives
 * ||KT z - z0s||^2 + lambda * ||A z - B||^2 (devicecalls.cu:762-785). Synchronises. */
int srps_depth_estimation(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_N,
                          const float* d_I, const float* d_xx, const float* d_yy, const float* d_dz,
                          const float* d_z0s, float* d_z, float K00, float K11,
                          int npix, int nimages, int nchannels, float* energy);
/* The same call with the reference's argument list (devicecalls.cuh:36; call site SRPS.cu:293), for a caller that keeps
 * building Dx, Dy, KT as CSR matrices: d_s, d_rho, d_N, d_I, d_xx, d_yy, d_dz, the three matrices as (row_ptr, col_ind, val,
 * n_rows, n_cols, nnz) each, d_z0s, d_z, K00, K11, npix, nimages, nchannels -- the float the reference returns comes back
 * through *energy.  The matrices are not used for the arithmetic (the operator is applied matrix-free from the bound mask):
 * they are CHECKED, row by row, against the structure srps_bind_grid derived -- SRPS_ERR_INVALID names the matrix that
 * does not describe the bound mask -- and the call is forwarded to srps_depth_estimation.  d_N is accepted and unused in
 * both calls: the linearised system takes N3 == 1 and dz, never N (devicecalls.cu:573). */
int srps_depth_estimation_csr(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_N, const float* d_I,
                              const float* d_xx, const float* d_yy, const float* d_dz,
                              const int* d_Dx_row_ptr, const int* d_Dx_col_ind, const float* d_Dx_val, int n_rows_Dx, int n_cols_Dx, int nnz_Dx,
                              const int* d_Dy_row_ptr, const int* d_Dy_col_ind, const float* d_Dy_val, int n_rows_Dy, int n_cols_Dy, int nnz_Dy,
                              const int* d_KT_row_ptr, const int* d_KT_col_ind, const float* d_KT_val, int n_rows_KT, int n_cols_KT, int nnz_KT,
                              const float* d_z0s, float* d_z, float K00, float K11, int npix, int nimages, int nchannels, float* energy);
/* zx = Dx z, zy = Dy z on the bound grid.
 * replaces: the two cuda_based_sparsemat_densevec_mul calls at SRPS.cu:264-265 / 310-311. */
int srps_gradient(srps_ctx* ctx, const float* d_z, int npix, float* d_zx, float* d_zy);
/* y = (KT^T KT + lambda A^T A) x for the tensor assembled by the last depth call (tests). */
int srps_depth_operator_apply(srps_ctx* ctx, const float* d_x, int npix, float* d_y);

/* ---- whole pipeline = SRPS::execute (SRPS.cu:84-370) ----------------------------------------
 * srps_problem mirrors DataHandler (Utilities.h:166-181) after the CPU pre-processing of
 * SRPS.cu:117-149 (zs_lr = smoothed LR depth, z_full = up-sampled HR depth).  All pointers
 * are HOST memory.  Image sharding (new): a rank passes only its images
 * [image_offset, image_offset + n_images) of n_images_total. */
typedef struct srps_problem {
    int h, w;               /* I_h, I_w */
    int n_channels;         /* I_c */
    int n_images;           /* images in I (this rank's shard) */
    int n_images_total;     /* I_n of the whole job (== n_images on one GPU) */
    int image_offset;       /* index of this shard's first image */
    int sf;
    const float* mask;      /* [h*w], {0,1} */
    const float* K;         /* [9] column-major: K[0]=fx K[4]=fy K[6]=cx K[7]=cy */
    const float* I;         /* [n_images][n_channels][h*w] or NULL (then srps_upload_image) */
    const float* zs_lr;     /* [(h/sf)*(w/sf)] */
    const float* z_full;    /* [h*w] */
    const unsigned char* I_u8;  /* [n_images][n_channels][h*w] or NULL: the images as the BYTES the image-folder loader read
                             * (Utilities.cpp:343 forms I = byte / 255.f from them); then I must be NULL.  A quarter of the bytes cross
                             * PCIe, the floats are formed on the device with the loader's expression (same bits), and the bytes
                             * are what the context's 8-bit image store keeps (option "image_store"). */
} srps_problem;

/* replaces: SRPS.cu:100-270 (mask indices, KT/Dx/Dy structure, compaction, s/rho init,
 * meshgrid, first normals).  The host arrays of `prob` are ordinary (pageable) memory and are no longer needed when the call
 * returns.  The device never maps them: they cross PCIe through a pinned buffer of the library's own, filled by a few host
 * threads (option "pin_uploads" = 1 registers the image array in place instead; DESIGN.md §4.5 and docs/HISTORY.md say why that is not the default).
 * The same holds for srps_upload_image, srps_get, srps_set and every other entry point that takes a host pointer.  Every transfer
 * takes a buffer of its own from a process-wide pool (round 5): no lock is held while bytes move, contexts on other threads are
 * never blocked by one context's transfer or wait.  With a full-frame mask and float images the images as they arrive ARE
 * I[n][c][P]: no compaction pass, no second copy on the device. */
int srps_setup(srps_ctx* ctx, const srps_problem* prob);
int srps_upload_image(srps_ctx* ctx, int local_index, const float* host_image /* [c][h*w] */);
int srps_upload_image_u8(srps_ctx* ctx, int local_index, const unsigned char* host_image /* [c][h*w] bytes; I = byte / 255.f */);
int srps_dims(srps_ctx* ctx, int* npix, int* npixs, int* grid_h, int* grid_w, int* n_images, int* n_channels);

/* one-GPU phases on the context's own state; each replaces the call at the cited line */
int srps_lighting(srps_ctx* ctx);                    /* SRPS.cu:281 */
int srps_albedo(srps_ctx* ctx);                      /* SRPS.cu:287 */
int srps_depth(srps_ctx* ctx, float* energy);        /* SRPS.cu:293 */
int srps_normals(srps_ctx* ctx);                     /* SRPS.cu:310-315 */

/* Sharded phases.  *_partial computes this rank's contribution into an exchange
 * buffer; the host sums the buffer over ranks (RCCL all-reduce on the context's stream) and
 * calls the matching *_finish.  srps_exchange returns the device pointer and length of the
 * buffer the last *_partial filled. */
int srps_lighting_local(srps_ctx* ctx);              /* s of the local images; other rows zeroed */
int srps_albedo_partial(srps_ctx* ctx);              /* num [C][P] of the local images; den -- which does not involve the
                                                      * images -- is formed completely on every rank and is not exchanged */
int srps_albedo_finish(srps_ctx* ctx);
int srps_depth_partial(srps_ctx* ctx);               /* q [3][P] (compact; one GPU: on the grid)  */
int srps_depth_solve(srps_ctx* ctx);                 /* rhs, residual, 101 CG steps               */
int srps_energy_partial(srps_ctx* ctx);              /* [2] floats: t1 (replicated), t2 (partial) */
int srps_energy_finish(srps_ctx* ctx, float* energy);
int srps_exchange(srps_ctx* ctx, const char* which /* "s","albedo","depth","energy" */,
                  void** d_ptr, size_t* n_floats);

/* ---- multi-GPU through the boundary (new: the reference is single-GPU, its only device code is
 * cudaSetDevice(Preferences::deviceId), SRPS.cu:88 -- the seam these calls belong to) ---------------------------------------
 * A context that holds a shard of the images (srps_problem.n_images < n_images_total) gets an RCCL communicator; the four
 * all-reduces of a pass (s, the albedo numerator, the compact q, the energy term) then run inside the library, as
 * ncclAllReduce on the context's stream (xGMI between the GPUs of a node).  librccl is resolved at run time: a process that
 * never calls these needs no RCCL.  One rank per device; one host thread per context (a thread per GPU in a one-process job).
 *   srps_comm_unique_id + srps_comm_init_rank : one process per GPU -- rank 0 makes the id (ncclGetUniqueId), the launcher
 *       distributes its SRPS_COMM_ID_BYTES bytes, every rank joins (ncclCommInitRank on the context's device);
 *   srps_comm_init_all : one process, n contexts on n different devices (ncclCommInitAll); rank = index in the array;
 *   srps_set_comm      : borrow an ncclComm_t the caller made itself (not destroyed with the context); NULL unbinds.
 * The communicators the library creates are destroyed by srps_comm_release / srps_destroy.
 * First contact with a multi-GPU node: the environment variable SRPS_FORCE_FAIL (a comma-separated list of "comm", "resident_strips",
 * "strips") makes the named stage fail on every rank where a real failure would be noticed -- srps_comm_init_rank / _init_all return
 * SRPS_ERR_UNSUPPORTED; the handshake of "cg_partition" = 2 reports a local failure and all ranks go on with the streaming strips; the
 * streaming strips are not taken and the replicated CG runs -- so that every fall-back can be run on purpose (tools/multi_gpu_first_contact.sh). */
#define SRPS_COMM_ID_BYTES 128
int srps_comm_unique_id(void* id /* [SRPS_COMM_ID_BYTES] */);
int srps_comm_init_rank(srps_ctx* ctx, const void* id, int rank, int world);
int srps_comm_init_all(srps_ctx* const* ctxs, int n);
int srps_set_comm(srps_ctx* ctx, void* rccl_comm /* ncclComm_t */, int rank, int world);
int srps_comm_release(srps_ctx* ctx);
int srps_comm_info(srps_ctx* ctx, int* rank, int* world /* 0: no communicator bound */);
/* The image-sharded pass over collectives of the caller's instead of RCCL (MPI, gloo, ...): two host functions on DEVICE pointers;
 * the library calls them with the stream drained, each returns 0 once its reads and writes are complete.
 *   allreduce(user, d_buf, n, f64): in-place sum over the ranks of n floats (f64 == 0) or doubles (f64 != 0)
 *   broadcast(user, d_buf, n, root): n floats from rank `root` to every rank
 * srps_execute_sharded and srps_all_reduce then use them; both NULL removes them.  (tests: two processes on one GPU over gloo) */
typedef int (*srps_host_allreduce_fn)(void* user, void* d_buf, size_t n, int f64);
typedef int (*srps_host_broadcast_fn)(void* user, float* d_buf, size_t n, int root);
int srps_set_host_collectives(srps_ctx* ctx, int rank, int world, srps_host_allreduce_fn allreduce, srps_host_broadcast_fn broadcast, void* user);
/* Sum of the exchange buffer `which` ("s", "albedo", "depth", "energy": see srps_exchange) over the ranks, in place, enqueued
 * on the context's stream: what a host that drives the phases itself calls between *_partial and *_finish. */
int srps_all_reduce(srps_ctx* ctx, const char* which);
/* The alternating loop of SRPS.cu:272-335 on a context that holds a shard: per pass lighting_local, all-reduce s,
 * albedo_partial, all-reduce num, albedo_finish, depth_partial, all-reduce q, depth_solve (replicated), energy_partial,
 * all-reduce of the energy term, normals -- one host synchronisation per pass, like srps_execute.  Every rank obtains the same
 * energies and takes the same stop decision.  Should a persistent kernel give up a wait on ANY rank, all ranks learn it
 * through the energy all-reduce, repeat the pass's tail with the streaming kernels together and stay replicas of each other.
 * With a one-rank communicator the results are those of srps_execute bit for bit.  Arguments as srps_execute. */
int srps_execute_sharded(srps_ctx* ctx, int max_outer, float* energies, int* n_outer);

/* The depth CG partitioned into column strips over the communicator's ranks (option "cg_partition" = 1; srps_strips.hip): rank r
 * owns the bounding-box columns [c_r, c_{r+1}) (cut at multiples of sf), runs the one-launch CG step on them, and between two
 * steps the ranks all-reduce the step's four sums (32 bytes) and exchange their edge columns of p, r and omega with their
 * neighbours (ncclSend / ncclRecv); after the last step the strips of the depth are gathered on every rank.  The recurrence is
 * devicecalls.cu:252-275 unchanged.  With it srps_depth_solve / srps_execute_sharded scale the CG itself with the GPUs; without
 * it (default) every rank runs the whole CG (replicas).
 * srps_strip_group_solve: the same solve on n contexts of ONE process that share a device and a stream (srps_set_stream), as a
 * stand-in for n ranks -- the collectives are device copies, the arithmetic is the multi-GPU path's.  Every context must have
 * gone through srps_depth_partial on the same problem; afterwards each holds the new depth as after srps_depth_solve. */
int srps_strip_group_solve(srps_ctx* const* ctxs, int n);
/* The same group with the RESIDENT depth-CG kernel on every strip (round 4): ranges of 256 x 64 tile columns, one persistent launch per
 * context on the context's OWN stream (the launches must run side by side: the contexts must not share a stream), sums and border
 * edges exchanged through each other's memory while the kernels run -- no host step between the 101 CG steps (devicecalls.cu:252-275
 * unchanged; the grid-wide sums are added in the single launch's order: the same bits as srps_depth_solve on one context).  The
 * contexts share a device (one-GPU test bed: all occupied tiles together <= its CUs) or sit on peer devices of this process (xGMI;
 * not exercised yet).  SRPS_ERR_UNSUPPORTED: does not fit, or the launches could not become resident together within spin_budget_ms
 * (nothing stored; use srps_strip_group_solve). */
int srps_strip_group_solve_resident(srps_ctx* const* ctxs, int n);
/* The two partitions as pure functions (no device needed): rank `rank` of `world` owns the grid columns [*c0, *c0 + *width)
 * (multiples of sf; sizes differ by at most one block column) resp. the images [*begin, *begin + *count) (contiguous; sizes differ
 * by at most one). */
int srps_strip_range(int grid_cols, int sf, int world, int rank, int* c0, int* width);
int srps_shard_range(int n_images, int world, int rank, int* begin, int* count);
/* The strips over a transport of the caller's instead of RCCL (MPI, gloo, shared memory ...): three host functions that work on
 * DEVICE pointers.  srps_depth_solve calls them between the launches with the context's stream drained; each returns 0 after
 * its reads and writes are complete.
 *   allreduce(user, d_in, d_out)        : d_out[0..3] = sum over the ranks of d_in[0..3] (doubles; the same bits on every rank)
 *   exchange(user, nbuf, send_left, recv_left, send_right, recv_right, n) : for b < nbuf send n floats from send_left[b] to the left
 *                                          neighbour and receive n into recv_left[b], likewise to the right; a NULL array: no
 *                                          neighbour on that side
 *   allgather(user, d_x, offset, count) : piece q = d_x[offset[q] .. + count[q]) is rank q's; every rank ends with all pieces
 * All three NULL removes the transport. */
typedef int (*srps_strip_allreduce_fn)(void* user, const double* d_in, double* d_out);
typedef int (*srps_strip_exchange_fn)(void* user, int nbuf, const float* const* d_send_left, float* const* d_recv_left,
                                      const float* const* d_send_right, float* const* d_recv_right, size_t n);
typedef int (*srps_strip_allgather_fn)(void* user, float* d_x, const size_t* offset, const size_t* count);
int srps_set_strip_transport(srps_ctx* ctx, int rank, int world, srps_strip_allreduce_fn allreduce, srps_strip_exchange_fn exchange,
                             srps_strip_allgather_fn allgather, void* user);

/* stop rule + loop of SRPS.cu:272-335 on one GPU.  max_outer <= 0: run to the reference's stop
 * rule (at most 11 passes).  energies (may be NULL) must hold max_outer values, or 12 when
 * max_outer <= 0; *n_outer receives the number of passes executed. */
int srps_execute(srps_ctx* ctx, int max_outer, float* energies, int* n_outer);

/* state read-back: name in {"z","rho","s","N","dz","zx","zy","xx","yy","z0s","I"}; host buffer
 * of n floats (must equal the array length). */
int srps_get(srps_ctx* ctx, const char* name, float* host, size_t n);
int srps_set(srps_ctx* ctx, const char* name, const float* host, size_t n);
/* The length (in floats) of a state array, without any side effect on what the context caches. */
int srps_array_size(srps_ctx* ctx, const char* name, size_t* n_floats);
/* The device array itself. The call drops the partial sums cached between phases ("fuse_energy_lighting"); a caller
 * that keeps the pointer and writes through it later must call this again (or srps_set) before the next phase.  Asking for
 * "I" also ends the use of the 8-bit image store (option "image_store") until the images are set again.  Asking for "N" or "dz"
 * ends the double-buffering of the normals ("fuse_normals" swaps two sets of arrays per pass): the pointer stays THE normals / dz of
 * the context until the next srps_setup, which invalidates every pointer obtained here. */
int srps_get_device_ptr(srps_ctx* ctx, const char* name, void** d_ptr, size_t* n_floats);
int srps_last_cg_iterations(srps_ctx* ctx, int* depth_iters, int* albedo_iters /*[8]*/, int* lighting_iters_max);

/* ---- tracing ------------------------------------------------------------------------------
 * replaces: the host Timer around every phase of the loop (Utilities.h:194-222, SRPS.cu:277-295), which needs a device
 * synchronisation per phase.  With option "phase_timing" = 1 every pipeline phase records a pair of HIP events on the
 * context's stream (no synchronisation); srps_get_timings waits for the stream and returns the milliseconds of the phases
 * run since the previous call (summed per phase; -1 for a phase that did not run).  With option "roctx" = 1 (default: the
 * environment variable SRPS_ROCTX) the same spans are roctx ranges, visible in rocprofv3 --marker-trace. */
enum { SRPS_PHASE_LIGHTING = 0, SRPS_PHASE_ALBEDO_SWEEP, SRPS_PHASE_ALBEDO_SOLVE, SRPS_PHASE_DEPTH_ASSEMBLY, SRPS_PHASE_DEPTH_SOLVE,
       SRPS_PHASE_ENERGY, SRPS_PHASE_NORMALS, SRPS_N_PHASES };
int srps_get_timings(srps_ctx* ctx, float* ms /* [SRPS_N_PHASES] */);
const char* srps_phase_name(int phase);

/* ---- measurement --------------------------------------------------------------------------
 * Runs `solves` depth-CG solves of exactly `iters_per_solve` steps on the current system
 * (state is restored afterwards) and reports wall seconds of the whole loop plus the average
 * HIP-event duration (microseconds) of the operator kernel and of the update kernel. */
int srps_bench_cg(srps_ctx* ctx, int solves, int iters_per_solve, double* seconds,
                  double* apply_kernel_us, double* update_kernel_us);
/* algorithmic bytes per launch of the operator / update kernel */
int srps_cg_bytes(srps_ctx* ctx, double* apply_bytes, double* update_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SRPS_H */
