// srps_internal.h -- context, grid geometry and launch helpers shared by the HIP sources.
// gfx950 (MI355X) only: wave = 64 lanes everywhere in this library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "srps.h"

namespace srps {

constexpr int WAVE = 64;
constexpr int PAD = 4;            // zero halo (rows and columns) around the bounding-box grid

// per-pixel structure byte of the grid layout (what make_gradient / KT encode as sparse
// matrices in the reference, SRPS.cu:23-71 and 170-193)
enum : uint8_t {
    F_MASK = 1,     // pixel is in the mask
    F_FX = 2,       // Dx row is forward:  x[i,j+1] - x[i,j]     (SRPS.cu:39-42)
    F_BX = 4,       // Dx row is backward: x[i,j]   - x[i,j-1]   (SRPS.cu:43-46)
    F_FY = 8,       // Dy row is forward:  x[i+1,j] - x[i,j]     (SRPS.cu:31-34)
    F_BY = 16,      // Dy row is backward: x[i,j]   - x[i-1,j]   (SRPS.cu:35-38)
    F_KB = 32       // pixel belongs to a fully masked sf x sf block (row of KT, SRPS.cu:176-183)
};

// CG scalars kept on the device (no host round trip inside the 101-step loop)
struct CgScalars {
    float r0;        // previous r.r      (dc.cu:273)
    float r1_last;   // last r.r seen by the update kernel
    int iters;       // steps executed    (dc.cu:254)
    int active;      // 1 while r1 > tol^2
    float alpha;     // step length of the last update kernel; x += alpha p is applied by the NEXT operator launch
    // status words of the persistent kernels (device_utils.h SpinState): a wait that outlived the spin budget
    int abort_flags;    // bit 0: depth CG (k_cg_resident), bit 1: albedo CG (k_dcg_persistent*); 0 = every wait completed
    int abort_arrived;  // blocks whose granule had arrived at the first wait that gave up
    int abort_gen;      // generation (wait number) of that wait: 1 = not all blocks became resident
    float alpha_hist[2];   // one-launch streaming step with the x update every SECOND launch (march_x2): launch k leaves alpha_{k-1} in slot (k - 1) & 1
};
enum { ABORT_DEPTH = 1, ABORT_ALBEDO = 2 };
// class of a tile of the resident CG (kernels_resident.hip: resident_body<.., RECT>), per tile shape, set by build_grid
enum : uint8_t { TILE_RECT = 1, TILE_BOTTOM_EMPTY = 2, TILE_RIGHT_EMPTY = 4, TILE_OCCUPIED = 8 /* the tile holds at least one masked pixel */ };

struct Grid {
    bool bound = false;
    int h = 0, w = 0, sf = 0;
    int P = 0, Ps = 0;
    int i_lo = 0, j_lo = 0, Hg = 0, Wg = 0;     // bounding box (aligned to sf)
    int Hs = 0, Ws = 0;                         // padded storage: plane = Ws columns of Hs rows
    size_t plane = 0;
    size_t used = 0;                            // Hs * (Wg + 2 PAD): the part of a plane that holds the grid
    int Hl = 0, Wl = 0;                         // LR bounding box = Hg/sf x Wg/sf
    // device
    int* d_gofp = nullptr;        // [P]   grid offset of compact pixel p
    int* d_imask = nullptr;       // [P]   HR linear index of compact pixel p (SRPS.cu:157-162)
    int* d_imasks = nullptr;      // [Ps]  LR linear index of complete block t (SRPS.cu:163-168)
    uint8_t* d_flags = nullptr;   // [plane]
    int* d_lr_index = nullptr;    // [Hl*Wl] compact LR index of the block, -1 if not fully masked
    uint8_t* d_tile_cls[3] = {nullptr, nullptr, nullptr};   // TILE_* bits of the resident CG's tiles: [0] 256 x 32, [1] 256 x 64, [2] 256 x 16
    int n_tiles[3] = {0, 0, 0}, n_rect_tiles[3] = {0, 0, 0};
    // The resident CG launches one block per OCCUPIED tile (a sparse mask in a large frame -- an ellipse, the reference's Mitten --
    // leaves bounding-box tiles empty: they hold no unknowns, need no CU, and their neighbours see an empty ring side)
    int* d_tile_list[3] = {nullptr, nullptr, nullptr};      // [n_occ] tile index (column of tiles * tiles per column + row of tiles), ascending
    int n_occ[3] = {0, 0, 0};
    // depth workspace (grid layout)
    float* d_M = nullptr;         // [6][plane]  photometric tensor, SoA
    float* d_q = nullptr;         // [3][plane]  (exchange buffer for the sharded depth phase)
    float* d_G = nullptr;         // [tensor_channels][plane]  g_c = (rho_c/dz)^2 (tensor-recompute form)
    size_t G_planes = 0;          // planes allocated in d_G
    float* d_tconsts = nullptr;   // [8][8] per-channel constants of the tensor-recompute form
    int tensor_channels = 0;      // channels of the last assembly
    bool M_valid = false;         // d_M holds the tensor of the last assembly
    float cx = 0.f, cy = 0.f;     // principal point used by the last assembly (xx = j - cx, yy = i - cy)
    std::vector<int> h_tile_list[3]; // host copies of d_tile_list[.] (build_grid): the resident strips cut them into ranges of tile columns
    float* d_x = nullptr;         // [plane] z on the grid
    float* d_x2 = nullptr;        // [plane] the resident CG stores its result here and the two planes swap roles: the iterate a persistent
                                  // launch started from survives it (an aborted launch is repeated by the streaming kernels from exactly that iterate)
    float* d_r = nullptr;         // [plane] rhs, then residual
    float* d_r2 = nullptr;        // [plane] second residual plane of the one-launch CG step
    float* d_p = nullptr;         // [2][plane] search direction, double-buffered by step parity
    float* d_w = nullptr;         // [plane] omega = A p
    float* d_w2 = nullptr;        // [plane] second omega plane of the one-launch CG step (omega of the previous step is read while the new one is written)
    float* d_part4 = nullptr;     // [2][4][n_part4] partial sums of the one-launch CG step (p.omega, r.omega, omega.omega, r.r per block)
    int n_part4 = 0;
    float* d_save = nullptr;      // [plane] bench: copy of x
    // reductions
    float* d_pw_part = nullptr;   // [nb_apply]
    float* d_rr_part = nullptr;   // [2][nb_update]
    float* d_misc_part = nullptr; // [4096] energy partials etc.
    CgScalars* d_scal = nullptr;
    int nb_apply = 0, nb_update = 0;
    // marching-kernel decomposition
    int seg_rows = 0, n_seg = 0, strip_cols = 0, n_strip = 0;
    // strip-partitioned CG (srps_strips.hip): this rank's columns [view_c0, view_c0 + view_w) of the grid (view_w == 0: the whole grid)
    int view_c0 = 0, view_w = 0;
    double* d_totals4 = nullptr;  // [2][4] the sums over all ranks of the launches of even / odd parity (null: the block partials of this grid are
                                  // summed by the next launch itself)
    // Everything above is carved out of ONE device allocation, kept while a later bind asks for no more than it holds (a re-setup
    // on the same frame size allocates nothing): hipMalloc / hipFree of ~25 arrays cost milliseconds per solve
    void* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
};

// scratch of the device-side structure build (kernels_structure.hip)
struct StructScratch {
    uint8_t* mb = nullptr;        // [w][h] the mask as bytes
    uint8_t* lr_full = nullptr;   // [w/sf][h/sf] complete sf x sf blocks of the image's LR grid
    int *col_count = nullptr, *col_lo = nullptr, *col_hi = nullptr, *col_start = nullptr, *lr_count = nullptr, *lr_start = nullptr;
    int *bad = nullptr, *header = nullptr, *n_rect = nullptr;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace srps

struct srps_ctx {
    int device = 0;
    int block_x = 256, block_y = 4;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    int albedo_mode = SRPS_ALBEDO_AUTO;       // pipeline: the fixed point (fused into the sweep where the context holds all images); operator: the reference's CG
    int apply_mode = SRPS_APPLY_AUTO;
    int march_tj = 0;                // strip width of the marching operator (multiple of 4); 0 = chosen by march_plan
    int keep_stored_tensor = 0;      // also write the 6-plane tensor when the recompute form is active (tests)
    int march_x2 = 2;                // 0 never, 1 always, 2 (default) when the step's planes exceed the Infinity Cache (march_x2_on).  One-launch streaming step: x is read and written by every SECOND launch only, which applies the two pending updates (x += alpha_{k-2} p_{k-2}, then alpha_{k-1} p_{k-1}: the same two fmas, the same bits) -- p_{k-2} is what the launch finds in the plane it is about to overwrite with p_k; 43 B per unknown and step instead of 45
    int march_snake = 2;             // 1: odd strips march right-to-left (halo columns shared through L2); 2 (default, round 6): the directions also alternate from CG step to CG step -- the far end of every strip is still in the Infinity Cache when the next step starts there (4096^2: 171 - 174 -> 154 - 160 us per step, 3584^2: 129 -> 105)
    int tensor_recompute = 1;        // rebuild M in the operator kernel from (rho_c/dz)^2 instead of streaming 6 planes
    int cg_max_iter = 100;           // dc.cu:231
    float cg_tol = 1e-9f;            // dc.cu:230
    bool cg_fixed = false;           // bench: run a fixed number of steps
    float lambda = 1.0f;             // dc.cu:644
    srps::Grid grid;
    // grow-only workspaces for the per-pixel phases
    srps::DevBuf ws_light, ws_albedo, ws_stage, ws_misc, ws_struct;
    srps::DevBuf ws_images;          // where srps_setup's image transfers land: [n][c][h w] floats or bytes.  With a full-frame mask and float images that IS
                                     // I[n][c][P]: the context's I then points here (I_in_ws_images) and no compaction pass runs
    hipStream_t aux_stream = nullptr;   // non-blocking: the structure build of srps_setup / srps_bind_grid runs here while the images cross PCIe on `stream`
    hipStream_t gather_stream = nullptr;   // non-blocking: the compaction of image n runs here while image n + 1 is copied
    hipEvent_t aux_event = nullptr;
    srps::DevBuf state_arena;        // the context's state arrays (srps_setup), kept across set-ups like the grid's arena
    int pin_uploads = 0;             // 1: srps_setup registers the caller's image array in place (hipHostRegister) and lets the device read it (see srps_xfer.hip for why not by default)
    float* h_pinned = nullptr;       // 256 floats of pinned host memory for scalar read-back
    // The scalars the host reads back after a pass live in ONE device record with the layout of h_pinned -- [0..3] energy terms,
    // [8] lighting iterations, [16..47] albedo CG records, [64..71] depth CG scalars -- so that one copy fetches them all.
    float* d_report = nullptr;       // 256 floats
    bool report_pending = false;     // the device record is newer than h_pinned
    int report_zero_copy = 1;        // the energy + lighting sweep's last block finishes the pass's report record and writes it into h_pinned itself
    unsigned report_seq = 0;         // sequence number of the last record armed that way; report_seq_armed != 0: srps_energy_finish waits for it
    unsigned report_seq_armed = 0;
    float* h_pinned_dev = nullptr;   // h_pinned as the device sees it
    // pipeline state (srps_setup); reference layouts
    bool have_state = false;
    int C = 0, N_local = 0, N_total = 0, img_offset = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    float op_cx = 0, op_cy = 0;      // srps_set_principal_point (operator-level depth call)
    bool op_pp_set = false;
    float *s = nullptr, *rho = nullptr, *z = nullptr, *Nrm = nullptr, *dz = nullptr;
    float *zx = nullptr, *zy = nullptr, *xx = nullptr, *yy = nullptr, *z0s = nullptr, *I = nullptr;
    float *Nrm2 = nullptr, *dz2 = nullptr;   // the fused energy + lighting sweep writes the normals and dz of the new depth here (the old ones stay current
                                     // until srps_normals, which swaps the two sets)
    bool nd_ptr_out = false;         // srps_get_device_ptr("N" | "dz") has handed out a pointer: no swapping of the two sets until the next set-up
    int fuse_normals = 1;            // option: that sweep also stores the normals and dz of the new depth (no k_normals launch per pass)
    bool normals_pending = false;    // Nrm2 / dz2 hold the normals and dz of the current depth, left by the sweep
    float* albedo_ex = nullptr;      // [2][C][P]  num, den
    float* q_ex = nullptr;           // [3][P] q of a shard in the compact layout: what travels in the all-reduce (allocated for shards only)
    bool q_in_exchange = false;      // srps_depth_partial left q in q_ex: srps_depth_solve scatters it onto the grid planes first
    float* energy_ex = nullptr;      // [2]
    int last_depth_iters = 0, last_light_iters = 0;
    int last_albedo_iters[8] = {0};
    int albedo_iters_pending = 0;    // channels whose counts still sit in the pinned buffer (persistent albedo CG)
    // what the last depth assembly was built from (srps_depth_operator_apply)
    bool tensor_valid = false;
    bool depth_assembled = false;    // SRPS_ALBEDO_FUSED: this pass's albedo sweep has left g and q on the grid (srps_depth_partial has nothing to do)
    // energy(k) + lighting(k+1) fusion: ws_light holds the lighting partial sums of the current rho, z, I
    int fuse_energy_lighting = 1;
    int light_bytes = 1;             // the tiled lighting sweep reads the 8-bit image store when the context holds one (round 4)
    bool I_in_ws_images = false;
    int light_run = 3;               // the tiled energy + lighting sweep, three channels: 3 = the contraction A'I on the matrix pipe (v_mfma_f32_4x4x1), a block's
                                     // waves decoupled, each with a quarter of the pixels and all the images (k_light_fused_mfw; more than 20 images: rounds); 1 = vector
                                     // form, a wave reads ONE image plane's four 1 KiB pieces back to back, channel by channel (k_light_fused_tile; one channel always).
                                     // Same box, 2048^2 x 20: 1: 0.247 - 0.264, 3: 0.216 - 0.247 ms (profiles/r05_ab_lighting_mfma.txt; earlier forms: docs/HISTORY.md)
    bool n3_one = false;             // the context's N[3] plane holds ones (set-up and every normals kernel write it so; false once the caller may have written N)
    int light_tiled = 1;             // the fused sweep as a tiled kernel (1 or 3 channels): geometry and normals once per pixel through LDS; 0: the sweep any other channel count takes (k_light_grouped)
    int light_grouped = 1;           // lighting sweep with the images of a batch dealt to four blocks per pixel range
    int coop_launch = 1;             // launch of the persistent kernels: 1 = hipLaunchCooperativeKernel (default; one cooperative queue
                                     // per device: two such kernels of this process cannot interleave their blocks and wait for each
                                     // other; +13 us of queue time before and after), 0 = plain launch behind the same occupancy
                                     // check -- the caller states that the device is exclusive to this context (option
                                     // "exclusive_device"), 2 = plain while this is the only live context of the process on its device.
                                     // Whatever the launch, every wait inside the kernels is bounded (spin_budget_ms).
    int host_wait_spin = 1;          // the pass's one wait for the device (srps_energy_finish) polls the stream instead of sleeping on it
    int spin_budget_ms = 200;        // a persistent launch whose waits are not all served within this time aborts (device_utils.h
                                     // SpinState); the host then repeats the phase with the streaming kernels
    int persistent_fallbacks = 0;    // aborted persistent launches so far (option "persistent_fallbacks", read-only)
    // multi-GPU (srps_comm.hip): the RCCL communicator of the image-sharded pass
    void* comm = nullptr;            // ncclComm_t
    bool comm_owned = false;         // created by srps_comm_init_rank / srps_comm_init_all (destroyed with the context), not borrowed (srps_set_comm)
    int comm_rank = 0, comm_world = 1;
    int cg_strips = 0;               // option "cg_partition": 1 = the depth CG partitioned into column strips over the communicator's ranks
                                     // (streaming step + collectives per step); 2 = the RESIDENT kernel on strips of tile columns, sums and border
                                     // edges through the ranks' shared exchange buffers (hipIpc), falling back to 1 / the replicated CG
    // the exchange buffer of the resident strips (cg_partition = 2): fine-grained device memory, exported to the other ranks of the
    // communicator by hipIpc handle; peer_base[q] = rank q's buffer as mapped into this process (own rank: the buffer itself)
    void* xg_buf = nullptr;
    size_t xg_bytes = 0;
    void* xg_peer[8] = {};
    int xg_peer_ipc[8] = {};         // 1: mapped with hipIpcOpenMemHandle (closed on release); 0: the pointer of a rank of this process
    int xg_fine = 0;                 // 1: xg_buf is fine-grained memory (coherent across devices while kernels run)
    int debug_ipc_same_process = 0;  // tests: map a same-process peer through its handle (HIP refuses: the failure must be recognised)
    int debug_foreign_pid_twin = 0;  // tests: this rank's handshake record carries a process number of its own -- a rank of ANOTHER pid namespace that happens to share the pid
    int xg_world = 0;                // ranks the peers were opened for (0: not open)
    int xg_failed = 0;               // the handshake or a launch failed once: not tried again on this context
    double* d_strip_tot = nullptr;   // [12]: [0..3] this rank's sums of a launch, [4..7] / [8..11] the sums over all ranks of the launches of even / odd
                                     // parity (Grid::d_totals4 points at [4])
    // the caller's collectives for the image-sharded pass (srps_set_host_collectives) instead of an RCCL communicator
    srps_host_allreduce_fn host_allreduce = nullptr;
    srps_host_broadcast_fn host_broadcast = nullptr;
    void* host_user = nullptr;
    // the caller's transport for the strips (srps_set_strip_transport) instead of the communicator
    srps_strip_allreduce_fn strip_allreduce = nullptr;
    srps_strip_exchange_fn strip_exchange = nullptr;
    srps_strip_allgather_fn strip_allgather = nullptr;
    void* strip_user = nullptr;
    int strip_rank = 0, strip_world = 1;
    int overlap_exchange = 0;        // option: srps_execute_sharded cuts the two sweeps that feed an all-reduce into pixel ranges and reduces a range
                                     // (on a second stream) while the next one is computed
    bool defer_shard_checks = false; // srps_execute_sharded: a shard's phases do not look at the abort flags themselves; the ranks decide together at the end of the pass
    bool plane_restored = false;     // the last look at the abort flags made the depth plane of the pass's start current again
    bool x_swapped = false;          // the resident CG launched since the abort flags were last looked at swapped grid.d_x and grid.d_x2
    int persistent_inflight = 0;     // ABORT_* bits of the persistent kernels launched since the abort flags were last looked at
    int cg_one_sync = 1;             // resident CG: r.r from r.r - 2 alpha r.w + alpha^2 w.w (one grid-wide wait per step)
    int cg_fused_step = 1;           // streaming depth CG: the whole step in one launch (kernels_march.hip MODE 3) instead of operator + update
    int cg_resident = 1;             // depth CG as one persistent launch with its state in registers + LDS, when the grid fits
    srps::DevBuf ws_resident;
    const void* res_tags_ptr = nullptr;      // the granule arrays the single resident launch last used (see resident_cg: launch-numbered generation tags)
    size_t res_tags_bytes = 0;
    unsigned long long res_tags_layout = 0;      // (tiles, tile columns, threads) of that launch: another layout in the same buffer is zeroed first
    unsigned res_launch_seq = 0;
    int cg_resident_debug = 0;       // timing experiments (kernels_resident.hip)
    int shard_range_check = 1;       // srps_execute_sharded verifies (one small all-reduce per call) that the ranks' image ranges tile the image set
    int debug_inject_abort = 0;      // test hook: ABORT_* bits the next persistent_aborts finds, as reported by another rank
    int cg_resident_tile = 0;        // 0: the smallest tile shape that fits the device, 256 | 512: threads per block of the forced shape
    int cg_resident_rect = 1;        // tiles inside the mask take the body without structure bits (0: every tile the general body)
    int albedo_channels_together = 1;   // persistent albedo CG, 3 channels, small masks: the channels share the grid-wide waits
    int albedo_one_sync = 1;         // persistent albedo CG: p.(D p) of the next direction predicted from three products summed with r.r
    int albedo_persistent = 1;       // albedo CG in registers (one cooperative launch) when the mask fits
    int num_cus = 256;
    bool light_cache_valid = false;
    // The 8-bit image store: when every sample of I is k / 255.f for a byte k -- what the reference's image loader produces
    // (Utilities.cpp:343) -- the two sweeps of a pass read the images as bytes (a quarter of the traffic) and form the same floats.
    int image_store = 1;                  // option "image_store": 0 floats only, 1 bytes whenever the samples allow it
    unsigned char* I8 = nullptr;          // [N_local][C][P] bytes
    size_t I8_cap = 0;                    // bytes allocated behind I8 (kept across set-ups)
    bool I8_cap_ok(size_t n) const { return I8 != nullptr && I8_cap >= n; }
    std::vector<hipEvent_t> ev_copied, ev_gathered;      // upload pipeline of srps_setup: per staging slot
    int i8_state = 0;                     // 0: not looked at since I last changed, 1: I8 holds I, 2: I is not representable
    // image sums of the depth right-hand side left by the albedo sweep of this pass (assemble_from_sums)
    int assemble_from_sums = 1;
    srps::DevBuf ws_ssum;                 // [C][3][P]
    bool ssum_valid = false;              // they belong to the current s and I
    bool plane_holds_z = false;           // the grid plane d_x holds the current z (left there by the last solve)
    bool grad_current = false;            // zx, zy (and the grid copy of z) belong to the current z: srps_normals need not redo them
    bool light_cache_normals = false;     // srps_normals ran on the depth the sums were taken from (Nrm is current)
    int light_cache_V = 0, light_cache_nblk = 0;
    // tracing: per-phase HIP events on the context's stream (option "phase_timing") and roctx ranges (option "roctx")
    int phase_timing = 0, roctx = 0;
    hipEvent_t ev_begin[SRPS_N_PHASES] = {}, ev_end[SRPS_N_PHASES] = {};
    unsigned ev_mask = 0;            // phases whose pair of events has been recorded since the last srps_get_timings
    bool ev_created = false;
};

namespace srps {

// phase spans for tracing: an RAII guard at the top of every pipeline phase
struct PhaseSpan {
    srps_ctx* c;
    int phase;
    bool pushed = false;
    PhaseSpan(srps_ctx* ctx, int ph);
    ~PhaseSpan();
};

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define SRPS_HIP(expr)                                                        \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) return srps::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)
#define SRPS_LAUNCH_CHECK() SRPS_HIP(hipGetLastError())
#define SRPS_REQUIRE(cond, code, ...)                  \
    do {                                               \
        if (!(cond)) {                                 \
            srps::set_error(__VA_ARGS__);              \
            return (code);                             \
        }                                              \
    } while (0)
#define SRPS_TRY(expr)                 \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != SRPS_OK) return rc_; \
    } while (0)

int ensure(DevBuf& b, size_t bytes);
// ---- copies between the caller's host arrays and the device, through the library's own pinned buffer (srps_xfer.hip) ----
int host_upload(srps_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, hipStream_t st);        // returns when h_src has been read and the copies have run
int host_download(srps_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, hipStream_t st);      // returns when h_dst holds the data (after the work queued on st)
int xfer_buffers_made();                                                                              // pinned transfer buffers the process has made (every transfer in flight holds its own)
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- kernel launchers (kernels_pixel.hip) ------------------------------------------------
int launch_fill(hipStream_t st, float* d, size_t n, float v);
int launch_pack_bytes(hipStream_t st, const float* d_I, size_t n, unsigned char* d_out, int* d_inexact);
const unsigned char* image_store_bytes(srps_ctx* ctx, const float* d_I);
int launch_gather_image(hipStream_t st, const float* d_full, const int* d_imask, int P, int C, size_t hw, float* d_out);
int launch_meshgrid_compact(hipStream_t st, const int* d_imask, int P, int h, float cx, float cy, float* xx, float* yy);
int launch_meshgrid_full(hipStream_t st, int w, int h, float K02, float K12, float* xx, float* yy);
int launch_mean_channels(hipStream_t st, const float* d_data, int h, int w, int nc, float* mean, uint8_t* flag);
int launch_final_sum(hipStream_t st, const float* part, int n, float* out);      // out[0] = sum(part[0..n)), fixed order, double
int launch_normals(hipStream_t st, const float* z, const float* zx, const float* zy, const float* xx,
                   const float* yy, int P, float fx, float fy, float* N, float* dz);
int lighting(srps_ctx* ctx, float* d_s, const float* d_rho, const float* d_N, const float* d_I, int P,
             int n_local, int C, int n_total, int img_offset, bool zero_nonlocal, bool use_cache = false);
// What the last block of the fused energy + lighting sweep does when `ticket` is set (one GPU, tiled sweep): it adds the partial sums
// of both energy terms -- the routine and the bits of k_final_sum --, stores them in the report record, copies the record's 80 floats
// into the host's pinned copy and then stores `seq` behind it: two single-block kernels and a copy less per pass, and the host
// sees the record the moment it is complete.
struct ReportFinish {
    unsigned* ticket = nullptr;      // arrivals of the sweep's blocks: zero before the launch, left zero by the last block
    const float* t1_part = nullptr;  // k_energy_t1's partial sums
    int n_t1 = 0;
    float* report = nullptr;         // the device record: [0] = term 1, [1] = term 2
    float* host_report = nullptr;    // the record in mapped host memory; host_report[REPORT_SEQ_AT] = seq when it is complete, [REPORT_CHECK_AT] = seq ^ xor of its words
    unsigned seq = 0;
};
constexpr int REPORT_FLOATS = 80, REPORT_SEQ_AT = 100, REPORT_CHECK_AT = 101, REPORT_TICKET_AT = 250;
int energy_light_fused(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I, const float* d_xx,
                       const float* d_yy, const float* d_dz, const float* d_z, const float* d_zx, const float* d_zy,
                       float fx, float fy, int P, int n_local, int C, int img_offset, float* d_out, const ReportFinish* fin = nullptr, bool* fin_armed = nullptr);
int albedo_numden(srps_ctx* ctx, const float* d_s, const float* d_N, const float* d_I, int P, int n_local,
                  int C, int s_img_offset, float* d_numden, float fx = 0.f, float fy = 0.f, float* d_ssum = nullptr, int n_total = 0,
                  int q0 = 0, int q1 = 0 /* > 0: the pixels [q0, q1) only, q0 a multiple of 1024 */);
int albedo_finish(srps_ctx* ctx, float* d_rho, const float* d_numden, int P, int C, bool pipeline = false);
int albedo_fused(srps_ctx* ctx, const float* d_s, const float* d_N, const float* d_I, int P, int n_img, int C, float* d_rho, const float* d_qc,
                 const float* d_xx, const float* d_yy, const float* d_dz, float fx, float fy);
int depth_fused_prepare(srps_ctx* ctx, const float* d_s, float fx, float fy, int C, int n_total, int n_local, int img_offset, float cx, float cy, bool* ok);
void albedo_iters_collect(srps_ctx* ctx);
int depth_assemble(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I, const float* d_xx,
                   const float* d_yy, const float* d_dz, float fx, float fy, int P, int n_local, int C,
                   int n_total, int img_offset, float cx, float cy, const float* d_ssum = nullptr, float* d_q_compact = nullptr,
                   int q0 = 0, int q1 = 0 /* > 0 (from sums only): the pixels [q0, q1); the constants are formed by the launch with q0 == 0 */);
int depth_q_scatter(srps_ctx* ctx, const float* d_q_compact);
int energy_photometric_partial(srps_ctx* ctx, const float* d_s, const float* d_rho, const float* d_I,
                               const float* d_xx, const float* d_yy, const float* d_dz, const float* d_z,
                               const float* d_zx, const float* d_zy, float fx, float fy, int P, int n_local,
                               int C, int img_offset, float* d_out /* one float, device */);

// ---- grid / CG (kernels_cg.hip) ---------------------------------------------------------
// launch of a kernel whose blocks wait for each other (grid-wide sums): all blocks must be resident at once
int contexts_on_device(int device);      // live srps contexts of this process on that device
int launch_persistent(srps_ctx* ctx, const void* fn, int blocks, int threads, void** args, size_t lds_bytes);
bool resident_supported(const srps_ctx* ctx);
bool resident_rect_active(const srps_ctx* ctx);
int resident_cg(srps_ctx* ctx, int max_steps, bool fixed_steps);
// one rank of a group of resident launches: what a tile-shape unit needs to build its kernel arguments (kernels_resident.hip)
struct ResidentGroupSpec {
    void* exch;                          // this rank's exchange buffer: ent | ent3 | halo
    void* peer_exch[7];                  // the other ranks', in rank order without this one
    int n_peers, left_peer, right_peer;  // indices into peer_exch of the ranks to the left / right (-1: none)
    int tiles, nbr, nbc;                 // the whole grid's tiling in the unit's shape
    int nb_total, list_base, blocks;     // granule slots of the group, this rank's first entry of the tile list, its blocks
    int bc_first, bc_last;               // its first / last column of tiles
    bool rect;
    int max_steps;
    bool fixed_steps;
};
size_t resident_group_bytes_n512(int tiles); size_t resident_group_bytes_n256(int tiles); size_t resident_group_bytes_n256c4(int tiles);
size_t resident_group_bytes_n512c4(int tiles); size_t resident_group_bytes_n512c2(int tiles);
int resident_group_launch_n512(srps_ctx* ctx, const ResidentGroupSpec& sp); int resident_group_launch_n256(srps_ctx* ctx, const ResidentGroupSpec& sp);
int resident_group_launch_n256c4(srps_ctx* ctx, const ResidentGroupSpec& sp); int resident_group_launch_n512c4(srps_ctx* ctx, const ResidentGroupSpec& sp);
int resident_group_launch_n512c2(srps_ctx* ctx, const ResidentGroupSpec& sp);
int resident_cg_group(srps_ctx* const* ctxs, int n, int max_steps, bool fixed_steps);
int resident_cg_rank(srps_ctx* ctx, int max_steps, bool fixed_steps);      // one RANK of such a group: the other ranks are other processes / devices (cg_partition = 2); SRPS_ERR_UNSUPPORTED: use another path
void resident_rank_release(srps_ctx* ctx);                                 // closes the peers' mappings, frees the exchange buffer      // the resident kernel on column strips of several contexts
bool resident_supported_n512(const srps_ctx* ctx);       // the three tile shapes (kernels_resident.hip, kernels_resident_n256.hip, kernels_resident_n256c4.hip)
bool resident_supported_n256(const srps_ctx* ctx);
bool resident_supported_n256c4(const srps_ctx* ctx);
bool resident_supported_n512c4(const srps_ctx* ctx);
bool resident_supported_n512c2(const srps_ctx* ctx);
int resident_cg_n512(srps_ctx* ctx, int max_steps, bool fixed_steps);
int resident_cg_n256(srps_ctx* ctx, int max_steps, bool fixed_steps);
int resident_cg_n256c4(srps_ctx* ctx, int max_steps, bool fixed_steps);
int resident_cg_n512c4(srps_ctx* ctx, int max_steps, bool fixed_steps);
int resident_cg_n512c2(srps_ctx* ctx, int max_steps, bool fixed_steps);
int grid_scatter(srps_ctx* ctx, const float* d_compact, float* d_plane);
int grid_gather(srps_ctx* ctx, const float* d_plane, float* d_compact);
int grid_gradient(srps_ctx* ctx, const float* d_plane, float* d_zx, float* d_zy, float* d_compact = nullptr);      // also gathers the plane when d_compact is given
int grid_rhs(srps_ctx* ctx, const float* d_z0s);                 // r = KT' z0s + lambda (Dx'q0 + Dy'q1 + q2)
int grid_residual(srps_ctx* ctx);                                // r -= A_ x ; rr_part[0]
int grid_apply_plain(srps_ctx* ctx, const float* d_in_plane, float* d_out_plane);
int grid_cg(srps_ctx* ctx, int max_steps, bool fixed_steps);     // the 101-step loop
int grid_energy_t1(srps_ctx* ctx, const float* d_z0s, float* d_out /* one float, device */, const float** part_out = nullptr, int* n_part_out = nullptr);      // part_out: the partial sums only, the caller adds them
int cg_launch_apply(srps_ctx* ctx, int k);
int cg_launch_update(srps_ctx* ctx, int k);
bool use_march(const srps_ctx* ctx);
int apply_blocks(const srps_ctx* ctx);      // partial sums the operator kernel in use leaves in d_pw_part

// ---- marching operator (kernels_march.hip) ------------------------------------------------
bool march_supported(const srps_ctx* ctx);
void march_plan(Grid& G, int strip_cols, int num_cus);
int march_blocks(const Grid& G);
int march_apply_plain(srps_ctx* ctx, const float* d_in_plane, float* d_out_plane);
int march_residual(srps_ctx* ctx);
int march_cg_apply(srps_ctx* ctx, int k);
int march_cg_step(srps_ctx* ctx, int k);
bool march_x2_on(const srps_ctx* ctx);      // the two-step x update is in use on the bound grid (option "march_x2")
bool cg_fused_step(const srps_ctx* ctx);      // the streaming CG runs one launch per step
int cg_flush_x(srps_ctx* ctx);
int march_recompute_channels(const srps_ctx* ctx);
int march_part4_totals(srps_ctx* ctx, int which, double* d_out);

int grid_need_M(srps_ctx* ctx);            // srps_api.hip: the stored 6-plane tensor, allocated on demand
// ---- structure build on the device (kernels_structure.hip) ----------------------------------
size_t struct_scratch_bytes(int h, int w, int sf);
StructScratch struct_scratch(void* base, int h, int w, int sf);
int struct_phase1(hipStream_t st, const float* d_mask, int h, int w, int sf, const StructScratch& s);
int struct_phase2(hipStream_t st, Grid& G, const StructScratch& s, int* d_imasks);
int launch_gather_index(hipStream_t st, const float* d_full, const int* d_index, int n, float* d_out);
int launch_gather_images(hipStream_t st, const float* d_full, const int* d_imask, int P, int C, size_t hw, int n_img, float* d_out);
int launch_gather_images_u8(hipStream_t st, const unsigned char* d_full, const int* d_imask, int P, int C, size_t hw, int n_img, float* d_out, unsigned char* d_out8);

// ---- RCCL (srps_comm.hip) -------------------------------------------------------------------
bool comm_bound(const srps_ctx* ctx);
int comm_all_reduce_sum(srps_ctx* ctx, float* d_buf, size_t n);      // in place, on the context's stream
int comm_all_reduce_pieces_on(srps_ctx* ctx, hipStream_t st, float* const* d_piece, const size_t* n, int pieces);
int comm_broadcast(srps_ctx* ctx, float* d_buf, size_t n, int root);
int comm_all_reduce_sum_f64(srps_ctx* ctx, const double* d_in, double* d_out, size_t n);
int comm_exchange(srps_ctx* ctx, int nbuf, const float* const* send_left, float* const* recv_left, int left,
                  const float* const* send_right, float* const* recv_right, int right, size_t n);
int comm_all_gather_pieces(srps_ctx* ctx, float* d_buf, const size_t* offset, const size_t* count);

// ---- strip-partitioned depth CG (srps_strips.hip) --------------------------------------------
// First contact with a multi-GPU node (round 6): SRPS_FORCE_FAIL=comm,resident_strips,strips in the environment makes the named stage fail
// on EVERY rank (all read the same environment) exactly where a real failure would be noticed, so that the fall-back behind it can be run
// on purpose: "comm" -- srps_comm_init_rank / _init_all refuse (the caller falls back to its own collectives); "resident_strips" -- the
// handshake of cg_partition = 2 reports a local failure (all ranks leave for the streaming strips together); "strips" -- the streaming
// strips are not taken (the replicated CG runs).  tools/multi_gpu_first_contact.sh, tests/test_gpu_distributed.py.
bool forced_failure(const char* stage);
bool strips_active(const srps_ctx* ctx);                       // option "cg_partition" = 1, a communicator of more than one rank, a grid the streaming step handles
int strips_bind_view(srps_ctx* ctx, int rank, int world);      // this rank's columns of the bound grid
void strips_clear_view(srps_ctx* ctx);
int strips_cg(srps_ctx* ctx, int max_steps, bool fixed_steps); // the CG of devicecalls.cu:252-275 on this rank's strip (RCCL between the steps)
int depth_solve_prepare(srps_ctx* ctx);                        // srps_api.hip: the two halves of srps_depth_solve around the CG
int depth_solve_finish(srps_ctx* ctx);
void comm_release(srps_ctx* ctx);

// ---- generic CSR (kernels_csr.hip) --------------------------------------------------------
int csr_spmv(srps_ctx* ctx, const int* rp, const int* ci, const float* v, int n_rows, int n_cols, int nnz,
             const float* x, int transpose, float* y);
int csr_matches_grid(srps_ctx* ctx, const int* dx_rp, const int* dx_ci, const float* dx_v, const int* dy_rp, const int* dy_ci, const float* dy_v,
                     const int* kt_rp, const int* kt_ci, const float* kt_v, int* h_err);
int csr_cg(srps_ctx* ctx, const int* rp, const int* ci, const float* v, int n, int nnz, float* x, float* b, int* iters);

}  // namespace srps
