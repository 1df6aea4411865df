// SRPS.cpp -- SRPS::execute (reference SRPS.cu:84-370) driving the HIP library through the C ABI.
#include "SRPS.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <thread>
#include "Preprocess.h"
#include "Visualize.h"

SRPS::SRPS(DataHandler& dh) { this->dh = &dh; }
SRPS::~SRPS() {
    if (shard_ctx.empty()) { if (ctx) srps_destroy(ctx); }
    else for (srps_ctx* c : shard_ctx) srps_destroy(c);
}

static std::string out_path(const char* name) { return Preferences::outDir + "/" + name; }

void SRPS::execute() {
    const float TOLERANCE = 5e-3f;          // SRPS.cu:85
    const int MAX_ITERATIONS = 10;          // SRPS.cu:86
    dh->validate();
    const int n_gpus = std::max(1, std::min(Preferences::numGpus, std::max(1, dh->I_n)));      // a rank without images has nothing to add
    const bool sharded = n_gpus > 1 || Preferences::forceSharded;
    if (!sharded) {
        if (!ctx) srps_check(srps_create(Preferences::deviceId, Preferences::blockX, Preferences::blockY, &ctx));   // SRPS.cu:88-98
        srps_check(srps_set_option(ctx, "exclusive_device", Preferences::exclusiveDevice ? 1 : 0));
    }

    // Depth mean, inpainting, smoothing, up-sampling (CPU) -- SRPS.cu:117-149
    std::cout << "Mean of depth values" << std::endl;
    std::cout << "Inpainting depth values" << std::endl;
    std::cout << "Smoothing depth" << std::endl;
    std::cout << "Resample depths" << std::endl;
    std::vector<float> zs, z_full;
    preprocess_depth(dh->z0.data(), dh->z0_h, dh->z0_w, dh->z0_n, dh->I_h, dh->I_w, zs, z_full);
    if (Preferences::writeOutputs) write_MAT_floats(zs.data(), zs.size(), out_path("zs_init.mat").c_str());   // SRPS.cu:143
    if (sharded) { execute_sharded(zs, z_full, n_gpus); return; }

    // mask indices, KT / Dx / Dy structure, compaction, initial values -- SRPS.cu:151-270
    std::cout << "Mask index calculation" << std::endl;
    std::cout << "Masked resample matrix" << std::endl;
    std::cout << "Masked gradient matrix" << std::endl;
    std::cout << "Initialization" << std::endl;
    srps_problem pr{};
    pr.h = dh->I_h; pr.w = dh->I_w; pr.n_channels = dh->I_c; pr.n_images = dh->I_n; pr.n_images_total = dh->I_n;
    pr.image_offset = 0; pr.sf = (int)dh->sf;
    pr.mask = dh->mask.data(); pr.K = dh->K.data(); pr.zs_lr = zs.data(); pr.z_full = z_full.data();
    if (dh->I_u8.size() == dh->I.size() && !dh->I_u8.empty()) pr.I_u8 = dh->I_u8.data();      // 8-bit image files: the bytes cross PCIe (I = byte / 255.f is formed on the device)
    else pr.I = dh->I.data();
    srps_check(srps_setup(ctx, &pr));
    int nimg = 0, nch = 0, gh = 0, gw = 0;
    srps_check(srps_dims(ctx, &npix, &npixs, &gh, &gw, &nimg, &nch));
    z.resize(npix); rho.resize((size_t)npix * nch); s.resize((size_t)nimg * nch * 4); N.resize((size_t)npix * 4);
    if (Preferences::writeOutputs) {
        srps_check(srps_get(ctx, "z", z.data(), z.size()));
        write_MAT_floats(z.data(), z.size(), out_path("z_init.mat").c_str());                                  // SRPS.cu:250
    }

    // pixel index list for the visualisations (imask of SRPS.cu:157-162)
    std::vector<int> imask;
    std::vector<float> N_init;
    if (Preferences::writeImages) {
        for (int i = 0; i < dh->I_h * dh->I_w; ++i) if (dh->mask[i] != 0.f) imask.push_back(i);
        N_init.resize(N.size());
        srps_check(srps_get(ctx, "N", N_init.data(), N_init.size()));                                          // d_init_N, SRPS.cu:270
    }

    // Core algorithm -- SRPS.cu:272-335
    float last_error = NAN;
    bool stop_loop = false;
    int iteration = 1;
    energies.clear();
    // The reference times every phase with a host Timer (SRPS.cu:277-295), which costs a device synchronisation per phase; here
    // the phases record HIP events on the stream (srps_get_timings) and the host waits once per pass, for the energy.
    srps_check(srps_set_option(ctx, "phase_timing", 1));
    do {
        float error = 0.f;
        srps_check(srps_lighting(ctx));                                                                        // SRPS.cu:281
        srps_check(srps_albedo(ctx));                                                                          // SRPS.cu:287
        srps_check(srps_depth(ctx, &error));                                                                   // SRPS.cu:293
        float ms[SRPS_N_PHASES];
        srps_check(srps_get_timings(ctx, ms));
        auto sec = [&](int a, int b = -1) { return 1e-3 * ((ms[a] > 0 ? ms[a] : 0.f) + (b >= 0 && ms[b] > 0 ? ms[b] : 0.f)); };
        printf("\n%-25s: %-6.6fs\n", "Lightning Estimation", sec(SRPS_PHASE_LIGHTING));
        printf("%-25s: %-6.6fs\n", "Albedo Estimation", sec(SRPS_PHASE_ALBEDO_SWEEP, SRPS_PHASE_ALBEDO_SOLVE));
        printf("%-25s: %-6.6fs\n", "Depth Estimation", sec(SRPS_PHASE_DEPTH_ASSEMBLY, SRPS_PHASE_DEPTH_SOLVE) + sec(SRPS_PHASE_ENERGY));

        const float rel_err = fabsf(last_error - error) / fabsf(error);                                        // SRPS.cu:298
        if (error > last_error || rel_err < TOLERANCE || iteration > MAX_ITERATIONS) stop_loop = true;       // SRPS.cu:299
        last_error = error;
        energies.push_back(error);
        printf("\nIteration %02d summary\n", iteration);
        printf("%-25s: %-6.3f\n", "Error", error);
        printf("%-25s: %-6.3f\n", "Relative Error", rel_err);
        srps_check(srps_normals(ctx));                                                                         // SRPS.cu:310-315
        iteration++;
        srps_check(srps_get(ctx, "s", s.data(), s.size()));
        srps_check(srps_get(ctx, "rho", rho.data(), rho.size()));
        srps_check(srps_get(ctx, "z", z.data(), z.size()));
        srps_check(srps_get(ctx, "N", N.data(), N.size()));
        if (Preferences::writeImages) {                                                                        // SRPS.cu:319-327, scale 0.425
            const float scale = 0.425f;
            png_write_rgb8(out_path("Normals-Initial.png"), resize_bilinear(normals_image(N_init.data(), imask, dh->I_h, dh->I_w), scale));
            png_write_rgb8(out_path("Normals-Current-Iteration.png"), resize_bilinear(normals_image(N.data(), imask, dh->I_h, dh->I_w), scale));
            png_write_rgb8(out_path("Albedo.png"), resize_bilinear(albedo_image(rho.data(), imask, dh->I_h, dh->I_w, nch), scale));
            png_write_rgb8(out_path("Depth.png"), resize_bilinear(depth_image(z.data(), imask, dh->I_h, dh->I_w), 0.4f));
        }
        if (Preferences::writeOutputs) {                                                                       // SRPS.cu:330-333
            write_MAT_floats(s.data(), s.size(), out_path("s.mat").c_str());
            write_MAT_floats(rho.data(), rho.size(), out_path("rho.mat").c_str());
            write_MAT_floats(z.data(), z.size(), out_path("z.mat").c_str());
            write_MAT_floats(N.data(), N.size(), out_path("N.mat").c_str());
        }
    } while (!stop_loop);
    std::cout << "Done!" << std::endl;                                                                         // SRPS.cu:337 (no waitKey)
}


// --gpus N: the images are sharded over N devices of this node, one host thread and one context per device, the four
// all-reduces of a pass inside the library (srps_execute_sharded: ncclAllReduce over xGMI on each context's stream).  The
// reference's only device code is cudaSetDevice(Preferences::deviceId) (SRPS.cu:88, set from Main.cpp:29): device r of the job
// is deviceId + r.  Every rank computes the same energies and takes the same stop decision (SRPS.cu:297-302); rank 0 prints
// and writes what the one-GPU loop prints and writes.
void SRPS::execute_sharded(const std::vector<float>& zs, const std::vector<float>& z_full, int n_gpus) {
    const float TOLERANCE = 5e-3f;          // SRPS.cu:85
    const int MAX_ITERATIONS = 10;          // SRPS.cu:86
    std::cout << "Mask index calculation" << std::endl;
    std::cout << "Masked resample matrix" << std::endl;
    std::cout << "Masked gradient matrix" << std::endl;
    std::cout << "Initialization" << std::endl;
    for (srps_ctx* c : shard_ctx) srps_destroy(c);
    shard_ctx.assign(n_gpus, nullptr);
    for (int r = 0; r < n_gpus; ++r) {
        srps_check(srps_create(Preferences::deviceId + r, Preferences::blockX, Preferences::blockY, &shard_ctx[r]));
        srps_check(srps_set_option(shard_ctx[r], "exclusive_device", Preferences::exclusiveDevice ? 1 : 0));
        // --partition strips: 2 = the resident kernel on strips of tile columns where they fit.  The ranks are threads of THIS process: the
        // library's handshake finds that out (process ids travel with the buffer handles) and reaches the other ranks' exchange buffers
        // through their pointers + hipDeviceEnablePeerAccess -- hipIpc handles open in other processes only.  Where the strips do not
        // fit (or no fine-grained memory is to be had across devices) the library falls back to the streaming strips (1) by itself.
        if (Preferences::partitionStrips) srps_check(srps_set_option(shard_ctx[r], "cg_partition", 2));
    }
    ctx = shard_ctx[0];
    srps_check(srps_comm_init_all(shard_ctx.data(), n_gpus));                  // ncclCommInitAll over the job's devices
    printf("Images sharded over %d GPU%s (devices %d..%d), RCCL all-reduce of the partial sums%s\n", n_gpus, n_gpus > 1 ? "s" : "",
           Preferences::deviceId, Preferences::deviceId + n_gpus - 1,
           Preferences::partitionStrips ? (n_gpus > 1 ? "; depth CG partitioned into column strips" : "; --partition strips has no effect on one device") : "");
    const size_t per_image = (size_t)dh->I_c * dh->I_h * dh->I_w;
    std::vector<int> imask;
    if (Preferences::writeImages)
        for (int i = 0; i < dh->I_h * dh->I_w; ++i) if (dh->mask[i] != 0.f) imask.push_back(i);
    energies.clear();

    auto rank_main = [&](int r) {
        srps_ctx* c = shard_ctx[r];
        // contiguous shards whose sizes differ by at most one (api.py shard_range)
        int lo = 0, cnt = 0;
        srps_check(srps_shard_range(dh->I_n, n_gpus, r, &lo, &cnt));
        srps_problem pr{};
        pr.h = dh->I_h; pr.w = dh->I_w; pr.n_channels = dh->I_c; pr.n_images = cnt; pr.n_images_total = dh->I_n;
        pr.image_offset = lo; pr.sf = (int)dh->sf;
        pr.mask = dh->mask.data(); pr.K = dh->K.data(); pr.zs_lr = zs.data(); pr.z_full = z_full.data();
        if (dh->I_u8.size() == dh->I.size() && !dh->I_u8.empty()) pr.I_u8 = dh->I_u8.data() + (size_t)lo * per_image;
        else pr.I = dh->I.data() + (size_t)lo * per_image;
        srps_check(srps_setup(c, &pr));
        std::vector<float> N_init;
        if (r == 0) {
            int nimg = 0, nch = 0, gh = 0, gw = 0;
            srps_check(srps_dims(c, &npix, &npixs, &gh, &gw, &nimg, &nch));
            z.resize(npix); rho.resize((size_t)npix * nch); s.resize((size_t)dh->I_n * nch * 4); N.resize((size_t)npix * 4);
            if (Preferences::writeOutputs) {
                srps_check(srps_get(c, "z", z.data(), z.size()));
                write_MAT_floats(z.data(), z.size(), out_path("z_init.mat").c_str());                          // SRPS.cu:250
            }
            if (Preferences::writeImages) { N_init.resize(N.size()); srps_check(srps_get(c, "N", N_init.data(), N_init.size())); }
            srps_check(srps_set_option(c, "phase_timing", 1));
        }
        float last_error = NAN;
        bool stop_loop = false;
        int iteration = 1;
        do {
            float error = 0.f;
            int done = 0;
            srps_check(srps_execute_sharded(c, 1, &error, &done));                                             // SRPS.cu:281-315, one pass
            const float rel_err = fabsf(last_error - error) / fabsf(error);                                    // SRPS.cu:298
            if (error > last_error || rel_err < TOLERANCE || iteration > MAX_ITERATIONS) stop_loop = true;   // SRPS.cu:299
            last_error = error;
            if (r == 0) {
                if (iteration == 1 && Preferences::partitionStrips && n_gpus > 1) {      // which form of the depth CG the pass really ran
                    int resident = 0, strips = 0;
                    srps_check(srps_get_option(c, "cg_partition_resident_active", &resident));
                    srps_check(srps_get_option(c, "cg_partition_active", &strips));
                    printf("Depth CG: %s\n", resident ? "the resident kernel on strips of tile columns (ranks of one process: peer pointers)"
                                                      : strips ? "streaming column strips (an all-reduce and a neighbour exchange per step)" : "replicated on every rank");
                }
                float ms[SRPS_N_PHASES];
                srps_check(srps_get_timings(c, ms));
                auto sec = [&](int a, int b = -1) { return 1e-3 * ((ms[a] > 0 ? ms[a] : 0.f) + (b >= 0 && ms[b] > 0 ? ms[b] : 0.f)); };
                printf("\n%-25s: %-6.6fs\n", "Lightning Estimation", sec(SRPS_PHASE_LIGHTING));
                printf("%-25s: %-6.6fs\n", "Albedo Estimation", sec(SRPS_PHASE_ALBEDO_SWEEP, SRPS_PHASE_ALBEDO_SOLVE));
                printf("%-25s: %-6.6fs\n", "Depth Estimation", sec(SRPS_PHASE_DEPTH_ASSEMBLY, SRPS_PHASE_DEPTH_SOLVE) + sec(SRPS_PHASE_ENERGY));
                energies.push_back(error);
                printf("\nIteration %02d summary\n", iteration);
                printf("%-25s: %-6.3f\n", "Error", error);
                printf("%-25s: %-6.3f\n", "Relative Error", rel_err);
                srps_check(srps_get(c, "s", s.data(), s.size()));
                srps_check(srps_get(c, "rho", rho.data(), rho.size()));
                srps_check(srps_get(c, "z", z.data(), z.size()));
                srps_check(srps_get(c, "N", N.data(), N.size()));
                if (Preferences::writeImages) {                                                                // SRPS.cu:319-327
                    const float scale = 0.425f;
                    png_write_rgb8(out_path("Normals-Initial.png"), resize_bilinear(normals_image(N_init.data(), imask, dh->I_h, dh->I_w), scale));
                    png_write_rgb8(out_path("Normals-Current-Iteration.png"), resize_bilinear(normals_image(N.data(), imask, dh->I_h, dh->I_w), scale));
                    png_write_rgb8(out_path("Albedo.png"), resize_bilinear(albedo_image(rho.data(), imask, dh->I_h, dh->I_w, dh->I_c), scale));
                    png_write_rgb8(out_path("Depth.png"), resize_bilinear(depth_image(z.data(), imask, dh->I_h, dh->I_w), 0.4f));
                }
                if (Preferences::writeOutputs) {                                                               // SRPS.cu:330-333
                    write_MAT_floats(s.data(), s.size(), out_path("s.mat").c_str());
                    write_MAT_floats(rho.data(), rho.size(), out_path("rho.mat").c_str());
                    write_MAT_floats(z.data(), z.size(), out_path("z.mat").c_str());
                    write_MAT_floats(N.data(), N.size(), out_path("N.mat").c_str());
                }
            }
            iteration++;
        } while (!stop_loop);
    };
    // A rank that fails cannot leave the others waiting in a collective for ever: the error is printed and the process ends
    // (the reference's error path is exit(1) too, Utilities.cpp:8-19).
    auto guarded = [&](int r) {
        try { rank_main(r); }
        catch (const std::exception& e) { std::cerr << "rank " << r << ": " << e.what() << std::endl; std::_Exit(1); }
    };
    std::vector<std::thread> th;
    for (int r = 1; r < n_gpus; ++r) th.emplace_back(guarded, r);
    guarded(0);
    for (auto& t : th) t.join();
    std::cout << "Done!" << std::endl;                                                                         // SRPS.cu:337
}
