// SRPS.h -- the reference's class surface (SRPS.h:10-18) on top of libsrps_hip.so
#pragma once
#include <vector>
#include "Utilities.h"

class SRPS {
private:
    DataHandler* dh;
    srps_ctx* ctx = nullptr;
    std::vector<srps_ctx*> shard_ctx;     // --gpus N: one context per device (shard_ctx[0] == ctx)
    void execute_sharded(const std::vector<float>& zs, const std::vector<float>& z_full, int n_gpus);

public:
    SRPS(DataHandler& dh);
    ~SRPS();
    void execute();                       // SRPS.cu:84-370

    // results of the last execute(), in the reference's device layouts (compact masked vectors)
    std::vector<float> z, rho, s, N;
    std::vector<float> energies;
    int npix = 0, npixs = 0;
};
