// Main.cpp -- command line of the reference (Main.cpp:9-44): --dstype|-t, --dsloc|-d, --device|-g,
// --blockx|-x, --blocky|-y, --help|-h|--usage; plus --outdir|-o and --no-output for the result dumps, --images, --exclusive,
// --gpus|-n N (images sharded over N devices of this node, RCCL inside the library), --partition images|strips (strips: the depth CG
// is also cut into column strips over the N devices, library option "cg_partition") and --sharded.
#include <algorithm>
#include <cstring>
#include <iostream>
#include <map>
#include "SRPS.h"
#include "Utilities.h"

static void print_message() {
    std::cout << "Usage: srps [params]\n\n"
                 "\t-h, --help, --usage\n\t\tprint help\n"
                 "\t-t, --dstype (value:matlab)\n\t\tdataset type, can be matlab or images\n"
                 "\t-d, --dsloc\n\t\tpath to dataset mat file or folder containing images\n"
                 "\t-g, --device (value:0)\n\t\tHIP device to run the application on\n"
                 "\t-x, --blockx (value:256)\n\t\tblock dimension x (advisory)\n"
                 "\t-y, --blocky (value:4)\n\t\tblock dimension y (advisory)\n"
                 "\t-o, --outdir (value:.)\n\t\tdirectory for zs_init/z_init/s/rho/z/N .mat dumps\n"
                 "\t--no-output\n\t\tdo not write .mat dumps\n"
                 "\t--images\n\t\twrite the reference's three views (normals initial/current, albedo) and the depth map as PNG\n"
                 "\t--exclusive\n\t\tnothing else uses the device: plain instead of cooperative launches of the persistent kernels\n"
                 "\t-n, --gpus (value:1)\n\t\tshard the images over this many devices of the node, starting at --device (RCCL all-reduce of the partial sums)\n"
                 "\t--partition (value:images)\n\t\twith --gpus N: images = shard the images, every device runs the whole depth CG; strips = the depth CG is also\n"
                 "\t\tpartitioned into column strips over the devices (4-double all-reduce + edge-column exchange per CG step)\n"
                 "\t--sharded\n\t\ttake the communicator path even with one GPU (diagnostic)\n";
}

int main(int argc, char* argv[]) {
    static const std::map<std::string, std::string> alias = {{"h", "help"}, {"usage", "help"}, {"t", "dstype"}, {"d", "dsloc"}, {"g", "device"},
                                                            {"x", "blockx"}, {"y", "blocky"}, {"o", "outdir"}, {"n", "gpus"}};
    std::map<std::string, std::string> opt = {{"dstype", "matlab"}, {"device", "0"}, {"blockx", "256"}, {"blocky", "4"}, {"outdir", "."}, {"gpus", "1"},
                                               {"partition", "images"}};   // Main.cpp:11-16
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a.rfind("--", 0) == 0) a = a.substr(2); else if (a.rfind("-", 0) == 0) a = a.substr(1); else continue;
        std::string key = a, val;
        const size_t eq = a.find('=');
        bool has_val = false;
        if (eq != std::string::npos) { key = a.substr(0, eq); val = a.substr(eq + 1); has_val = true; }
        auto al = alias.find(key);
        if (al != alias.end()) key = al->second;
        if (key == "help" || key == "no-output" || key == "images" || key == "exclusive" || key == "sharded") { opt[key] = "true"; continue; }
        if (!has_val && i + 1 < argc) val = argv[++i];
        if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
        opt[key] = val;
    }
    if (opt.count("help") || !opt.count("dsloc")) {                 // Main.cpp:19-26
        print_message();
        return 0;
    }
    // every number of the command line is parsed where a bad one ends in the usage text and exit code 1, not in std::terminate
    // (round-3 advisor finding: --gpus x)
    auto number = [&](const char* key) {
        const std::string& v = opt[key];
        size_t used = 0;
        int n = 0;
        try { n = std::stoi(v, &used); } catch (const std::exception&) { used = 0; }
        if (used == 0 || used != v.size()) throw std::invalid_argument(std::string("--") + key + ": '" + v + "' is not a number");
        return n;
    };
    try {
        Preferences::blockX = number("blockx");                     // Main.cpp:27-29
        Preferences::blockY = number("blocky");
        Preferences::deviceId = number("device");
        Preferences::numGpus = number("gpus");
        if (Preferences::numGpus < 1) throw std::invalid_argument("--gpus: at least 1");
        if (opt["partition"] != "images" && opt["partition"] != "strips") throw std::invalid_argument("--partition: images or strips, got '" + opt["partition"] + "'");
        Preferences::partitionStrips = opt["partition"] == "strips";
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        print_message();
        return 1;
    }
    Preferences::outDir = opt["outdir"];
    Preferences::writeOutputs = !opt.count("no-output");
    Preferences::writeImages = opt.count("images") > 0;
    Preferences::exclusiveDevice = opt.count("exclusive") > 0;
    Preferences::forceSharded = opt.count("sharded") > 0;
    // the devices the job names must exist -- asked after the data set has loaded (a bad path fails without touching a device) and
    // before any context is made
    auto check_devices = [] {
        int ndev = 0;
        srps_check(srps_device_count(&ndev));
        if (Preferences::deviceId < 0 || Preferences::deviceId + Preferences::numGpus > ndev)
            throw std::invalid_argument("--device " + std::to_string(Preferences::deviceId) + " --gpus " + std::to_string(Preferences::numGpus) + ": this node shows " +
                                        std::to_string(ndev) + " HIP device(s)");
    };
    try {
        if (opt["dstype"] == "matlab") {                            // Main.cpp:31-36
            MatFileDataHandler dh;
            dh.loadDataFromMatFiles(opt["dsloc"].c_str());
            check_devices();
            SRPS srps(dh);
            srps.execute();
        } else if (opt["dstype"] == "images") {                     // Main.cpp:37-42
            ImageDataHandler dh;
            dh.loadDataFromImages(opt["dsloc"].c_str());
            check_devices();
            SRPS srps(dh);
            srps.execute();
        }
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;                         // the reference lets it terminate(); exit code 1 either way
        return 1;
    }
    return 0;
}
