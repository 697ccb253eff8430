// gridgen.cpp -- the role of the reference's experient/main.cpp (:131-168): build the 2-D and 3-D
// wavelet tiles (tile 128, seed 12345) and the Perlin table, then write the five 256x256 raw grids
// for octaves 3, 4, 5 into <outdir>/ -- 15 batched GPU launches instead of ~1M scalar calls.
//
//   gridgen [outdir=result_raw] [--fast] [--size N]
//   gridgen [outdir] --gpus N --lattice L [--octave O]     the sharded dense-grid path (BASELINE configs[4]):
//       ONE L^3 3-D wavelet lattice in z-slabs over N GPUs, one fresh child process per GPU (the parent never touches
//       a GPU and never execs after one did), every rank its slab with wn_eval3d_grid and no collective, then ONE
//       gather of the slabs on rank 0 through the C ABI (include/wnoise_shard.h: wn_comm_create, wn_gather_volume =
//       grouped ncclSend / ncclRecv over RCCL).  Rank 0 writes plane 0 and the last plane as raw float32 files and
//       prints one JSON line with the timings.
//
// By default all 15 files are byte-identical to the reference's committed
// experient/result_raw/*.raw.  --fast routes the 3-D sliced wavelet grids through the separable
// brick kernel (within 1e-5 of the reference); --exact is accepted and is the default.
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <fstream>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "noise_grid.h"
#include "wnoise_shard.h"

namespace {

// One rank of the sharded run (a fresh process): device = rank, its z-slab, the gather, rank 0's files.
int shard_rank(const std::string &outdir, int world, int rank, int lattice, int octave, const std::string &id_file)
{
    using clock = std::chrono::steady_clock;
    auto shard_check = [](int rc, const char *what) {
        if (rc != WN_OK) throw std::runtime_error(std::string(what) + ": " + wn_shard_last_error());
    };
    try {
        int devices = 0;
        wnhost::check(wn_device_count(&devices), "wn_device_count");
        wnhost::check(wn_device_set(rank % devices), "wn_device_set");
        // the communicator id: rank 0 draws it and publishes it through a file, the others wait for the file
        unsigned char id[WN_COMM_ID_BYTES];
        if (rank == 0) {
            shard_check(wn_comm_unique_id(id), "wn_comm_unique_id");
            const std::string tmp = id_file + ".tmp";
            std::ofstream(tmp, std::ios::binary).write(reinterpret_cast<const char *>(id), sizeof(id));
            std::rename(tmp.c_str(), id_file.c_str());
        } else {
            for (int tries = 0;; ++tries) {
                std::ifstream f(id_file, std::ios::binary);
                if (f && f.read(reinterpret_cast<char *>(id), sizeof(id))) break;
                if (tries > 6000) throw std::runtime_error("no communicator id from rank 0 within 60 s");
                usleep(10000);
            }
        }
        wn_comm *comm = nullptr;
        shard_check(wn_comm_create(&comm, world, rank, id), "wn_comm_create");

        WaveletNoise noise(128, 12345); // every rank regenerates the tile from the seed (experient/main.cpp:137-138,143-144)
        noise.generateNoiseTile3D();
        int z0 = 0, z1 = 0;
        shard_check(wn_shard_bounds(lattice, world, rank, &z0, &z1), "wn_shard_bounds");
        wn_grid g = wnhost::lattice2d(lattice, octave, 2.0f, 1.0f / std::sqrt(0.18402f), WN_GRID_DEFAULT);
        g.z0 = z0;
        g.z1 = z1;
        const size_t plane = (size_t)lattice * lattice;
        wnhost::DeviceBuffer slab(std::max<size_t>(1, (size_t)(z1 - z0) * plane) * sizeof(float));
        std::unique_ptr<wnhost::DeviceBuffer> volume;
        if (rank == 0) volume.reset(new wnhost::DeviceBuffer(plane * lattice * sizeof(float)));
        const auto t0 = clock::now();
        if (z1 > z0) wnhost::check(wn_eval3d_grid(noise.tile(3), &g, slab.as<float>(), nullptr), "wn_eval3d_grid");
        wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
        const auto t1 = clock::now();
        shard_check(wn_gather_volume(comm, slab.as<float>(), lattice, lattice, lattice, 0,
                                     rank == 0 ? volume->as<float>() : nullptr, 0, nullptr), "wn_gather_volume");
        wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
        const auto t2 = clock::now();
        if (rank == 0) {
            std::vector<float> host(plane);
            for (int z : {0, lattice - 1}) {
                wnhost::check(wn_copy_d2h(host.data(), volume->as<float>() + (size_t)z * plane, plane * sizeof(float), nullptr), "wn_copy_d2h");
                wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
                std::ofstream(outdir + "/wavelet_noise_3D_lattice_" + std::to_string(lattice) + "_octave_" + std::to_string(octave) +
                                  "_plane_" + std::to_string(z) + ".raw", std::ios::binary)
                    .write(reinterpret_cast<const char *>(host.data()), plane * sizeof(float));
            }
            const double ms_eval = std::chrono::duration<double, std::milli>(t1 - t0).count();
            const double ms_gather = std::chrono::duration<double, std::milli>(t2 - t1).count();
            std::cout << "{\"lattice\": " << lattice << ", \"octave\": " << octave << ", \"ranks\": " << world
                      << ", \"gathered_planes\": " << lattice << ", \"rank0_slab_ms\": " << ms_eval
                      << ", \"gather_ms\": " << ms_gather << ", \"gather\": \"wn_gather_volume: grouped ncclSend/ncclRecv (RCCL)\"}"
                      << std::endl;
        }
        wn_comm_destroy(comm);
    } catch (const std::exception &e) {
        std::cerr << "gridgen rank " << rank << ": " << e.what() << std::endl;
        return 1;
    }
    return 0;
}

// The parent of the sharded run: starts one fresh child per rank (fork + exec of this program BEFORE anything touched a
// GPU in this process) and waits for them.
int shard_parent(const char *self, const std::string &outdir, int world, int lattice, int octave)
{
    const std::string id_file = outdir + "/.gridgen_comm_id." + std::to_string((long)getpid());
    std::remove(id_file.c_str());
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) {
            const std::string w = std::to_string(world), rk = std::to_string(r), l = std::to_string(lattice), o = std::to_string(octave);
            execl(self, self, outdir.c_str(), "--shard-rank", rk.c_str(), "--gpus", w.c_str(), "--lattice", l.c_str(),
                  "--octave", o.c_str(), "--id-file", id_file.c_str(), (char *)nullptr);
            perror("execl");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    std::remove(id_file.c_str());
    return rc;
}

} // namespace

int main(int argc, char **argv)
{
    std::string outdir = "result_raw";
    int flags = WN_GRID_EXACT, image = 256;
    int gpus = 0, lattice = 0, shard_octave = 4, shard_rank_id = -1;
    std::string id_file;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--exact")) flags = WN_GRID_EXACT;
        else if (!std::strcmp(argv[i], "--fast")) flags = WN_GRID_DEFAULT;
        else if (!std::strcmp(argv[i], "--size") && i + 1 < argc) image = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--lattice") && i + 1 < argc) lattice = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--octave") && i + 1 < argc) shard_octave = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--shard-rank") && i + 1 < argc) shard_rank_id = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--id-file") && i + 1 < argc) id_file = argv[++i];
        else outdir = argv[i];
    }
    if (gpus > 0) { // the sharded dense-grid path
        if (lattice <= 0) lattice = 2048;
        mkdir(outdir.c_str(), 0755);
        if (shard_rank_id >= 0) return shard_rank(outdir, gpus, shard_rank_id, lattice, shard_octave, id_file);
        return shard_parent("/proc/self/exe", outdir, gpus, lattice, shard_octave);
    }
    std::cout << "=== Wavelet & Perlin Noise Comparison Generation (MI355X) ===" << std::endl;
    mkdir(outdir.c_str(), 0755);

    const int TILE_SIZE = 128;          // experient/main.cpp:137
    const unsigned int SEED = 12345;    // :138
    try {
        WaveletNoise noise2D(TILE_SIZE, SEED);
        noise2D.generateNoiseTile2D();
        WaveletNoise noise3D(TILE_SIZE, SEED);
        noise3D.generateNoiseTile3D();
        PerlinNoise perlin(SEED);
        for (int octave : {3, 4, 5}) {
            const std::string o = std::to_string(octave);
            generate2DOctaveBandNoise(image, octave, outdir + "/wavelet_noise_2D_octave_" + o + ".raw", noise2D);
            generate3DSlicedOctaveBandNoise(image, octave, outdir + "/wavelet_noise_3Dsliced_octave_" + o + ".raw", noise3D, flags);
            generate3DProjectedOctaveBandNoise(image, octave, outdir + "/wavelet_noise_3Dprojected_octave_" + o + ".raw", noise3D);
            generatePerlinNoise2D(image, octave, outdir + "/perlin_noise_2D_octave_" + o + ".raw", perlin);
            generatePerlinNoise3DSliced(image, octave, outdir + "/perlin_noise_3Dsliced_octave_" + o + ".raw", perlin);
        }
    } catch (const std::exception &e) {
        std::cerr << "gridgen: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "=== Generation Complete ===" << std::endl;
    return 0;
}
