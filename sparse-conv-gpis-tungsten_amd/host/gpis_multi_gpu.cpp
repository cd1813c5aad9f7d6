// gpis_multi_gpu.cpp — the multi-GPU tile driver in C++ (north_star: host code stays C++): one process per GPU, RCCL over xGMI.
//
// What it replaces in the reference: the tile loop of PathTraceIntegrator (src/core/integrators/path_tracer/
// PathTraceIntegrator.cpp:237-245), which deals image tiles to the threads of one CPU.  Here 16-pixel tile rows are dealt
// round-robin to the ranks (gpis_scene_s.shard_index / shard_count: every rank renders its rows as ONE batch through
// gpis_render_scene_s); ranks exchange nothing while marching.  Two collectives per job, both RCCL:
//   * ncclBroadcast of the parameter block from rank 0 (so that every rank provably renders the same medium),
//   * one grouped ncclSend / ncclRecv of each rank's DISJOINT tile rows to rank 0 per frame (a gather, no sum: the assembled
//     frame is bit-identical to the single-GPU frame), inside the timed region.
// The same scheme as sparse-conv-gpis-tungsten_amd/dist.py (torch.distributed), which bench.py uses; this program is the C++
// form of it and prints the same one-line JSON.
//
// Process model: the launcher forks the N ranks BEFORE anything touches the GPU and waits for them; rank 0 creates the
// ncclUniqueId and hands it to the others through pipes the launcher opened.  Nothing is exec'ed.
//
//   gpis_multi_gpu --gpus N [--config C1|C3] [--width W --height H --spp S] [--steps K --warmup W] [--guide half:ppc|off]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpis.h"

namespace {

struct Options {
    int gpus = 1, steps = 2, warmup = 1, width = 1920, height = 1080, spp = 64, guide_half = 16, guide_ppc = 64;
    std::string config = "C1";
};

#define CHECK_HIP(x)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (x);                                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define CHECK_NCCL(x)                                                                                        \
    do {                                                                                                     \
        ncclResult_t e_ = (x);                                                                               \
        if (e_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(e_)); exit(3); } \
    } while (0)
#define CHECK_GPIS(x)                                                                                        \
    do {                                                                                                     \
        if ((x) != GPIS_OK) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, gpis_last_error()); exit(4); } \
    } while (0)
int g_rank = 0;

// scene-S medium of the BASELINE configurations (the same values as bindings.py: params_for_config)
gpis_params params_for(const std::string &config)
{
    gpis_params p;
    gpis_default_params(&p);
    for (int c = 0; c < 3; ++c) { p.sigma_a[c] = 0.f; p.sigma_s[c] = 1.f; }
    p.density = 1.f; p.step_size = 0.01f; p.min_step = 8; p.seed = 7; p.max_bounces = 1024;
    p.sigma = 0.1f; p.length_scale = 0.05f; p.local_scale = 3.f;
    p.mean.type = GPIS_MEAN_SPHERICAL; p.mean.radius = 1.f;
    p.isotropic_3d_sampling = 1;
    if (config == "C3") {           // proc_nonstationary, multi-resolution grid, ls ramp bottom_top 0.5 .. 2 over y in [-1, 1], rho 64, per path
        p.impulse_density = 64.f; p.correlation_context = GPIS_CTX_RENEWAL; p.single_realization = 0;
        p.nonstationary = 1; p.multi_resolution_grid = 1; p.ls_ramp_type = GPIS_RAMP_BOTTOM_TOP;
        p.ls_min = 0.5; p.ls_max = 2.0; p.ls_start = -1.0; p.ls_end = 1.0;
    } else {                        // C1: the headline
        p.impulse_density = 32.f; p.correlation_context = GPIS_CTX_RENEWAL; p.single_realization = 1;
    }
    return p;
}

// image rows of tile row t
inline void tile_rows(const gpis_scene_s &s, uint32_t t, uint32_t &y0, uint32_t &ny)
{
    y0 = s.y_begin + t * s.tile_size;
    const uint32_t end = s.y_begin + s.y_count;
    ny = y0 + s.tile_size <= end ? s.tile_size : end - y0;
}

int run_rank(const Options &o, int rank, int world, const ncclUniqueId &id)
{
    g_rank = rank;
    CHECK_HIP(hipSetDevice(rank));
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    // parameter block: rank 0's, broadcast
    gpis_params p = params_for(o.config);
    if (rank != 0) memset(&p, 0, sizeof p);
    void *d_p;
    CHECK_HIP(hipMalloc(&d_p, sizeof p));
    CHECK_HIP(hipMemcpy(d_p, &p, sizeof p, hipMemcpyHostToDevice));
    CHECK_NCCL(ncclBroadcast(d_p, d_p, sizeof p, ncclChar, 0, comm, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(&p, d_p, sizeof p, hipMemcpyDeviceToHost));

    gpis_medium *m = nullptr;
    CHECK_GPIS(gpis_create(&p, rank, &m));
    bool guided = false;
    if (o.guide_half > 0 && p.single_realization) {
        if (gpis_build_guide(m, o.guide_half, o.guide_ppc) == GPIS_OK) guided = true;
        else fprintf(stderr, "rank %d: no guide field (%s)\n", rank, gpis_last_error());
    }

    gpis_scene_s scene;
    gpis_default_scene_s(&scene, (uint32_t)o.width, (uint32_t)o.height, (uint32_t)o.spp);
    scene.shard_index = (uint32_t)rank;
    scene.shard_count = (uint32_t)world;
    // 16-pixel tile rows (the reference's tile size) unless that leaves a rank with fewer than 16 of them: 1080 rows over 8 ranks are
    // 9 rows for four ranks and 8 for the others in 16-pixel rows (6 % imbalance), 17 and 16 in 8-pixel rows (dist.py: rows_tile)
    while (scene.tile_size > 1 && scene.y_count / scene.tile_size < 16u * (uint32_t)world) scene.tile_size /= 2;
    const size_t n_pix = (size_t)o.width * (size_t)o.height;
    float *d_rad;
    CHECK_HIP(hipMalloc((void **)&d_rad, n_pix * sizeof(float)));
    double *d_t;                                    // [elapsed max over ranks]
    CHECK_HIP(hipMalloc((void **)&d_t, 2 * sizeof(double)));

    const uint32_t n_tile_rows = (scene.y_count + scene.tile_size - 1) / scene.tile_size;
    size_t wire_bytes = 0;
    auto barrier = [&]() {                          // an all-reduce of one word is the barrier
        CHECK_NCCL(ncclAllReduce(d_t + 1, d_t + 1, 1, ncclDouble, ncclSum, comm, st));
        CHECK_HIP(hipStreamSynchronize(st));
    };
    double render_s = 0., gather_s = 0.;
    auto frame = [&]() {
        const auto a = std::chrono::steady_clock::now();
        CHECK_HIP(hipMemsetAsync(d_rad, 0, n_pix * sizeof(float), st));
        CHECK_GPIS(gpis_render_scene_s(m, &scene, d_rad, nullptr, st));
        if (world > 1) CHECK_HIP(hipStreamSynchronize(st));       // so that the two clocks below mean what they say
        const auto b = std::chrono::steady_clock::now();
        if (world > 1) {
            wire_bytes = 0;
            CHECK_NCCL(ncclGroupStart());
            for (uint32_t t = 0; t < n_tile_rows; ++t) {
                const int owner = (int)(t % (uint32_t)world);
                if (owner == 0) continue;
                uint32_t y0, ny;
                tile_rows(scene, t, y0, ny);
                float *rows = d_rad + (size_t)y0 * (size_t)o.width;
                const size_t count = (size_t)ny * (size_t)o.width;
                if (rank == owner) { CHECK_NCCL(ncclSend(rows, count, ncclFloat, 0, comm, st)); wire_bytes += count * sizeof(float); }
                else if (rank == 0) { CHECK_NCCL(ncclRecv(rows, count, ncclFloat, owner, comm, st)); wire_bytes += count * sizeof(float); }
            }
            CHECK_NCCL(ncclGroupEnd());
            CHECK_HIP(hipStreamSynchronize(st));
        }
        const auto c = std::chrono::steady_clock::now();
        render_s += std::chrono::duration<double>(b - a).count();
        gather_s += std::chrono::duration<double>(c - b).count();
    };

    for (int i = 0; i < o.warmup; ++i) frame();
    render_s = gather_s = 0.;
    CHECK_HIP(hipDeviceSynchronize());
    barrier();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < o.steps; ++i) frame();
    CHECK_HIP(hipDeviceSynchronize());
    barrier();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // max over ranks of the elapsed time; per-rank render / gather seconds gathered on rank 0
    CHECK_HIP(hipMemcpy(d_t, &dt, sizeof dt, hipMemcpyHostToDevice));
    CHECK_NCCL(ncclAllReduce(d_t, d_t, 1, ncclDouble, ncclMax, comm, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipMemcpy(&dt, d_t, sizeof dt, hipMemcpyDeviceToHost));
    double mine[2] = {render_s / o.steps, gather_s / o.steps};
    double *d_all;
    CHECK_HIP(hipMalloc((void **)&d_all, (size_t)world * 2 * sizeof(double)));
    CHECK_HIP(hipMemcpy(d_all + 2 * rank, mine, sizeof mine, hipMemcpyHostToDevice));
    CHECK_NCCL(ncclAllGather(d_all + 2 * rank, d_all, 2, ncclDouble, comm, st));
    CHECK_HIP(hipStreamSynchronize(st));
    std::vector<double> all((size_t)world * 2);
    CHECK_HIP(hipMemcpy(all.data(), d_all, all.size() * sizeof(double), hipMemcpyDeviceToHost));

    if (rank == 0) {
        // a checksum of the assembled frame, so that two runs (or this program and bench.py) can be compared
        std::vector<float> rad(n_pix);
        CHECK_HIP(hipMemcpy(rad.data(), d_rad, n_pix * sizeof(float), hipMemcpyDeviceToHost));
        double sum = 0.;
        for (float v : rad) sum += (double)v;
        const double samples = (double)n_pix * (double)o.spp * (double)o.steps;
        printf("{\"metric\": \"Msamples/s (primary rays x spp / s)\", \"value\": %.6f, \"unit\": \"Msamples/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
               "\"ms_per_step\": %.3f, \"higher_is_better\": true, \"scaling\": \"strong\", \"vs_baseline\": null, \"dtype\": \"f32\", \"data\": \"synthetic\", "
               "\"config\": {\"workload\": \"%s: scene S %dx%d, %d spp\", \"sharding\": \"16-pixel tile rows round-robin, one batch per rank, one gather of "
               "disjoint rows (RCCL send/recv group)\", \"guide\": %s, \"driver\": \"host/gpis_multi_gpu.cpp\"}, \"radiance_sum\": %.17g, \"wire_bytes_per_frame\": %zu, \"per_rank\": [",
               samples / dt / 1e6, world, o.steps, o.warmup, dt / o.steps * 1e3, o.config.c_str(), o.width, o.height, o.spp,
               guided ? "true" : "false", sum, wire_bytes);
        for (int r = 0; r < world; ++r)
            printf("%s{\"rank\": %d, \"render_ms\": %.3f, \"gather_ms\": %.3f}", r ? ", " : "", r, all[2 * r] * 1e3, all[2 * r + 1] * 1e3);
        printf("]}\n");
        fflush(stdout);
    }
    gpis_destroy(m);
    CHECK_HIP(hipFree(d_rad)); CHECK_HIP(hipFree(d_t)); CHECK_HIP(hipFree(d_all)); CHECK_HIP(hipFree(d_p));
    ncclCommDestroy(comm);
    return 0;
}

}   // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "missing value after %s\n", a.c_str()); exit(1); } return argv[++i]; };
        if (a == "--gpus") o.gpus = atoi(next());
        else if (a == "--steps") o.steps = atoi(next());
        else if (a == "--warmup") o.warmup = atoi(next());
        else if (a == "--width") o.width = atoi(next());
        else if (a == "--height") o.height = atoi(next());
        else if (a == "--spp") o.spp = atoi(next());
        else if (a == "--config") o.config = next();
        else if (a == "--guide") {
            const std::string g = next();
            if (g == "off") o.guide_half = 0;
            else if (sscanf(g.c_str(), "%d:%d", &o.guide_half, &o.guide_ppc) != 2) { fprintf(stderr, "--guide half:ppc | off\n"); return 1; }
        } else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
    }
    if (o.gpus < 1 || o.steps < 1 || o.warmup < 0 || (o.config != "C1" && o.config != "C3")) { fprintf(stderr, "bad arguments\n"); return 1; }
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);    // dmabuf IPC (the only mode the host driver supports)

    // pipes from rank 0 to every other rank for the ncclUniqueId; forked before any HIP / RCCL call
    const int world = o.gpus;
    std::vector<int> rd((size_t)world, -1), wr((size_t)world, -1);
    for (int r = 1; r < world; ++r) {
        int fd[2];
        if (pipe(fd) != 0) { perror("pipe"); return 1; }
        rd[r] = fd[0]; wr[r] = fd[1];
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) { perror("fork"); return 1; }
        if (pid == 0) {
            ncclUniqueId id;
            if (r == 0) {
                g_rank = 0;
                CHECK_NCCL(ncclGetUniqueId(&id));
                for (int k = 1; k < world; ++k)
                    if (write(wr[k], &id, sizeof id) != (ssize_t)sizeof id) { perror("write"); _exit(5); }
            } else {
                if (read(rd[r], &id, sizeof id) != (ssize_t)sizeof id) { perror("read"); _exit(5); }
            }
            _exit(run_rank(o, r, world, id));
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    return rc;
}
