// gpis_hip.hip — kernels and C ABI (include/gpis.h) of the MI355X sparse-convolution GPIS path.
//
// Build (see __graft_entry__.build):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -I include -o libgpis_hip.so gpis_hip.hip
//
// Layout in HBM: rays and results are arrays of the 128-/96-byte PODs of gpis.h (a wave reads
// 64 consecutive records = 8 KiB / 6 KiB contiguous); the medium's constants live in one
// DevModel block read through scalar loads; counters are two 64-bit words updated once per wave.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "gpis.h"
#include "gpis_device.hpp"
#include "gpis_fast.hpp"      // FastTable, fast_supported (no kernel of it is instantiated here)
#include "gpis_guide.hpp"     // GuideField, guide_free
#include "gpis_launch.hpp"    // the march kernels live in the tu_*.hip translation units

#pragma clang fp contract(off)

using namespace gpis;
using gpis::launch::grid_of;
constexpr int kBlock = 64;   // one wave per workgroup (gpis_lane.hpp)
constexpr unsigned kPersistSlots = 256;   // ring of ray counters: one per persistent launch in flight

// ======================================================================================
// errors
// ======================================================================================
static thread_local char g_err[512] = "";

static int set_err(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(GPIS_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *gpis_last_error(void) { return g_err; }

// ======================================================================================
// handle
// ======================================================================================
struct SceneConst {   // host-precomputed scene-S constants (host libm: tanf, normalisation)
    gpis_scene_s s;
    float plane_dist, ratio, psx;
    float light[3];
};

// Pixel `local` of a driver call -> index y*width + x in the image.  A call renders the image rows
// [y_begin, y_begin + y_count); with shard_count > 1 only the tile rows t (tile_size pixels high, counted
// from y_begin) with t % shard_count == shard_index — the interleaved tile-row split of the multi-GPU
// driver, kept in ONE batch per rank so that every stage stays one launch.
__host__ __device__ inline size_t scene_pixel(const gpis_scene_s &s, size_t local)
{
    if (s.shard_count <= 1u)
        return (size_t)s.y_begin * s.width + local;
    const size_t ly = local / s.width, x = local % s.width;
    const size_t lt = ly / s.tile_size, r = ly % s.tile_size;
    return ((size_t)s.y_begin + (lt * s.shard_count + s.shard_index) * s.tile_size + r) * s.width + x;
}
// rows this call renders
static size_t scene_rows(const gpis_scene_s &s)
{
    if (s.shard_count <= 1u)
        return s.y_count;
    size_t rows = 0;
    for (size_t t = s.shard_index, y0 = (size_t)s.shard_index * s.tile_size; y0 < s.y_count; t += s.shard_count, y0 = t * s.tile_size)
        rows += (s.y_count - y0 < s.tile_size) ? s.y_count - y0 : s.tile_size;
    return rows;
}
static bool scene_args_ok(const gpis_scene_s *s)
{
    return s->width > 0 && s->height > 0 && s->spp_count > 0 && s->y_begin + s->y_count <= s->height &&
           (s->shard_count <= 1u || (s->shard_index < s->shard_count && s->tile_size > 0));
}

struct gpis_medium {
    gpis_params params;
    DevModel host_model;
    gpis_derived derived;
    int device;
    DevModel *d_model;
    Counters *d_counters;
    FastTable fast;          // single-realization wave-cooperative path (gpis_fast.hpp); enabled == 0 when unused
    GuideField guide;        // certified guide field (gpis_guide.hpp); enabled == 0 until gpis_build_guide
    unsigned long long *d_guide_cnt;
    uint64_t selfcheck_tabulated = 0;   // points of the last gpis_guide_selfcheck that fell into tabulated bricks
    GuideField *d_guide;     // device copy of `guide` (the resident guided kernels read it through scalar loads instead of 14 kernel-argument SGPRs)
    float *d_grid_vox = nullptr;      // GridNonstationaryCovariance voxels (gpis_set_variance_grid)
    unsigned lambert_calls = 0;       // gpis_render_scene_s calls so far (the first one works in small chunks, see lambert_ws_plan)
    void *fs_ws = nullptr;            // function-space workspace: one FsGlob per resident workgroup (gpis_fs.hpp)
    unsigned fs_ws_blocks = 0;
    // staging for the *_host entries and workspace for the renderer (grown on demand)
    // slots 0-2: *_host staging; 3: renderer workspace (Lambert driver: primary rays + jitters + mask); 4: wavefront march;
    // 5, 6: the Lambert driver's segment results and its shadow-ray arrays — separate allocations, so that each can be made while
    // the kernel that does not need it yet runs (gpis_render_scene_s) or ahead of time from another thread (gpis_reserve_scene_workspace)
    static constexpr int kStageSlots = 7;
    void *stage[kStageSlots];
    size_t stage_bytes[kStageSlots];
    std::mutex stage_mu[kStageSlots];   // guards the growth of one slot (the reserve entry does not take `mu`)
    std::mutex mu;
    std::mutex fs_host_mu;   // serialises the function-space host entries, which own fs_stage[] end to end
    void *fs_stage[3] = {nullptr, nullptr, nullptr};
    size_t fs_stage_bytes[3] = {0, 0, 0};
    // optional per-kernel timing (gpis_set_profiling): event pairs around each march launch
    int batch_hint;          // gpis_set_batch_order: which form of the guided march the *_batch / *_host entries use
    // tuning options (gpis_set_option; defaults from the GPIS_* environment variables read ONCE in gpis_create)
    long long opt[GPIS_OPT_COUNT_];
    // the renderer's workspace (stage[3]) and the wavefront march's (stage[4]) are shared by all callers of this
    // handle: the last user records an event, the next one makes its stream wait for it
    hipEvent_t ws_event[2];
    bool ws_event_set[2];
    // persistent march (gpis_persist.inc): ring of ray counters (one per launch in flight), resident-wave budget
    // pipelined host path (gpis_sample_distance_host / gpis_transmittance_host): two slots, each with its own stream,
    // pinned staging and device buffers, so that chunk k's H2D copy overlaps chunk k-1's kernel and chunk k-2's D2H
    struct HostSlot {
        hipStream_t stream;
        hipEvent_t done;
        char *pin_in, *pin_out, *pin_aux;       // pinned host staging (used when the caller's memory is pageable)
        char *dev_in, *dev_out, *dev_aux;
        size_t cap;                             // chunk capacity in records
    } hs[2];
    unsigned int *d_next;
    unsigned next_slot;
    int persist_waves[8];    // cached occupancy * CUs per kernel instance; 0 = not queried yet
    int n_cus;
    bool profiling;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events[3];     // 0: sampleDistance, 1: transmittance, 2: the NEE driver's neePDF / neeGrad launches
    size_t events_used[3];
    double prof_ms[3];
    uint64_t prof_launches[3];
};

static int ensure_stage(gpis_medium *m, int slot, size_t bytes, bool exact = false)
{
    std::lock_guard<std::mutex> lock(m->stage_mu[slot]);
    if (m->stage_bytes[slot] >= bytes)
        return GPIS_OK;
    if (m->stage[slot])
        HIP_TRY(hipFree(m->stage[slot]));
    m->stage[slot] = nullptr;
    m->stage_bytes[slot] = 0;
    size_t want = exact ? bytes : bytes + bytes / 4 + 4096;     // exact: whole-frame workspaces (tens of GB; allocation time is per byte)
    HIP_TRY(hipMalloc(&m->stage[slot], want));
    m->stage_bytes[slot] = want;
    return GPIS_OK;
}
static size_t stage_size(gpis_medium *m, int slot)
{
    std::lock_guard<std::mutex> lock(m->stage_mu[slot]);
    return m->stage_bytes[slot];
}

// ======================================================================================
// host-side "fromJson": precompute the constants the reference computes once
// (GPF.cpp:654-679, 696-709, 741-760; SCN.cpp:8-37; GPM.cpp:152-158)
// ======================================================================================
static inline float hsum3e(float a, float b, float c) { return a + (b + c); }
#define HM(m, r, c) ((m)[3 * (r) + (c)])
static float hcof(const float *m, int i, int j)
{
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return HM(m, i1, j1) * HM(m, i2, j2) - HM(m, i1, j2) * HM(m, i2, j1);
}
static void hinverse3(const float *m, float *r)
{
    float c0 = hcof(m, 0, 0), c1 = hcof(m, 1, 0), c2 = hcof(m, 2, 0);
    float det = hsum3e(c0 * HM(m, 0, 0), c1 * HM(m, 1, 0), c2 * HM(m, 2, 0));
    float invdet = 1.0f / det;
    HM(r, 1, 0) = hcof(m, 0, 1) * invdet; HM(r, 1, 1) = hcof(m, 1, 1) * invdet; HM(r, 2, 0) = hcof(m, 0, 2) * invdet;
    HM(r, 1, 2) = hcof(m, 2, 1) * invdet; HM(r, 2, 1) = hcof(m, 1, 2) * invdet; HM(r, 2, 2) = hcof(m, 2, 2) * invdet;
    HM(r, 0, 0) = c0 * invdet; HM(r, 0, 1) = c1 * invdet; HM(r, 0, 2) = c2 * invdet;
}
static float hdet3(const float *m)
{
    auto h = [&](int a, int b, int c) { return HM(m, 0, a) * (HM(m, 1, b) * HM(m, 2, c) - HM(m, 1, c) * HM(m, 2, b)); };
    return h(0, 1, 2) - h(1, 0, 2) + h(2, 0, 1);
}

static int build_model(const gpis_params &P, DevModel &M, gpis_derived &D)
{
    memset(&M, 0, sizeof M);
    memset(&D, 0, sizeof D);
    M.single_realization = P.single_realization != 0;
    M.iso3d = P.isotropic_3d_sampling != 0;
    M.sampling_1d = P.sampling_1d != 0;
    M.correlation_xy = P.correlation_xy != 0;
    M.ctx = P.correlation_context;
    M.activate_conditioning = !M.single_realization && (M.ctx == GPIS_CTX_RENEWAL || M.ctx == GPIS_CTX_RENEWAL_PLUS);
    M.scheme_1d_eff = (!M.single_realization && M.sampling_1d) ? P.scheme_1d : GPIS_UNI;
    M.nonstationary = P.nonstationary != 0;
    M.multi_resolution_grid = P.multi_resolution_grid != 0;
    M.multi_res = M.nonstationary && M.multi_resolution_grid;
    M.use_aniso_mtx = P.use_aniso_mtx != 0;
    M.surf_vol_phase_separate = P.surf_vol_phase_separate != 0;
    M.surf_vol_phase_amp_thresh = P.surf_vol_phase_amp_thresh;
    M.has_mean_additional = P.has_mean_additional != 0;
    M.max_bounces = P.max_bounces;
    M.seed = P.seed;
    M.n_impulses = (uint32_t)P.impulse_density;
    M.min_step = P.min_step;
    M.step_size = P.step_size;
    M.impulse_density = P.impulse_density;
    M.sigma = M.nonstationary ? (float)(1.0 * (double)P.sigma) : P.sigma;
    M.sigma_raw = P.sigma;

    // SquaredExponentialCovariance::fromJson
    float l_conv = P.length_scale * sqrtf(2.f) / 2;
    float l2w[9] = {0}, w2l[9] = {0};
    float l_aniso[3] = {0, 0, 0};
    if (!M.use_aniso_mtx) {
        for (int i = 0; i < 3; ++i) {
            l_aniso[i] = l_conv * P.aniso[i];
            float inv = 1.0f / l_aniso[i];
            if (std::isinf(inv) || std::isnan(inv)) inv = 0;
            HM(l2w, i, i) = l_aniso[i];
            HM(w2l, i, i) = inv;
            HM(M.invcov_world, i, i) = inv * inv;
        }
        M.cov_det_sqrt_world = (double)(l_aniso[0] * l_aniso[1] * l_aniso[2]);
        float a = l_aniso[0] > l_aniso[1] ? l_aniso[0] : l_aniso[1];
        M.mtx_factor = a > l_aniso[2] ? a : l_aniso[2];
    } else {
        for (int i = 0; i < 9; ++i) l2w[i] = l_conv * P.aniso_mtx[i];
        hinverse3(l2w, w2l);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                HM(M.invcov_world, r, c) = hsum3e(HM(w2l, 0, r) * HM(w2l, 0, c), HM(w2l, 1, r) * HM(w2l, 1, c), HM(w2l, 2, r) * HM(w2l, 2, c));
        float det = hdet3(M.invcov_world);
        M.cov_det_sqrt_world = 1.0 / (double)sqrtf(det);
        float e0 = (HM(l2w, 0, 0) + HM(l2w, 0, 1)) + HM(l2w, 0, 2);
        float e1 = (HM(l2w, 1, 0) + HM(l2w, 1, 1)) + HM(l2w, 1, 2);
        float e2 = (HM(l2w, 2, 0) + HM(l2w, 2, 1)) + HM(l2w, 2, 2);
        float a = e0 > e1 ? e0 : e1;
        M.mtx_factor = a > e2 ? a : e2;
    }
    memcpy(M.w2l, w2l, sizeof w2l);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            HM(M.w2l_T, r, c) = HM(w2l, c, r);
            HM(M.l2w_T, r, c) = HM(l2w, c, r);
        }
    M.kernel_scale = P.local_scale;
    M.pi_pow_1_5 = pow(M_PI, 1.5);
    M.sqrt_pi = sqrt(M_PI);

    // ramp + multi-resolution tables (host libm so that they equal the reference's values)
    M.ls_ramp_type = P.ls_ramp_type;
    M.ls_min = P.ls_min; M.ls_max = P.ls_max;
    M.ls_scale = 1.0 / (P.ls_end - P.ls_start);
    M.ls_offset = -P.ls_start * M.ls_scale;
    { double mn = P.ls_min + 1., mx = P.ls_max + 1.; M.ls_log_min2 = log(mn * mn); M.ls_log_max2 = log(mx * mx); }
    M.ls_maxval = (float)(P.ls_max > P.ls_min ? P.ls_max : P.ls_min);
    // the procedural fields in their general form (host libm for the logs, as the reference computes them per call)
    auto make_ramp = [](int enabled, int type, double mn, double mx, double st, double en, double mn2, double mx2, double st2, double en2) {
        DevRamp R;
        memset(&R, 0, sizeof R);
        R.enabled = enabled; R.type = type;
        R.scale = 1.0 / (en - st); R.offset = -st * R.scale;
        R.scale2 = 1.0 / (en2 - st2); R.offset2 = -st2 * R.scale2;
        const double a = mn + 1., b = mx + 1., a2 = mn2 + 1., b2 = mx2 + 1.;
        R.log_min2 = log(a * a); R.log_max2 = log(b * b);
        R.log2_min2 = log(a2 * a2); R.log2_max2 = log(b2 * b2);
        R.vmin = mn; R.vmax = mx;
        return R;
    };
    M.ls = make_ramp(M.nonstationary, P.ls_ramp_type, P.ls_min, P.ls_max, P.ls_start, P.ls_end, P.ls_min2, P.ls_max2, P.ls_start2, P.ls_end2);
    M.var = make_ramp(P.var.enabled != 0, P.var.type, P.var.min, P.var.max, P.var.start, P.var.end, P.var.min2, P.var.max2, P.var.start2, P.var.end2);
    M.aniso = make_ramp(P.aniso_field.enabled != 0, P.aniso_field.type, P.aniso_field.min, P.aniso_field.max, P.aniso_field.start, P.aniso_field.end,
                        P.aniso_field.min2, P.aniso_field.max2, P.aniso_field.start2, P.aniso_field.end2);
    M.color = make_ramp(P.mean_color.enabled != 0, P.mean_color.type, P.mean_color.min, P.mean_color.max, P.mean_color.start, P.mean_color.end,
                        P.mean_color.min2, P.mean_color.max2, P.mean_color.start2, P.mean_color.end2);
    M.emission = make_ramp(P.mean_emission.enabled != 0, P.mean_emission.type, P.mean_emission.min, P.mean_emission.max, P.mean_emission.start, P.mean_emission.end,
                           P.mean_emission.min2, P.mean_emission.max2, P.mean_emission.start2, P.mean_emission.end2);
    if (P.ls_ramp_type == GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT)      // ProceduralNoiseVec::maxVal, GPF.cpp:124-138
        M.ls_maxval = (float)((P.ls_max > P.ls_min ? P.ls_max : P.ls_min) * (P.ls_max2 > P.ls_min2 ? P.ls_max2 : P.ls_min2));
    if (P.ls_ramp_type == GPIS_NOISE_SANDSTONE || P.ls_ramp_type == GPIS_NOISE_RUST)
        M.ls_maxval = 1.f;
    {   // sandstone / rust anywhere: the launcher routes the medium to the all-features path instance (gpis_device.hpp: GPIS_FLAG_fbm_noise)
        const DevRamp *fields[5] = {&M.ls, &M.var, &M.aniso, &M.color, &M.emission};
        M.fbm_noise = 0;
        for (const DevRamp *f : fields)
            if (f->enabled && f->type >= GPIS_NOISE_SANDSTONE) M.fbm_noise = 1;
    }
    {   // GridNonstationaryCovariance (GPF.cpp:1326-1427): variance and kernel scale from a voxel grid (gpis_set_variance_grid)
        memset(&M.grid, 0, sizeof M.grid);
        M.grid.on = P.grid_nonstationary != 0;
        M.grid.offset = P.grid_offset; M.grid.scale = P.grid_scale;
        M.grid.separate = P.grid_surf_vol_amp_separate != 0;
        M.grid.thresh = P.grid_surf_vol_amp_thresh;
        M.grid.surf_amp = P.grid_surf_amp_scale; M.grid.vol_amp = P.grid_vol_amp_scale;
        M.grid.surf_ls = P.grid_surf_ls_scale; M.grid.vol_ls = P.grid_vol_ls_scale;
        if (M.grid.on) {
            M.fbm_noise = 1;                                     // all-features path instance
            M.ls.enabled = 0;
            // sparseConvNoiseMaxLateralScale, GPF.cpp:1422-1427
            M.ls_maxval = M.grid.separate ? (P.grid_surf_ls_scale < P.grid_vol_ls_scale ? P.grid_vol_ls_scale : P.grid_surf_ls_scale) : 1.f;
        }
    }
    const float base = 2.5f;
    M.log_base = logf(base);
    for (int l = kLevelMin; l <= kLevelMax; ++l) {
        float sc = powf(base, (float)l);
        M.level_scale[l - kLevelMin] = sc;
        M.level_addseed[l - kLevelMin] = (int)floorf(logf(sc) / logf(base));
    }
    float wss = M.nonstationary ? M.ls_maxval : 1.f;
    M.world_addseed = (int)floorf(logf(wss) / logf(base));

    // stationary constants
    auto variance3d = [&](float dens, float R, bool isIdentity, float globalScale, float localScale) {
        double idua = dens / (R * R * R);
        double cds = 1.0;
        if (!isIdentity) {
            cds = M.cov_det_sqrt_world;
            cds *= pow(globalScale, 3);
        }
        cds *= pow(localScale, 3);
        return (float)(idua * (M.pi_pow_1_5 * cds));
    };
    M.radius_iso = M.kernel_scale;
    M.radius_world = M.kernel_scale * 1.0f * M.mtx_factor;
    M.norm3d_world = sqrtf(variance3d(P.impulse_density, M.radius_world, false, 1.0f, 1.0f));
    {   // bound on ab^T A ab inside the unit ball, for the grid space the cooperative kernels use (A as in coop_eval_noise3d)
        const float amax = M.iso3d ? 0.5f : 0.5f * fmaxf(M.invcov_world[0], fmaxf(M.invcov_world[4], M.invcov_world[8]));
        const float R = M.iso3d ? M.kernel_scale : M.radius_world;
        const float v = amax * R * R * 1.01f;
        M.exp_arg_max = (v == v && v >= 0.f) ? v : 3.4e38f;
    }
    M.norm3d_iso = sqrtf(variance3d(P.impulse_density, M.radius_iso, true, 1.0f, 1.0f));
    // Matérn / Gabor kernels: their own radius and variance (GPF.cpp:1020-1046, 1127-1138, 1192-1203); host libm as the reference
    M.kernel_type = P.kernel_type;
    M.matern_v = P.matern_v;
    M.k_l = P.length_scale;
    M.fs_n = P.fs_sample_points;
    M.fs_step = P.fs_step_size;
    for (int i = 0; i < 3; ++i) M.fs_aniso[i] = P.aniso[i];
    M.gabor_a = (float)(1.0 / P.gabor_a_inv);
    M.gabor_f = (float)(1.0 / P.gabor_f_inv);
    {
        float l2 = 0.f; l2 += P.gabor_omega[0] * P.gabor_omega[0]; l2 += P.gabor_omega[1] * P.gabor_omega[1]; l2 += P.gabor_omega[2] * P.gabor_omega[2];
        const float inv = 1.0f / sqrtf(l2);
        for (int i = 0; i < 3; ++i) M.gabor_omega[i] = P.gabor_omega[i] * inv;
    }
    if (P.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL) {
        const float l = P.length_scale, a = M.gabor_a, f = M.gabor_f;
        double iks;
        if (P.kernel_type == GPIS_KERNEL_MATERN) {
            float la[3];
            for (int i = 0; i < 3; ++i) {
                la[i] = l / sqrtf(P.aniso[i]);
                if (std::isinf(la[i]) || std::isnan(la[i])) la[i] = 0;
            }
            float mx = la[0] > la[1] ? la[0] : la[1];
            mx = mx > la[2] ? mx : la[2];
            M.radius_world = (float)(M.kernel_scale * 1.0f * sqrt(2) / 2 * mx);
            iks = P.matern_v == 0.5 ? 2.0 * M_PI * l : (P.matern_v == 1.5 ? pow(M_PI * l, 3) / (24 * sqrt(3)) : M_PI * pow(l, 3) / (5 * sqrt(5)));      // GPF.cpp:1029-1046
        } else if (P.kernel_type == GPIS_KERNEL_GABOR_ANISO) {
            M.radius_world = (float)(M.kernel_scale * sqrt(2) / 2 * 1.0 / a);
            const float q = f / a;
            iks = pow(1.0 / a, 3) * (1 + exp(-2.0 * M_PI * (q * q))) / (4 * sqrt(2));
        } else {
            M.radius_world = (float)(M.kernel_scale * sqrt(2) / 4 * 1.0 / a);
            iks = 2 * sqrt(2) * M_PI * (f * f) / a * (1 - exp(-2 * M_PI * f / (a * a)));
        }
        const float R = M.radius_world;
        const double idua = P.impulse_density / (R * R * R);
        M.norm3d_world = sqrtf((float)(idua * iks));
        M.radius_iso = M.kernel_scale;      // splattingKernelRadius(true, .) of these kernels; unused (world space only)
    }
    {
        double idua = P.impulse_density / M.radius_iso;
        M.norm1d = sqrtf((float)(idua * (M.sqrt_pi * 1.0f)));
    }
    // prepareForRender
    bool all_zero = true;
    for (int c = 0; c < 3; ++c) {
        float sa = P.sigma_a[c] * P.density, ss = P.sigma_s[c] * P.density;
        float st = sa + ss;
        M.sigma_s_over_t[c] = ss / st;
        if (ss != 0.0f) all_zero = false;
    }
    M.absorption_only = all_zero;
    M.mean[0] = P.mean;
    M.mean[1] = P.mean_additional;
    for (int w = 0; w < 2; ++w) {
        const gpis_mean &mu = M.mean[w];
        double l2 = 0.; l2 += mu.dir[0] * mu.dir[0]; l2 += mu.dir[1] * mu.dir[1]; l2 += mu.dir[2] * mu.dir[2];
        double len = sqrt(l2);
        double inv = len > 0 ? 1.0 / len : 0.0;
        for (int k = 0; k < 3; ++k) M.lin_dir[w][k] = mu.dir[k] * inv;
    }

    memcpy(D.world_to_local, w2l, sizeof w2l);
    memcpy(D.local_to_world, l2w, sizeof l2w);
    if (!M.nonstationary) {
        D.kernel_radius_world = M.radius_world;
        D.kernel_radius_iso = M.radius_iso;
        D.norm3d_world = M.norm3d_world;
        D.norm3d_iso = M.norm3d_iso;
        D.norm1d = M.norm1d;
    } else {
        float ls = (float)((double)1.0f * (M.multi_resolution_grid ? 1.0 : (double)M.ls_maxval));
        ls *= M.aniso.enabled ? 1.5f : 1.f;          // sparseConvNoiseMaxAnisotropyScale(), GPF.cpp:1743-1747
        D.kernel_radius_world = M.kernel_scale * ls * M.mtx_factor;
        D.kernel_radius_iso = M.kernel_scale;
        D.norm3d_world = 0.f; D.norm3d_iso = 0.f; D.norm1d = 0.f;   // position dependent: not constants of the medium
    }
    D.impulses_per_cell = M.n_impulses;
    D.activate_conditioning = M.activate_conditioning;
    D.effective_scheme_1d = M.scheme_1d_eff;
    D.multi_resolution = M.multi_res;
    return GPIS_OK;
}

// MeanFunction::color / emission at double-precision points (GPF.hpp:849-857; ramp noises: three equal components)
__global__ void __launch_bounds__(256) k_mean_color_emission(const DevModel *__restrict__ Mp, size_t n, const double *__restrict__ p3,
                                                             float *__restrict__ color3, float *__restrict__ emission3)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevModel &M = *Mp;
    const V3d p{p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]};
    if (color3) {
        double c[3] = {1., 1., 1.};
        if (M.color.enabled) field_vec(M.color, p, true, c);
        for (int k = 0; k < 3; ++k) color3[3 * i + k] = (float)c[k];
    }
    if (emission3) {
        double e[3] = {0., 0., 0.};
        if (M.emission.enabled) field_vec(M.emission, p, true, e);
        for (int k = 0; k < 3; ++k) emission3[3 * i + k] = (float)e[k];
    }
}
__global__ void k_xxhash32(size_t n, int arity, const uint32_t *__restrict__ w, uint32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *p = w + i * (size_t)arity;
    out[i] = arity == 1 ? xxhash32_1(p[0]) : arity == 2 ? xxhash32_2(p[0], p[1]) : arity == 3 ? xxhash32_3(p[0], p[1], p[2]) : xxhash32_4(p[0], p[1], p[2], p[3]);
}
__global__ void k_pcg32_stream(size_t n, const uint64_t *__restrict__ state, uint32_t count, uint32_t *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg32 s;
    s.set_state(state[i]);
    for (uint32_t k = 0; k < count; ++k)
        out[i * (size_t)count + k] = s.next_i();
}

// ======================================================================================
// scene-S tile → ray-batch driver.  Sample index within a chunk:
// ((y*W + x) * spp_count + k): the 64 lanes of a wave carry consecutive spp of one pixel.
// ======================================================================================
__device__ __forceinline__ bool sphere_chord(V3 o, V3 d, float R, float &t0, float &t1)
{
    double ox = o.x, oy = o.y, oz = o.z, dx = d.x, dy = d.y, dz = d.z;
    double a = dx * dx + dy * dy + dz * dz;
    double b = ox * dx + oy * dy + oz * dz;
    double c = ox * ox + oy * oy + oz * oz - (double)R * (double)R;
    double disc = b * b - a * c;
    if (!(disc > 0.0))
        return false;
    double sq = sqrt(disc);
    double ta = (-b - sq) / a, tb = (-b + sq) / a;
    if (tb <= 0.0)
        return false;
    if (ta < 0.0) ta = 0.0;
    t0 = (float)ta; t1 = (float)tb;
    return true;
}

__global__ void __launch_bounds__(256) k_scene_primary(SceneConst sc, size_t first_pixel, size_t n_samples,
                                                       gpis_ray_in *__restrict__ rays, float *__restrict__ u_shadow, uint8_t *__restrict__ valid)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const gpis_scene_s &s = sc.s;
    size_t pix = scene_pixel(s, first_pixel + i / s.spp_count);
    uint32_t k = (uint32_t)(i % s.spp_count);
    uint32_t x = (uint32_t)(pix % s.width), y = (uint32_t)(pix / s.width);
    uint32_t spp = s.spp_begin + k;
    Pcg32 g;
    g.set_state((uint64_t)(uint32_t)(xxhash32_4(x, y, spp, s.scene_seed) + 1u));
    float jx = normalized_uint(g.next_i()), jy = normalized_uint(g.next_i());
    float u0 = normalized_uint(g.next_i()), u1 = normalized_uint(g.next_i());
    V3 local = normalized(v3(-1.0f + ((float)x + jx) * 2.0f * sc.psx, sc.ratio - ((float)y + jy) * 2.0f * sc.psx, sc.plane_dist));
    V3 d = v3(local.x, local.y, -local.z);
    V3 o = v3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
    gpis_ray_in r;
    memset(&r, 0, sizeof r);
    r.pos[0] = o.x; r.pos[1] = o.y; r.pos[2] = o.z;
    r.dir[0] = d.x; r.dir[1] = d.y; r.dir[2] = d.z;
    r.pixel[0] = x; r.pixel[1] = y; r.spp = spp; r.segment = 0;
    r.scene_seed = s.scene_seed; r.info_t = 0.f; r.u_jitter = u0;
    r.first_scatter = 1;
    float t0 = 0.f, t1 = 0.f;
    bool hit = sphere_chord(o, d, s.bound_radius, t0, t1);
    r.near_t = t0; r.far_t = t1;
    rays[i] = r;
    u_shadow[i] = u1;
    valid[i] = hit ? 1 : 0;
}

// `shadow` may be `prim` itself (the Lambert driver writes each shadow ray over the primary ray it came from: the same thread has
// read that record by then, and nothing reads the primary rays afterwards) — hence no __restrict__ on the two.
__global__ void __launch_bounds__(256) k_scene_shade(SceneConst sc, size_t n_samples, const gpis_ray_in *prim,
                                                     const gpis_seg_out *__restrict__ seg, const float *__restrict__ u_shadow,
                                                     const uint8_t *__restrict__ valid, gpis_ray_in *shadow,
                                                     float *__restrict__ cosl, uint8_t *__restrict__ valid2, uint8_t *__restrict__ hit)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    uint8_t v2 = 0, h = 0;
    float c = 0.f;
    if (valid[i]) {
        gpis_seg_out o = seg[i];
        if (o.ok && !o.exited) {
            h = 1;
            V3 l = v3(sc.light[0], sc.light[1], sc.light[2]);
            double ax = o.aniso[0], ay = o.aniso[1], az = o.aniso[2];
            double len = sqrt(ax * ax + ay * ay + az * az);
            V3 nn = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
            c = dot(nn, l);
            float t0, t1;
            if (c > 0.f && sphere_chord(v3(o.p[0], o.p[1], o.p[2]), l, sc.s.bound_radius, t0, t1)) {
                gpis_ray_in p = prim[i];
                gpis_ray_in sh;
                memset(&sh, 0, sizeof sh);
                sh.pos[0] = o.p[0]; sh.pos[1] = o.p[1]; sh.pos[2] = o.p[2];
                sh.dir[0] = l.x; sh.dir[1] = l.y; sh.dir[2] = l.z;
                sh.near_t = 0.f; sh.far_t = t1;
                sh.pixel[0] = p.pixel[0]; sh.pixel[1] = p.pixel[1]; sh.spp = p.spp;
                sh.segment = p.segment + 1;
                sh.scene_seed = p.scene_seed;
                sh.info_t = p.info_t + o.sample_t;
                sh.u_jitter = u_shadow[i];
                sh.first_scatter = 0;
                sh.bounce = p.bounce + 1;
                sh.last_val = o.last_val;
                sh.last_gp_id = o.gp_id;
                sh.last_aniso[0] = o.aniso[0]; sh.last_aniso[1] = o.aniso[1]; sh.last_aniso[2] = o.aniso[2];
                shadow[i] = sh;
                v2 = 1;
            }
        }
    }
    cosl[i] = c;
    valid2[i] = v2;
    hit[i] = h;
}

// one lane per pixel: sequential sum over its spp, in sample order (matches the CPU estimator)
__global__ void __launch_bounds__(256) k_scene_accumulate(SceneConst sc, size_t first_pixel, size_t n_pixels, const float *__restrict__ cosl,
                                                          const uint8_t *__restrict__ valid2, const uint8_t *__restrict__ vis,
                                                          const uint8_t *__restrict__ hit, float *__restrict__ radiance_sum,
                                                          uint32_t *__restrict__ hit_count)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pixels) return;
    const uint32_t spp = sc.s.spp_count;
    float acc = 0.f;
    uint32_t hits = 0;
    if ((spp & 3u) == 0u) {
        // four samples per load (the arrays start 256-byte aligned and a pixel's run is a multiple of 4): the same sum in the same
        // order with a quarter of the load instructions (a thread walks its own 64-byte lines; 4.7 -> 2 ms per C1 frame)
        for (uint32_t k = 0; k < spp; k += 4) {
            const size_t i = j * spp + k;
            const uint32_t h4 = *reinterpret_cast<const uint32_t *>(hit + i), v4 = *reinterpret_cast<const uint32_t *>(valid2 + i),
                           s4 = *reinterpret_cast<const uint32_t *>(vis + i);
            const float4 c4 = *reinterpret_cast<const float4 *>(cosl + i);
            const float c[4] = {c4.x, c4.y, c4.z, c4.w};
            for (int u = 0; u < 4; ++u) {
                hits += (h4 >> (8 * u)) & 0xFFu;
                if ((v4 >> (8 * u)) & 0xFFu)
                    acc += c[u] * (((s4 >> (8 * u)) & 0xFFu) ? 1.f : 0.f) * sc.s.light_radiance;
            }
        }
    } else {
        for (uint32_t k = 0; k < spp; ++k) {
            size_t i = j * spp + k;
            hits += hit[i];
            if (valid2[i])
                acc += cosl[i] * (vis[i] ? 1.f : 0.f) * sc.s.light_radiance;
        }
    }
    const size_t pix = scene_pixel(sc.s, first_pixel + j);
    radiance_sum[pix] += acc;
    if (hit_count) hit_count[pix] += hits;
}


// --------------------------------------------------------------------------------------
// multi-bounce wavefront driver (gpis_render_scene_s_paths): PathTracer.cpp:62-75,
// TraceBase.cpp:346-386 and 539-563 with a Lambertian micro-surface (BRDFPhaseFunction.cpp:27-96,
// LambertBsdf.cpp:27-47) and a directional (Dirac) light.  Path state lives in SoA arrays; the
// medium kernels are the same batch entries a host integrator would call.
// --------------------------------------------------------------------------------------
struct PathArrays {
    gpis_ray_in *rays;       // current segment of every path (rewritten in place at each bounce)
    gpis_seg_out *seg;
    gpis_ray_in *shadow;
    uint64_t *rng;           // PCG32 state of the sample's stream
    float *throughput, *emission, *contrib;
    uint8_t *alive, *nee, *vis;
};

__global__ void __launch_bounds__(256) k_paths_begin(SceneConst sc, size_t first_pixel, size_t n_samples, PathArrays a)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const gpis_scene_s &s = sc.s;
    size_t pix = scene_pixel(s, first_pixel + i / s.spp_count);
    uint32_t k = (uint32_t)(i % s.spp_count);
    uint32_t x = (uint32_t)(pix % s.width), y = (uint32_t)(pix / s.width);
    uint32_t spp = s.spp_begin + k;
    Pcg32 g;
    g.set_state((uint64_t)(uint32_t)(xxhash32_4(x, y, spp, s.scene_seed) + 1u));
    float jx = normalized_uint(g.next_i()), jy = normalized_uint(g.next_i());
    float u0 = normalized_uint(g.next_i());
    V3 local = normalized(v3(-1.0f + ((float)x + jx) * 2.0f * sc.psx, sc.ratio - ((float)y + jy) * 2.0f * sc.psx, sc.plane_dist));
    V3 d = v3(local.x, local.y, -local.z);
    V3 o = v3(s.cam_pos[0], s.cam_pos[1], s.cam_pos[2]);
    gpis_ray_in r;
    memset(&r, 0, sizeof r);
    r.pos[0] = o.x; r.pos[1] = o.y; r.pos[2] = o.z;
    r.dir[0] = d.x; r.dir[1] = d.y; r.dir[2] = d.z;
    r.pixel[0] = x; r.pixel[1] = y; r.spp = spp; r.segment = 0;
    r.scene_seed = s.scene_seed; r.info_t = 0.f; r.u_jitter = u0;
    r.first_scatter = 1;
    float t0 = 0.f, t1 = 0.f;
    bool hit = sphere_chord(o, d, s.bound_radius, t0, t1);
    r.near_t = t0; r.far_t = t1;
    a.rays[i] = r;
    a.rng[i] = g.state;
    a.throughput[i] = 1.f;
    a.emission[i] = 0.f;
    a.alive[i] = hit ? 1 : 0;
}

// Regrouping of a bounce's segments (k_paths_keys → radix sort → k_paths_gather).  Key = Morton code of
// the lattice cell the segment starts in, computed in the space the medium's grid lives in: in
// isotropic-ray space every ray travels along the lattice's +z axis (SCN.cpp:296-301: the frame's normal
// is the whitened direction), so segments that start in the same lattice cell stay neighbours for
// their whole length, whatever their world-space directions; in world space the direction octant is
// appended.  Dead paths get the largest key: the sort is also the compaction.
__device__ __forceinline__ uint32_t spread3(uint32_t v)   // 9 bits -> every third bit
{
    v &= 0x1FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void __launch_bounds__(256) k_paths_keys(const DevModel *__restrict__ Mp, float cell_size, size_t n, const gpis_ray_in *__restrict__ rays,
                                                    const uint8_t *__restrict__ alive, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t key = 0xFFFFFFFFu;
    if (alive[i]) {
        const DevModel &M = *Mp;
        const V3 p = v3(rays[i].pos[0], rays[i].pos[1], rays[i].pos[2]);
        const V3 d = v3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        V3 u = p;
        if (M.iso3d) {
            const Frame coord = frame_from_normal(normalized(generic::cov_pos_w2l(M, d, 1.0f)));
            u = to_local(coord, generic::cov_pos_w2l(M, p, 1.0f));
        }
        const float inv = 1.0f / cell_size;
        const int cx = (int)floorf(fminf(fmaxf(u.x * inv, -255.f), 255.f)) + 256;
        const int cy = (int)floorf(fminf(fmaxf(u.y * inv, -255.f), 255.f)) + 256;
        const int cz = (int)floorf(fminf(fmaxf(u.z * inv, -255.f), 255.f)) + 256;
        const uint32_t morton = (spread3((uint32_t)cx) << 2) | (spread3((uint32_t)cy) << 1) | spread3((uint32_t)cz);
        const uint32_t oct = M.iso3d ? 0u : ((d.x < 0.f ? 4u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 1u : 0u));
        key = (morton << 3) | oct;
    }
    keys[i] = key;
    vals[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k_paths_gather(size_t n, const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ order,
                                                      const gpis_ray_in *__restrict__ rays, gpis_ray_in *__restrict__ rays_sorted,
                                                      uint8_t *__restrict__ live_sorted)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const bool live = keys_sorted[j] != 0xFFFFFFFFu;
    live_sorted[j] = live ? 1 : 0;
    if (live)
        rays_sorted[j] = rays[order[j]];
}

// after sampleDistance of segment `bounce`: next-event estimation set-up + the bounce itself.
// Slot j of the batch (seg, shadow, nee, contrib, vis, and rays_in) belongs to path i = order[j]
// (identity when order is null); the path state (rng, throughput, emission, alive, next ray) is
// indexed by i.
__global__ void __launch_bounds__(256) k_paths_shade(SceneConst sc, size_t n_samples, int bounce, int max_bounces, float albedo, PathArrays a,
                                                     const uint32_t *__restrict__ order, const gpis_ray_in *__restrict__ rays_in,
                                                     const uint8_t *__restrict__ live)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_samples) return;
    uint8_t nee = 0;
    if (live ? !live[j] : !a.alive[j]) { a.nee[j] = 0; return; }
    const size_t i = order ? (size_t)order[j] : j;
    const gpis_seg_out o = a.seg[j];
    if (!o.ok) { a.alive[i] = 0; a.nee[j] = 0; return; }
    float thr = a.throughput[i] * o.weight[0];
    if (o.exited) { a.alive[i] = 0; a.nee[j] = 0; a.throughput[i] = thr; return; }
    const gpis_ray_in ray = rays_in[j];
    Pcg32 g;
    g.state = a.rng[i];
    const V3 l = v3(sc.light[0], sc.light[1], sc.light[2]);
    const double ax = o.aniso[0], ay = o.aniso[1], az = o.aniso[2];
    const double len = sqrt(ax * ax + ay * ay + az * az);
    const V3 n = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
    const Frame fr = frame_from_normal(n);
    const V3 dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
    const V3 wi = normalized(to_local(fr, v3(-dir.x, -dir.y, -dir.z)));
    const V3 p = v3(o.p[0], o.p[1], o.p[2]);
    gpis_ray_in next;
    memset(&next, 0, sizeof next);
    next.pos[0] = p.x; next.pos[1] = p.y; next.pos[2] = p.z;
    next.near_t = 0.f;
    next.pixel[0] = ray.pixel[0]; next.pixel[1] = ray.pixel[1]; next.spp = ray.spp;
    next.scene_seed = ray.scene_seed;
    next.info_t = ray.info_t + o.sample_t;
    next.first_scatter = 0;
    next.bounce = ray.bounce + 1;
    next.last_val = o.last_val;
    next.last_gp_id = o.gp_id;
    next.last_aniso[0] = o.aniso[0]; next.last_aniso[1] = o.aniso[1]; next.last_aniso[2] = o.aniso[2];
    if (bounce < max_bounces - 1) {
        const V3 wo = normalized(to_local(fr, l));
        if (wi.z > 0.0f && wo.z > 0.0f) {
            const float f = albedo * (1.0f / 3.1415926536f) * wo.z;
            float t0, t1;
            if (sphere_chord(p, l, sc.s.bound_radius, t0, t1)) {
                gpis_ray_in sh = next;
                sh.dir[0] = l.x; sh.dir[1] = l.y; sh.dir[2] = l.z;
                sh.far_t = t1;
                sh.segment = (uint32_t)bounce + 1;
                sh.u_jitter = normalized_uint(g.next_i());
                a.shadow[j] = sh;
                a.contrib[j] = thr * (f * sc.s.light_radiance);
                nee = 1;
            }
        }
    }
    a.nee[j] = nee;
    bool alive = wi.z > 0.0f;
    if (alive) {
        float dx, dy, d2;
        do {
            dx = 2.f * normalized_uint(g.next_i()) - 1.f;
            dy = 2.f * normalized_uint(g.next_i()) - 1.f;
            d2 = dx * dx + dy * dy;
        } while (!(d2 < 1.f));
        const float rem = 1.0f - d2;
        const V3 w = normalized(to_global(fr, v3(dx, dy, sqrtf(rem > 0.f ? rem : 0.f))));
        thr *= albedo;
        float t0, t1;
        alive = sphere_chord(p, w, sc.s.bound_radius, t0, t1);
        if (alive) {
            next.dir[0] = w.x; next.dir[1] = w.y; next.dir[2] = w.z;
            next.far_t = t1;
            next.segment = (uint32_t)bounce + 1;
            next.u_jitter = normalized_uint(g.next_i());
            a.rays[i] = next;
        }
    }
    a.alive[i] = alive ? 1 : 0;
    a.throughput[i] = thr;
    a.rng[i] = g.state;
}

// slot k of the (possibly regrouped) shadow batch -> slot j of the bounce batch -> path i
__global__ void __launch_bounds__(256) k_paths_nee_add(size_t n_samples, PathArrays a, const uint32_t *__restrict__ order,
                                                       const uint32_t *__restrict__ shadow_order, const uint8_t *__restrict__ shadow_live,
                                                       const uint8_t *__restrict__ vis)
{
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_samples) return;
    if (shadow_order ? !shadow_live[k] : !a.nee[k]) return;
    const size_t j = shadow_order ? (size_t)shadow_order[k] : k;
    a.emission[order ? (size_t)order[j] : j] += vis[k] ? a.contrib[j] : 0.f;
}

__global__ void __launch_bounds__(256) k_paths_accumulate(SceneConst sc, size_t first_pixel, size_t n_pixels, const float *__restrict__ emission,
                                                          float *__restrict__ radiance_sum)
{
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pixels) return;
    const uint32_t spp = sc.s.spp_count;
    float acc = 0.f;
    for (uint32_t k = 0; k < spp; ++k)
        acc += emission[j * spp + k];
    radiance_sum[scene_pixel(sc.s, first_pixel + j)] += acc;
}

// --------------------------------------------------------------------------------------
// scene S with the specular NEE coupling (gpis_render_scene_s_nee): TraceBase.cpp:346-420,
// BRDFPhaseFunction.cpp:27-96, ConductorBsdf.cpp:59-139, Fresnel.hpp:102-123.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ float conductor_reflectance(float eta, float k, float cosThetaI)
{
    if (eta == 0 && k == 0)
        return 1;
    float cosThetaISq = cosThetaI * cosThetaI;
    float sinThetaISq = 1.0f - cosThetaISq > 0.0f ? 1.0f - cosThetaISq : 0.0f;
    float sinThetaIQu = sinThetaISq * sinThetaISq;
    float innerTerm = eta * eta - k * k - sinThetaISq;
    float q = innerTerm * innerTerm + 4.0f * eta * eta * k * k;
    float aSqPlusBSq = sqrtf(q > 0.0f ? q : 0.0f);
    float h = (aSqPlusBSq + innerTerm) * 0.5f;
    float a = sqrtf(h > 0.0f ? h : 0.0f);
    float Rs = ((aSqPlusBSq + cosThetaISq) - (2.0f * a * cosThetaI)) /
               ((aSqPlusBSq + cosThetaISq) + (2.0f * a * cosThetaI));
    float Rp = ((cosThetaISq * aSqPlusBSq + sinThetaIQu) - (2.0f * a * cosThetaI * sinThetaISq)) /
               ((cosThetaISq * aSqPlusBSq + sinThetaIQu) + (2.0f * a * cosThetaI * sinThetaISq));
    return 0.5f * (Rs + Rs * Rp);
}
__device__ __forceinline__ float power_heuristic(float pdf0, float pdf1) { return (pdf0 * pdf0) / (pdf0 * pdf0 + pdf1 * pdf1); }

struct NeeAux {            // per sample, between the set-up and the shading kernel
    float d[3];            // sampled light direction
    float w[3];            // mirror direction of the phase sample
    float F;               // albedo * conductorReflectance(wi.z)
    float v1, v2;          // the next two draws of the sample's stream (march jitters of the shadow segments)
    int32_t scheme;
};
struct NeeArrays {
    gpis_cond_coeff *coeff;
    gpis_nee_query *q_half, *q_normal;
    NeeAux *aux;
    gpis_ray_in *shadow_light, *shadow_phase;
    float *pdf_half, *grad_half, *pdf_normal;
    float *contrib_light, *contrib_phase;
    uint8_t *want_light, *want_phase, *want_pdf_normal, *go_light, *go_phase, *vis_light, *vis_phase;
};

__global__ void __launch_bounds__(256) k_nee_setup(SceneConst sc, gpis_surface_s sf, size_t n_samples, PathArrays a, NeeArrays b)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    uint8_t wl = 0, wp = 0, wn = 0;
    const gpis_seg_out o = a.seg[i];
    if (a.alive[i] && o.ok && !o.exited) {
        const gpis_ray_in ray = a.rays[i];
        Pcg32 g;
        g.state = a.rng[i];
        NeeAux x;
        x.scheme = o.scheme;
        const V3 capDir = v3(sc.light[0], sc.light[1], sc.light[2]);
        const double ax = o.aniso[0], ay = o.aniso[1], az = o.aniso[2];
        const double len = sqrt(ax * ax + ay * ay + az * az);
        const V3 n = v3((float)(ax / len), (float)(ay / len), (float)(az / len));
        const Frame fr = frame_from_normal(n);
        const V3 dir = v3(ray.dir[0], ray.dir[1], ray.dir[2]);
        const V3 wi = normalized(to_local(fr, v3(-dir.x, -dir.y, -dir.z)));
        x.F = sf.albedo * conductor_reflectance(sf.eta, sf.k, wi.z);
        gpis_nee_query q;
        memset(&q, 0, sizeof q);
        q.ray_dir[0] = dir.x; q.ray_dir[1] = dir.y; q.ray_dir[2] = dir.z;
        q.p[0] = o.p[0]; q.p[1] = o.p[1]; q.p[2] = o.p[2];
        q.t_segment = o.sample_t;
        q.info_t = ray.info_t + o.sample_t;
        q.pixel[0] = ray.pixel[0]; q.pixel[1] = ray.pixel[1]; q.spp = ray.spp; q.segment = ray.segment;
        q.scene_seed = ray.scene_seed;
        q.coeff = b.coeff[i];
        x.d[0] = x.d[1] = x.d[2] = 0.f;
        if (x.scheme != GPIS_UNI) {     // volumeLightSample: a direction inside the cap
            const float z = normalized_uint(g.next_i()) * (1.0f - sf.cap_cos) + sf.cap_cos;
            float dx, dy, d2;
            do {
                dx = 2.f * normalized_uint(g.next_i()) - 1.f;
                dy = 2.f * normalized_uint(g.next_i()) - 1.f;
                d2 = dx * dx + dy * dy;
            } while (!(d2 < 1.f) || !(d2 > 1e-12f));
            const float rr = 1.0f - z * z;
            const float rad = sqrtf(rr > 0.f ? rr : 0.f) / sqrtf(d2);
            const Frame cf = frame_from_normal(capDir);
            const V3 d = to_global(cf, v3(dx * rad, dy * rad, z));
            const V3 wo = normalized(to_local(fr, d));
            const V3 nl = (wi + wo) * 0.5f;
            const V3 nw = normalized(to_global(fr, nl));
            x.d[0] = d.x; x.d[1] = d.y; x.d[2] = d.z;
            gpis_nee_query qh = q;
            qh.normal[0] = nw.x; qh.normal[1] = nw.y; qh.normal[2] = nw.z;
            b.q_half[i] = qh;
            wl = 1;
        }
        x.w[0] = x.w[1] = x.w[2] = 0.f;
        if (x.scheme != GPIS_NEE) {     // volumePhaseSample: mirror about the sampled normal
            const V3 w = normalized(to_global(fr, v3(-wi.x, -wi.y, wi.z)));
            x.w[0] = w.x; x.w[1] = w.y; x.w[2] = w.z;
            float t0, t1;
            if (!(dot(w, capDir) < sf.cap_cos) && sphere_chord(v3(o.p[0], o.p[1], o.p[2]), w, sc.s.bound_radius, t0, t1)) {
                wp = 1;
                if (x.scheme != GPIS_UNI) {
                    gpis_nee_query qn = q;
                    qn.normal[0] = n.x; qn.normal[1] = n.y; qn.normal[2] = n.z;
                    b.q_normal[i] = qn;
                    wn = 1;
                }
            }
        }
        x.v1 = normalized_uint(g.next_i());
        x.v2 = normalized_uint(g.next_i());
        b.aux[i] = x;
    }
    b.want_light[i] = wl;
    b.want_phase[i] = wp;
    b.want_pdf_normal[i] = wn;
}

__global__ void __launch_bounds__(256) k_nee_shade(SceneConst sc, gpis_surface_s sf, size_t n_samples, PathArrays a, NeeArrays b)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    uint8_t gl = 0, gp = 0;
    if (b.want_light[i] || b.want_phase[i]) {
        const gpis_seg_out o = a.seg[i];
        const gpis_ray_in ray = a.rays[i];
        const NeeAux x = b.aux[i];
        const float pdf_l = (0.5f * (1.0f / 3.1415926536f)) / (1.0f - sf.cap_cos);
        const V3 p = v3(o.p[0], o.p[1], o.p[2]);
        gpis_ray_in sh0;
        memset(&sh0, 0, sizeof sh0);
        sh0.pos[0] = p.x; sh0.pos[1] = p.y; sh0.pos[2] = p.z;
        sh0.near_t = 0.f;
        sh0.pixel[0] = ray.pixel[0]; sh0.pixel[1] = ray.pixel[1]; sh0.spp = ray.spp;
        sh0.segment = ray.segment + 1;
        sh0.scene_seed = ray.scene_seed;
        sh0.info_t = ray.info_t + o.sample_t;
        sh0.first_scatter = 0;
        sh0.bounce = ray.bounce + 1;
        sh0.last_val = o.last_val;
        sh0.last_gp_id = o.gp_id;
        sh0.last_aniso[0] = o.aniso[0]; sh0.last_aniso[1] = o.aniso[1]; sh0.last_aniso[2] = o.aniso[2];
        bool light_drew = false;
        if (b.want_light[i]) {
            const float pdf = b.pdf_half[i];
            const float f = x.F * pdf;
            float t0, t1;
            const V3 d = v3(x.d[0], x.d[1], x.d[2]);
            if (f != 0.0f && sphere_chord(p, d, sc.s.bound_radius, t0, t1)) {
                gpis_ray_in sh = sh0;
                sh.dir[0] = d.x; sh.dir[1] = d.y; sh.dir[2] = d.z;
                sh.far_t = t1;
                sh.u_jitter = x.v1;
                sh.last_aniso[0] = (double)b.grad_half[3 * i]; sh.last_aniso[1] = (double)b.grad_half[3 * i + 1]; sh.last_aniso[2] = (double)b.grad_half[3 * i + 2];
                b.shadow_light[i] = sh;
                const float e = 1.f * sf.cap_radiance;
                float lightF = f * e / pdf_l;
                if (x.scheme != GPIS_NEE)
                    lightF *= power_heuristic(pdf_l, pdf);
                b.contrib_light[i] = lightF;
                light_drew = true;
                gl = 1;
            }
        }
        if (b.want_phase[i]) {
            float t0, t1;
            const V3 w = v3(x.w[0], x.w[1], x.w[2]);
            (void)sphere_chord(p, w, sc.s.bound_radius, t0, t1);   // known to succeed (k_nee_setup)
            gpis_ray_in sh = sh0;
            sh.dir[0] = w.x; sh.dir[1] = w.y; sh.dir[2] = w.z;
            sh.far_t = t1;
            sh.u_jitter = light_drew ? x.v2 : x.v1;
            b.shadow_phase[i] = sh;
            const float e = 1.f * sf.cap_radiance;
            float phaseF = e * x.F;
            if (x.scheme != GPIS_UNI)
                phaseF *= power_heuristic(b.pdf_normal[i], pdf_l);
            b.contrib_phase[i] = phaseF;
            gp = 1;
        }
    }
    b.go_light[i] = gl;
    b.go_phase[i] = gp;
}

__global__ void __launch_bounds__(256) k_nee_gather(float cap_radiance, size_t n_samples, PathArrays a, NeeArrays b)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    float L = 0.f;
    if (cap_radiance != 0.0f) {     // e == 0 ends both estimators (TraceBase.cpp:374, 410)
        if (b.go_light[i] && b.vis_light[i]) L += b.contrib_light[i];
        if (b.go_phase[i] && b.vis_phase[i]) L += b.contrib_phase[i];
    }
    a.emission[i] = L;
}

// ======================================================================================
// C ABI
// ======================================================================================

extern "C" const char *gpis_abi_sizes(void)
{
    static char buf[512];
    snprintf(buf, sizeof buf,
             "gpis_params=%zu,gpis_mean=%zu,gpis_ray_in=%zu,gpis_seg_out=%zu,gpis_cond_coeff=%zu,gpis_query=%zu,"
             "gpis_nee_query=%zu,gpis_derived=%zu,gpis_scene_s=%zu,gpis_surface_s=%zu,gpis_ramp=%zu,gpis_fs_state=%zu,gpis_guide_info=%zu",
             sizeof(gpis_params), sizeof(gpis_mean), sizeof(gpis_ray_in), sizeof(gpis_seg_out), sizeof(gpis_cond_coeff),
             sizeof(gpis_query), sizeof(gpis_nee_query), sizeof(gpis_derived), sizeof(gpis_scene_s), sizeof(gpis_surface_s), sizeof(gpis_ramp), sizeof(gpis_fs_state), sizeof(gpis_guide_info));
    return buf;
}

extern "C" void gpis_default_params(gpis_params *p)
{
    memset(p, 0, sizeof *p);
    p->abi_version = GPIS_ABI_VERSION;
    p->step_size = 0.01f; p->min_step = 8; p->seed = 0; p->impulse_density = 3.0f;
    p->scheme_1d = GPIS_UNI;
    p->correlation_context = GPIS_CTX_RENEWAL_PLUS;
    p->max_bounces = 1024;
    p->density = 1.f;
    p->sigma = 1.f; p->length_scale = 1.f;
    p->aniso[0] = p->aniso[1] = p->aniso[2] = 1.f;
    p->aniso_mtx[0] = p->aniso_mtx[4] = p->aniso_mtx[8] = 1.f;
    p->local_scale = 3.0f;
    p->ls_min = 1.; p->ls_max = 500.; p->ls_start = 0.; p->ls_end = 1.;
    p->ls_min2 = 1.; p->ls_max2 = 500.; p->ls_start2 = 0.; p->ls_end2 = 1.;          // GPF.hpp:694-699
    p->matern_v = 0.5f; p->gabor_a_inv = 1.f; p->gabor_f_inv = 1.f; p->gabor_omega[0] = 1.f;   // GPF.hpp:1964, 2041, 2079
    p->fs_sample_points = 32; p->fs_step_size = 0.;                                  // FunctionSpace...cpp:24-26
    p->grid_scale = 1.f; p->grid_surf_vol_amp_thresh = 1.f;                          // GPF.hpp:2327-2335
    p->grid_surf_amp_scale = p->grid_vol_amp_scale = p->grid_surf_ls_scale = p->grid_vol_ls_scale = 1.f;
    gpis_ramp *ramps[4] = {&p->var, &p->mean_color, &p->mean_emission, &p->aniso_field};
    for (gpis_ramp *r : ramps) { r->min = 1.; r->max = 500.; r->start = 0.; r->end = 1.; r->min2 = 1.; r->max2 = 500.; r->start2 = 0.; r->end2 = 1.; }
    p->mean.type = GPIS_MEAN_SPHERICAL; p->mean.radius = 1.f;
    p->mean.scale = 1.f; p->mean.min = -FLT_MAX; p->mean.dir[0] = 1.;
    p->mean_additional = p->mean;
}

extern "C" int gpis_create(const gpis_params *params, int device, gpis_medium **out)
{
    if (!params || !out) return set_err(GPIS_ERR_INVALID_ARG, "gpis_create: null argument");
    if (params->abi_version != GPIS_ABI_VERSION) return set_err(GPIS_ERR_INVALID_ARG, "gpis_create: abi_version %u != %d", params->abi_version, GPIS_ABI_VERSION);
    // same failure points as the reference's JSON parsing (GPM.cpp:40, SCNM.cpp:44)
    if (params->correlation_context < 0 || params->correlation_context > 3) return set_err(GPIS_ERR_INVALID_ARG, "Invalid correlation context: '%d'", params->correlation_context);
    if (params->scheme_1d < 0 || params->scheme_1d > 2) return set_err(GPIS_ERR_INVALID_ARG, "Invalid sparse conv sampling scheme: '%d'", params->scheme_1d);
    if (!(params->impulse_density >= 0.f) || params->impulse_density > 4096.f) return set_err(GPIS_ERR_INVALID_ARG, "impulse_density out of range");
    if (params->mean.type < 0 || params->mean.type > 2 || (params->has_mean_additional && (params->mean_additional.type < 0 || params->mean_additional.type > 2)))
        return set_err(GPIS_ERR_INVALID_ARG, "invalid mean type");
    if (params->nonstationary && (params->ls_ramp_type < 0 || params->ls_ramp_type > GPIS_NOISE_RUST)) return set_err(GPIS_ERR_INVALID_ARG, "invalid ls noise type");
    {
        const gpis_ramp *ramps[3] = {&params->var, &params->mean_color, &params->mean_emission};
        for (const gpis_ramp *r : ramps)
            if (r->enabled && (r->type < 0 || r->type > GPIS_NOISE_RUST)) return set_err(GPIS_ERR_INVALID_ARG, "invalid procedural noise type");
        if (params->var.enabled && !params->nonstationary) return set_err(GPIS_ERR_INVALID_ARG, "a \"var\" field needs the proc_nonstationary wrapper");
        if (params->aniso_field.enabled) {
            if (params->aniso_field.type < 0 || params->aniso_field.type > GPIS_NOISE_RUST) return set_err(GPIS_ERR_INVALID_ARG, "invalid procedural noise type");
            if (!params->nonstationary) return set_err(GPIS_ERR_INVALID_ARG, "an \"aniso\" field needs the proc_nonstationary wrapper");
            if (params->sampling_1d) return set_err(GPIS_ERR_INVALID_ARG, "an \"aniso\" field is built for 3D sampling only (GPF.cpp:1691-1727 are outside the built scope)");
        }
    }
    if (params->grid_nonstationary && (!params->nonstationary || params->var.enabled || params->aniso_field.enabled))
        return set_err(GPIS_ERR_INVALID_ARG, "the grid flavour of the non-stationary wrapper needs nonstationary = 1 and carries no var / aniso field");
    if (params->kernel_type < 0 || params->kernel_type > 3) return set_err(GPIS_ERR_INVALID_ARG, "invalid kernel type");
    if (params->kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL) {
        if (params->kernel_type == GPIS_KERNEL_MATERN && params->matern_v != 0.5f && params->matern_v != 1.5f && params->matern_v != 2.5f)
            return set_err(GPIS_ERR_UNSUPPORTED, "Matern kernel only implemented for v = 0.5, 1.5, 2.5! (GPF.cpp:1000)");
        if (params->isotropic_3d_sampling || params->sampling_1d || params->nonstationary || params->correlation_context == GPIS_CTX_RENEWAL_PLUS)
            return set_err(GPIS_ERR_UNSUPPORTED, "Matern / Gabor kernels: world-space 3D sampling with correlation context none / global / renewal only "
                                                 "(they define no isotropic transform, 1D kernel or second derivative, GPF.hpp:2002-2110)");
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return set_err(GPIS_ERR_NO_DEVICE, "gpis_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= count) return set_err(GPIS_ERR_INVALID_ARG, "gpis_create: device %d out of range (%d devices)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return set_err(GPIS_ERR_NO_DEVICE, "gpis_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    gpis_medium *m = new (std::nothrow) gpis_medium();
    if (!m) return set_err(GPIS_ERR_DEVICE, "out of host memory");
    m->params = *params;
    m->device = device;
    for (int i = 0; i < gpis_medium::kStageSlots; ++i) { m->stage[i] = nullptr; m->stage_bytes[i] = 0; }
    m->d_model = nullptr; m->d_counters = nullptr; m->d_guide_cnt = nullptr; m->d_guide = nullptr;
    m->batch_hint = GPIS_ORDER_COHERENT;
    for (int k = 0; k < 2; ++k) { m->ws_event[k] = nullptr; m->ws_event_set[k] = false; }
    memset(m->hs, 0, sizeof m->hs);
    m->d_next = nullptr; m->next_slot = 0; m->n_cus = prop.multiProcessorCount;
    for (int k = 0; k < 8; ++k) m->persist_waves[k] = 0;
    {   // diagnostic overrides, read here and nowhere else (include/gpis.h: gpis_set_option)
        m->opt[GPIS_OPT_MARCH_FORM] = GPIS_MARCH_FORM_AUTO;
        if (const char *e = getenv("GPIS_MARCH")) {
            if (strcmp(e, "resident") == 0) m->opt[GPIS_OPT_MARCH_FORM] = GPIS_MARCH_FORM_RESIDENT;
            else if (strcmp(e, "wave") == 0) m->opt[GPIS_OPT_MARCH_FORM] = GPIS_MARCH_FORM_WAVE;
        }
        m->opt[GPIS_OPT_WAVE_TAIL] = 262144;
        if (const char *e = getenv("GPIS_WAVE_TAIL")) m->opt[GPIS_OPT_WAVE_TAIL] = atoll(e) < 0 ? 0 : atoll(e);
        const char *e1 = getenv("GPIS_PATHS_SORT"), *e2 = getenv("GPIS_PATHS_PRESORT");
        m->opt[GPIS_OPT_PATHS_SORT] = !(e1 && e1[0] == '0');
        m->opt[GPIS_OPT_PATHS_PRESORT] = !(e2 && e2[0] == '0');
        m->opt[GPIS_OPT_PERSISTENT] = 1;
        if (const char *e = getenv("GPIS_PERSIST")) m->opt[GPIS_OPT_PERSISTENT] = e[0] != '0';
        m->opt[GPIS_OPT_RANGE_LEN] = 0;       // measured on C1: refill breaks the depth coherence the cooperative evaluator lives on
        if (const char *e = getenv("GPIS_RANGE_LEN")) { long long v = atoll(e); m->opt[GPIS_OPT_RANGE_LEN] = v <= 0 ? 0 : ((v + 63) / 64) * 64; }
        m->opt[GPIS_OPT_DEFER_GRAD] = 0;
        if (const char *e = getenv("GPIS_DEFER_GRAD")) m->opt[GPIS_OPT_DEFER_GRAD] = e[0] != '0';
        m->opt[GPIS_OPT_SOLO_MAX] = -1;
        if (const char *e = getenv("GPIS_SOLO_MAX")) m->opt[GPIS_OPT_SOLO_MAX] = atoll(e);
        m->opt[GPIS_OPT_CHUNK_LOG2] = 0;
        if (const char *e = getenv("GPIS_CHUNK_LOG2")) { int l = atoi(e); m->opt[GPIS_OPT_CHUNK_LOG2] = l < 16 ? 16 : (l > 28 ? 28 : l); }
    }
    memset(&m->guide, 0, sizeof m->guide);
    m->profiling = false;
    for (int k = 0; k < 3; ++k) { m->events_used[k] = 0; m->prof_ms[k] = 0.; m->prof_launches[k] = 0; }
    memset(&m->fast, 0, sizeof m->fast);
    int st = build_model(*params, m->host_model, m->derived);
    if (st != GPIS_OK) { delete m; return st; }
    hipError_t e = hipMalloc(&m->d_model, sizeof(DevModel));
    if (e == hipSuccess) e = hipMalloc(&m->d_counters, 2 * sizeof(Counters));
    if (e == hipSuccess) e = hipMalloc(&m->d_guide_cnt, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(m->d_guide_cnt, 0, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&m->d_guide, sizeof(GuideField));
    if (e == hipSuccess) e = hipMemset(m->d_guide, 0, sizeof(GuideField));
    if (e == hipSuccess) e = hipMalloc(&m->d_next, kPersistSlots * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemcpy(m->d_model, &m->host_model, sizeof(DevModel), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(m->d_counters, 0, 2 * sizeof(Counters));
    if (e != hipSuccess) {
        set_err(GPIS_ERR_DEVICE, "gpis_create: %s", hipGetErrorString(e));
        if (m->d_model) (void)hipFree(m->d_model);
        if (m->d_counters) (void)hipFree(m->d_counters);
        if (m->d_guide_cnt) (void)hipFree(m->d_guide_cnt);
        if (m->d_guide) (void)hipFree(m->d_guide);
        if (m->d_next) (void)hipFree(m->d_next);
        delete m;
        return GPIS_ERR_DEVICE;
    }
    st = launch::fast_table_build(m->host_model, &m->fast);
    if (st != GPIS_OK) {
        set_err(st, "gpis_create: building the cell table failed");
        (void)hipFree(m->d_model); (void)hipFree(m->d_counters); (void)hipFree(m->d_guide_cnt); (void)hipFree(m->d_guide); (void)hipFree(m->d_next);
        delete m;
        return st;
    }
    m->derived.fast_path = m->fast.enabled;
    *out = m;
    return GPIS_OK;
}

extern "C" int gpis_destroy(gpis_medium *m)
{
    if (!m) return GPIS_OK;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    fast_table_free(&m->fast);
    guide_free(&m->guide);
    if (m->d_guide_cnt) (void)hipFree(m->d_guide_cnt);
    if (m->d_guide) (void)hipFree(m->d_guide);
    if (m->fs_ws) (void)hipFree(m->fs_ws);
    if (m->d_grid_vox) (void)hipFree(m->d_grid_vox);
    for (int k = 0; k < 3; ++k)
        if (m->fs_stage[k]) (void)hipFree(m->fs_stage[k]);
    for (int k = 0; k < 3; ++k)
        for (auto &ev : m->events[k]) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (int i = 0; i < gpis_medium::kStageSlots; ++i)
        if (m->stage[i]) (void)hipFree(m->stage[i]);
    for (int k = 0; k < 2; ++k)
        if (m->ws_event[k]) (void)hipEventDestroy(m->ws_event[k]);
    for (int k = 0; k < 2; ++k) {
        gpis_medium::HostSlot &S = m->hs[k];
        if (S.pin_in) (void)hipHostFree(S.pin_in);
        if (S.pin_out) (void)hipHostFree(S.pin_out);
        if (S.pin_aux) (void)hipHostFree(S.pin_aux);
        if (S.dev_in) (void)hipFree(S.dev_in);
        if (S.dev_out) (void)hipFree(S.dev_out);
        if (S.dev_aux) (void)hipFree(S.dev_aux);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.stream) (void)hipStreamDestroy(S.stream);
    }
    if (m->d_model) (void)hipFree(m->d_model);
    if (m->d_counters) (void)hipFree(m->d_counters);
    if (m->d_next) (void)hipFree(m->d_next);
    delete m;
    return GPIS_OK;
}

extern "C" int gpis_get_derived(const gpis_medium *m, gpis_derived *out)
{
    if (!m || !out) return set_err(GPIS_ERR_INVALID_ARG, "null argument");
    *out = m->derived;
    return GPIS_OK;
}

#define CHECK_ARGS(cond)                                                        \
    do {                                                                        \
        if (!(cond)) return set_err(GPIS_ERR_INVALID_ARG, "%s: invalid argument (%s)", __func__, #cond); \
    } while (0)

static int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return set_err(GPIS_ERR_DEVICE, "%s launch: %s", what, hipGetErrorString(e));
    return GPIS_OK;
}

struct ProfScope {   // records an event pair around one march-kernel launch when profiling is on
    gpis_medium *m; int kind; hipStream_t s; bool on;
    ProfScope(gpis_medium *m_, int kind_, hipStream_t s_) : m(m_), kind(kind_), s(s_), on(m_->profiling)
    {
        if (!on) return;
        if (m->events_used[kind] == m->events[kind].size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            m->events[kind].emplace_back(a, b);
        }
        (void)hipEventRecord(m->events[kind][m->events_used[kind]].first, s);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(m->events[kind][m->events_used[kind]].second, s);
        m->events_used[kind]++;
    }
};

// The guided march as a wavefront (gpis_wave.hpp): step → sort → eval until no ray requests a value,
// then (sampleDistance) one sorted gradient pass.  Synchronises the stream once per iteration to read the
// request count.  GPIS_MARCH=resident selects the one-kernel state machine instead.
// Which form runs: the resident kernels win on batches whose waves are already coherent (camera rays in
// pixel order: 131 vs 181 ms on C1), the wavefront on scattered rays (1.4x on the second bounce), so the
// caller's hint decides; GPIS_MARCH=wave|resident overrides it.
enum MarchHint { MARCH_COHERENT = GPIS_ORDER_COHERENT, MARCH_SCATTERED = GPIS_ORDER_SCATTERED };
static bool wave_march_selected(const gpis_medium *m, int hint)
{
    if (m->opt[GPIS_OPT_MARCH_FORM] == GPIS_MARCH_FORM_RESIDENT) return false;
    if (m->opt[GPIS_OPT_MARCH_FORM] == GPIS_MARCH_FORM_WAVE) return true;
    return hint == MARCH_SCATTERED;
}
// shared-workspace hand-over between callers on different streams (slot 0: stage[3], slot 1: stage[4])
static int ws_acquire(gpis_medium *m, int slot, hipStream_t s)
{
    if (m->ws_event_set[slot])
        HIP_TRY(hipStreamWaitEvent(s, m->ws_event[slot], 0));
    return GPIS_OK;
}
static int ws_release(gpis_medium *m, int slot, hipStream_t s)
{
    if (!m->ws_event[slot])
        HIP_TRY(hipEventCreateWithFlags(&m->ws_event[slot], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->ws_event[slot], s));
    m->ws_event_set[slot] = true;
    return GPIS_OK;
}
static int wave_march(gpis_medium *m, size_t n, const gpis_ray_in *rays, const uint8_t *mask, bool want_sample, gpis_seg_out *out,
                      gpis_cond_coeff *coeff, uint8_t *visible, hipStream_t s)
{
    if (n > 0x7FFFFFF0ull) return set_err(GPIS_ERR_INVALID_ARG, "wave march: batch too large (the radix sort takes an int count)");
    size_t temp_bytes = 0;
    if (sort_pairs_u32(nullptr, temp_bytes, nullptr, nullptr, nullptr, nullptr, n, s) != hipSuccess)
        return set_err(GPIS_ERR_DEVICE, "radix sort scratch query failed");
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_cnt = carve(64), o_state = carve(n * launch::wave_state_bytes()), o_k0 = carve(n * 4), o_v0 = carve(n * 4), o_k1 = carve(n * 4),
                 o_v1 = carve(n * 4), o_temp = carve(temp_bytes);
    int rc = ensure_stage(m, 4, off);
    if (rc) return rc;
    if ((rc = ws_acquire(m, 1, s))) return rc;
    char *ws = (char *)m->stage[4];
    unsigned long long *d_req = (unsigned long long *)(ws + o_cnt);
    uint32_t *k0 = (uint32_t *)(ws + o_k0), *v0 = (uint32_t *)(ws + o_v0), *k1 = (uint32_t *)(ws + o_k1), *v1 = (uint32_t *)(ws + o_v1);
    const launch::WaveBufs wb{ws + o_state, k0, v0, k1, v1, d_req};
    const bool small_arg = m->host_model.exp_arg_max < 100.f;
    Counters *cnt = m->d_counters + (want_sample ? 0 : 1);
    auto read_requests = [&](size_t &n_req) -> int {
        unsigned long long h = 0;
        HIP_TRY(hipMemcpyAsync(&h, d_req, sizeof h, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemsetAsync(d_req, 0, sizeof h, s));
        HIP_TRY(hipStreamSynchronize(s));
        n_req = (size_t)h;
        return GPIS_OK;
    };
    HIP_TRY(hipMemsetAsync(d_req, 0, sizeof(unsigned long long), s));
    const size_t tail_limit = (size_t)m->opt[GPIS_OPT_WAVE_TAIL];      // default 262144; measured on the multi-bounce driver: 16 Ki 55.9, 64 Ki 61.5, 256 Ki 65.0, 1 Mi 64.5 M paths/s
    size_t n_active = n;
    const uint32_t *active = nullptr;
    int init = 1;
    for (;;) {
        launch::wave_step(want_sample, m->d_model, m->guide, n_active, active, init, rays, mask, wb, m->d_guide_cnt, s);
        if ((rc = launch_check("k_wave_step"))) return rc;
        size_t n_req = 0;
        if ((rc = read_requests(n_req))) return rc;
        if (n_req == 0)
            break;
        size_t tb = temp_bytes;
        if (sort_pairs_u32(ws + o_temp, tb, k0, k1, v0, v1, n_active, s) != hipSuccess)
            return set_err(GPIS_ERR_DEVICE, "radix sort failed");
        launch::wave_eval(small_arg, m->d_model, m->fast, m->guide, n_req, v1, rays, wb, cnt, s);
        if ((rc = launch_check("k_wave_eval"))) return rc;
        active = v1;            // the sorted requesters are the rays still marching
        n_active = n_req;
        init = 0;
        if (n_active <= tail_limit) {
            // few rays left: one wave per ray runs them to the end of their value requests
            launch::wave_tail(want_sample, m->d_model, m->fast, m->guide, n_active, active, rays, wb, cnt, m->d_guide_cnt, s);
            if ((rc = launch_check("k_wave_tail"))) return rc;
            break;
        }
    }
    if (want_sample) {
        launch::wave_grad_keys(m->d_model, m->guide, n, rays, wb, s);
        if ((rc = launch_check("k_wave_grad_keys"))) return rc;
        size_t n_grad = 0;
        if ((rc = read_requests(n_grad))) return rc;
        if (n_grad) {
            size_t tb = temp_bytes;
            if (sort_pairs_u32(ws + o_temp, tb, k0, k1, v0, v1, n, s) != hipSuccess)
                return set_err(GPIS_ERR_DEVICE, "radix sort failed");
            launch::wave_grad(small_arg, m->d_model, m->fast, m->guide, n_grad, v1, rays, wb, cnt, s);
            if ((rc = launch_check("k_wave_grad"))) return rc;
        }
        launch::wave_finish_sd(m->d_model, n, rays, mask, wb, out, coeff, cnt, s);
        if ((rc = launch_check("k_wave_finish_sd"))) return rc;
        return ws_release(m, 1, s);
    }
    launch::wave_finish_tr(n, mask, wb, visible, cnt, s);
    if ((rc = launch_check("k_wave_finish_tr"))) return rc;
    return ws_release(m, 1, s);
}

// ---- persistent march launch (per-path media) ---------------------------------------------------------------
// the lattice sums of gpis_persist.inc have a diagonal kernel matrix: every medium except world-space 3D with a full anisoMtx
static bool persist_supported(const DevModel &H) { return !H.aniso.enabled && !H.fbm_noise && (H.sampling_1d || H.iso3d || !H.use_aniso_mtx); }
// sideways (lane = impulse) evaluation pays while at most this many lanes of a wave have a job: lockstep costs
// 27 n g instructions per round whatever the number of jobs, one sideways job ~20.6 cells x (c0 + c1 n)
// (g = 61 generator, c0 = 135, c1 = 0.31: counted on the gfx950 ISA of these loops)
static int persist_solo_max(const gpis_medium *m)
{
    const DevModel &H = m->host_model;
    if (H.sampling_1d || H.n_impulses > 64u || H.n_impulses == 0u) return 0;     // the sideways form holds one cell's impulses in one wave
    if (m->opt[GPIS_OPT_SOLO_MAX] >= 0) return (int)m->opt[GPIS_OPT_SOLO_MAX];
    const double lock = 27.0 * H.n_impulses * 61.0, side = 20.6 * (135.0 + 0.31 * H.n_impulses);
    int t = (int)(lock / side);
    return t < 0 ? 0 : (t > 48 ? 48 : t);
}
template <bool WANT_SAMPLE>
static int launch_persist(gpis_medium *m, PersistArgs a, hipStream_t s)
{
    const DevModel &H = m->host_model;
    if (a.n >= 0xFFFF0000ull) return set_err(GPIS_ERR_INVALID_ARG, "persistent march: batch too large for the 32-bit ray counter");
    a.solo_max = persist_solo_max(m);
    int inst = launch::INST_GENERIC;
    if (H.fbm_noise) inst = launch::INST_GENERIC;          // sandstone / rust fields: all-features instance only
    else if (H.sampling_1d) inst = launch::INST_1D;
    else if (H.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL) inst = launch::INST_GENERIC;
    else if (!H.nonstationary && !H.multi_res && !H.multi_resolution_grid) inst = launch::INST_3D;
    else if (H.nonstationary && H.multi_res && H.multi_resolution_grid) inst = launch::INST_3D_MULTIRES;
    const int slot_id = (WANT_SAMPLE ? 0 : 4) + inst;
    if (m->persist_waves[slot_id] == 0) {
        int nb = launch::persist_blocks_per_cu(inst, WANT_SAMPLE);
        if (nb <= 0) nb = 8;
        m->persist_waves[slot_id] = nb * m->n_cus;
        if (getenv("GPIS_DEBUG")) fprintf(stderr, "gpis: persistent march instance %d (%s): %d resident waves per CU x %d CUs\n", inst, WANT_SAMPLE ? "sampleDistance" : "transmittance", nb, m->n_cus);
    }
    const size_t waves_needed = (a.n + kBlock - 1) / kBlock;
    const unsigned grid = (unsigned)(waves_needed < (size_t)m->persist_waves[slot_id] ? waves_needed : (size_t)m->persist_waves[slot_id]);
    a.next = m->d_next + (m->next_slot++ % kPersistSlots);
    HIP_TRY(hipMemsetAsync(a.next, 0, sizeof(unsigned int), s));
    launch::persist_march(inst, WANT_SAMPLE, grid, m->d_model, a, s);
    return launch_check("k_persist_march");
}
// neePDF / neeGrad: the 1D-sampling instance for the media the NEE estimators are built for, else the all-features one
static int nee_instance(const DevModel &H) { return (H.sampling_1d && !H.fbm_noise) ? launch::INST_1D : launch::INST_GENERIC; }
// the lane-per-ray instance whose compile-time flags equal the medium's
static int lane_instance(const DevModel &H)
{
    if (H.fbm_noise) return launch::INST_GENERIC;           // sandstone / rust fields: all-features instance only
    if (H.sampling_1d) return launch::INST_1D;
    if (!H.nonstationary && !H.multi_res && !H.multi_resolution_grid && H.kernel_type == GPIS_KERNEL_SQUARED_EXPONENTIAL) return launch::INST_3D;
    if (H.nonstationary && H.multi_res && H.multi_resolution_grid && !H.aniso.enabled) return launch::INST_3D_MULTIRES;
    return launch::INST_GENERIC;
}

static int sample_distance_impl(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                                const uint8_t *mask, hipStream_t s, int hint = MARCH_COHERENT)
{
    if (n == 0) return GPIS_OK;
    ProfScope prof(m, 0, s);
    if (m->guide.enabled && wave_march_selected(m, hint))
        return wave_march(m, n, rays, mask, true, out, coeff, nullptr, s);
    if (m->guide.enabled && m->opt[GPIS_OPT_RANGE_LEN] > 0) {
        // guided march with in-wave refill (gpis_guide_range.hpp) + one coherent gradient pass
        launch::range_sample_distance(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->guide, n, rays, out, coeff, mask, m->d_counters, m->d_guide_cnt,
                                      (uint32_t)m->opt[GPIS_OPT_RANGE_LEN], s);
        return launch_check("k_guided_range_sd / k_guided_range_grad");
    }
    if (m->guide.enabled) {
        if (m->opt[GPIS_OPT_DEFER_GRAD]) {
            launch::guided_sample_distance_nograd(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->d_guide, n, rays, out, coeff, mask, m->d_counters, m->d_guide_cnt, s);
            int rc = launch_check("k_guided_sample_distance_nograd");
            if (rc) return rc;
            launch::range_grad(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->guide, n, rays, out, coeff, mask, m->d_counters, s);
            return launch_check("k_guided_range_grad");
        }
        launch::guided_sample_distance(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->d_guide, n, rays, out, coeff, mask, m->d_counters, m->d_guide_cnt, s);
        return launch_check("k_guided_sample_distance");
    }
    if (m->fast.enabled) {
        launch::fast_sample_distance(m->d_model, m->fast, n, rays, out, coeff, mask, m->d_counters, s);
        return launch_check("k_fast_sample_distance");
    }
    // pick the instance whose compile-time flags equal the medium's
    const DevModel &H = m->host_model;
    if (m->opt[GPIS_OPT_PERSISTENT] && persist_supported(H)) {
        PersistArgs a{};
        a.n = n; a.rays = rays; a.out = out; a.coeff = coeff; a.visible = nullptr; a.mask = mask; a.cnt = m->d_counters;
        return launch_persist<true>(m, a, s);
    }
    launch::lane_sample_distance(lane_instance(H), m->d_model, n, rays, out, coeff, mask, m->d_counters, s);
    return launch_check("k_sample_distance");
}
static int transmittance_impl(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint8_t *visible, const uint8_t *mask, hipStream_t s,
                              int hint = MARCH_COHERENT)
{
    if (n == 0) return GPIS_OK;
    ProfScope prof(m, 1, s);
    if (m->guide.enabled && wave_march_selected(m, hint))
        return wave_march(m, n, rays, mask, false, nullptr, nullptr, visible, s);
    if (m->guide.enabled && m->opt[GPIS_OPT_RANGE_LEN] > 0) {
        launch::range_transmittance(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->guide, n, rays, visible, mask, m->d_counters + 1, m->d_guide_cnt,
                                    (uint32_t)m->opt[GPIS_OPT_RANGE_LEN], s);
        return launch_check("k_guided_range_tr");
    }
    if (m->guide.enabled) {
        launch::guided_transmittance(m->host_model.exp_arg_max < 100.f, m->d_model, m->fast, m->d_guide, n, rays, visible, mask, m->d_counters + 1, m->d_guide_cnt, s);
        return launch_check("k_guided_transmittance");
    }
    if (m->fast.enabled) {
        launch::fast_transmittance(m->d_model, m->fast, n, rays, visible, mask, m->d_counters + 1, s);
        return launch_check("k_fast_transmittance");
    }
    const DevModel &H = m->host_model;
    if (m->opt[GPIS_OPT_PERSISTENT] && persist_supported(H)) {
        PersistArgs a{};
        a.n = n; a.rays = rays; a.out = nullptr; a.coeff = nullptr; a.visible = visible; a.mask = mask; a.cnt = m->d_counters + 1;
        return launch_persist<false>(m, a, s);
    }
    launch::lane_transmittance(lane_instance(H), m->d_model, n, rays, visible, mask, m->d_counters + 1, s);
    return launch_check("k_transmittance");
}

extern "C" int gpis_sample_distance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (rays && out)));
    std::lock_guard<std::mutex> lock(m->mu);      // the handle's event lists / wavefront workspace are shared; held while ENQUEUEING only
    HIP_TRY(hipSetDevice(m->device));
    return sample_distance_impl(m, n, rays, out, coeff, nullptr, (hipStream_t)stream, m->batch_hint);
}
extern "C" int gpis_transmittance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint8_t *visible, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (rays && visible)));
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    return transmittance_impl(m, n, rays, visible, nullptr, (hipStream_t)stream, m->batch_hint);
}
extern "C" int gpis_set_batch_order(gpis_medium *m, int order)
{
    CHECK_ARGS(m && (order == GPIS_ORDER_COHERENT || order == GPIS_ORDER_SCATTERED));
    std::lock_guard<std::mutex> lock(m->mu);
    m->batch_hint = order;
    return GPIS_OK;
}
extern "C" int gpis_set_option(gpis_medium *m, int option, long long value)
{
    CHECK_ARGS(m && option >= 0 && option < GPIS_OPT_COUNT_);
    switch (option) {
    case GPIS_OPT_MARCH_FORM: CHECK_ARGS(value >= GPIS_MARCH_FORM_AUTO && value <= GPIS_MARCH_FORM_WAVE); break;
    case GPIS_OPT_WAVE_TAIL: CHECK_ARGS(value >= 0); break;
    case GPIS_OPT_PATHS_SORT: case GPIS_OPT_PATHS_PRESORT: case GPIS_OPT_PERSISTENT: case GPIS_OPT_DEFER_GRAD: CHECK_ARGS(value == 0 || value == 1); break;
    case GPIS_OPT_CHUNK_LOG2: CHECK_ARGS(value == 0 || (value >= 16 && value <= 28)); break;
    case GPIS_OPT_SOLO_MAX: CHECK_ARGS(value >= -1 && value <= 64); break;
    case GPIS_OPT_RANGE_LEN: CHECK_ARGS(value >= 0 && value <= (1 << 24) && value % 64 == 0); break;
    default: break;
    }
    std::lock_guard<std::mutex> lock(m->mu);
    m->opt[option] = value;
    return GPIS_OK;
}
extern "C" int gpis_get_option(gpis_medium *m, int option, long long *value)
{
    CHECK_ARGS(m && value && option >= 0 && option < GPIS_OPT_COUNT_);
    std::lock_guard<std::mutex> lock(m->mu);
    *value = m->opt[option];
    return GPIS_OK;
}
extern "C" int gpis_eval_value_batch(gpis_medium *m, size_t n, const gpis_query *q, float *value, int32_t *gp_id, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (q && value)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    launch::eval_value(m->d_model, n, q, value, gp_id, m->d_counters, (hipStream_t)stream);
    return launch_check("k_eval_value");
}
extern "C" int gpis_eval_gradient_batch(gpis_medium *m, size_t n, const gpis_query *q, float *grad3, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (q && grad3)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    launch::eval_gradient(m->d_model, n, q, grad3, m->d_counters, (hipStream_t)stream);
    return launch_check("k_eval_gradient");
}
extern "C" int gpis_conditioning_batch(gpis_medium *m, size_t n, const gpis_query *q, const float *target_val, const float *target_grad3,
                                       gpis_cond_coeff *coeff_out, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (q && target_val && target_grad3 && coeff_out)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    launch::conditioning(m->d_model, n, q, target_val, target_grad3, coeff_out, m->d_counters, (hipStream_t)stream);
    return launch_check("k_conditioning");
}
extern "C" int gpis_nee_pdf_batch(gpis_medium *m, size_t n, const gpis_nee_query *q, float *pdf, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (q && pdf)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    launch::nee(nee_instance(m->host_model), m->d_model, n, q, pdf, nullptr, m->d_counters, nullptr, (hipStream_t)stream);
    return launch_check("k_nee");
}
extern "C" int gpis_nee_grad_batch(gpis_medium *m, size_t n, const gpis_nee_query *q, float *grad3, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (q && grad3)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    launch::nee(nee_instance(m->host_model), m->d_model, n, q, nullptr, grad3, m->d_counters, nullptr, (hipStream_t)stream);
    return launch_check("k_nee");
}
extern "C" int gpis_mean_color_emission_batch(gpis_medium *m, size_t n, const double *p3, float *color3, float *emission3, void *stream)
{
    CHECK_ARGS(m && (n == 0 || p3));
    if (n == 0 || (!color3 && !emission3)) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    k_mean_color_emission<<<grid_of(n, 256), 256, 0, (hipStream_t)stream>>>(m->d_model, n, p3, color3, emission3);
    return launch_check("k_mean_color_emission");
}
// function-space comparison path: gpis_fs.hpp
static int fs_check(gpis_medium *m)
{
    const DevModel &H = m->host_model;
    if (H.fs_n < 2 || H.fs_n > GPIS_FS_MAX_POINTS || !(H.fs_step >= 0) || H.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL || H.nonstationary ||
        H.use_aniso_mtx || H.has_mean_additional || (H.color.enabled && H.color.type >= GPIS_NOISE_SANDSTONE))
        return set_err(GPIS_ERR_INVALID_ARG, "function-space path: squared-exponential covariance, analytic mean (ramp colour), 2..64 sample points");
    return GPIS_OK;
}
// 37 KB of LDS per workgroup: four one-wave workgroups (one per SIMD) are resident per CU and walk the batch; each owns one
// slice of the L2-resident workspace
static int fs_workspace(gpis_medium *m, unsigned &cap)
{
    cap = (unsigned)(m->n_cus > 0 ? m->n_cus : 256) * 4u;
    std::lock_guard<std::mutex> lock(m->mu);
    if (m->fs_ws_blocks < cap) {
        if (m->fs_ws) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(m->fs_ws); m->fs_ws = nullptr; m->fs_ws_blocks = 0; }
        if (hipMalloc(&m->fs_ws, (size_t)cap * launch::fs_workspace_bytes_per_block()) != hipSuccess) { (void)hipGetLastError(); return set_err(GPIS_ERR_DEVICE, "function-space workspace allocation failed"); }
        m->fs_ws_blocks = cap;
    }
    return GPIS_OK;
}
static int fs_launch(bool want_sample, gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out, uint8_t *visible, hipStream_t s)
{
    HIP_TRY(hipSetDevice(m->device));
    unsigned cap = 0;
    if (int st = fs_workspace(m, cap)) return st;
    const unsigned grid = (unsigned)(n < cap ? n : cap);
    launch::fs_march(want_sample, grid, m->d_model, n, rays, states, out, visible, m->fs_ws, s);
    return launch_check("k_fs_march");
}
#ifdef GPIS_FS_PROF
extern "C" int gpis_fs_prof_read(unsigned long long *out16, int reset) { return launch::fs_prof_read(out16, reset); }
#endif
extern "C" int gpis_fs_linalg_batch(gpis_medium *m, int op, int n, size_t count, const double *in, double *out, double *evals, void *stream)
{
    CHECK_ARGS(m && op >= GPIS_FS_OP_EIGH && op <= GPIS_FS_OP_PINV && n >= 1 && n <= GPIS_FS_MAX_CTX && (count == 0 || (in && out)));
    if (count == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    unsigned cap = 0;
    if (int st = fs_workspace(m, cap)) return st;
    launch::fs_linalg((unsigned)(count < cap ? count : cap), op, n, count, in, out, evals, m->fs_ws, (hipStream_t)stream);
    return launch_check("k_fs_linalg");
}
extern "C" int gpis_sort_pairs_u32(size_t n, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, void *stream)
{
    CHECK_ARGS(n == 0 || (keys_in && vals_in && keys_out && vals_out));
    size_t tb = 0;
    if (sort_pairs_u32(nullptr, tb, nullptr, nullptr, nullptr, nullptr, n, (hipStream_t)stream) != hipSuccess)
        return set_err(GPIS_ERR_INVALID_ARG, "gpis_sort_pairs_u32: batch too large");
    if (n == 0) return GPIS_OK;
    void *temp = nullptr;
    HIP_TRY(hipMalloc(&temp, tb));
    hipError_t e = sort_pairs_u32(temp, tb, keys_in, keys_out, vals_in, vals_out, n, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(temp);
    if (e != hipSuccess) return set_err(GPIS_ERR_DEVICE, "gpis_sort_pairs_u32: %s", hipGetErrorString(e));
    return GPIS_OK;
}
extern "C" int gpis_libm_batch(int fn, size_t n, const double *x, const double *y, double *out, double *out2, void *stream)
{
    CHECK_ARGS(fn >= GPIS_LIBM_EXP && fn <= GPIS_LIBM_SINCOSF && (n == 0 || (x && out)) && (fn != GPIS_LIBM_POW || n == 0 || y) && ((fn != GPIS_LIBM_SINCOS && fn != GPIS_LIBM_SINCOSF) || n == 0 || out2));
    if (n == 0) return GPIS_OK;
    launch::libm_eval(fn, n, x, y, out, out2, (hipStream_t)stream);
    return launch_check("k_libm");
}
extern "C" int gpis_fs_sample_distance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (rays && states && out)));
    if (int st = fs_check(m)) return st;
    if (n == 0) return GPIS_OK;
    return fs_launch(true, m, n, rays, states, out, nullptr, (hipStream_t)stream);
}
extern "C" int gpis_fs_transmittance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, uint8_t *visible, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (rays && states && visible)));
    if (int st = fs_check(m)) return st;
    if (n == 0) return GPIS_OK;
    return fs_launch(false, m, n, rays, states, nullptr, visible, (hipStream_t)stream);
}
// host-pointer forms of the two function-space entries: one staging round trip per call (stage[0..2] are shared by the
// host conveniences of this handle, so the whole call holds the handle's host mutex; the workspace has its own lock)
static int fs_host(gpis_medium *m, bool want_sample, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, void *out)
{
    if (int st = fs_check(m)) return st;
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->fs_host_mu);
    HIP_TRY(hipSetDevice(m->device));
    const size_t out_rec = want_sample ? sizeof(gpis_seg_out) : 1;
    const size_t need[3] = {n * sizeof(gpis_ray_in), n * sizeof(gpis_fs_state), n * out_rec};
    for (int k = 0; k < 3; ++k)
        if (m->fs_stage_bytes[k] < need[k]) {
            if (m->fs_stage[k]) { HIP_TRY(hipFree(m->fs_stage[k])); m->fs_stage[k] = nullptr; m->fs_stage_bytes[k] = 0; }
            HIP_TRY(hipMalloc(&m->fs_stage[k], need[k] + need[k] / 4 + 4096));
            m->fs_stage_bytes[k] = need[k] + need[k] / 4 + 4096;
        }
    void *d_rays = m->fs_stage[0], *d_states = m->fs_stage[1], *d_out = m->fs_stage[2];
    HIP_TRY(hipMemcpy(d_rays, rays, n * sizeof(gpis_ray_in), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_states, states, n * sizeof(gpis_fs_state), hipMemcpyHostToDevice));
    int st = fs_launch(want_sample, m, n, (const gpis_ray_in *)d_rays, (gpis_fs_state *)d_states, want_sample ? (gpis_seg_out *)d_out : nullptr,
                       want_sample ? nullptr : (uint8_t *)d_out, nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(states, d_states, n * sizeof(gpis_fs_state), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out, d_out, n * out_rec, hipMemcpyDeviceToHost));
    return GPIS_OK;
}
extern "C" int gpis_fs_sample_distance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out)
{
    CHECK_ARGS(m && (n == 0 || (rays && states && out)));
    return fs_host(m, true, n, rays, states, out);
}
extern "C" int gpis_fs_transmittance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, uint8_t *visible)
{
    CHECK_ARGS(m && (n == 0 || (rays && states && visible)));
    return fs_host(m, false, n, rays, states, visible);
}
extern "C" int gpis_mean_color_emission_host(gpis_medium *m, size_t n, const double *p3, float *color3, float *emission3)
{
    CHECK_ARGS(m && (n == 0 || p3));
    if (n == 0 || (!color3 && !emission3)) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    int st;
    if ((st = ensure_stage(m, 0, 3 * n * sizeof(double))) || (st = ensure_stage(m, 1, 3 * n * sizeof(float))) || (st = ensure_stage(m, 2, 3 * n * sizeof(float))))
        return st;
    HIP_TRY(hipMemcpy(m->stage[0], p3, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    st = gpis_mean_color_emission_batch(m, n, (const double *)m->stage[0], color3 ? (float *)m->stage[1] : nullptr, emission3 ? (float *)m->stage[2] : nullptr, nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    if (color3) HIP_TRY(hipMemcpy(color3, m->stage[1], 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (emission3) HIP_TRY(hipMemcpy(emission3, m->stage[2], 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return GPIS_OK;
}
extern "C" int gpis_xxhash32_batch(gpis_medium *m, size_t n, int arity, const uint32_t *words, uint32_t *out, void *stream)
{
    CHECK_ARGS(m && arity >= 1 && arity <= 4 && (n == 0 || (words && out)));
    if (n == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    k_xxhash32<<<grid_of(n, 256), 256, 0, (hipStream_t)stream>>>(n, arity, words, out);
    return launch_check("k_xxhash32");
}
extern "C" int gpis_pcg32_stream_batch(gpis_medium *m, size_t n, const uint64_t *state, uint32_t count, uint32_t *out, void *stream)
{
    CHECK_ARGS(m && (n == 0 || (state && out)));
    if (n == 0 || count == 0) return GPIS_OK;
    HIP_TRY(hipSetDevice(m->device));
    k_pcg32_stream<<<grid_of(n, 256), 256, 0, (hipStream_t)stream>>>(n, state, count, out);
    return launch_check("k_pcg32_stream");
}

// ---- host-pointer conveniences -----------------------------------------------------------
// ---- pipelined host path ------------------------------------------------------------------------------------
// Records move in chunks through two slots (streams): H2D of chunk k, the march kernel of chunk k-1 and D2H of chunk k-2
// overlap.  Caller memory that is pinned (gpis_alloc_host, hipHostMalloc, hipHostRegister) is DMA'd directly; pageable
// memory is staged through the slot's own pinned buffers (one memcpy each way).  A batch of one costs one small H2D,
// one launch, one D2H and one stream synchronise.
// records per chunk: 32 MiB of rays in, 24 MiB of results out.  Measured on C1 (guided march, pinned caller memory, 1 Mi rays):
// 64 Ki-record chunks 74 M segments/s — each chunk's kernel is too small to fill the chip (61 M segments/s for a lone 64 Ki batch
// against 361 M for 1 Mi) — so chunks are as large as the overlap of three stages allows
constexpr size_t kHostChunk = 262144;
constexpr size_t kInRec = sizeof(gpis_ray_in), kOutRecMax = sizeof(gpis_seg_out), kAuxRec = sizeof(gpis_cond_coeff);

static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
static int host_slot_ensure(gpis_medium *m, int k, size_t cap)
{
    gpis_medium::HostSlot &S = m->hs[k];
    if (!S.stream) HIP_TRY(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    if (!S.done) HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    if (S.cap >= cap) return GPIS_OK;
    HIP_TRY(hipStreamSynchronize(S.stream));
    if (S.pin_in) { (void)hipHostFree(S.pin_in); (void)hipHostFree(S.pin_out); (void)hipHostFree(S.pin_aux); (void)hipFree(S.dev_in); (void)hipFree(S.dev_out); (void)hipFree(S.dev_aux); }
    S.pin_in = S.pin_out = S.pin_aux = S.dev_in = S.dev_out = S.dev_aux = nullptr; S.cap = 0;
    HIP_TRY(hipHostMalloc((void **)&S.pin_in, cap * kInRec, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&S.pin_out, cap * kOutRecMax, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&S.pin_aux, cap * kAuxRec, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&S.dev_in, cap * kInRec));
    HIP_TRY(hipMalloc((void **)&S.dev_out, cap * kOutRecMax));
    HIP_TRY(hipMalloc((void **)&S.dev_aux, cap * kAuxRec));
    S.cap = cap;
    return GPIS_OK;
}
// sample == true: sampleDistance (out = gpis_seg_out, aux = gpis_cond_coeff or null); false: transmittance (out = uint8_t)
static int host_march(gpis_medium *m, bool sample, size_t n, const gpis_ray_in *rays, void *out, gpis_cond_coeff *aux)
{
    const size_t out_rec = sample ? sizeof(gpis_seg_out) : 1;
    const size_t chunk = n < kHostChunk ? n : kHostChunk;
    const size_t cap = chunk <= 256 ? 256 : (chunk <= 16384 ? 16384 : kHostChunk);   // small batches keep a small slot
    int st;
    if ((st = host_slot_ensure(m, 0, cap)) || (n > chunk && (st = host_slot_ensure(m, 1, cap))))
        return st;
    // whether the caller's buffers are pinned decides between DMA from them and a copy through the slot's pinned staging; small
    // batches (the Medium adapter's batch of one) go through the staging without asking — three pointer queries cost more than
    // the copies they would save
    const bool ask = n >= 4096;
    const bool in_pinned = ask && is_pinned_host(rays), out_pinned = ask && is_pinned_host(out), aux_pinned = ask && aux && is_pinned_host(aux);
    const size_t n_chunks = (n + chunk - 1) / chunk;
    struct Pending { size_t first, count; bool live; } pend[2] = {{0, 0, false}, {0, 0, false}};
    auto retire = [&](int k) -> int {      // chunk in slot k has finished: hand its results to the caller
        gpis_medium::HostSlot &S = m->hs[k];
        if (!pend[k].live) return GPIS_OK;
        HIP_TRY(hipEventSynchronize(S.done));
        if (!out_pinned) memcpy((char *)out + pend[k].first * out_rec, S.pin_out, pend[k].count * out_rec);
        if (aux && !aux_pinned) memcpy(aux + pend[k].first, S.pin_aux, pend[k].count * kAuxRec);
        pend[k].live = false;
        return GPIS_OK;
    };
    for (size_t c = 0; c < n_chunks; ++c) {
        const int k = (int)(c & 1);
        gpis_medium::HostSlot &S = m->hs[k];
        if ((st = retire(k))) return st;
        const size_t first = c * chunk, count = n - first < chunk ? n - first : chunk;
        const void *src = rays + first;
        if (!in_pinned) { memcpy(S.pin_in, src, count * kInRec); src = S.pin_in; }
        HIP_TRY(hipMemcpyAsync(S.dev_in, src, count * kInRec, hipMemcpyHostToDevice, S.stream));
        if (sample)
            st = sample_distance_impl(m, count, (const gpis_ray_in *)S.dev_in, (gpis_seg_out *)S.dev_out, aux ? (gpis_cond_coeff *)S.dev_aux : nullptr, nullptr, S.stream);
        else
            st = transmittance_impl(m, count, (const gpis_ray_in *)S.dev_in, (uint8_t *)S.dev_out, nullptr, S.stream);
        if (st) return st;
        HIP_TRY(hipMemcpyAsync(out_pinned ? (char *)out + first * out_rec : S.pin_out, S.dev_out, count * out_rec, hipMemcpyDeviceToHost, S.stream));
        if (aux) HIP_TRY(hipMemcpyAsync(aux_pinned ? (char *)(aux + first) : S.pin_aux, S.dev_aux, count * kAuxRec, hipMemcpyDeviceToHost, S.stream));
        HIP_TRY(hipEventRecord(S.done, S.stream));
        pend[k] = {first, count, true};
    }
    if ((st = retire((int)(n_chunks & 1)))) return st;       // the older chunk first
    return retire((int)((n_chunks + 1) & 1));
}
extern "C" int gpis_sample_distance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff)
{
    CHECK_ARGS(m && (n == 0 || (rays && out)));
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    return host_march(m, true, n, rays, out, coeff);
}
extern "C" int gpis_transmittance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint8_t *visible)
{
    CHECK_ARGS(m && (n == 0 || (rays && visible)));
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    return host_march(m, false, n, rays, visible, nullptr);
}
extern "C" void *gpis_alloc_host(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void gpis_free_host(void *p)
{
    if (p) (void)hipHostFree(p);
}
extern "C" int gpis_eval_value_host(gpis_medium *m, size_t n, const gpis_query *q, float *value, int32_t *gp_id)
{
    CHECK_ARGS(m && (n == 0 || (q && value)));
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    int st;
    if ((st = ensure_stage(m, 0, n * sizeof(gpis_query))) || (st = ensure_stage(m, 1, n * sizeof(float))) || (st = ensure_stage(m, 2, n * sizeof(int32_t))))
        return st;
    HIP_TRY(hipMemcpy(m->stage[0], q, n * sizeof(gpis_query), hipMemcpyHostToDevice));
    st = gpis_eval_value_batch(m, n, (const gpis_query *)m->stage[0], (float *)m->stage[1], (int32_t *)m->stage[2], nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(value, m->stage[1], n * sizeof(float), hipMemcpyDeviceToHost));
    if (gp_id) HIP_TRY(hipMemcpy(gp_id, m->stage[2], n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GPIS_OK;
}
extern "C" int gpis_eval_gradient_host(gpis_medium *m, size_t n, const gpis_query *q, float *grad3)
{
    CHECK_ARGS(m && (n == 0 || (q && grad3)));
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    int st;
    if ((st = ensure_stage(m, 0, n * sizeof(gpis_query))) || (st = ensure_stage(m, 1, 3 * n * sizeof(float))))
        return st;
    HIP_TRY(hipMemcpy(m->stage[0], q, n * sizeof(gpis_query), hipMemcpyHostToDevice));
    st = gpis_eval_gradient_batch(m, n, (const gpis_query *)m->stage[0], (float *)m->stage[1], nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(grad3, m->stage[1], 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    return GPIS_OK;
}

extern "C" int gpis_conditioning_host(gpis_medium *m, size_t n, const gpis_query *q, const float *target_val, const float *target_grad3,
                                      gpis_cond_coeff *coeff_out)
{
    CHECK_ARGS(m && (n == 0 || (q && target_val && target_grad3 && coeff_out)));
    if (n == 0) return GPIS_OK;
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    int st;
    if ((st = ensure_stage(m, 0, n * sizeof(gpis_query))) || (st = ensure_stage(m, 1, 4 * n * sizeof(float))) ||
        (st = ensure_stage(m, 2, n * sizeof(gpis_cond_coeff))))
        return st;
    float *d_tv = (float *)m->stage[1], *d_tg = d_tv + n;
    HIP_TRY(hipMemcpy(m->stage[0], q, n * sizeof(gpis_query), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tv, target_val, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_tg, target_grad3, 3 * n * sizeof(float), hipMemcpyHostToDevice));
    st = gpis_conditioning_batch(m, n, (const gpis_query *)m->stage[0], d_tv, d_tg, (gpis_cond_coeff *)m->stage[2], nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(coeff_out, m->stage[2], n * sizeof(gpis_cond_coeff), hipMemcpyDeviceToHost));
    return GPIS_OK;
}
static int nee_host(gpis_medium *m, size_t n, const gpis_nee_query *q, float *pdf, float *grad3)
{
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    int st;
    if ((st = ensure_stage(m, 0, n * sizeof(gpis_nee_query))) || (st = ensure_stage(m, 1, 3 * n * sizeof(float))))
        return st;
    HIP_TRY(hipMemcpy(m->stage[0], q, n * sizeof(gpis_nee_query), hipMemcpyHostToDevice));
    st = pdf ? gpis_nee_pdf_batch(m, n, (const gpis_nee_query *)m->stage[0], (float *)m->stage[1], nullptr)
             : gpis_nee_grad_batch(m, n, (const gpis_nee_query *)m->stage[0], (float *)m->stage[1], nullptr);
    if (st) return st;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(pdf ? pdf : grad3, m->stage[1], (pdf ? 1 : 3) * n * sizeof(float), hipMemcpyDeviceToHost));
    return GPIS_OK;
}
extern "C" int gpis_nee_pdf_host(gpis_medium *m, size_t n, const gpis_nee_query *q, float *pdf)
{
    CHECK_ARGS(m && (n == 0 || (q && pdf)));
    return n ? nee_host(m, n, q, pdf, nullptr) : GPIS_OK;
}
extern "C" int gpis_nee_grad_host(gpis_medium *m, size_t n, const gpis_nee_query *q, float *grad3)
{
    CHECK_ARGS(m && (n == 0 || (q && grad3)));
    return n ? nee_host(m, n, q, nullptr, grad3) : GPIS_OK;
}

// ---- measurement -------------------------------------------------------------------------
extern "C" int gpis_get_counters(gpis_medium *m, uint64_t *n_eval, uint64_t *n_seg)
{
    CHECK_ARGS(m);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    Counters c[2];
    HIP_TRY(hipMemcpy(c, m->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (n_eval) *n_eval = c[0].n_eval + c[1].n_eval;
    if (n_seg) *n_seg = c[0].n_seg + c[1].n_seg;
    return GPIS_OK;
}
extern "C" int gpis_reset_counters(gpis_medium *m)
{
    CHECK_ARGS(m);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(m->d_counters, 0, 2 * sizeof(Counters)));
    HIP_TRY(hipMemset(m->d_guide_cnt, 0, 8 * sizeof(unsigned long long)));
    for (int k = 0; k < 3; ++k) { m->events_used[k] = 0; m->prof_ms[k] = 0.; m->prof_launches[k] = 0; }
    return GPIS_OK;
}

// ---- guide field ---------------------------------------------------------------------------
extern "C" int gpis_build_guide(gpis_medium *m, int half_extent_cells, int points_per_cell)
{
    CHECK_ARGS(m);
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!m->fast.enabled)
        return set_err(GPIS_ERR_UNSUPPORTED, "gpis_build_guide: the medium is not covered by the wave-cooperative path (single_realization, 3D, stationary SE, diagonal anisotropy)");
    int st = launch::guide_build(m->host_model, m->d_model, m->fast, half_extent_cells, points_per_cell, &m->guide, getenv("GPIS_GUIDE_DENSE") == nullptr);
    if (st != GPIS_OK)
        return set_err(st, "gpis_build_guide(half=%d, ppc=%d) failed: %s", half_extent_cells, points_per_cell,
                       st == GPIS_ERR_UNSUPPORTED ? "unsupported arguments" : hipGetErrorString(hipGetLastError()));
    HIP_TRY(hipMemcpy(m->d_guide, &m->guide, sizeof(GuideField), hipMemcpyHostToDevice));
    return GPIS_OK;
}
extern "C" int gpis_drop_guide(gpis_medium *m)
{
    CHECK_ARGS(m);
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    guide_free(&m->guide);
    HIP_TRY(hipMemcpy(m->d_guide, &m->guide, sizeof(GuideField), hipMemcpyHostToDevice));
    return GPIS_OK;
}
extern "C" int gpis_get_guide_steps(gpis_medium *m, uint64_t *n_guide)
{
    CHECK_ARGS(m && n_guide);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, m->d_guide_cnt, sizeof v, hipMemcpyDeviceToHost));
    *n_guide = v;
    return GPIS_OK;
}
extern "C" int gpis_guide_selfcheck(gpis_medium *m, size_t n, const float *points3, uint64_t *checked, uint64_t *violations,
                                    float *max_ratio, float *mean_bound, void *stream)
{
    CHECK_ARGS(m && (n == 0 || points3));
    if (!m->guide.enabled) return set_err(GPIS_ERR_UNSUPPORTED, "gpis_guide_selfcheck: no guide field built");
    HIP_TRY(hipSetDevice(m->device));
    // scratch: [0] = points inside the field, [1] = violations, word 2 = {max ratio bits, sum of bounds}, [3] = points in tabulated bricks
    unsigned long long *st = m->d_guide_cnt + 1;
    HIP_TRY(hipMemsetAsync(st, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));
    if (n)
        launch::guide_selfcheck(m->d_model, m->fast, m->guide, n, points3, st, (float *)(st + 2), (float *)(st + 2) + 1, (hipStream_t)stream);
    int rc = launch_check("k_guide_selfcheck");
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long h[4];
    HIP_TRY(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    float fr[2];
    memcpy(fr, &h[2], sizeof fr);
    if (checked) *checked = h[0];
    if (violations) *violations = h[1];
    if (max_ratio) *max_ratio = fr[0];
    if (mean_bound) *mean_bound = h[3] ? fr[1] / (float)h[3] : 0.f;
    m->selfcheck_tabulated = h[3];
    return GPIS_OK;
}
extern "C" int gpis_get_guide_info(gpis_medium *m, gpis_guide_info *out)
{
    CHECK_ARGS(m && out);
    memset(out, 0, sizeof *out);
    if (!m->guide.enabled) return GPIS_OK;
    const GuideField &F = m->guide;
    const uint64_t nblk = (uint64_t)(F.side / 4) * (F.side / 4) * (F.side / 4);
    out->half_extent_cells = F.half; out->points_per_cell = F.ppc;
    out->bricks_total = nblk / 64; out->bricks_allocated = F.n_alloc; out->bricks_usable = F.n_usable;
    out->bytes_samples = (uint64_t)(F.n_alloc ? F.n_alloc : 1u) * kBrickFloats * 4;
    out->bytes_bounds = nblk * 8;
    out->bytes_dense = (uint64_t)F.side * F.side * F.side * 4;
    out->selfcheck_points_tabulated = m->selfcheck_tabulated;
    return GPIS_OK;
}

extern "C" int gpis_guide_raycheck(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint32_t steps, uint64_t *certified,
                                   uint64_t *violations, void *stream)
{
    CHECK_ARGS(m && (n == 0 || rays));
    if (!m->guide.enabled) return set_err(GPIS_ERR_UNSUPPORTED, "gpis_guide_raycheck: no guide field built");
    HIP_TRY(hipSetDevice(m->device));
    unsigned long long *st = m->d_guide_cnt + 1;
    HIP_TRY(hipMemsetAsync(st, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream));
    if (n)
        launch::guide_raycheck(m->d_model, m->fast, m->guide, n, rays, steps, st, (hipStream_t)stream);
    int rc = launch_check("k_guide_raycheck");
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
    if (certified) *certified = h[0];
    if (violations) *violations = h[1];
    return GPIS_OK;
}

extern "C" int gpis_set_profiling(gpis_medium *m, int enable)
{
    CHECK_ARGS(m);
    m->profiling = enable != 0;
    return GPIS_OK;
}
extern "C" int gpis_get_kernel_profile(gpis_medium *m, int which, double *total_ms, uint64_t *launches, uint64_t *n_eval, uint64_t *n_seg)
{
    CHECK_ARGS(m && which >= 0 && which <= 2);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    for (size_t i = 0; i < m->events_used[which]; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, m->events[which][i].first, m->events[which][i].second));
        m->prof_ms[which] += (double)ms;
        m->prof_launches[which]++;
    }
    m->events_used[which] = 0;
    Counters c[2];
    HIP_TRY(hipMemcpy(c, m->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (total_ms) *total_ms = m->prof_ms[which];
    if (launches) *launches = m->prof_launches[which];
    if (n_eval) *n_eval = which < 2 ? c[which].n_eval : 0;      // the NEE queries count into the sampleDistance pair
    if (n_seg) *n_seg = which < 2 ? c[which].n_seg : 0;
    return GPIS_OK;
}

#ifdef GPIS_FAST_STATS
// diagnostic build only: read and clear the cooperative loop's work counters (of the guided sampleDistance kernel's TU)
extern "C" int gpis_debug_fast_stats(uint64_t *out32)
{
    unsigned long long h[32];
    int st = launch::fast_stats_read(h);
    if (st != GPIS_OK) return st;
    for (int i = 0; i < 32; ++i) out32[i] = h[i];
    return GPIS_OK;
}
#endif

// ---- scene S driver ----------------------------------------------------------------------
extern "C" void gpis_default_scene_s(gpis_scene_s *s, uint32_t width, uint32_t height, uint32_t spp)
{
    memset(s, 0, sizeof *s);
    s->width = width; s->height = height;
    s->spp_begin = 0; s->spp_count = spp;
    s->scene_seed = 0xBA5EBA11u;
    s->tile_size = 16;
    s->cam_pos[0] = 0.f; s->cam_pos[1] = 0.f; s->cam_pos[2] = 4.f;
    s->cam_fov_deg = 35.f;
    s->bound_radius = 1.5f;
    s->light_dir[0] = 0.5f; s->light_dir[1] = 0.7f; s->light_dir[2] = 0.5f;
    s->light_radiance = 1.f;
    s->y_begin = 0; s->y_count = height;
    s->shard_index = 0; s->shard_count = 1;
}

// Samples per chunk of the tile drivers: 2^default_log2 unless GPIS_CHUNK_LOG2 says otherwise.  Chunks are
// sized for the 288 GB of an MI355X — one launch per stage and frame where it fits — because every launch
// ends in a tail of half-empty waves and the wavefront march in a tail of thin iterations (C1 frame:
// 8 Mi-sample chunks 268.8, 32 Mi 284.8, 128 Mi 291.3 Msamples/s; multi-bounce 65 → 80 M paths/s).
static int chunk_log2(const gpis_medium *m, int default_log2)
{
    return m->opt[GPIS_OPT_CHUNK_LOG2] ? (int)m->opt[GPIS_OPT_CHUNK_LOG2] : default_log2;
}
// grows stage[slot] to `bytes` if the device has the memory; false (and no error state) otherwise
static bool try_stage(gpis_medium *m, int slot, size_t bytes)
{
    if (stage_size(m, slot) >= bytes)
        return true;
    if (ensure_stage(m, slot, bytes) == GPIS_OK)
        return true;
    (void)hipGetLastError();
    return false;
}

static SceneConst make_scene_const(const gpis_scene_s *s)
{
    SceneConst sc;
    sc.s = *s;
    const float pi_f = 3.1415926536f;
    float fov_rad = s->cam_fov_deg * (pi_f / 180.0f);
    sc.plane_dist = 1.0f / tanf(fov_rad * 0.5f);
    sc.ratio = (float)s->height / (float)s->width;
    sc.psx = 1.0f / (float)s->width;
    {
        float lx = s->light_dir[0], ly = s->light_dir[1], lz = s->light_dir[2];
        float l2 = 0.f; l2 += lx * lx; l2 += ly * ly; l2 += lz * lz;
        float inv = 1.0f / sqrtf(l2);
        sc.light[0] = lx * inv; sc.light[1] = ly * inv; sc.light[2] = lz * inv;
    }
    return sc;
}

// Workspace of the Lambert driver, 236 B per sample in three allocations: slot 3 [primary rays, overwritten in place by the shadow
// rays | shadow jitter | mask], slot 5 [segment results], slot 6 [cos | mask | visible | hit].  The chunk is the largest the
// device can hold, starting from the whole frame (2^27 samples = 31.3 GB; 48.3 GB with separate shadow rays until round 3).
struct LambertWs {
    size_t chunk_pixels, ns_max;
    size_t b_prim, b_seg, b_shadow;                    // bytes of slots 3, 5, 6
    size_t o_prim, o_us, o_v1;                         // offsets inside slot 3
    size_t o_cos, o_v2, o_vis, o_hit;                  // offsets inside slot 6
};
// First-use policy: memory this process allocates for the first time costs 15-30 ms per GB on these boxes (DESIGN.md §6 "cold
// frame"), so the whole-frame workspace (31 GB) would triple the latency of a handle's FIRST frame.  That frame therefore runs in
// 16 Mi-sample chunks (4 GB, -4 % throughput); from the second call on — the handle is evidently reused — the workspace grows to
// the whole frame.  A workspace that is already large enough (gpis_reserve_scene_workspace, an earlier larger call) is used as is,
// and GPIS_CHUNK_LOG2 / GPIS_OPT_CHUNK_LOG2 override the policy.
static int lambert_ws_plan(gpis_medium *m, const gpis_scene_s *s, size_t total_pixels, LambertWs &W, bool first_call = false)
{
    int l0 = chunk_log2(m, 27);
    if (first_call && !m->opt[GPIS_OPT_CHUNK_LOG2]) {
        const size_t whole = (total_pixels * s->spp_count < ((size_t)1 << 27) ? total_pixels * s->spp_count : ((size_t)1 << 27));
        const bool have_whole = stage_size(m, 3) >= whole * (sizeof(gpis_ray_in) + 5) && stage_size(m, 5) >= whole * sizeof(gpis_seg_out) && stage_size(m, 6) >= whole * 7;
        if (!have_whole) l0 = 24;
    }
    for (int l = l0;; --l) {
        W.chunk_pixels = ((size_t)1 << l) / s->spp_count;
        if (W.chunk_pixels < 1) W.chunk_pixels = 1;
        if (W.chunk_pixels > total_pixels) W.chunk_pixels = total_pixels;
        const size_t n = W.ns_max = W.chunk_pixels * s->spp_count;
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        W.o_prim = carve(n * sizeof(gpis_ray_in)); W.o_us = carve(n * 4); W.o_v1 = carve(n);
        W.b_prim = off;
        W.b_seg = n * sizeof(gpis_seg_out);
        off = 0;
        W.o_cos = carve(n * 4); W.o_v2 = carve(n); W.o_vis = carve(n); W.o_hit = carve(n);
        W.b_shadow = off;
        // what is still to be allocated must fit the free memory (with 1 GiB to spare), otherwise halve the chunk
        const size_t have3 = stage_size(m, 3), have5 = stage_size(m, 5), have6 = stage_size(m, 6);
        const size_t missing = (have3 >= W.b_prim ? 0 : W.b_prim) + (have5 >= W.b_seg ? 0 : W.b_seg) + (have6 >= W.b_shadow ? 0 : W.b_shadow);
        if (missing == 0) return GPIS_OK;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t reclaim = (have3 >= W.b_prim ? 0 : have3) + (have5 >= W.b_seg ? 0 : have5) + (have6 >= W.b_shadow ? 0 : have6);   // freed on growth
        if (missing + ((size_t)1 << 30) <= free_b + reclaim) return GPIS_OK;
        if (l <= 20) return set_err(GPIS_ERR_DEVICE, "scene driver: no memory for a 1 Mi-sample workspace");
    }
}

extern "C" int gpis_set_variance_grid(gpis_medium *m, const gpis_variance_grid *g, const float *voxels)
{
    CHECK_ARGS(m && g && voxels);
    CHECK_ARGS(g->dims[0] >= 1 && g->dims[1] >= 1 && g->dims[2] >= 1 && (g->interpolate == 0 || g->interpolate == 1));
    if (!m->host_model.grid.on) return set_err(GPIS_ERR_INVALID_ARG, "gpis_set_variance_grid: the medium was not created with grid_nonstationary = 1");
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = (size_t)g->dims[0] * (size_t)g->dims[1] * (size_t)g->dims[2];
    float *d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(float)));
    if (hipMemcpy(d, voxels, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return set_err(GPIS_ERR_DEVICE, "gpis_set_variance_grid: copy failed"); }
    if (m->d_grid_vox) HIP_TRY(hipFree(m->d_grid_vox));
    m->d_grid_vox = d;
    DevGrid &G = m->host_model.grid;
    G.vox = d;
    G.interpolate = g->interpolate;
    for (int c = 0; c < 3; ++c) { G.dims[c] = g->dims[c]; G.origin[c] = g->origin[c]; G.lo[c] = g->bounds_min[c] + 2; G.hi[c] = g->bounds_max[c] - 3; }
    for (int i = 0; i < 12; ++i) G.T[i] = g->inv_natural_transform[i];
    HIP_TRY(hipMemcpy(m->d_model, &m->host_model, sizeof(DevModel), hipMemcpyHostToDevice));
    return GPIS_OK;
}

extern "C" int gpis_reserve_scene_workspace(gpis_medium *m, const gpis_scene_s *s)
{
    CHECK_ARGS(m && s);
    CHECK_ARGS(scene_args_ok(s));
    HIP_TRY(hipSetDevice(m->device));
    LambertWs W;
    if (int rc = lambert_ws_plan(m, s, scene_rows(*s) * s->width, W)) return rc;
    if (int rc = ensure_stage(m, 3, W.b_prim, true)) return rc;      // in the order the first frame needs them
    if (int rc = ensure_stage(m, 5, W.b_seg, true)) return rc;
    return ensure_stage(m, 6, W.b_shadow, true);
}

extern "C" int gpis_render_scene_s(gpis_medium *m, const gpis_scene_s *s, float *radiance_sum, uint32_t *hit_count, void *stream)
{
    CHECK_ARGS(m && s && radiance_sum);
    CHECK_ARGS(scene_args_ok(s));
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    SceneConst sc = make_scene_const(s);
    const size_t total_pixels = scene_rows(*s) * s->width;
    LambertWs W;
    if (int rc0 = lambert_ws_plan(m, s, total_pixels, W, m->lambert_calls == 0)) return rc0;
    ++m->lambert_calls;
    int rc = GPIS_OK;
    if ((rc = ws_acquire(m, 0, st))) return rc;
    // each of the three allocations is made (first frame only) right before the first kernel that needs it, i.e. while the kernels
    // launched so far run
    if ((rc = ensure_stage(m, 3, W.b_prim, true))) return rc;
    char *ws3 = (char *)m->stage[3];
    gpis_ray_in *prim = (gpis_ray_in *)(ws3 + W.o_prim);
    float *us = (float *)(ws3 + W.o_us);
    uint8_t *v1 = (uint8_t *)(ws3 + W.o_v1);
    const size_t chunk_pixels = W.chunk_pixels;
    for (size_t p0 = 0; p0 < total_pixels; p0 += chunk_pixels) {
        size_t np = total_pixels - p0 < chunk_pixels ? total_pixels - p0 : chunk_pixels;
        size_t ns = np * s->spp_count;
        k_scene_primary<<<grid_of(ns, 256), 256, 0, st>>>(sc, p0, ns, prim, us, v1);
        if ((rc = launch_check("k_scene_primary"))) return rc;
        if ((rc = ensure_stage(m, 5, W.b_seg, true))) return rc;
        gpis_seg_out *seg = (gpis_seg_out *)m->stage[5];
        if ((rc = sample_distance_impl(m, ns, prim, seg, nullptr, v1, st))) return rc;
        if ((rc = ensure_stage(m, 6, W.b_shadow, true))) return rc;
        char *ws6 = (char *)m->stage[6];
        gpis_ray_in *sh = prim;
        float *cosl = (float *)(ws6 + W.o_cos);
        uint8_t *v2 = (uint8_t *)(ws6 + W.o_v2), *vis = (uint8_t *)(ws6 + W.o_vis), *hit = (uint8_t *)(ws6 + W.o_hit);
        k_scene_shade<<<grid_of(ns, 256), 256, 0, st>>>(sc, ns, prim, seg, us, v1, sh, cosl, v2, hit);
        if ((rc = launch_check("k_scene_shade"))) return rc;
        if ((rc = transmittance_impl(m, ns, sh, vis, v2, st))) return rc;
        k_scene_accumulate<<<grid_of(np, 256), 256, 0, st>>>(sc, p0, np, cosl, v2, vis, hit, radiance_sum, hit_count);
        if ((rc = launch_check("k_scene_accumulate"))) return rc;
    }
    return ws_release(m, 0, st);
}

extern "C" int gpis_render_scene_s_paths(gpis_medium *m, const gpis_scene_s *s, int max_path_bounces, float albedo, float *radiance_sum, void *stream)
{
    CHECK_ARGS(m && s && radiance_sum && max_path_bounces >= 1);
    CHECK_ARGS(scene_args_ok(s));
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    SceneConst sc = make_scene_const(s);
    const size_t total_pixels = scene_rows(*s) * s->width;
    size_t chunk_pixels = ((size_t)1 << chunk_log2(m, 25)) / s->spp_count;
    if (chunk_pixels < 1) chunk_pixels = 1;
    if (chunk_pixels > total_pixels) chunk_pixels = total_pixels;
    const size_t ns_max = chunk_pixels * s->spp_count;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_rays = carve(ns_max * sizeof(gpis_ray_in)), o_seg = carve(ns_max * sizeof(gpis_seg_out)), o_sh = carve(ns_max * sizeof(gpis_ray_in));
    size_t o_rng = carve(ns_max * 8), o_thr = carve(ns_max * 4), o_em = carve(ns_max * 4), o_con = carve(ns_max * 4);
    size_t o_alive = carve(ns_max), o_nee = carve(ns_max), o_vis = carve(ns_max);
    // regrouping of secondary segments (GPIS_PATHS_SORT=0 keeps the sample order: results are identical)
    const bool regroup = m->opt[GPIS_OPT_PATHS_SORT] != 0 && max_path_bounces > 2;
    const bool presort = m->opt[GPIS_OPT_PATHS_PRESORT] != 0;
    size_t sort_temp_bytes = 0;
    size_t o_keys = 0, o_vals = 0, o_keys2 = 0, o_order = 0, o_rsorted = 0, o_live = 0, o_temp = 0, o_order2 = 0, o_live2 = 0;
    if (regroup) {
        if (sort_pairs_u32(nullptr, sort_temp_bytes, nullptr, nullptr, nullptr, nullptr, ns_max, st) != hipSuccess)
            return set_err(GPIS_ERR_DEVICE, "radix sort scratch query failed");
        o_keys = carve(ns_max * 4); o_vals = carve(ns_max * 4); o_keys2 = carve(ns_max * 4); o_order = carve(ns_max * 4);
        o_rsorted = carve(ns_max * sizeof(gpis_ray_in)); o_live = carve(ns_max); o_temp = carve(sort_temp_bytes);
        o_order2 = carve(ns_max * 4); o_live2 = carve(ns_max);
    }
    int rc = ensure_stage(m, 3, off);
    if (rc) return rc;
    if ((rc = ws_acquire(m, 0, st))) return rc;
    char *ws = (char *)m->stage[3];
    uint32_t *keys = (uint32_t *)(ws + o_keys), *vals = (uint32_t *)(ws + o_vals), *keys2 = (uint32_t *)(ws + o_keys2), *order = (uint32_t *)(ws + o_order);
    gpis_ray_in *rays_sorted = (gpis_ray_in *)(ws + o_rsorted);
    uint8_t *live_sorted = (uint8_t *)(ws + o_live), *live2 = (uint8_t *)(ws + o_live2);
    uint32_t *order2 = (uint32_t *)(ws + o_order2);
    const DevModel &H = m->host_model;
    const float cell_size = H.iso3d ? (H.radius_iso > 0.f ? H.radius_iso : H.kernel_scale) : (H.radius_world > 0.f ? H.radius_world : 0.1f);
    PathArrays a;
    a.rays = (gpis_ray_in *)(ws + o_rays); a.seg = (gpis_seg_out *)(ws + o_seg); a.shadow = (gpis_ray_in *)(ws + o_sh);
    a.rng = (uint64_t *)(ws + o_rng);
    a.throughput = (float *)(ws + o_thr); a.emission = (float *)(ws + o_em); a.contrib = (float *)(ws + o_con);
    a.alive = (uint8_t *)(ws + o_alive); a.nee = (uint8_t *)(ws + o_nee); a.vis = (uint8_t *)(ws + o_vis);
    for (size_t p0 = 0; p0 < total_pixels; p0 += chunk_pixels) {
        size_t np = total_pixels - p0 < chunk_pixels ? total_pixels - p0 : chunk_pixels;
        size_t ns = np * s->spp_count;
        k_paths_begin<<<grid_of(ns, 256), 256, 0, st>>>(sc, p0, ns, a);
        if ((rc = launch_check("k_paths_begin"))) return rc;
        // the segment of bounce max-1 cannot contribute (no NEE there, TraceBase.cpp:546, and the
        // light is a Dirac delta), so it is not traced
        for (int bounce = 0; bounce + 1 < max_path_bounces; ++bounce) {
            // primary segments are coherent as generated (consecutive spp of a pixel); later ones are regrouped
            // Secondary segments are regrouped here by start cell (compaction + locality of the guide steps;
            // GPIS_PATHS_PRESORT=0 skips it when the wavefront march runs, which regroups the exact work
            // itself).  Measured at 4 bounces, C1 1080p x16: sample order + resident march 22.1 M paths/s,
            // regrouped + resident 40.4, wavefront march alone 56.7, regrouped + wavefront 61.3.
            const bool wave = bounce > 0 && m->guide.enabled && wave_march_selected(m, MARCH_SCATTERED);
            const bool sorted = regroup && bounce > 0 && (!wave || presort);
            const int hint = bounce > 0 ? MARCH_SCATTERED : MARCH_COHERENT;
            // src rays + mask -> order, regrouped copy, live flags of the regrouped slots
            auto regroup_batch = [&](const gpis_ray_in *src, const uint8_t *mask, uint32_t *ord, gpis_ray_in *dst, uint8_t *live_out) -> int {
                k_paths_keys<<<grid_of(ns, 256), 256, 0, st>>>(m->d_model, cell_size, ns, src, mask, keys, vals);
                int r = launch_check("k_paths_keys");
                if (r) return r;
                size_t tb = sort_temp_bytes;
                if (sort_pairs_u32(ws + o_temp, tb, keys, keys2, vals, ord, ns, st) != hipSuccess)
                    return set_err(GPIS_ERR_DEVICE, "radix sort failed");
                k_paths_gather<<<grid_of(ns, 256), 256, 0, st>>>(ns, keys2, ord, src, dst, live_out);
                return launch_check("k_paths_gather");
            };
            if (sorted && (rc = regroup_batch(a.rays, a.alive, order, rays_sorted, live_sorted))) return rc;
            const gpis_ray_in *rays_in = sorted ? rays_sorted : a.rays;
            const uint8_t *live = sorted ? live_sorted : a.alive;
            if ((rc = sample_distance_impl(m, ns, rays_in, a.seg, nullptr, live, st, hint))) return rc;
            k_paths_shade<<<grid_of(ns, 256), 256, 0, st>>>(sc, ns, bounce, max_path_bounces, albedo, a, sorted ? order : nullptr, rays_in,
                                                           sorted ? live_sorted : nullptr);
            if ((rc = launch_check("k_paths_shade"))) return rc;
            if (sorted) {
                // the shadow segments share one direction but start where the bounce segments ended: regroup
                // them by their own lattice cells (the bounce batch's copy of the rays is free again)
                if ((rc = regroup_batch(a.shadow, a.nee, order2, rays_sorted, live2))) return rc;
                if ((rc = transmittance_impl(m, ns, rays_sorted, a.vis, live2, st, hint))) return rc;
                k_paths_nee_add<<<grid_of(ns, 256), 256, 0, st>>>(ns, a, order, order2, live2, a.vis);
            } else {
                if ((rc = transmittance_impl(m, ns, a.shadow, a.vis, a.nee, st, hint))) return rc;
                k_paths_nee_add<<<grid_of(ns, 256), 256, 0, st>>>(ns, a, sorted ? order : nullptr, nullptr, nullptr, a.vis);
            }
            if ((rc = launch_check("k_paths_nee_add"))) return rc;
        }
        k_paths_accumulate<<<grid_of(np, 256), 256, 0, st>>>(sc, p0, np, a.emission, radiance_sum);
        if ((rc = launch_check("k_paths_accumulate"))) return rc;
    }
    return ws_release(m, 0, st);
}

extern "C" int gpis_render_scene_s_nee(gpis_medium *m, const gpis_scene_s *s, const gpis_surface_s *surf, float *radiance_sum, void *stream)
{
    CHECK_ARGS(m && s && surf && radiance_sum);
    CHECK_ARGS(scene_args_ok(s));
    CHECK_ARGS(surf->cap_cos < 1.0f && surf->cap_cos > -1.0f);
    std::lock_guard<std::mutex> lock(m->mu);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    SceneConst sc = make_scene_const(s);
    const size_t total_pixels = scene_rows(*s) * s->width;
    // ≈ 800 B per sample (rays, results, the two query records, two shadow rays, masks): the largest chunk the device can hold,
    // starting from the whole frame (2^27 samples ≈ 106 GB for C2's 1920x1080x64) — every launch ends in a tail of half-empty waves
    PathArrays a;
    NeeArrays b;
    size_t chunk_pixels = 0, ns_max = 0, off = 0;
    size_t o_rays = 0, o_seg = 0, o_rng = 0, o_thr = 0, o_em = 0, o_alive = 0, o_coeff = 0, o_qh = 0, o_qn = 0, o_aux = 0, o_shl = 0, o_shp = 0;
    size_t o_ph = 0, o_gh = 0, o_pn = 0, o_cl = 0, o_cp = 0, o_f[7] = {0};
    int rc = GPIS_OK;
    for (int l = chunk_log2(m, 27);; --l) {
        chunk_pixels = ((size_t)1 << l) / s->spp_count;
        if (chunk_pixels < 1) chunk_pixels = 1;
        if (chunk_pixels > total_pixels) chunk_pixels = total_pixels;
        ns_max = chunk_pixels * s->spp_count;
        off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        o_rays = carve(ns_max * sizeof(gpis_ray_in)); o_seg = carve(ns_max * sizeof(gpis_seg_out));
        o_rng = carve(ns_max * 8); o_thr = carve(ns_max * 4); o_em = carve(ns_max * 4); o_alive = carve(ns_max);
        o_coeff = carve(ns_max * sizeof(gpis_cond_coeff)); o_qh = carve(ns_max * sizeof(gpis_nee_query)); o_qn = carve(ns_max * sizeof(gpis_nee_query));
        o_aux = carve(ns_max * sizeof(NeeAux)); o_shl = carve(ns_max * sizeof(gpis_ray_in)); o_shp = carve(ns_max * sizeof(gpis_ray_in));
        o_ph = carve(ns_max * 4); o_gh = carve(ns_max * 12); o_pn = carve(ns_max * 4); o_cl = carve(ns_max * 4); o_cp = carve(ns_max * 4);
        for (int k = 0; k < 7; ++k) o_f[k] = carve(ns_max);
        if (try_stage(m, 3, off))
            break;
        if (l <= 20) return set_err(GPIS_ERR_DEVICE, "NEE driver: no memory for a 1 Mi-sample workspace");
    }
    if ((rc = ws_acquire(m, 0, st))) return rc;
    char *ws = (char *)m->stage[3];
    a.rays = (gpis_ray_in *)(ws + o_rays); a.seg = (gpis_seg_out *)(ws + o_seg); a.shadow = nullptr;
    a.rng = (uint64_t *)(ws + o_rng); a.throughput = (float *)(ws + o_thr); a.emission = (float *)(ws + o_em); a.contrib = nullptr;
    a.alive = (uint8_t *)(ws + o_alive); a.nee = nullptr; a.vis = nullptr;
    b.coeff = (gpis_cond_coeff *)(ws + o_coeff); b.q_half = (gpis_nee_query *)(ws + o_qh); b.q_normal = (gpis_nee_query *)(ws + o_qn);
    b.aux = (NeeAux *)(ws + o_aux); b.shadow_light = (gpis_ray_in *)(ws + o_shl); b.shadow_phase = (gpis_ray_in *)(ws + o_shp);
    b.pdf_half = (float *)(ws + o_ph); b.grad_half = (float *)(ws + o_gh); b.pdf_normal = (float *)(ws + o_pn);
    b.contrib_light = (float *)(ws + o_cl); b.contrib_phase = (float *)(ws + o_cp);
    b.want_light = (uint8_t *)(ws + o_f[0]); b.want_phase = (uint8_t *)(ws + o_f[1]); b.want_pdf_normal = (uint8_t *)(ws + o_f[2]);
    b.go_light = (uint8_t *)(ws + o_f[3]); b.go_phase = (uint8_t *)(ws + o_f[4]); b.vis_light = (uint8_t *)(ws + o_f[5]); b.vis_phase = (uint8_t *)(ws + o_f[6]);
    for (size_t p0 = 0; p0 < total_pixels; p0 += chunk_pixels) {
        size_t np = total_pixels - p0 < chunk_pixels ? total_pixels - p0 : chunk_pixels;
        size_t ns = np * s->spp_count;
        k_paths_begin<<<grid_of(ns, 256), 256, 0, st>>>(sc, p0, ns, a);
        if ((rc = launch_check("k_paths_begin"))) return rc;
        if ((rc = sample_distance_impl(m, ns, a.rays, a.seg, b.coeff, a.alive, st))) return rc;
        k_nee_setup<<<grid_of(ns, 256), 256, 0, st>>>(sc, *surf, ns, a, b);
        if ((rc = launch_check("k_nee_setup"))) return rc;
        {
            ProfScope prof(m, 2, st);
            launch::nee(nee_instance(m->host_model), m->d_model, ns, b.q_half, b.pdf_half, b.grad_half, m->d_counters, b.want_light, st);
            if ((rc = launch_check("k_nee"))) return rc;
            launch::nee(nee_instance(m->host_model), m->d_model, ns, b.q_normal, b.pdf_normal, nullptr, m->d_counters, b.want_pdf_normal, st);
            if ((rc = launch_check("k_nee"))) return rc;
        }
        k_nee_shade<<<grid_of(ns, 256), 256, 0, st>>>(sc, *surf, ns, a, b);
        if ((rc = launch_check("k_nee_shade"))) return rc;
        if ((rc = transmittance_impl(m, ns, b.shadow_light, b.vis_light, b.go_light, st))) return rc;
        if ((rc = transmittance_impl(m, ns, b.shadow_phase, b.vis_phase, b.go_phase, st))) return rc;
        k_nee_gather<<<grid_of(ns, 256), 256, 0, st>>>(surf->cap_radiance, ns, a, b);
        if ((rc = launch_check("k_nee_gather"))) return rc;
        k_paths_accumulate<<<grid_of(np, 256), 256, 0, st>>>(sc, p0, np, a.emission, radiance_sum);
        if ((rc = launch_check("k_paths_accumulate"))) return rc;
    }
    return ws_release(m, 0, st);
}
