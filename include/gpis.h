/*
 * gpis.h — C ABI of the MI355X-native sparse-convolution GPIS hot path.
 *
 * This is the drop-in boundary: the entry points below are what a binding in the
 * reference (a `Medium` subclass registered in MediumFactory.cpp:13-22) would call in
 * place of the CPU code in
 *   src/core/media/SparseConvolutionNoiseMedium.cpp   (SCNM.cpp)
 *   src/core/media/GaussianProcessMedium.cpp          (GPM.cpp)
 *   src/core/math/SparseConvolutionNoise.cpp          (SCN.cpp)
 *   src/core/math/GPFunctions.cpp                     (GPF.cpp)
 * Plain pointers and sizes only; no C++ types, no torch types, no exceptions.
 *
 * Memory convention: every `*_batch` entry takes DEVICE pointers (hipMalloc'd, or a
 * torch tensor's data_ptr) and enqueues on `stream` (a hipStream_t passed as void*,
 * NULL = the null stream).  The resident-kernel form of the march (the default) never
 * synchronises; the wavefront form (GPIS_ORDER_SCATTERED with a guide field, or
 * GPIS_MARCH_FORM_WAVE) synchronises `stream` once per step/sort/evaluate iteration to read
 * how many rays are left.  The `*_host` variants take HOST pointers, stage through an
 * internal device workspace and synchronise before returning — they are what a
 * batch-of-one `Medium` adapter uses.
 *
 * Threading: a handle may be used from several host threads on distinct streams, as Tungsten's
 * render workers use one `const Medium`.  Entry points serialise on a per-handle
 * mutex while they ENQUEUE (kernels of different callers still overlap on the device); the
 * workspaces the handle owns (tile drivers, wavefront march) are handed from one caller's stream
 * to the next with an event, so a second caller never overwrites a workspace still in use.
 * The `*_host` entries are the exception: they hold the mutex END TO END (upload, kernel, event
 * wait, download) because the two staging slots are state of the handle — concurrent batch-of-one
 * callers of one handle (Tungsten's render workers through the Medium adapter) are served one
 * after the other, 150 us each.  That path is a compatibility path, not a throughput path:
 * a renderer that wants the GPU busy batches its rays.  The function-space
 * host entries serialise the same way on their own lock.
 * gpis_destroy / gpis_build_guide / gpis_drop_guide must not race with other calls on the handle.
 *
 * Environment: a few diagnostic overrides are read ONCE, in gpis_create, as the initial values
 * of the options below and of the cell table: GPIS_MARCH=resident|wave, GPIS_WAVE_TAIL=<rays>,
 * GPIS_PATHS_SORT=0, GPIS_PATHS_PRESORT=0, GPIS_CHUNK_LOG2=<16..28>, GPIS_DISABLE_FAST=1,
 * GPIS_DISABLE_TABLE=1, GPIS_TABLE_HALF_EXTENT=<cells>, GPIS_PERSIST=0, GPIS_SOLO_MAX=<lanes>, GPIS_RANGE_LEN=<rays>.  Nothing is read from
 * the environment after gpis_create.  Results never depend on any of them.
 *
 * All functions return GPIS_OK (0) or a negative gpis_status; gpis_last_error()
 * returns a thread-local message for the last failure.
 */
#ifndef GPIS_H_
#define GPIS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPIS_ABI_VERSION 3

typedef enum gpis_status {
    GPIS_OK = 0,
    GPIS_ERR_INVALID_ARG = -1,   /* bad pointer / size / enum */
    GPIS_ERR_UNSUPPORTED = -2,   /* a parameter combination outside the built scope */
    GPIS_ERR_DEVICE = -3,        /* HIP runtime error (message in gpis_last_error) */
    GPIS_ERR_NO_DEVICE = -4      /* no gfx950 device / extension cannot run */
} gpis_status;

/* GPCorrelationContext, src/core/math/GaussianProcess.hpp:26-31 (same numeric order). */
typedef enum gpis_corr_ctx {
    GPIS_CTX_GLOBAL = 0,
    GPIS_CTX_RENEWAL_PLUS = 1,
    GPIS_CTX_RENEWAL = 2,
    GPIS_CTX_NONE = 3
} gpis_corr_ctx;

/* SparseConv1DSamplingScheme, src/core/media/Medium.hpp:40-44. */
typedef enum gpis_scheme_1d { GPIS_UNI = 0, GPIS_NEE = 1, GPIS_MIS = 2 } gpis_scheme_1d;

/* Mean functions, src/core/math/GPFunctions.hpp:867-1005. */
typedef enum gpis_mean_type {
    GPIS_MEAN_HOMOGENEOUS = 0,   /* offset                       GPF.hpp:867-901  */
    GPIS_MEAN_SPHERICAL = 1,     /* |p-c| - r                    GPF.hpp:903-945  */
    GPIS_MEAN_LINEAR = 2         /* max((p-ref).dir*scale, min)  GPF.hpp:947-1005 */
} gpis_mean_type;

/* NoiseType of ProceduralNoise / ProceduralNoiseVec (GPF.hpp:613-650, 702-739): the four ramps (GPF.cpp:57-69, 91-103) and
 * the two fbm noises over 3D simplex noise (GPF.cpp:70-83, 104-117; math/SdfFunctions.cpp:199-296). */
typedef enum gpis_ramp_type {
    GPIS_RAMP_BOTTOM_TOP = 0,    /* along y */
    GPIS_RAMP_LEFT_RIGHT = 1,    /* along x */
    GPIS_RAMP_FRONT_BACK = 2,    /* along z */
    GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT = 3,  /* product of a y ramp and an x ramp ("min2".."end2"), GPF.cpp:96-103 */
    GPIS_NOISE_SANDSTONE = 4,    /* "sandstone": three nested fbm, scalar fields lerp(min, max, .), vector fields clamp(0.2 col) */
    GPIS_NOISE_RUST = 5          /* "rust": smoothStep(0.4, 0.6, fbm -/+ 0.1 fbm(25 p)), scalar lerp(min, max, .), vector lerp((0.278, 0.212, 0.141), 1, .) */
} gpis_ramp_type;

/* A procedural field of type "noise" (ProceduralNoise / ProceduralNoiseVec, GPF.hpp:596-776, GPF.cpp:43-138).  Used for the
 * scalar fields "var" (variance) and "aniso" (ProceduralNoise: 2 fbm octaves) and for the vector fields "ls" (the kernel scale
 * is the largest component, GPF.cpp:1729-1735) and the mean's "color" / "emission" (ProceduralNoiseVec: 10 octaves; the three
 * components are equal for the ramp noises and differ for sandstone / rust).  On the device `sin` / `sqrt` of the fbm hash are
 * ocml's, not glibc's: values agree with the CPU to the stated tolerance, not bit for bit. */
typedef struct gpis_ramp {
    int32_t enabled;
    int32_t type;                    /* gpis_ramp_type */
    double min, max, start, end;     /* "min", "max", "start", "end"      (defaults 1, 500, 0, 1) */
    double min2, max2, start2, end2; /* "min2", "max2", "start2", "end2"  (bottom_top_left_right) */
} gpis_ramp;

/* Stationary covariance kernels with a sparse-convolution splatting kernel (GPF.cpp).  Matérn with v = 1.5 needs Boost's
 * cyl_bessel_k (GPF.cpp:1055, 1073), which is neither vendored nor installed: outside the built scope.  Matérn / Gabor kernels
 * define no isotropic-space transform, no 1D kernel and no second derivative (GPF.hpp:2002-2110), so they are accepted for
 * world-space 3D sampling with correlation context none / global / renewal. */
typedef enum gpis_kernel_type {
    GPIS_KERNEL_SQUARED_EXPONENTIAL = 0,   /* GPF.cpp:654-865 */
    GPIS_KERNEL_MATERN = 1,                /* "matern": "sigma", "v" (0.5 or 2.5), "lengthScale", "aniso"     GPF.cpp:866-1082 */
    GPIS_KERNEL_GABOR_ANISO = 2,           /* "gabor_aniso": "sigma", "a_inv", "f_inv", "omega"               GPF.cpp:1086-1150 */
    GPIS_KERNEL_GABOR_ISO = 3              /* "gabor_iso": "sigma", "a_inv", "f_inv"                          GPF.cpp:1155-1214 */
} gpis_kernel_type;

typedef struct gpis_mean {
    int32_t type;          /* gpis_mean_type */
    float radius;          /* spherical */
    float offset;          /* homogeneous */
    float scale;           /* linear */
    float min;             /* linear (default -FLT_MAX) */
    int32_t _pad;
    double center[3];      /* spherical centre / linear reference_point */
    double dir[3];         /* linear direction (normalised by gpis_create as LinearMean::fromJson does) */
} gpis_mean;

/*
 * All JSON keys of the hot path (SURVEY.md §5 "Config / flags"):
 *   medium:      SCNM.cpp:57-73, GPM.cpp:97-126, Medium.cpp:37
 *   covariance:  GPF.cpp:654-679 (squared_exponential), GPF.hpp:1481-1484 (localScale),
 *                GPF.hpp:1134-1137 (lateralScale)
 *   wrapper:     GPF.hpp:2211-2217 + GPF.cpp:1590-1606 (proc_nonstationary: ls ramp, multiResolutionGrid)
 */
typedef struct gpis_params {
    uint32_t abi_version;            /* must be GPIS_ABI_VERSION */
    /* --- sparse_conv_noise medium --- */
    float step_size;                 /* "step_size"  default 0.01 */
    uint32_t min_step;               /* "min_step"   default 8 */
    uint32_t seed;                   /* "seed"       _globalSeed */
    float impulse_density;           /* "impulse_density" */
    int32_t single_realization;      /* "single_realization" */
    int32_t isotropic_3d_sampling;   /* "isotropic_3D_sampling" (iso-RAY space: SCN.cpp:17 hard-wires it) */
    int32_t sampling_1d;             /* "1D_sampling" */
    int32_t scheme_1d;               /* "1D_sampling_scheme" gpis_scheme_1d */
    int32_t correlation_xy;          /* "1D_gradient_correlationXY" */
    int32_t surf_vol_phase_separate; /* "surf_vol_phase_separate" */
    float surf_vol_phase_amp_thresh; /* "surf_vol_phase_amp_thresh" */
    /* --- gaussian_process medium base --- */
    int32_t correlation_context;     /* "correlation_context" gpis_corr_ctx (required) */
    int32_t max_bounces;             /* "max_bounces" default 1024 */
    float sigma_a[3];                /* "sigma_a" */
    float sigma_s[3];                /* "sigma_s" */
    float density;                   /* "density" */
    /* --- squared_exponential covariance --- */
    float sigma;                     /* "sigma" */
    float length_scale;              /* "lengthScale" */
    float aniso[3];                  /* "aniso" */
    int32_t use_aniso_mtx;           /* "useAnisoMtx" */
    float aniso_mtx[9];              /* "anisoMtx" row-major, used when use_aniso_mtx */
    float local_scale;               /* "localScale" (_kernelScale, default 3) */
    /* --- proc_nonstationary wrapper (0 = plain stationary kernel) --- */
    int32_t nonstationary;           /* 1: ProceduralNonstationaryCovariance with an "ls" ramp */
    int32_t multi_resolution_grid;   /* "multiResolutionGrid" */
    int32_t ls_ramp_type;            /* gpis_ramp_type */
    int32_t _pad0;
    double ls_min, ls_max;           /* ramp "min","max" */
    double ls_start, ls_end;         /* ramp "start","end" */
    /* --- mean(s) --- */
    gpis_mean mean;                  /* GaussianProcess::_mean,  _id = 0 */
    int32_t has_mean_additional;     /* GaussianProcess::_mean_additional (CSG min), _id_additional = 1 */
    int32_t _pad1;
    gpis_mean mean_additional;
    /* --- rest of the proc_nonstationary wrapper and of the mean (GPF.cpp:1590-1606, GPF.hpp:803-862) --- */
    double ls_min2, ls_max2;         /* second ramp of an "ls" field of type bottom_top_left_right */
    double ls_start2, ls_end2;
    gpis_ramp var;                   /* "var": getVariance(p) (GPF.cpp:1638-1641); the noise amplitude becomes var(p) * sigma (GPF.cpp:1235-1237) */
    gpis_ramp mean_color;            /* mean "color":    MediumSample.weight *= color(p) on a hit (GPM.cpp:316) */
    gpis_ramp mean_emission;         /* mean "emission": MediumSample.emission (GPM.cpp:317) — gpis_mean_color_emission_* */
    /* --- other stationary kernels (sigma / length_scale / aniso above are shared) --- */
    int32_t kernel_type;             /* gpis_kernel_type */
    float matern_v;                  /* "v" */
    float gabor_a_inv, gabor_f_inv;  /* "a_inv", "f_inv" (the reference stores a = 1/a_inv, f = 1/f_inv) */
    float gabor_omega[3];            /* "omega" (normalised at construction, GPF.cpp:1095) */
    int32_t _pad2;
    /* --- "aniso" field of proc_nonstationary (GPF.cpp:1600-1602): an angle in units of pi/2 that turns the in-plane anisotropy
     *     (axis ratio 1.5 : 1/1.5 : 1, hard-coded in the reference, GPF.hpp:2372) of the splatting kernel about z
     *     (getNonstationaryAniso3D, GPF.cpp:1678-1689).  Built for 3D sampling (world and isotropic-ray space); with 1D sampling
     *     the medium is refused (getNonstationaryAniso1D / ...CovSplatCov1D, GPF.cpp:1691-1727, are outside the built scope). --- */
    gpis_ramp aniso_field;
    /* --- FunctionSpaceGaussianProcessMedium (FunctionSpaceGaussianProcessMedium.cpp:34-43): read by the
     *     gpis_fs_* entry points only.  "skip_space" = 0 and "step_size_cov" (dead code in the reference, :98-106) are not carried. --- */
    int32_t fs_sample_points;        /* "sample_points" (2 .. GPIS_FS_MAX_POINTS; the reference's default is 32) */
    int32_t _pad3;
    double fs_step_size;             /* "step_size" of the function-space medium (0 = the whole segment in one batch of points) */
    /* --- GridNonstationaryCovariance (GPF.cpp:1326-1427; needs nonstationary = 1): the variance comes from a voxel grid
     *     (gpis_set_variance_grid; 1 until one is set, GPF.cpp:1386-1391) and the kernel scale from a threshold on it.  The "ls",
     *     "var" and "aniso" fields of the procedural wrapper are not read in this flavour. --- */
    int32_t grid_nonstationary;      /* 1: this flavour of the non-stationary wrapper */
    int32_t grid_surf_vol_amp_separate; /* "surf_vol_amp_separate" */
    float grid_offset, grid_scale;   /* "offset", "scale": (density + offset) * scale */
    float grid_surf_vol_amp_thresh;  /* "surf_vol_amp_thresh" */
    float grid_surf_amp_scale, grid_vol_amp_scale;   /* "surf_amp_scale", "vol_amp_scale" */
    float grid_surf_ls_scale, grid_vol_ls_scale;     /* "surf_ls_scale", "vol_ls_scale" */
    int32_t _pad4;
} gpis_params;

/* The voxel grid of a GridNonstationaryCovariance, as a dense array: what VdbGrid::density(p) (VdbGrid.cpp:405-431) returns for
 * an index-space point — p clamped to [bounds_min + 2, bounds_max - 3], then OpenVDB's PointSampler (interpolate 0) or BoxSampler
 * (1: trilinear, weights in double, each stage narrowed to float).  OpenVDB is not vendored by the reference and absent here: the
 * two samplers are restated from its published source (tools/Interpolation.h); PARITY UNPINNED for this lookup.  "quadratic" is
 * refused.  voxel (i, j, k) of the index-space box origin + [0, dims) is voxels[i + dims[0] * (j + dims[1] * k)]; anything
 * outside it is 0 (the tree's background). */
typedef struct gpis_variance_grid {
    int32_t dims[3];
    int32_t interpolate;             /* 0 "point", 1 "linear" */
    int32_t origin[3];               /* index-space coordinate of voxel (0, 0, 0) */
    int32_t _pad;
    float bounds_min[3], bounds_max[3];   /* VdbGrid::bounds() (index space) */
    float inv_natural_transform[16]; /* row-major Mat4f, world -> index space (VdbGrid::invNaturalTransform) */
} gpis_variance_grid;
/* (set with gpis_set_variance_grid, declared with the other entry points below) */

/* The function-space comparison path (FunctionSpaceGaussianProcessMedium.cpp:58-282): the field is sampled at `sample_points`
 * positions of the segment from the multivariate normal the GP prior (or its conditional, given the previous segment's
 * values) defines there — covariance build, pseudo-inverse through a symmetric eigen-decomposition, Cholesky (or the
 * eigen square root when it fails), GaussianProcess.cpp:590-753, Gaussian.cpp:121-232.
 * Built for: squared-exponential covariance in its GP form (sigma^2 exp(-d^T diag(aniso) d / (2 l^2)), GPF.hpp:1602-1605,
 * GPF.cpp:770-772), the analytic means, all four correlation contexts, normal sampling "ConditionedGaussian".
 *
 * gpis_fs_state = the caller-owned MediumState part this medium adds: state.gpContext (GPContextFunctionSpace,
 * GaussianProcessMedium.hpp:23-33: points, derivative kinds and the GPRealNodeValues) and the path's PathSampleGenerator
 * (a PCG32 stream, UniformPathSampler), from which the medium draws an unbounded number of variates.  It is a VALUE: a shadow
 * segment works on a copy and never alters the path's context (in the reference the copy of MediumState shares the
 * GPRealNode, which applyMemory mutates in place, GaussianProcess.cpp:134-168). */
#define GPIS_FS_MAX_POINTS 64
#define GPIS_FS_MAX_CTX 66
typedef struct gpis_fs_state {
    uint64_t sampler_state;          /* PCG32 state (UniformSampler.hpp:41-75) */
    int32_t has_context;             /* state.gpContext != nullptr */
    int32_t is_intersect;            /* GPRealNodeValues::_isIntersect */
    int32_t n_points;                /* ctxt->points.size() */
    int32_t n_values;                /* ctxt->values rows (differs from n_points only transiently) */
    double sampled_grad[3];          /* GPRealNodeValues::_sampledGrad */
    double points[GPIS_FS_MAX_CTX][3];
    double values[GPIS_FS_MAX_CTX];
    int32_t derivs[GPIS_FS_MAX_CTX]; /* 0 = Derivative::None, 1 = Derivative::First */
} gpis_fs_state;

/*
 * One ray segment handed to Medium::sampleDistance / Medium::transmittance
 * (Medium.hpp:104-108): the Ray (math/Ray.hpp), the caller-owned Medium::MediumState
 * (Medium.hpp:59-88) and RayInfo (MediumSample.hpp:14-18), plus the single
 * PathSampleGenerator::next1D() the path consumes (SCNM.cpp:129).  128 bytes.
 */
typedef struct gpis_ray_in {
    float pos[3];
    float dir[3];
    float near_t;
    float far_t;              /* may be +inf: clamped to near_t+2000 (GPM.cpp:229-231) */
    uint32_t pixel[2];        /* info.pixelSampleSegment.xy */
    uint32_t spp;             /* info.pixelSampleSegment.z  */
    uint32_t segment;         /* info.pixelSampleSegment.w  (bounce index, PathTracer.cpp:64) */
    uint32_t scene_seed;      /* info.sceneSeed */
    float info_t;             /* info.t */
    float u_jitter;           /* sampler.next1D() */
    uint32_t first_scatter;   /* state.firstScatter */
    int32_t bounce;           /* state.bounce */
    float last_val;           /* state.lastVal */
    int32_t last_gp_id;       /* state.lastGPId */
    int32_t _pad;
    double last_aniso[3];     /* state.lastAniso */
    double _reserved[3];
} gpis_ray_in;

/*
 * Result of one sampleDistance call: the MediumSample fields the path writes
 * (GPM.cpp:224-340) and the MediumState updates the caller must apply:
 *   state.lastAniso = aniso; state.lastVal = last_val; state.lastGPId = gp_id;
 *   state.firstScatter = false (when gradient_sampled); state.bounce++; state.info.t += sample_t.
 * 96 bytes.
 */
typedef struct gpis_seg_out {
    double t;                 /* intersect t (double, SCNM.cpp:102) */
    double aniso[3];          /* sample.aniso = sampled gradient */
    float sample_t;           /* sample.t = min(float(t), maxT) */
    float continued_t;        /* sample.continuedT */
    float weight[3];          /* sample.weight */
    float continued_weight[3];
    float p[3];               /* sample.p */
    float last_val;           /* state.lastVal after the call */
    int32_t exited;           /* sample.exited */
    int32_t ok;               /* return value of sampleDistance (0 → caller terminates the path) */
    int32_t gp_id;            /* sample.gpId */
    int32_t scheme;           /* sample.sparseConv1DSamplingScheme */
} gpis_seg_out;

/*
 * The realization handle MediumSample.ctxt points to (GPContextSparseConvNoise,
 * SparseConvolutionNoiseMedium.hpp:11-16) reduced to its mutable part: the
 * pathwise-update conditioning coefficients (SCN.hpp:7-21).  32 bytes.
 */
typedef struct gpis_cond_coeff {
    float value_scale;
    float gradient_scale[3];
    float ray_origin[3];
    uint32_t n_evals;         /* noise evaluations spent on this segment (diagnostic) */
} gpis_cond_coeff;

/*
 * A single noise query: SparseConvolutionNoiseRealization::evaluateValue /
 * evaluateGradient (SCN.cpp:73-99) — unit-test surface.  `t_segment` is the
 * `t` argument (distance along the current segment); `info_t` is RayInfo::t.  96 bytes.
 */
typedef struct gpis_query {
    float p[3];
    float dir[3];
    float t_segment;
    float info_t;
    uint32_t pixel[2];
    uint32_t spp;
    uint32_t segment;
    uint32_t scene_seed;
    uint32_t _pad[3];
    gpis_cond_coeff coeff;    /* conditioning state to evaluate under (all-zero = unconditioned) */
} gpis_query;

/* Input of neePDF / neeGrad (SCN.cpp:652-743). 96 bytes. */
typedef struct gpis_nee_query {
    float ray_dir[3];
    float normal[3];
    float p[3];
    float t_segment;          /* neePDF's `tSegment` */
    float info_t;
    uint32_t pixel[2];
    uint32_t spp;
    uint32_t segment;
    uint32_t scene_seed;
    gpis_cond_coeff coeff;
} gpis_nee_query;

typedef struct gpis_medium gpis_medium;   /* opaque handle */

/* Derived constants the path precomputes once (GPF.cpp:654-679, 696-709, 741-760);
 * exposed so tests can pin them against the oracle. */
typedef struct gpis_derived {
    float world_to_local[9];   /* row-major */
    float local_to_world[9];
    float kernel_radius_world; /* splattingKernelRadius(false, 1) */
    float kernel_radius_iso;   /* splattingKernelRadius(true, 1)  */
    float norm3d_world;        /* sqrt(sparseConvNoiseVariance3D(world)) at scale 1; 0 for non-stationary media (position dependent) */
    float norm3d_iso;          /* sqrt(sparseConvNoiseVariance3D(iso)); 0 for non-stationary media */
    float norm1d;              /* sqrt(sparseConvNoiseVariance1D); 0 for non-stationary media */
    uint32_t impulses_per_cell;
    int32_t activate_conditioning; /* SCN.cpp:21 */
    int32_t effective_scheme_1d;   /* SCN.cpp:23-26 */
    int32_t multi_resolution;      /* SCN.cpp:30 */
    int32_t fast_path;             /* 1 when the single-realization cell-table kernels are in use */
} gpis_derived;

/* ---- lifetime ------------------------------------------------------------------- */

/* Replaces SparseConvolutionNoiseMedium::fromJson + prepareForRender
 * (SCNM.cpp:57-73, GPM.cpp:97-126,152-158).  `device` = HIP device ordinal. */
int gpis_create(const gpis_params *params, int device, gpis_medium **out);
int gpis_destroy(gpis_medium *m);
int gpis_get_derived(const gpis_medium *m, gpis_derived *out);
const char *gpis_last_error(void);
/* Fills `p` with the reference's defaults (SCNM.cpp:17-34, GPM.cpp:86-95, GPF.hpp:1729,1784). */
void gpis_default_params(gpis_params *p);

/* ---- the hot path (device pointers) --------------------------------------------- */

/* Medium::sampleDistance for n independent segments (GPM.cpp:221-341 → SCNM.cpp:102-183,
 * 93-100).  `coeff` may be NULL. */
int gpis_sample_distance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays,
                               gpis_seg_out *out, gpis_cond_coeff *coeff, void *stream);

/* Medium::transmittance (GPM.cpp:343-393): visible[i] = 1 if the segment exits, else 0
 * (the reference returns Vec3f(1) / Vec3f(0)). */
int gpis_transmittance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays,
                             uint8_t *visible, void *stream);

/* SparseConvolutionNoiseRealization::evaluateValue / evaluateGradient (SCN.cpp:73-99). */
int gpis_eval_value_batch(gpis_medium *m, size_t n, const gpis_query *q,
                          float *value, int32_t *gp_id, void *stream);
int gpis_eval_gradient_batch(gpis_medium *m, size_t n, const gpis_query *q,
                             float *grad3, void *stream);

/* SparseConvolutionNoiseRealization::conditioning (SCN.cpp:431-595): q[i].p/dir/... is the
 * segment start; target_val / target_grad3 are state.lastVal / state.lastAniso. */
int gpis_conditioning_batch(gpis_medium *m, size_t n, const gpis_query *q,
                            const float *target_val, const float *target_grad3,
                            gpis_cond_coeff *coeff_out, void *stream);

/* SparseConvolutionNoiseRealization::neePDF / neeGrad (SCN.cpp:652-743). */
int gpis_nee_pdf_batch(gpis_medium *m, size_t n, const gpis_nee_query *q, float *pdf, void *stream);
int gpis_nee_grad_batch(gpis_medium *m, size_t n, const gpis_nee_query *q, float *grad3, void *stream);

/* MeanFunction::color / emission (GPF.hpp:849-857) at n points (xyz triples in DOUBLE, as the reference evaluates them at
 * ro + rd * t, GPM.cpp:316-317): the factor sampleDistance has already applied to `weight` on a hit, and the value a binding
 * stores in MediumSample.emission.  color3 / emission3: n xyz triples of float; either may be NULL. */
int gpis_mean_color_emission_batch(gpis_medium *m, size_t n, const double *p3, float *color3, float *emission3, void *stream);
int gpis_mean_color_emission_host(gpis_medium *m, size_t n, const double *p3, float *color3, float *emission3);

/* Medium::sampleDistance / transmittance of the function-space medium (GaussianProcessMedium.cpp:221-393 over
 * FunctionSpaceGaussianProcessMedium::intersectGP / sampleGradient).  rays[i].u_jitter is not used (every variate comes from
 * states[i].sampler_state); states are read and written in place (device pointers).  The two entries share one device
 * workspace per handle: launches of the SAME handle must be ordered (one stream, or an event between them). */
int gpis_fs_sample_distance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out, void *stream);
int gpis_fs_transmittance_batch(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, uint8_t *visible, void *stream);
/* The same two entries for host pointers (synchronous; what the Medium binding of the function-space medium calls with a batch
 * of one, integration/HipFunctionSpaceMedium.cpp): rays, states and results are staged through the handle's device buffers. */
int gpis_fs_sample_distance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out);
int gpis_fs_transmittance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, uint8_t *visible);

/* Test surface of the function-space path's dense linear algebra: `count` n x n column-major matrices of doubles (device
 * pointers, 1 <= n <= GPIS_FS_MAX_CTX), one wave each.  op GPIS_FS_OP_EIGH: Eigen::SelfAdjointEigenSolver<MatrixXd> (lower
 * triangle read; eigenvectors -> out, eigenvalues ascending -> evals, which may be NULL); GPIS_FS_OP_NORM_TRANSFORM:
 * MultivariateNormalDistribution's normTransform (Eigen::LLT, else eigenvectors x sqrt(max(eigenvalues, 0)), Gaussian.cpp:
 * 121-167); GPIS_FS_OP_PINV: pseudo_inverse (GaussianProcess.cpp:645-662).  Results are bit-identical to the reference's
 * vendored Eigen under the reference's build flags (tests/golden/ref_fs_primitives.npz). */
typedef enum gpis_fs_linalg_op { GPIS_FS_OP_EIGH = 0, GPIS_FS_OP_NORM_TRANSFORM = 1, GPIS_FS_OP_PINV = 2 } gpis_fs_linalg_op;
int gpis_fs_linalg_batch(gpis_medium *m, int op, int n, size_t count, const double *in, double *out, double *evals, void *stream);

/* Test surface of csrc/gpis_libm.hpp, the device's bit-for-bit restatement of the host libm (glibc 2.35, x86-64 with FMA) for the
 * double-precision functions this path calls: out[i] = fn(x[i]) (fn(x[i], y[i]) for GPIS_LIBM_POW; sin -> out, cos -> out2 for
 * GPIS_LIBM_SINCOS and GPIS_LIBM_SINCOSF; GPIS_LIBM_LOGF and GPIS_LIBM_SINCOSF round x to float first and widen the results).  Device pointers; y and out2 may be NULL where
 * unused.  Needs no medium handle; runs on the current device. */
typedef enum gpis_libm_fn { GPIS_LIBM_EXP = 0, GPIS_LIBM_LOG = 1, GPIS_LIBM_LOGF = 2, GPIS_LIBM_SIN = 3, GPIS_LIBM_COS = 4, GPIS_LIBM_SINCOS = 5,
                            GPIS_LIBM_POW = 6, GPIS_LIBM_SINCOSF = 7 } gpis_libm_fn;
int gpis_libm_batch(int fn, size_t n, const double *x, const double *y, double *out, double *out2, void *stream);

/* Test surface of the library's own radix sort (csrc/gpis_sort.hip: stable LSD sort of (uint32 key, uint32 value) pairs, used by
 * the wavefront march and the multi-bounce driver to regroup rays by lattice cell).  Device pointers; scratch is allocated and
 * freed inside the call. */
int gpis_sort_pairs_u32(size_t n, const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, void *stream);

/* Bit-exact primitives (MathUtil.hpp:179-224, UniformSampler.hpp:41-75, BitManip.hpp:47-50):
 * out[i] = xxhash32 of `arity` (1..4) words at words[i*arity..]; and the PCG32 stream
 * after set_state(state[i]) — `count` raw nextI() draws each. */
int gpis_xxhash32_batch(gpis_medium *m, size_t n, int arity, const uint32_t *words,
                        uint32_t *out, void *stream);
int gpis_pcg32_stream_batch(gpis_medium *m, size_t n, const uint64_t *state, uint32_t count,
                            uint32_t *out, void *stream);

/* ---- host-pointer entries (synchronous) ------------------------------------------
 * What a `Medium` adapter inside the reference calls (a batch of one per sampleDistance / transmittance), and what a
 * host-side wavefront integrator calls with whole batches.  The two march entries move their records in chunks of
 * 262 144 through two streams (H2D, kernel and D2H of consecutive chunks overlap).  Buffers obtained from
 * gpis_alloc_host (pinned memory) are transferred by DMA directly; any other host memory is staged through pinned
 * buffers of the handle with one extra memcpy each way. */
void *gpis_alloc_host(size_t bytes);      /* pinned host memory; NULL on failure */
void gpis_free_host(void *p);
int gpis_sample_distance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays,
                              gpis_seg_out *out, gpis_cond_coeff *coeff);
int gpis_transmittance_host(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint8_t *visible);
int gpis_eval_value_host(gpis_medium *m, size_t n, const gpis_query *q, float *value, int32_t *gp_id);
int gpis_eval_gradient_host(gpis_medium *m, size_t n, const gpis_query *q, float *grad3);
int gpis_conditioning_host(gpis_medium *m, size_t n, const gpis_query *q, const float *target_val,
                           const float *target_grad3, gpis_cond_coeff *coeff_out);
int gpis_nee_pdf_host(gpis_medium *m, size_t n, const gpis_nee_query *q, float *pdf);
int gpis_nee_grad_host(gpis_medium *m, size_t n, const gpis_nee_query *q, float *grad3);

/* ---- measurement ---------------------------------------------------------------- */

/* Device counters: noise evaluations (evaluateValue/evaluateGradient-equivalents,
 * conditioning evaluations included) and segments, accumulated since the last reset.
 * Synchronises the device. */
int gpis_get_counters(gpis_medium *m, uint64_t *n_eval, uint64_t *n_seg);
int gpis_reset_counters(gpis_medium *m);

/* Per-kernel timing with HIP events recorded on the launch stream around every march-kernel
 * launch (off by default; events cost a few microseconds per launch).  `which`: 0 = the
 * sampleDistance kernel, 1 = the transmittance kernel, 2 = the neePDF / neeGrad launches of
 * gpis_render_scene_s_nee (time and launches only).  Returns the summed kernel time, the
 * number of launches, and that kernel's share of the evaluation / segment counters since the
 * last gpis_reset_counters.  Synchronises the device. */
int gpis_set_profiling(gpis_medium *m, int enable);

/* How the rays of the *_batch / *_host march entries are ordered.  Results never depend on it; with a
 * guide field built it selects the form of the march: GPIS_ORDER_COHERENT (default) — the 64 consecutive
 * records of a wave are neighbours in lattice space (camera rays in pixel order): one resident kernel per
 * batch; GPIS_ORDER_SCATTERED — arbitrary order (bounced rays of a wavefront integrator): the march
 * runs as step / sort / evaluate iterations that regroup the exact evaluations by lattice cell. */
enum { GPIS_ORDER_COHERENT = 0, GPIS_ORDER_SCATTERED = 1 };
int gpis_set_batch_order(gpis_medium *m, int order);

/* Tuning options of a handle (speed only — results never depend on them). */
typedef enum gpis_option {
    GPIS_OPT_MARCH_FORM = 0,     /* form of the guided march: gpis_march_form; AUTO follows gpis_set_batch_order */
    GPIS_OPT_WAVE_TAIL = 1,      /* wavefront march: below this many active rays one wave finishes one ray (default 262144; 0 = never) */
    GPIS_OPT_PATHS_SORT = 2,     /* gpis_render_scene_s_paths: regroup secondary segments by lattice cell (default 1) */
    GPIS_OPT_PATHS_PRESORT = 3,  /* ... also when the wavefront march (which regroups the exact work itself) runs (default 1) */
    GPIS_OPT_CHUNK_LOG2 = 4,     /* log2 of the tile drivers' samples per chunk (16..28; 0 = as large as the device holds) */
    GPIS_OPT_PERSISTENT = 5,     /* per-path media: 1 (default) = persistent refilling march kernels, 0 = one ray per lane per launch */
    GPIS_OPT_SOLO_MAX = 6,       // persistent march: evaluate sideways (lane = impulse) while at most this many lanes of a wave have a
                                 //   pending evaluation; -1 (default) = derived from impulse_density
    GPIS_OPT_RANGE_LEN = 7,      // guided march (resident form): 0 (default) = one ray per lane per launch; > 0 = rays per wave with
                                 //   in-wave refill as segments finish (multiple of 64; measured slower on camera rays)
    GPIS_OPT_DEFER_GRAD = 8,     // guided march (resident form): 0 (default) = the segment's gradient is evaluated at the end of the march kernel;
                                 //   1 = by a second kernel over the pending records (measured: DESIGN.md 8, the spill experiment)
    GPIS_OPT_COUNT_
} gpis_option;
typedef enum gpis_march_form { GPIS_MARCH_FORM_AUTO = 0, GPIS_MARCH_FORM_RESIDENT = 1, GPIS_MARCH_FORM_WAVE = 2 } gpis_march_form;
int gpis_set_option(gpis_medium *m, int option, long long value);
int gpis_get_option(gpis_medium *m, int option, long long *value);
int gpis_get_kernel_profile(gpis_medium *m, int which, double *total_ms, uint64_t *launches,
                            uint64_t *n_eval, uint64_t *n_seg);

/* ---- certified guide field (single-realization media) ------------------------ */

/* Tabulates the medium's lattice noise on a grid of `points_per_cell` (8/16/32/64) samples per cell
 * over |grid coordinate| < half_extent_cells, with a rigorous per-block error bound, and switches
 * sampleDistance / transmittance to the guided march: steps whose sign the table certifies cost one
 * lookup, all others run the exact evaluation, so results are unchanged.  Memory:
 * (2*half*ppc)^3 * 4 bytes (4.3 GB for half=16, ppc=32).  Returns GPIS_ERR_UNSUPPORTED for media
 * the wave-cooperative path does not cover.  gpis_drop_guide frees it. */
int gpis_build_guide(gpis_medium *m, int half_extent_cells, int points_per_cell);
int gpis_drop_guide(gpis_medium *m);
/* Size of the guide field: the samples are stored in bricks of 16^3 grid points and only the bricks in which the certificate can be
 * asked for more than the mean (|mean| within sigma * bound(|N|) / norm somewhere in the brick) are tabulated; everywhere else
 * every step is decided by the mean alone.  bytes_dense = what the full grid would take.  All zero without a guide field. */
typedef struct gpis_guide_info {
    int32_t half_extent_cells, points_per_cell;
    uint64_t bricks_total, bricks_allocated, bricks_usable;
    uint64_t bytes_samples, bytes_bounds, bytes_dense;
    uint64_t selfcheck_points_tabulated;      /* of the last gpis_guide_selfcheck: points that fell into tabulated bricks */
} gpis_guide_info;
int gpis_get_guide_info(gpis_medium *m, gpis_guide_info *out);
/* March steps certified by the guide since the last gpis_reset_counters: each stands for one
 * evaluateValue call of the reference (exact evaluations stay in n_eval). */
int gpis_get_guide_steps(gpis_medium *m, uint64_t *n_guide);
/* Test surface: evaluates the exact lattice sum and the guide at n grid-space points (device
 * pointer, xyz triples; consecutive groups of 64 must lie within one cell of each other) and reports
 * how many violate |exact - guide| <= bound (must be 0), the largest ratio and the mean bound. */
int gpis_guide_selfcheck(gpis_medium *m, size_t n, const float *points3, uint64_t *checked,
                         uint64_t *violations, float *max_ratio, float *mean_bound, void *stream);

/* Test surface for the march's certificate: for each ray (device pointer) the first `steps` march
 * positions of SCNM.cpp:129-132 are classified by the guide, and every certified sign is compared with
 * the sign of the exact evaluateValue at the same position; `violations` must be 0. */
int gpis_guide_raycheck(gpis_medium *m, size_t n, const gpis_ray_in *rays, uint32_t steps,
                        uint64_t *certified, uint64_t *violations, void *stream);

/* ---- tile → ray-batch driver (SURVEY.md §8d "Scene S", §8f-1) -------------------- */

typedef struct gpis_scene_s {
    uint32_t width, height;
    uint32_t spp_begin, spp_count;    /* sample indices [spp_begin, spp_begin+spp_count) */
    uint32_t scene_seed;              /* 0xBA5EBA11 */
    uint32_t tile_size;               /* 16 (PathTraceIntegrator.hpp:27) */
    float cam_pos[3];                 /* (0,0,4) */
    float cam_fov_deg;                /* 35 */
    float bound_radius;               /* 1.5 */
    float light_dir[3];               /* (0.5,0.7,0.5)/|.| (normalised by the driver) */
    float light_radiance;             /* 1 */
    uint32_t y_begin, y_count;        /* image rows this call covers */
    uint32_t shard_index, shard_count; /* shard_count > 1: of those rows, only the tile rows t (tile_size pixels high, counted
                                          from y_begin) with t % shard_count == shard_index — the interleaved tile-row split of
                                          the multi-GPU driver; 0 or 1 = all rows */
} gpis_scene_s;

void gpis_default_scene_s(gpis_scene_s *s, uint32_t width, uint32_t height, uint32_t spp);

/* Copies `voxels` (host pointer, dims[0] * dims[1] * dims[2] floats) to the device and switches the lookup on.  The medium must
 * have been created with grid_nonstationary = 1.  Not to be called while work of this handle is in flight. */
int gpis_set_variance_grid(gpis_medium *m, const gpis_variance_grid *g, const float *voxels);

/* Allocates the workspace gpis_render_scene_s needs for `s` (236 B per sample of the largest chunk the device holds: 31.3 GB for
 * a whole 1920x1080x64 frame) ahead of the first frame.  Optional: the render entry allocates on demand, each of its three arrays
 * right before the first kernel that needs it.  This entry does not take the handle's lock: call it from a second host thread
 * while gpis_build_guide runs (allocation time is per byte and overlaps the guide build's kernels) — not concurrently with a
 * render call that uses a DIFFERENT scene size on the same handle. */
int gpis_reserve_scene_workspace(gpis_medium *m, const gpis_scene_s *s);

/* Renders rows [y_begin, y_begin+y_count) × all columns × spp_count samples of scene S and
 * ACCUMULATES sum-of-radiance into radiance_sum[height*width] (float, device pointer,
 * indexed y*width+x; caller divides by total spp).  Runs primary sampleDistance, shading,
 * one shadow transmittance per hit.  hit_count (device, may be NULL) accumulates per-pixel hits. */
int gpis_render_scene_s(gpis_medium *m, const gpis_scene_s *s, float *radiance_sum,
                        uint32_t *hit_count, void *stream);

/* Specular micro-surface + light of finite solid angle for gpis_render_scene_s_nee. 32 bytes. */
typedef struct gpis_surface_s {
    float eta, k;             /* ConductorBsdf "eta", "k" (one channel; Fresnel.hpp:102-123) */
    float albedo;
    float cap_cos;            /* the light: an infinite spherical cap about scene.light_dir, cos of its half-angle
                                 (InfiniteSphereCap "cap_angle") */
    float cap_radiance;
    float _pad[3];
} gpis_surface_s;

/* Scene S with the reference's specular NEE coupling: per sample the primary
 * sampleDistance, then TraceBase::volumeEstimateDirect = volumeLightSample + volumePhaseSample
 * (TraceBase.cpp:346-420) for a BRDFPhaseFunction over a ConductorBsdf (BRDFPhaseFunction.cpp:27-96,
 * ConductorBsdf.cpp:59-139) and one InfiniteSphereCap light:
 *   light sample (schemes NEE, MIS): f = albedo * F(wi.z) * neePDF(half-vector normal); the shadow segment
 *     runs on a state copy whose lastAniso = neeGrad(half-vector normal); weight power-heuristic unless NEE;
 *   phase sample (schemes UNI, MIS): mirror direction about the sampled normal, pdf = neePDF(normal), counts
 *     when it falls inside the cap.
 * The scheme is the one sampleDistance reports per sample (gpis_seg_out.scheme).  The cap direction is drawn
 * with a rejection-sampled azimuth (sqrt only) instead of SampleWarp::uniformSphericalCap's sin/cos. */
int gpis_render_scene_s_nee(gpis_medium *m, const gpis_scene_s *s, const gpis_surface_s *surf,
                            float *radiance_sum, void *stream);

/* Multi-bounce wavefront driver on scene S: per sample, up to `max_path_bounces`
 * medium interactions following PathTracer::traceSample / TraceBase::handleVolume
 * (PathTracer.cpp:62-75, TraceBase.cpp:539-563):
 *   sampleDistance (segment word = bounce, PathTracer.cpp:64) → on a hit: next-event estimation of
 *   the directional light through a state copy with segment+1 (TraceBase.cpp:546-549, 346-386:
 *   BRDFPhaseFunction/Lambert eval = albedo/pi * cos, Dirac light → no MIS) with one shadow
 *   transmittance, then a cosine-weighted bounce about the sampled normal (throughput *= albedo).
 * The bounce direction is drawn by rejection from the unit disk (sqrt only) instead of
 * SampleWarp::cosineHemisphere's sin/cos so that CPU and GPU agree bit for bit.
 * Accumulates sum-of-radiance into radiance_sum[height*width] (device pointer). */
int gpis_render_scene_s_paths(gpis_medium *m, const gpis_scene_s *s, int max_path_bounces, float albedo,
                              float *radiance_sum, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GPIS_H_ */
