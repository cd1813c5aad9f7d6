// HipFunctionSpaceMedium.cpp — see the header.  Compiles against the reference's real headers with the same command as
// HipSparseConvNoiseMedium.cpp (tests/test_integration_compile.py).  Links against libgpis_hip.so.
#include "HipFunctionSpaceMedium.hpp"
#include "TungstenJsonAccess.hpp"

#include "io/JsonObject.hpp"
#include "sampling/UniformSampler.hpp"

#include <cmath>
#include <cstring>

namespace Tungsten {

GPContextHipFs::GPContextHipFs() { std::memset(&st, 0, sizeof st); }
void GPContextHipFs::reset() { std::memset(&st, 0, sizeof st); }

HipFunctionSpaceMedium::HipFunctionSpaceMedium()
: _handle(nullptr),
  _device(0),
  _sigmaA(0.0f),
  _sigmaS(0.0f),
  _sigmaT(0.0f),
  _absorptionOnly(true)
{
    gpis_default_params(&_params);
    _params.correlation_context = GPIS_CTX_RENEWAL_PLUS;   // FunctionSpaceGaussianProcessMedium.cpp:20-24
    _params.fs_sample_points = 32;                          // :25
    _params.fs_step_size = 0.0;                             // :27
}

HipFunctionSpaceMedium::~HipFunctionSpaceMedium()
{
    teardownAfterRender();
}

void HipFunctionSpaceMedium::fromJson(JsonPtr value, const Scene &scene)
{
    Medium::fromJson(value, scene);      // phase_function, transmittance, max_bounces (Medium.cpp:29-38)
    _params.max_bounces = _maxBounce;

    // GaussianProcessMedium::fromJson (GaussianProcessMedium.cpp:97-126) and GaussianProcess::fromJson for an inline object
    // (GaussianProcess.cpp:172-190) through the shared key table (include/gpis_json.hpp); the device path of this medium is built
    // for an analytic mean and a stationary squared-exponential covariance (include/gpis.h "function-space comparison path")
    TungstenJson::vec3f(value, "sigma_a", _params.sigma_a);
    TungstenJson::vec3f(value, "sigma_s", _params.sigma_s);
    TungstenJson::num(value, "density", _params.density);
    std::string ctxtString = "goldfish";
    value.getField("correlation_context", ctxtString);
    _params.correlation_context = gpis_json::correlationContext<TungstenJson>(ctxtString);
    if (auto gp = value["gaussian_process"]) {
        if (!gp.isObject())
            FAIL("hip_function_space_gaussian_process: \"gaussian_process\" must be an inline object");
        gpis_json::readGaussianProcess<TungstenJson>(gp, _params);
        if (_params.nonstationary || _params.kernel_type != GPIS_KERNEL_SQUARED_EXPONENTIAL || _params.has_mean_additional)
            FAIL("hip_function_space_gaussian_process: built for one analytic mean and a stationary squared-exponential covariance");
    }
    _phaseFunctions.clear();
    _phaseFunctions.push_back(_phaseFunction);

    // FunctionSpaceGaussianProcessMedium::fromJson, FunctionSpaceGaussianProcessMedium.cpp:34-43
    value.getField("sample_points", _params.fs_sample_points);
    value.getField("step_size", _params.fs_step_size);
    double stepSizeCov = 0.0, skipSpace = 0.0;
    value.getField("step_size_cov", stepSizeCov);
    value.getField("skip_space", skipSpace);
    if (stepSizeCov != 0.0 || skipSpace != 0.0)
        FAIL("hip_function_space_gaussian_process: \"step_size_cov\" and \"skip_space\" are outside the built scope (both 0 in the device path)");
    int device = _device;
    value.getField("hip_device", device);
    _device = device;
}

rapidjson::Value HipFunctionSpaceMedium::toJson(Allocator &allocator) const
{
    static const char *ctxNames[] = {"global", "renewal+", "renewal", "none"};
    return JsonObject{Medium::toJson(allocator), allocator,
        "type", "hip_function_space_gaussian_process",
        "sigma_a", Vec3f(_params.sigma_a[0], _params.sigma_a[1], _params.sigma_a[2]),
        "sigma_s", Vec3f(_params.sigma_s[0], _params.sigma_s[1], _params.sigma_s[2]),
        "density", _params.density,
        "correlation_context", ctxNames[_params.correlation_context],
        "sample_points", _params.fs_sample_points,
        "step_size_cov", 0.0,
        "step_size", _params.fs_step_size,
        "skip_space", 0.0
    };
}

void HipFunctionSpaceMedium::prepareForRender()
{
    teardownAfterRender();
    _sigmaA = Vec3f(_params.sigma_a[0], _params.sigma_a[1], _params.sigma_a[2])*_params.density;
    _sigmaS = Vec3f(_params.sigma_s[0], _params.sigma_s[1], _params.sigma_s[2])*_params.density;
    _sigmaT = _sigmaA + _sigmaS;
    _absorptionOnly = _sigmaS == 0.0f;
    _params.single_realization = 0;        // every path samples its own realisation of the process
    if (gpis_create(&_params, _device, &_handle) != GPIS_OK) {
        _handle = nullptr;
        FAIL("hip_function_space_gaussian_process: gpis_create failed: %s", gpis_last_error());
    }
}

void HipFunctionSpaceMedium::teardownAfterRender()
{
    if (_handle)
        gpis_destroy(_handle);
    _handle = nullptr;
}

void HipFunctionSpaceMedium::fillRay(const Ray &ray, const MediumState &state, gpis_ray_in &r) const
{
    std::memset(&r, 0, sizeof r);
    for (int i = 0; i < 3; ++i) {
        r.pos[i] = ray.pos()[i];
        r.dir[i] = ray.dir()[i];
        r.last_aniso[i] = state.lastAniso[i];
    }
    r.near_t = ray.nearT();
    r.far_t = ray.farT();
    r.pixel[0] = state.info.pixelSampleSegment.x();
    r.pixel[1] = state.info.pixelSampleSegment.y();
    r.spp = state.info.pixelSampleSegment.z();
    r.segment = state.info.pixelSampleSegment.w();
    r.scene_seed = state.info.sceneSeed;
    r.info_t = state.info.t;
    r.first_scatter = state.firstScatter ? 1u : 0u;
    r.bounce = state.bounce;
    r.last_val = state.lastVal;
    r.last_gp_id = state.lastGPId;
}

// PCG32's step is s' = s * M + inc (UniformSampler.hpp:43-44); the only public way to set a generator's state is the constructor,
// which takes two steps after storing its seed (:23-27) — so the seed that leaves the generator AT `target` is two steps back.
static uint64 twoStepsBack(uint64 target, uint64 sequence)
{
    const uint64 mult = 6364136223846793005ULL, inc = sequence | 1;
    uint64 inv = mult;                         // Newton iteration for the inverse of an odd number modulo 2^64
    for (int i = 0; i < 6; ++i)
        inv *= 2 - mult*inv;
    uint64 s = (target - inc)*inv;
    return (s - inc)*inv;
}

template<typename Call>
bool HipFunctionSpaceMedium::withContext(PathSampleGenerator &sampler, MediumState &state, Call call) const
{
    UniformSampler &gen = sampler.uniformGenerator();
    if (gen.sequence() != 0)
        FAIL("hip_function_space_gaussian_process: the device's PCG32 stream has sequence 0 (UniformSampler's default)");
    gpis_fs_state fs;
    auto old = std::dynamic_pointer_cast<GPContextHipFs>(state.gpContext);
    if (old)
        fs = old->st;
    else
        std::memset(&fs, 0, sizeof fs);
    fs.sampler_state = gen.state();
    bool ok = call(fs);
    gen = UniformSampler(twoStepsBack(fs.sampler_state, 0));
    // intersectGP leaves a NEW context object in the state it was given (FunctionSpaceGaussianProcessMedium.cpp:254-259, 273-278):
    // a shadow segment on a copy of MediumState (TraceBase.cpp:79-85) does not touch the path's own context
    auto ctxt = std::make_shared<GPContextHipFs>();
    ctxt->st = fs;
    state.gpContext = ctxt;
    return ok;
}

// GaussianProcessMedium::sampleDistance, GaussianProcessMedium.cpp:221-341 over FunctionSpaceGaussianProcessMedium::intersectGP /
// sampleGradient: the batches of sample_points values, the conditioning and the gradient sample run on the device; the MediumState /
// MediumSample writes below are the reference's.
bool HipFunctionSpaceMedium::sampleDistance(PathSampleGenerator &sampler, const Ray &ray,
        MediumState &state, MediumSample &sample) const
{
    sample.emission = Vec3f(0.0f);
    if (state.bounce >= _maxBounce)
        return false;

    float maxT = ray.farT();
    if (!std::isfinite(maxT))
        maxT = float(double(ray.nearT()) + 2000);
    if (maxT == 0.f) {
        sample.t = maxT;
        sample.weight = Vec3f(1.f);
        sample.pdf = 1.0f;
        sample.exited = true;
        sample.p = ray.pos() + sample.t*ray.dir();
        sample.phase = _phaseFunction.get();
        sample.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
        return true;
    }

    gpis_ray_in r;
    gpis_seg_out o;
    fillRay(ray, state, r);
    withContext(sampler, state, [&](gpis_fs_state &fs) {
        if (gpis_fs_sample_distance_host(_handle, 1, &r, &fs, &o) != GPIS_OK)
            FAIL("gpis_fs_sample_distance_host: %s", gpis_last_error());
        return true;
    });

    state.lastGPId = o.gp_id;
    state.lastVal = o.last_val;
    state.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
    sample.exited = o.exited != 0;
    if (_absorptionOnly) {
        state.lastAniso = Vec3d(o.aniso[0], o.aniso[1], o.aniso[2]);
        if (o.weight[0] == 0.f)
            state.firstScatter = false;
    } else {
        state.lastAniso = sample.aniso = Vec3d(o.aniso[0], o.aniso[1], o.aniso[2]);
        state.firstScatter = false;
    }
    if (!o.ok)
        return false;

    sample.t = o.sample_t;
    sample.continuedT = o.continued_t;
    sample.weight = Vec3f(o.weight[0], o.weight[1], o.weight[2]);
    sample.continuedWeight = Vec3f(o.continued_weight[0], o.continued_weight[1], o.continued_weight[2]);
    sample.pdf = 1.0f;
    sample.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
    if (!_absorptionOnly)
        state.advance();
    sample.p = Vec3f(o.p[0], o.p[1], o.p[2]);
    sample.phase = _phaseFunctions[size_t(state.lastGPId) < _phaseFunctions.size() ? state.lastGPId : 0].get();
    sample.gpId = state.lastGPId;
    sample.ctxt = state.gpContext.get();
    state.info.t += sample.t;
    sample.rayInfo = state.info;
    return true;
}

// GaussianProcessMedium::transmittance, GaussianProcessMedium.cpp:343-393: 1 if the segment left the medium, else 0; firstScatter and
// lastAniso change on a hit only (:371-381)
Vec3f HipFunctionSpaceMedium::transmittance(PathSampleGenerator &sampler, const Ray &ray, bool /*startOnSurface*/,
        bool /*endOnSurface*/, MediumState *state) const
{
    gpis_ray_in r;
    uint8_t visible = 0;
    fillRay(ray, *state, r);
    bool ok = withContext(sampler, *state, [&](gpis_fs_state &fs) {
        return gpis_fs_transmittance_host(_handle, 1, &r, &fs, &visible) == GPIS_OK;
    });
    if (!ok)
        return Vec3f(0.0f);
    if (!visible)
        state->firstScatter = false;
    return visible ? Vec3f(1.0f) : Vec3f(0.0f);
}

}
