// HipSparseConvNoiseMedium.cpp — see the header.  Compiles against the reference's real headers:
//   g++ -std=c++17 -c -DCONSTEXPR=constexpr -DRAPIDJSON_HAS_STDSTRING=1
//       -I<ref>/src/core -I<ref>/src/thirdparty -I<ref>/src/thirdparty/eigen -I<ref>/src -I<repo>/include
// (tests/test_integration_compile.py does exactly that).  Links against libgpis_hip.so.
#include "HipSparseConvNoiseMedium.hpp"

#include "io/JsonObject.hpp"
#include "TungstenJsonAccess.hpp"

#include <cmath>
#include <cstring>

namespace Tungsten {

int HipSparseConvNoiseMedium::stringToCorrelationContext(const std::string &name) { return gpis_json::correlationContext<TungstenJson>(name); }
int HipSparseConvNoiseMedium::stringToSamplingScheme1D(const std::string &name) { return gpis_json::samplingScheme1D<TungstenJson>(name); }

HipSparseConvNoiseMedium::HipSparseConvNoiseMedium()
: _handle(nullptr),
  _device(0),
  _sigmaA(0.0f),
  _sigmaS(0.0f),
  _sigmaT(0.0f),
  _absorptionOnly(true)
{
    gpis_default_params(&_params);     // the reference's defaults (SparseConvolutionNoiseMedium.cpp:17-34)
}

HipSparseConvNoiseMedium::~HipSparseConvNoiseMedium()
{
    teardownAfterRender();
}

void HipSparseConvNoiseMedium::fromJson(JsonPtr value, const Scene &scene)
{
    Medium::fromJson(value, scene);      // phase_function, transmittance, max_bounces (Medium.cpp:29-38)
    _params.max_bounces = _maxBounce;

    // GaussianProcessMedium::fromJson (GaussianProcessMedium.cpp:97-126), GaussianProcess::fromJson for an INLINE object
    // (GaussianProcess.cpp:172-190) and SparseConvolutionNoiseMedium::fromJson (SparseConvolutionNoiseMedium.cpp:57-73): the key
    // table of include/gpis_json.hpp.  The reference resolves "gaussian_process" through Scene::fetchGaussianProcess, which also
    // accepts the NAME of a process declared at scene level; a maintainer who wants that form adds accessors to GaussianProcess
    // and fills _params from them here (GaussianProcess.hpp keeps _mean / _cov public).
    if (auto gp = value["gaussian_process"])
        if (!gp.isObject())
            FAIL("hip_sparse_conv_noise: \"gaussian_process\" must be an inline object");
    gpis_json::readMedium<TungstenJson>(value, _params);
    _phaseFunctions.clear();
    _phaseFunctions.push_back(_phaseFunction);     // "We always have the default one"
    int device = _device;
    value.getField("hip_device", device);
    _device = device;
}

rapidjson::Value HipSparseConvNoiseMedium::toJson(Allocator &allocator) const
{
    static const char *ctxNames[] = {"global", "renewal+", "renewal", "none"};
    static const char *schemeNames[] = {"uni", "nee", "mis"};
    return JsonObject{Medium::toJson(allocator), allocator,
        "type", "hip_sparse_conv_noise",
        "sigma_a", Vec3f(_params.sigma_a[0], _params.sigma_a[1], _params.sigma_a[2]),
        "sigma_s", Vec3f(_params.sigma_s[0], _params.sigma_s[1], _params.sigma_s[2]),
        "density", _params.density,
        "correlation_context", ctxNames[_params.correlation_context],
        "step_size", _params.step_size,
        "min_step", _params.min_step,
        "seed", _params.seed,
        "impulse_density", _params.impulse_density,
        "single_realization", _params.single_realization != 0,
        "isotropic_3D_sampling", _params.isotropic_3d_sampling != 0,
        "1D_sampling", _params.sampling_1d != 0,
        "1D_sampling_scheme", schemeNames[_params.scheme_1d],
        "1D_gradient_correlationXY", _params.correlation_xy != 0,
        "surf_vol_phase_separate", _params.surf_vol_phase_separate != 0,
        "surf_vol_phase_amp_thresh", _params.surf_vol_phase_amp_thresh
    };
}

void HipSparseConvNoiseMedium::prepareForRender()
{
    teardownAfterRender();
    _sigmaA = Vec3f(_params.sigma_a[0], _params.sigma_a[1], _params.sigma_a[2])*_params.density;
    _sigmaS = Vec3f(_params.sigma_s[0], _params.sigma_s[1], _params.sigma_s[2])*_params.density;
    _sigmaT = _sigmaA + _sigmaS;
    _absorptionOnly = _sigmaS == 0.0f;
    if (gpis_create(&_params, _device, &_handle) != GPIS_OK) {
        _handle = nullptr;
        FAIL("hip_sparse_conv_noise: gpis_create failed: %s", gpis_last_error());
    }
}

void HipSparseConvNoiseMedium::teardownAfterRender()
{
    if (_handle)
        gpis_destroy(_handle);
    _handle = nullptr;
}

void HipSparseConvNoiseMedium::fillRay(const Ray &ray, const MediumState &state, float u, gpis_ray_in &r) const
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
    r.u_jitter = u;
    r.first_scatter = state.firstScatter ? 1u : 0u;
    r.bounce = state.bounce;
    r.last_val = state.lastVal;
    r.last_gp_id = state.lastGPId;
}

// GaussianProcessMedium::sampleDistance, GaussianProcessMedium.cpp:221-341: the march, the gradient and the
// validity checks run on the device; the MediumState / MediumSample writes below are the reference's.
bool HipSparseConvNoiseMedium::sampleDistance(PathSampleGenerator &sampler, const Ray &ray,
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
    // (GaussianProcessMedium.cpp:251-252 tests maxT against infinity AFTER the clamp above: dead code in the
    //  reference, so an infinite absorbing segment is marched over its 2000 units there and here)

    gpis_ray_in r;
    gpis_seg_out o;
    auto ctxt = std::make_shared<GPContextHip>();
    ctxt->medium = this;
    fillRay(ray, state, sampler.next1D(), r);     // the path's ONE next1D() (SparseConvolutionNoiseMedium.cpp:129)
    if (gpis_sample_distance_host(_handle, 1, &r, &o, &ctxt->coeff) != GPIS_OK)
        FAIL("gpis_sample_distance_host: %s", gpis_last_error());

    // intersectGP's state writes (SparseConvolutionNoiseMedium.cpp:162-181) and sampleDistance's own
    state.gpContext = ctxt;
    state.lastGPId = o.gp_id;
    state.lastVal = o.last_val;
    state.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme(o.scheme);
    sample.exited = o.exited != 0;
    if (_absorptionOnly) {
        // GaussianProcessMedium.cpp:250-258: the weight is transmittance(), which writes lastAniso / firstScatter
        // on a hit only (:371-381) — o.aniso is the incoming lastAniso otherwise; sample.aniso stays untouched
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
    sample.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme(o.scheme);
    if (!_absorptionOnly)
        state.advance();
    sample.p = Vec3f(o.p[0], o.p[1], o.p[2]);
    if (!o.exited && _params.mean_emission.enabled) {        // sample.emission = emission(ro + rd*t), GaussianProcessMedium.cpp:317
        Vec3d ip = Vec3d(ray.pos()) + Vec3d(ray.dir()).normalized()*o.t;
        double p3[3] = {ip.x(), ip.y(), ip.z()};
        float e[3] = {0.0f, 0.0f, 0.0f};
        if (gpis_mean_color_emission_host(_handle, 1, p3, nullptr, e) != GPIS_OK)
            FAIL("gpis_mean_color_emission_host: %s", gpis_last_error());
        sample.emission = Vec3f(e[0], e[1], e[2]);
    }
    sample.phase = _phaseFunctions[size_t(state.lastGPId) < _phaseFunctions.size() ? state.lastGPId : 0].get();
    sample.gpId = state.lastGPId;
    sample.ctxt = state.gpContext.get();
    state.info.t += sample.t;
    sample.rayInfo = state.info;
    return true;
}

// GaussianProcessMedium::transmittance, GaussianProcessMedium.cpp:343-393
Vec3f HipSparseConvNoiseMedium::transmittance(PathSampleGenerator &sampler, const Ray &ray, bool /*startOnSurface*/,
        bool /*endOnSurface*/, MediumState *state) const
{
    gpis_ray_in r;
    uint8_t visible = 0;
    fillRay(ray, *state, sampler.next1D(), r);
    if (gpis_transmittance_host(_handle, 1, &r, &visible) != GPIS_OK)
        return Vec3f(0.0f);
    // GaussianProcessMedium.cpp:371-381: firstScatter is cleared (and lastAniso set to the hit's gradient) on a hit only.  The device
    // skips that gradient — transmittance() cannot return it and the reference's callers run shadow segments on a state COPY
    // (TraceBase.cpp:79-85, 291-293) — so lastAniso is left as it was: an integrator that reads it after a blocked shadow segment
    // needs sampleDistance instead.
    if (!visible)
        state->firstScatter = false;
    return visible ? Vec3f(1.0f) : Vec3f(0.0f);
}

static void fillNee(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info,
        const gpis_cond_coeff &coeff, gpis_nee_query &q)
{
    std::memset(&q, 0, sizeof q);
    for (int i = 0; i < 3; ++i) {
        q.ray_dir[i] = rayDir[i];
        q.normal[i] = normal[i];
        q.p[i] = p[i];
    }
    q.t_segment = tSegment;
    q.info_t = info.t;
    q.pixel[0] = info.pixelSampleSegment.x();
    q.pixel[1] = info.pixelSampleSegment.y();
    q.spp = info.pixelSampleSegment.z();
    q.segment = info.pixelSampleSegment.w();
    q.scene_seed = info.sceneSeed;
    q.coeff = coeff;
}

float GPContextHip::neePDF(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info) const
{
    gpis_nee_query q;
    fillNee(rayDir, normal, p, tSegment, info, coeff, q);
    float pdf = 0.0f;
    if (gpis_nee_pdf_host(medium->handle(), 1, &q, &pdf) != GPIS_OK)
        FAIL("gpis_nee_pdf_host: %s", gpis_last_error());
    return pdf;
}

Vec3f GPContextHip::neeGrad(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, const RayInfo &info) const
{
    gpis_nee_query q;
    fillNee(rayDir, normal, p, 0.0f, info, coeff, q);
    float g[3] = {0.0f, 0.0f, 0.0f};
    if (gpis_nee_grad_host(medium->handle(), 1, &q, g) != GPIS_OK)
        FAIL("gpis_nee_grad_host: %s", gpis_last_error());
    return Vec3f(g[0], g[1], g[2]);
}

}
