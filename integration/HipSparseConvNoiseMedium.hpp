// HipSparseConvNoiseMedium.hpp — the binding a Tungsten maintainer drops into src/core/media/.
//
// A `Tungsten::Medium` subclass (src/core/media/Medium.hpp:50-115) that forwards the hot path of
// `SparseConvolutionNoiseMedium` (src/core/media/SparseConvolutionNoiseMedium.cpp,
// GaussianProcessMedium.cpp:221-398) to the C ABI of include/gpis.h.  It includes ONLY headers of the
// reference that compile without Boost / FFTW / OpenVDB (Medium.hpp, MediumSample.hpp, Ray.hpp,
// PathSampleGenerator.hpp, JsonObject.hpp) — not GaussianProcessMedium.hpp — so this translation unit
// is compiled against the real interface by tests/test_integration_compile.py wherever
// /root/reference exists.  Registration: one row in MediumFactory.cpp:13-22.
#ifndef HIPSPARSECONVNOISEMEDIUM_HPP_
#define HIPSPARSECONVNOISEMEDIUM_HPP_

#include "media/Medium.hpp"
#include "samplerecords/MediumSample.hpp"
#include "math/Ray.hpp"
#include "sampling/PathSampleGenerator.hpp"

#include <gpis.h>

#include <memory>
#include <string>
#include <vector>

namespace Tungsten {

// What MediumSample::ctxt / MediumState::gpContext point to for this medium (the role of
// GPContextSparseConvNoise, SparseConvolutionNoiseMedium.hpp:11-16): the segment's pathwise
// conditioning coefficients (SparseConvolutionNoise.hpp:7-21) plus the medium that evaluates them.
class HipSparseConvNoiseMedium;
struct GPContextHip : public GPContext
{
    gpis_cond_coeff coeff;
    const HipSparseConvNoiseMedium *medium;
    virtual void reset() override {}          // "Don't reset the realization" (SparseConvolutionNoiseMedium.hpp:14)

    // The two calls ConductorBsdf.cpp:68-137 / MirrorBsdf.cpp:40-109 make through the context
    // (SparseConvolutionNoise.cpp:652-743).
    float neePDF(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info) const;
    Vec3f neeGrad(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, const RayInfo &info) const;
};

class HipSparseConvNoiseMedium : public Medium
{
    gpis_params _params;
    gpis_medium *_handle;
    int _device;
    Vec3f _sigmaA, _sigmaS, _sigmaT;
    bool _absorptionOnly;
    std::vector<std::shared_ptr<PhaseFunction>> _phaseFunctions;   // GaussianProcessMedium.hpp: index = gpId

    static int stringToCorrelationContext(const std::string &name);   // GaussianProcessMedium.cpp:30-41
    static int stringToSamplingScheme1D(const std::string &name);     // SparseConvolutionNoiseMedium.cpp:36-45
    void fillRay(const Ray &ray, const MediumState &state, float u, gpis_ray_in &r) const;

public:
    HipSparseConvNoiseMedium();
    virtual ~HipSparseConvNoiseMedium();

    virtual void fromJson(JsonPtr value, const Scene &scene) override;
    virtual rapidjson::Value toJson(Allocator &allocator) const override;

    virtual bool isHomogeneous() const override { return false; }      // GaussianProcessMedium.cpp:147-150

    virtual void prepareForRender() override;                          // GaussianProcessMedium.cpp:152-158 + gpis_create
    virtual void teardownAfterRender() override;

    virtual Vec3f sigmaA(Vec3f /*p*/) const override { return _sigmaA; }
    virtual Vec3f sigmaS(Vec3f /*p*/) const override { return _sigmaS; }
    virtual Vec3f sigmaT(Vec3f /*p*/) const override { return _sigmaT; }

    virtual bool sampleDistance(PathSampleGenerator &sampler, const Ray &ray,
            MediumState &state, MediumSample &sample) const override;
    virtual Vec3f transmittance(PathSampleGenerator &sampler, const Ray &ray, bool startOnSurface,
            bool endOnSurface, MediumState *state) const override;
    virtual float pdf(PathSampleGenerator &/*sampler*/, const Ray &/*ray*/, bool /*startOnSurface*/,
            bool /*endOnSurface*/) const override { return 1.0f; }     // GaussianProcessMedium.cpp:395-398

    void setDevice(int device) { _device = device; }
    gpis_medium *handle() const { return _handle; }
    const gpis_params &params() const { return _params; }
};

}

#endif /* HIPSPARSECONVNOISEMEDIUM_HPP_ */
