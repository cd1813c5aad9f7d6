// HipFunctionSpaceMedium.hpp — the binding of the function-space comparison medium a Tungsten maintainer drops into
// src/core/media/ next to HipSparseConvNoiseMedium.
//
// A `Tungsten::Medium` subclass (src/core/media/Medium.hpp:50-115) that forwards the hot path of
// `FunctionSpaceGaussianProcessMedium` (src/core/media/FunctionSpaceGaussianProcessMedium.cpp:34-43, 58-345 under
// GaussianProcessMedium.cpp:221-398) to gpis_fs_sample_distance_host / gpis_fs_transmittance_host of include/gpis.h.
// Like HipSparseConvNoiseMedium it includes only headers of the reference that compile without Boost / FFTW / OpenVDB, so
// tests/test_integration_compile.py compiles it against the real interface.  Registration: one row in MediumFactory.cpp:13-22.
//
// Where the medium's variates come from: the reference draws an unbounded number of normal variates per segment from the
// caller's PathSampleGenerator (rand_normal_2(PathSampleGenerator &), sampling/Gaussian.cpp:36-49).  The device draws them from a
// PCG32 stream, so the binding hands over the state of sampler.uniformGenerator() (UniformSampler.hpp:41-75) and puts the advanced
// state back afterwards: with UniformPathSampler — whose next1D() IS that generator — the draws are the reference's own sequence.
#ifndef HIPFUNCTIONSPACEMEDIUM_HPP_
#define HIPFUNCTIONSPACEMEDIUM_HPP_

#include "media/Medium.hpp"
#include "samplerecords/MediumSample.hpp"
#include "math/Ray.hpp"
#include "sampling/PathSampleGenerator.hpp"

#include <gpis.h>

#include <memory>
#include <string>
#include <vector>

namespace Tungsten {

// What MediumState::gpContext points to for this medium: the role of GPContextFunctionSpace (GaussianProcessMedium.hpp:23-33: the
// previous segment's points, derivative kinds and sampled values), as the POD the device reads and writes.
struct GPContextHipFs : public GPContext
{
    gpis_fs_state st;
    GPContextHipFs();
    virtual void reset() override;             // points.clear(); values.reset(); derivs.clear()
};

class HipFunctionSpaceMedium : public Medium
{
    gpis_params _params;
    gpis_medium *_handle;
    int _device;
    Vec3f _sigmaA, _sigmaS, _sigmaT;
    bool _absorptionOnly;
    std::vector<std::shared_ptr<PhaseFunction>> _phaseFunctions;

    void fillRay(const Ray &ray, const MediumState &state, gpis_ray_in &r) const;
    // reads the context out of `state`, runs `call` on it with the sampler's generator state, stores the new context and state back
    template<typename Call>
    bool withContext(PathSampleGenerator &sampler, MediumState &state, Call call) const;

public:
    HipFunctionSpaceMedium();
    virtual ~HipFunctionSpaceMedium();

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

#endif /* HIPFUNCTIONSPACEMEDIUM_HPP_ */
