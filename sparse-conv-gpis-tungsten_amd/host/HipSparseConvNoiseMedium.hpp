// HipSparseConvNoiseMedium.hpp — host-side C++ mirror of the reference's plugin interface for the
// sparse-convolution GPIS path, on top of the C ABI in include/gpis.h.
//
// The class below has the same member functions, argument meaning and error behaviour as
// `Tungsten::SparseConvolutionNoiseMedium` (src/core/media/SparseConvolutionNoiseMedium.hpp:18-53)
// seen through `Tungsten::Medium` (src/core/media/Medium.hpp:50-115), so that PathTracer.cpp:68 and
// TraceBase.cpp:118-122 can call it unchanged.  It is written against small stand-alone mirrors of
// the reference's record types (this repository cannot include Tungsten's headers); INTEGRATION.md
// shows the five-line variant that derives from the real `Tungsten::Medium` inside the reference
// tree.  A batch of one goes through gpis_*_host; the *Batch members hand whole ray batches to the
// device entry points (what the tile driver uses).
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gpis.h"

namespace gpis_host {

// ---- mirrors of the reference's records ---------------------------------------------------
struct Vec3f { float x = 0, y = 0, z = 0; };
struct Vec3d { double x = 0, y = 0, z = 0; };

struct Ray {                       // src/core/math/Ray.hpp:13-24
    Vec3f pos, dir;
    float nearT = 1e-4f, farT = 0;
    Ray() { farT = infinity(); }
    Ray(Vec3f p, Vec3f d, float n = 1e-4f, float f = infinity()) : pos(p), dir(d), nearT(n), farT(f) {}
    static float infinity();
};

struct RayInfo {                   // src/core/samplerecords/MediumSample.hpp:14-18
    uint32_t pixelSampleSegment[4] = {0, 0, 0, 0};
    uint32_t sceneSeed = 0;
    float t = 0;
};

enum class SparseConv1DSamplingScheme { UNI, NEE, MIS };                  // Medium.hpp:40-44
enum class GPCorrelationContext { Global, RenewalPlus, Renewal, None };   // GaussianProcess.hpp:26-31

// The realization handle MediumSample.ctxt points to (GPContextSparseConvNoise,
// SparseConvolutionNoiseMedium.hpp:11-16): here the conditioning coefficients of the segment.
struct GPContextSparseConvNoise {
    gpis_cond_coeff coeff{};
    void reset() {}   // "Don't reset the realization"
};

struct MediumState {               // Medium.hpp:59-88
    bool firstScatter = true;
    int component = 0;
    int bounce = 0;
    int lastGPId = 0;
    Vec3d lastAniso;
    float lastVal = 0;
    RayInfo info;
    std::shared_ptr<GPContextSparseConvNoise> gpContext;
    SparseConv1DSamplingScheme sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
    void reset()
    {
        firstScatter = true;
        bounce = 0;
        if (gpContext) gpContext->reset();
        lastGPId = 0;
        sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
    }
    void advance()
    {
        firstScatter = false;
        sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
        bounce++;
    }
};

struct MediumSample {              // MediumSample.hpp:21-37 (phase is the index into the medium's phase list)
    int phase = 0;
    Vec3f p;
    float continuedT = 0;
    Vec3f continuedWeight;
    float t = 0;
    Vec3f weight;
    Vec3f emission;
    float pdf = 0;
    bool exited = false;
    Vec3d aniso;
    int gpId = 0;
    SparseConv1DSamplingScheme sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
    GPContextSparseConvNoise *ctxt = nullptr;
    RayInfo rayInfo;
};

// The one thing the path needs from the reference's PathSampleGenerator (SCNM.cpp:129).
struct PathSampleGenerator {
    virtual ~PathSampleGenerator() = default;
    virtual float next1D() = 0;
};

class HipSparseConvNoiseMedium {
public:
    HipSparseConvNoiseMedium();                      // the reference's defaults (SCNM.cpp:17-34)
    ~HipSparseConvNoiseMedium();
    HipSparseConvNoiseMedium(const HipSparseConvNoiseMedium &) = delete;
    HipSparseConvNoiseMedium &operator=(const HipSparseConvNoiseMedium &) = delete;

    // JSON of a `{"type": "sparse_conv_noise", ...}` medium object, with the gaussian process inlined:
    //   "gaussian_process": {"mean": {"type": "spherical", ...}, "covariance": {"type": "squared_exponential", ...}}
    // Same keys as SCNM.cpp:57-73, GPM.cpp:97-126, GPF.cpp:654-679 / 1590-1606; an invalid
    // "correlation_context" / "1D_sampling_scheme" string throws std::runtime_error like FAIL().
    void fromJson(const std::string &json);
    void setParams(const gpis_params &p) { _params = p; }
    const gpis_params &params() const { return _params; }

    bool isHomogeneous() const { return false; }                  // GPM.cpp:147-150
    void prepareForRender(int device = 0);                        // GPM.cpp:152-158 + device handle
    void teardownAfterRender();
    Vec3f sigmaA(Vec3f p) const;
    Vec3f sigmaS(Vec3f p) const;
    Vec3f sigmaT(Vec3f p) const;

    // Medium.hpp:104-108 / GPM.cpp:221-398
    bool sampleDistance(PathSampleGenerator &sampler, const Ray &ray, MediumState &state, MediumSample &sample) const;
    Vec3f transmittance(PathSampleGenerator &sampler, const Ray &ray, bool startOnSurface, bool endOnSurface, MediumState *state) const;
    float pdf(PathSampleGenerator &sampler, const Ray &ray, bool startOnSurface, bool endOnSurface) const { return 1.0f; }

    // Batched forms (host arrays): element i uses jitter u[i].  Return values as above, per element.
    void sampleDistanceBatch(size_t n, const float *u, const Ray *rays, MediumState *states, MediumSample *samples, uint8_t *ok) const;
    void transmittanceBatch(size_t n, const float *u, const Ray *rays, const MediumState *states, uint8_t *visible) const;

    // The calls ConductorBsdf/MirrorBsdf make through sample.ctxt (ConductorBsdf.cpp:68-137 →
    // SCN.cpp:652-743).
    float neePDF(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info, const GPContextSparseConvNoise &ctxt) const;
    Vec3f neeGrad(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, const RayInfo &info, const GPContextSparseConvNoise &ctxt) const;

    gpis_medium *handle() const { return _handle; }

    static GPCorrelationContext stringToCorrelationContext(const std::string &name);   // GPM.cpp:30-41
    static SparseConv1DSamplingScheme stringToSamplingScheme1D(const std::string &name); // SCNM.cpp:36-45

private:
    void requireHandle() const;
    static void fillRay(const Ray &ray, const MediumState &state, float u, gpis_ray_in &r);
    void applyResult(const Ray &ray, const gpis_seg_out &o, const gpis_cond_coeff &c, MediumState &state, MediumSample &sample) const;

    gpis_params _params;
    gpis_medium *_handle = nullptr;
    float _sigmaA[3] = {0, 0, 0}, _sigmaS[3] = {0, 0, 0}, _sigmaT[3] = {0, 0, 0};
};

}   // namespace gpis_host
