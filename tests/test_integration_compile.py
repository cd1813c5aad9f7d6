"""The binding of INTEGRATION.md (integration/HipSparseConvNoiseMedium.{hpp,cpp}) compiled against the
reference's REAL plugin interface: Medium.hpp:50-115, MediumSample.hpp:14-37, Ray.hpp,
PathSampleGenerator.hpp, JsonPtr/JsonObject — the vendored headers only, no stand-ins.  Runs where
/root/reference exists (this container); skipped on the GPU box, which has no reference tree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _compile(tmp_path, extra_src=None, unit="HipSparseConvNoiseMedium.cpp"):
    src = os.path.join(ROOT, "integration", unit)
    if extra_src is not None:
        src = extra_src
    obj = str(tmp_path / "binding.o")
    cmd = ["g++", "-std=c++17", "-c", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror=overloaded-virtual",
           "-DCONSTEXPR=constexpr", "-DRAPIDJSON_HAS_STDSTRING=1",
           "-I", os.path.join(REF, "core"), "-isystem", os.path.join(REF, "thirdparty"),
           "-isystem", os.path.join(REF, "thirdparty", "eigen"), "-I", REF,
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "integration"), "-o", obj, src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    return r, obj


def test_binding_compiles_against_the_real_medium_interface(tmp_path):
    r, obj = _compile(tmp_path)
    assert r.returncode == 0, r.stderr[-4000:]
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    # every pure virtual of Tungsten::Medium (Medium.hpp:96-108) is defined with the reference's own signature
    for needle in (
        "Tungsten::HipSparseConvNoiseMedium::sampleDistance(Tungsten::PathSampleGenerator&, Tungsten::Ray const&, "
        "Tungsten::Medium::MediumState&, Tungsten::MediumSample&) const",
        "Tungsten::HipSparseConvNoiseMedium::transmittance(Tungsten::PathSampleGenerator&, Tungsten::Ray const&, bool, bool, "
        "Tungsten::Medium::MediumState*) const",
        "Tungsten::HipSparseConvNoiseMedium::fromJson(Tungsten::JsonPtr, Tungsten::Scene const&)",
        "Tungsten::HipSparseConvNoiseMedium::prepareForRender()",
    ):
        assert needle in syms, needle
    # and the only undefined gpis_* symbols are entry points include/gpis.h declares
    header = open(os.path.join(ROOT, "include", "gpis.h")).read()
    undefined = [l.split()[-1] for l in syms.splitlines() if " U gpis_" in l]
    assert undefined, "the binding must call into the C ABI"
    for u in undefined:
        assert u + "(" in header, u


def test_binding_is_instantiable_as_a_medium(tmp_path):
    """A translation unit that constructs the class through the factory's `std::make_shared<T>` form
    (MediumFactory.cpp:13-22) and uses the state records the way PathTracer.cpp:42-70 does: fails to
    compile if a pure virtual is left unimplemented or a field name/type drifted."""
    tu = tmp_path / "use.cpp"
    tu.write_text(r'''
#include "HipSparseConvNoiseMedium.hpp"
#include <memory>
using namespace Tungsten;
std::shared_ptr<Medium> make() { return std::make_shared<HipSparseConvNoiseMedium>(); }
bool drive(const Medium &m, PathSampleGenerator &sampler, const Ray &ray)
{
    Medium::MediumState state;
    state.reset();
    state.info.pixelSampleSegment = Vec4u(1u, 2u, 3u, 0u);
    state.info.sceneSeed = 7u;
    state.info.t = 0.0f;
    MediumSample sample;
    bool ok = m.sampleDistance(sampler, ray, state, sample);
    Medium::MediumState shadow = state;                 // TraceBase.cpp:79-85: shadow rays run on a copy
    shadow.info.pixelSampleSegment.w() += 1;
    Vec3f tr = m.transmittance(sampler, ray, false, false, &shadow);
    GPContextHip *ctxt = static_cast<GPContextHip *>(sample.ctxt);   // ConductorBsdf.cpp:72-73's cast
    float pdf = ctxt ? ctxt->neePDF(ray.dir(), Vec3f(0.f, 0.f, 1.f), sample.p, sample.t, sample.rayInfo) : 0.f;
    SparseConv1DSamplingScheme s = sample.sparseConv1DSamplingScheme;
    Vec3d n = sample.aniso;
    PhaseFunction *ph = sample.phase;
    return ok && tr.x() > 0.f && pdf >= 0.f && s == SparseConv1DSamplingScheme::UNI && n.x() == n.x() && ph != nullptr && sample.gpId == state.lastGPId;
}
''')
    r, _ = _compile(tmp_path, str(tu))
    assert r.returncode == 0, r.stderr[-4000:]


def test_function_space_binding_compiles_and_is_instantiable(tmp_path):
    """integration/HipFunctionSpaceMedium.{hpp,cpp}: the Medium subclass for FunctionSpaceGaussianProcessMedium
    (FunctionSpaceGaussianProcessMedium.cpp:34-43, 58-345) against the real Medium.hpp, PathSampleGenerator.hpp and UniformSampler.hpp
    (it moves the state of sampler.uniformGenerator() to the device and back)."""
    r, obj = _compile(tmp_path, unit="HipFunctionSpaceMedium.cpp")
    assert r.returncode == 0, r.stderr[-4000:]
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    for needle in (
        "Tungsten::HipFunctionSpaceMedium::sampleDistance(Tungsten::PathSampleGenerator&, Tungsten::Ray const&, "
        "Tungsten::Medium::MediumState&, Tungsten::MediumSample&) const",
        "Tungsten::HipFunctionSpaceMedium::transmittance(Tungsten::PathSampleGenerator&, Tungsten::Ray const&, bool, bool, "
        "Tungsten::Medium::MediumState*) const",
        "Tungsten::HipFunctionSpaceMedium::fromJson(Tungsten::JsonPtr, Tungsten::Scene const&)",
    ):
        assert needle in syms, needle
    header = open(os.path.join(ROOT, "include", "gpis.h")).read()
    undefined = [l.split()[-1] for l in syms.splitlines() if " U gpis_" in l]
    assert "gpis_fs_sample_distance_host" in undefined and "gpis_fs_transmittance_host" in undefined
    for u in undefined:
        assert u + "(" in header, u
    tu = tmp_path / "use_fs.cpp"
    tu.write_text(r'''
#include "HipFunctionSpaceMedium.hpp"
#include "sampling/UniformPathSampler.hpp"
#include <memory>
using namespace Tungsten;
std::shared_ptr<Medium> make() { return std::make_shared<HipFunctionSpaceMedium>(); }
bool drive(const Medium &m, const Ray &ray)
{
    UniformPathSampler sampler(0xBA5EBA11u);           // its next1D() is the generator whose state the binding hands to the device
    Medium::MediumState state;
    state.reset();
    MediumSample sample;
    bool ok = m.sampleDistance(sampler, ray, state, sample);
    Medium::MediumState shadow = state;                 // TraceBase.cpp:79-85: shadow rays run on a copy
    Vec3f tr = m.transmittance(sampler, ray, false, false, &shadow);
    GPContextHipFs *ctxt = dynamic_cast<GPContextHipFs *>(state.gpContext.get());
    return ok && tr.x() >= 0.f && ctxt && ctxt->st.has_context && shadow.gpContext != state.gpContext && sample.gpId == state.lastGPId;
}
''')
    r, _ = _compile(tmp_path, str(tu))
    assert r.returncode == 0, r.stderr[-4000:]

