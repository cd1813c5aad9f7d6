// host_adapter_test — exercises HipSparseConvNoiseMedium the way PathTracer.cpp / TraceBase.cpp do.
//   host_adapter_test parse     JSON handling and error behaviour only (no GPU needed)
//   host_adapter_test gpu       adds a small path-traced loop on device 0 and checks the adapter's
//                               batch-of-one results against the C ABI's batch entry
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "HipSparseConvNoiseMedium.hpp"

using namespace gpis_host;

static int g_fail = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) { std::printf("CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); ++g_fail; } \
    } while (0)

static const char *kSceneC1 = R"({
  "type": "sparse_conv_noise", "sigma_a": 0, "sigma_s": 1, "density": 1, "max_bounces": 1024,
  "correlation_context": "renewal", "step_size": 0.01, "min_step": 8, "seed": 7,
  "impulse_density": 32, "single_realization": true, "isotropic_3D_sampling": true,
  "gaussian_process": {
     "mean": {"type": "spherical", "center": [0, 0, 0], "radius": 1},
     "covariance": {"type": "squared_exponential", "sigma": 0.1, "lengthScale": 0.05, "aniso": [1, 1, 1], "localScale": 3}
  }
})";

struct ConstSampler : PathSampleGenerator {
    float v;
    int calls = 0;
    explicit ConstSampler(float x) : v(x) {}
    float next1D() override { ++calls; return v; }
};

static bool throws(const std::string &json, const char *needle)
{
    HipSparseConvNoiseMedium m;
    try {
        m.fromJson(json);
    } catch (const std::runtime_error &e) {
        return std::strstr(e.what(), needle) != nullptr;
    }
    return false;
}

static void test_parse()
{
    HipSparseConvNoiseMedium m;
    m.fromJson(kSceneC1);
    const gpis_params &p = m.params();
    CHECK(p.impulse_density == 32.f && p.seed == 7u && p.single_realization == 1 && p.isotropic_3d_sampling == 1);
    CHECK(p.correlation_context == GPIS_CTX_RENEWAL && p.sigma == 0.1f && p.length_scale == 0.05f);
    CHECK(p.mean.type == GPIS_MEAN_SPHERICAL && p.mean.radius == 1.f && p.sigma_s[2] == 1.f);
    // the reference FAILs on these (GPM.cpp:40, SCNM.cpp:44); correlation_context is effectively required (GPM.cpp:104)
    CHECK(throws(R"({"correlation_context": "sometimes"})", "Invalid correlation context"));
    CHECK(throws(R"({"step_size": 0.01})", "Invalid correlation context: 'goldfish'"));
    CHECK(throws(R"({"correlation_context": "none", "1D_sampling_scheme": "both"})", "Invalid sparse conv sampling scheme"));
    CHECK(throws(R"({"correlation_context": "none", "gaussian_process": {"covariance": {"type": "thin_plate"}}})", "Unsupported covariance"));
    HipSparseConvNoiseMedium c6;
    c6.fromJson(R"({"correlation_context": "none", "gaussian_process": {"covariance": {"type": "matern", "sigma": 0.2, "v": 2.5, "lengthScale": 0.07, "aniso": [1, 2, 1]}}})");
    CHECK(c6.params().kernel_type == GPIS_KERNEL_MATERN && c6.params().matern_v == 2.5f && c6.params().length_scale == 0.07f && c6.params().aniso[1] == 2.f);
    c6.fromJson(R"({"correlation_context": "none", "gaussian_process": {"covariance": {"type": "gabor_aniso", "sigma": 0.1, "a_inv": 0.08, "f_inv": 0.06, "omega": [0, 1, 0]}}})");
    CHECK(c6.params().kernel_type == GPIS_KERNEL_GABOR_ANISO && c6.params().gabor_a_inv == 0.08f && c6.params().gabor_omega[1] == 1.f);
    CHECK(HipSparseConvNoiseMedium::stringToCorrelationContext("renewal+") == GPCorrelationContext::RenewalPlus);
    CHECK(HipSparseConvNoiseMedium::stringToSamplingScheme1D("MIS") == SparseConv1DSamplingScheme::MIS);
    HipSparseConvNoiseMedium c3;
    c3.fromJson(R"({"correlation_context": "renewal", "gaussian_process": {"covariance": {"type": "proc_nonstationary",
        "multiResolutionGrid": true, "cov": {"type": "squared_exponential", "sigma": 0.1, "lengthScale": 0.05},
        "ls": {"type": "noise", "noise": "bottom_top", "min": 0.5, "max": 2, "start": -1, "end": 1}}}})");
    CHECK(c3.params().nonstationary == 1 && c3.params().multi_resolution_grid == 1 && c3.params().ls_max == 2.0);
    // the rest of the wrapper and of the mean: "var", a bottom_top_left_right "ls", mean "color" / "emission" (GPF.cpp:1590-1606, GPF.hpp:808-818)
    HipSparseConvNoiseMedium c5;
    c5.fromJson(R"({"correlation_context": "renewal", "gaussian_process": {
        "mean": {"type": "spherical", "radius": 1, "color": {"type": "noise", "noise": "left_right", "min": 0.2, "max": 0.9, "start": -1, "end": 1},
                 "emission": {"type": "noise", "noise": "bottom_top", "min": 0, "max": 2}},
        "covariance": {"type": "proc_nonstationary", "cov": {"type": "squared_exponential", "sigma": 0.1, "lengthScale": 0.05},
        "ls": {"type": "noise", "noise": "bottom_top_left_right", "min": 0.5, "max": 2, "start": -1, "end": 1, "min2": 0.8, "max2": 1.5, "start2": -2, "end2": 2},
        "var": {"type": "noise", "noise": "front_back", "min": 0.4, "max": 1.8, "start": -1, "end": 1}}}})");
    CHECK(c5.params().ls_ramp_type == GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT && c5.params().ls_max2 == 1.5 && c5.params().ls_start2 == -2.0);
    CHECK(c5.params().var.enabled == 1 && c5.params().var.type == GPIS_RAMP_FRONT_BACK && c5.params().var.max == 1.8);
    CHECK(c5.params().mean_color.enabled == 1 && c5.params().mean_color.type == GPIS_RAMP_LEFT_RIGHT && c5.params().mean_color.min == 0.2);
    CHECK(c5.params().mean_emission.enabled == 1 && c5.params().mean_emission.max == 2.0 && c5.params().mean_emission.end == 1.0);
    // round 3: the fbm noises, Matern's v = 1.5 and the grid flavour of the wrapper come through the key table both adapters share
    HipSparseConvNoiseMedium r6;
    r6.fromJson(R"({"correlation_context": "none", "gaussian_process": {"mean": {"color": {"noise": "rust"}},
        "covariance": {"type": "matern", "v": 1.5, "sigma": 0.2, "lengthScale": 0.1}}})");
    CHECK(r6.params().mean_color.enabled == 1 && r6.params().mean_color.type == GPIS_NOISE_RUST);
    CHECK(r6.params().kernel_type == GPIS_KERNEL_MATERN && r6.params().matern_v == 1.5f);
    HipSparseConvNoiseMedium r7;
    r7.fromJson(R"({"correlation_context": "renewal", "gaussian_process": {"covariance": {"type": "grid_nonstationary",
        "cov": {"type": "squared_exponential", "sigma": 0.1, "lengthScale": 0.05}, "offset": 0.1, "scale": 1.25,
        "surf_vol_amp_separate": true, "surf_vol_amp_thresh": 1.4, "surf_ls_scale": 0.7, "vol_ls_scale": 1.6}}})");
    CHECK(r7.params().nonstationary == 1 && r7.params().grid_nonstationary == 1 && r7.params().grid_surf_vol_amp_separate == 1);
    CHECK(r7.params().grid_scale == 1.25f && r7.params().grid_vol_ls_scale == 1.6f && r7.params().sigma == 0.1f);
    CHECK(throws(R"({"correlation_context": "none", "gaussian_process": {"mean": {"color": {"noise": "marble"}}}})", "Invalid noise typ function"));
    // calling the path before prepareForRender fails loudly
    ConstSampler s(0.5f);
    MediumState st;
    MediumSample smp;
    bool threw = false;
    try { m.sampleDistance(s, Ray(), st, smp); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
}

static void test_gpu()
{
    HipSparseConvNoiseMedium m;
    m.fromJson(kSceneC1);
    m.prepareForRender(0);
    CHECK(!m.isHomogeneous());
    Vec3f o; o.x = 0; o.y = 0; o.z = 4;
    const int n = 96;
    std::vector<gpis_ray_in> in(n);
    std::vector<gpis_seg_out> want(n);
    int hits = 0, exits = 0, shadows = 0;
    for (int i = 0; i < n; ++i) {
        // PathTracer::traceSample: fresh state per sample, segment word = bounce (PathTracer.cpp:46-48, 64)
        MediumState st;
        st.info.pixelSampleSegment[0] = 100 + i; st.info.pixelSampleSegment[1] = 50; st.info.pixelSampleSegment[2] = 3; st.info.pixelSampleSegment[3] = 0;
        st.info.sceneSeed = 0xBA5EBA11u;
        st.reset();
        Vec3f d; d.x = -0.5f + 0.0105f * i; d.y = 0.02f; d.z = -1.f;
        float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        d.x /= len; d.y /= len; d.z /= len;
        Ray ray(o, d, 2.4f, 5.6f);
        ConstSampler samp(0.25f + 0.005f * i);
        MediumSample smp;
        MediumState before = st;
        bool ok = m.sampleDistance(samp, ray, st, smp);
        CHECK(samp.calls == 1);
        // the same segment through the C ABI directly
        gpis_ray_in &r = in[i];
        std::memset(&r, 0, sizeof r);
        r.pos[0] = o.x; r.pos[1] = o.y; r.pos[2] = o.z; r.dir[0] = d.x; r.dir[1] = d.y; r.dir[2] = d.z;
        r.near_t = 2.4f; r.far_t = 5.6f; r.pixel[0] = 100 + i; r.pixel[1] = 50; r.spp = 3; r.scene_seed = 0xBA5EBA11u;
        r.u_jitter = 0.25f + 0.005f * i; r.first_scatter = 1;
        CHECK(gpis_sample_distance_host(m.handle(), 1, &r, &want[i], nullptr) == GPIS_OK);
        CHECK(ok == (want[i].ok != 0));
        CHECK(smp.exited == (want[i].exited != 0));
        if (ok) {
            CHECK(smp.t == want[i].sample_t && smp.aniso.x == want[i].aniso[0] && smp.aniso.z == want[i].aniso[2]);
            CHECK(st.bounce == before.bounce + 1 && !st.firstScatter && st.info.t == want[i].sample_t);
            CHECK(smp.ctxt == st.gpContext.get() && smp.rayInfo.t == st.info.t);
            CHECK(smp.weight.x == 1.f && smp.pdf == 1.f);
        }
        (smp.exited ? exits : hits)++;
        if (ok && !smp.exited && shadows < 24) {
            // TraceBase::handleVolume → volumeLightSample → generalizedShadowRay: state copy, segment + 1
            MediumState shadowState = st;
            shadowState.info.pixelSampleSegment[3] += 1;
            Vec3f l; l.x = 0.5025f; l.y = 0.7035f; l.z = 0.5025f;
            Ray shadow(smp.p, l, 0.f, 1.6f);
            ConstSampler s2(0.6f);
            Vec3f tr = m.transmittance(s2, shadow, false, false, &shadowState);
            CHECK((tr.x == 0.f || tr.x == 1.f) && tr.x == tr.y && tr.y == tr.z);
            CHECK(m.pdf(s2, shadow, false, false) == 1.0f);
            ++shadows;
        }
    }
    CHECK(hits > 10 && exits > 0);
    // batch form agrees with the batch of one
    std::vector<Ray> rays(n);
    std::vector<MediumState> states(n);
    std::vector<MediumSample> samples(n);
    std::vector<float> u(n);
    std::vector<uint8_t> ok(n);
    for (int i = 0; i < n; ++i) {
        Vec3f d; d.x = in[i].dir[0]; d.y = in[i].dir[1]; d.z = in[i].dir[2];
        rays[i] = Ray(o, d, 2.4f, 5.6f);
        states[i].info.pixelSampleSegment[0] = 100 + i; states[i].info.pixelSampleSegment[1] = 50; states[i].info.pixelSampleSegment[2] = 3;
        states[i].info.sceneSeed = 0xBA5EBA11u;
        u[i] = in[i].u_jitter;
    }
    m.sampleDistanceBatch(n, u.data(), rays.data(), states.data(), samples.data(), ok.data());
    for (int i = 0; i < n; ++i)
        CHECK((ok[i] != 0) == (want[i].ok != 0) && (!ok[i] || samples[i].t == want[i].sample_t));
    // bounce limit and maxT == 0 (GPM.cpp:235-248)
    MediumState st;
    st.bounce = 1024;
    MediumSample smp;
    ConstSampler s3(0.1f);
    CHECK(!m.sampleDistance(s3, Ray(o, rays[0].dir, 2.4f, 5.6f), st, smp) && st.bounce == 1024 && s3.calls == 0);
    MediumState st0;
    CHECK(m.sampleDistance(s3, Ray(o, rays[0].dir, 0.f, 0.f), st0, smp) && smp.exited && smp.t == 0.f && st0.bounce == 0 && st0.firstScatter);
    m.teardownAfterRender();

    // absorption-only medium (sigma_s = 0: the reference's default when the key is missing, GPM.cpp:87-88):
    // sampleDistance returns weight = transmittance and does NOT advance the state (GPM.cpp:250-258)
    HipSparseConvNoiseMedium a;
    std::string js = kSceneC1;
    js.replace(js.find("\"sigma_a\": 0, \"sigma_s\": 1"), std::strlen("\"sigma_a\": 0, \"sigma_s\": 1"), "\"sigma_a\": 1, \"sigma_s\": 0");
    a.fromJson(js);
    a.prepareForRender(0);
    int blocked = 0, clear = 0;
    for (int i = 0; i < 48; ++i) {
        MediumState sa;
        sa.reset();
        sa.info.pixelSampleSegment[0] = 7 + i; sa.info.sceneSeed = 0xBA5EBA11u;
        Vec3f d; d.x = -0.4f + 0.017f * i; d.y = 0.01f; d.z = -1.f;
        float len = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
        d.x /= len; d.y /= len; d.z /= len;
        ConstSampler sm(0.3f);
        MediumSample out;
        bool ok = a.sampleDistance(sm, Ray(o, d, 2.4f, 5.6f), sa, out);
        CHECK(ok && out.exited && out.t == 5.6f && sm.calls == 1);
        CHECK(sa.bounce == 0);                                     // no advance()
        CHECK(out.weight.x == 0.f || out.weight.x == 1.f);
        CHECK(sa.firstScatter == (out.weight.x == 1.f));           // cleared by the hit inside transmittance() only
        CHECK(sa.info.t == 5.6f && out.rayInfo.t == 5.6f);
        (out.weight.x == 0.f ? blocked : clear)++;
    }
    CHECK(blocked > 5 && clear > 5);
    a.teardownAfterRender();
}

int main(int argc, char **argv)
{
    std::string mode = argc > 1 ? argv[1] : "parse";
    try {
        test_parse();
        if (mode == "gpu") test_gpu();
    } catch (const std::exception &e) {
        std::printf("unexpected exception: %s\n", e.what());
        return 2;
    }
    std::printf("%s: %s (%d failed checks)\n", mode.c_str(), g_fail ? "FAILED" : "ok", g_fail);
    return g_fail ? 1 : 0;
}
