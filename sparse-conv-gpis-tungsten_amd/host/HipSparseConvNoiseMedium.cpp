// HipSparseConvNoiseMedium.cpp — see the header.  Plain C++17 (g++), links libgpis_hip.so.
#include "HipSparseConvNoiseMedium.hpp"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>

namespace gpis_host {

float Ray::infinity() { return std::numeric_limits<float>::infinity(); }

// ---------------------------------------------------------------------------------------------
// a small JSON reader (objects, arrays, numbers, strings, booleans) — the reference parses with
// rapidjson through JsonPtr::getField; only the keys of the hot path are consumed here
// ---------------------------------------------------------------------------------------------
namespace {

struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue *get(const std::string &key) const
    {
        for (auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const std::string &s;
    size_t i = 0;
    explicit JParser(const std::string &text) : s(text) {}
    [[noreturn]] void fail(const char *what) const { throw std::runtime_error(std::string("JSON parse error: ") + what + " at offset " + std::to_string(i)); }
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    JValue parse()
    {
        ws();
        if (i >= s.size()) fail("unexpected end");
        char c = s[i];
        JValue v;
        if (c == '{') {
            v.kind = JValue::Object;
            ++i; ws();
            if (i < s.size() && s[i] == '}') { ++i; return v; }
            for (;;) {
                ws();
                JValue k = parse();
                if (k.kind != JValue::String) fail("object key must be a string");
                ws();
                if (i >= s.size() || s[i] != ':') fail("expected ':'");
                ++i;
                v.obj.emplace_back(k.str, parse());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = JValue::Array;
            ++i; ws();
            if (i < s.size() && s[i] == ']') { ++i; return v; }
            for (;;) {
                v.arr.push_back(parse());
                ws();
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = JValue::String;
            ++i;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) ++i;
                v.str.push_back(s[i++]);
            }
            if (i >= s.size()) fail("unterminated string");
            ++i;
        } else if (!s.compare(i, 4, "true")) { v.kind = JValue::Bool; v.b = true; i += 4; }
        else if (!s.compare(i, 5, "false")) { v.kind = JValue::Bool; v.b = false; i += 5; }
        else if (!s.compare(i, 4, "null")) { i += 4; }
        else {
            char *end = nullptr;
            v.num = std::strtod(s.c_str() + i, &end);
            if (end == s.c_str() + i) fail("unexpected character");
            v.kind = JValue::Number;
            i = (size_t)(end - s.c_str());
        }
        return v;
    }
};

template <typename T>
void getNum(const JValue &o, const char *key, T &dst)
{
    if (const JValue *v = o.get(key)) {
        if (v->kind == JValue::Number) dst = (T)v->num;
        else if (v->kind == JValue::Bool) dst = (T)(v->b ? 1 : 0);
    }
}
// Vec3f fields accept a scalar or a 3-array, as Tungsten's JsonPtr does for Vec3f
void getVec3(const JValue &o, const char *key, float *dst)
{
    if (const JValue *v = o.get(key)) {
        if (v->kind == JValue::Number) dst[0] = dst[1] = dst[2] = (float)v->num;
        else if (v->kind == JValue::Array && v->arr.size() == 3)
            for (int i = 0; i < 3; ++i) dst[i] = (float)v->arr[i].num;
    }
}
void getVec3d(const JValue &o, const char *key, double *dst)
{
    if (const JValue *v = o.get(key)) {
        if (v->kind == JValue::Number) dst[0] = dst[1] = dst[2] = v->num;
        else if (v->kind == JValue::Array && v->arr.size() == 3)
            for (int i = 0; i < 3; ++i) dst[i] = v->arr[i].num;
    }
}

int noiseType(const std::string &noise)            // ProceduralNoise(Vec)::stringToNoiseType, GPF.hpp:644-661
{
    if (noise == "bottom_top") return GPIS_RAMP_BOTTOM_TOP;
    if (noise == "left_right") return GPIS_RAMP_LEFT_RIGHT;
    if (noise == "front_back") return GPIS_RAMP_FRONT_BACK;
    if (noise == "bottom_top_left_right") return GPIS_RAMP_BOTTOM_TOP_LEFT_RIGHT;
    if (noise == "sandstone" || noise == "rust") throw std::runtime_error("noise type '" + noise + "' is outside the built scope");
    throw std::runtime_error("Invalid noise typ function: '" + noise + "'");
}
void readRamp(const JValue &v, gpis_ramp &r)         // ProceduralNoise::fromJson, GPF.hpp:671-688
{
    r.enabled = 1;
    std::string noise = "bottom_top";
    if (const JValue *n = v.get("noise")) noise = n->str;
    r.type = noiseType(noise);
    getNum(v, "min", r.min); getNum(v, "max", r.max); getNum(v, "start", r.start); getNum(v, "end", r.end);
    getNum(v, "min2", r.min2); getNum(v, "max2", r.max2); getNum(v, "start2", r.start2); getNum(v, "end2", r.end2);
}

void readMean(const JValue &m, gpis_mean &dst)
{
    std::string type = "spherical";
    if (const JValue *t = m.get("type")) type = t->str;
    if (type == "homogeneous") {                     // GPF.hpp:871-874
        dst.type = GPIS_MEAN_HOMOGENEOUS;
        getNum(m, "offset", dst.offset);
    } else if (type == "spherical") {                // GPF.hpp:908-912
        dst.type = GPIS_MEAN_SPHERICAL;
        getVec3d(m, "center", dst.center);
        getNum(m, "radius", dst.radius);
    } else if (type == "linear") {                   // GPF.hpp:953-962
        dst.type = GPIS_MEAN_LINEAR;
        getVec3d(m, "reference_point", dst.center);
        getVec3d(m, "direction", dst.dir);
        getNum(m, "scale", dst.scale);
        getNum(m, "min", dst.min);
    } else {
        throw std::runtime_error("Unsupported mean function type: '" + type + "'");
    }
}

void readSE(const JValue &c, gpis_params &p)          // GPF.cpp:654-679, GPF.hpp:1481-1484
{
    getNum(c, "sigma", p.sigma);
    getNum(c, "lengthScale", p.length_scale);
    getVec3(c, "aniso", p.aniso);
    getNum(c, "useAnisoMtx", p.use_aniso_mtx);
    getNum(c, "localScale", p.local_scale);
    if (const JValue *m = c.get("anisoMtx")) {
        if (m->kind == JValue::Array && m->arr.size() == 9)
            for (int i = 0; i < 9; ++i) p.aniso_mtx[i] = (float)m->arr[i].num;
    }
}

}   // namespace

// ---------------------------------------------------------------------------------------------
GPCorrelationContext HipSparseConvNoiseMedium::stringToCorrelationContext(const std::string &name)
{
    if (name == "global") return GPCorrelationContext::Global;
    if (name == "renewal+") return GPCorrelationContext::RenewalPlus;
    if (name == "renewal") return GPCorrelationContext::Renewal;
    if (name == "none") return GPCorrelationContext::None;
    throw std::runtime_error("Invalid correlation context: '" + name + "'");
}
SparseConv1DSamplingScheme HipSparseConvNoiseMedium::stringToSamplingScheme1D(const std::string &name)
{
    if (name == "uni" || name == "UNI") return SparseConv1DSamplingScheme::UNI;
    if (name == "nee" || name == "NEE") return SparseConv1DSamplingScheme::NEE;
    if (name == "mis" || name == "MIS") return SparseConv1DSamplingScheme::MIS;
    throw std::runtime_error("Invalid sparse conv sampling scheme: '" + name + "'");
}

HipSparseConvNoiseMedium::HipSparseConvNoiseMedium() { gpis_default_params(&_params); }
HipSparseConvNoiseMedium::~HipSparseConvNoiseMedium() { teardownAfterRender(); }

void HipSparseConvNoiseMedium::fromJson(const std::string &json)
{
    JParser parser(json);
    JValue v = parser.parse();
    if (v.kind != JValue::Object) throw std::runtime_error("medium JSON must be an object");
    gpis_params &p = _params;
    // Medium::fromJson (Medium.cpp:29-38) and GaussianProcessMedium::fromJson (GPM.cpp:97-126)
    getNum(v, "max_bounces", p.max_bounces);
    getVec3(v, "sigma_a", p.sigma_a);
    getVec3(v, "sigma_s", p.sigma_s);
    getNum(v, "density", p.density);
    std::string ctxt = "goldfish";   // the reference's (invalid) default: the key is effectively required
    if (const JValue *c = v.get("correlation_context")) ctxt = c->str;
    p.correlation_context = (int32_t)stringToCorrelationContext(ctxt);
    // SparseConvolutionNoiseMedium::fromJson (SCNM.cpp:57-73)
    getNum(v, "step_size", p.step_size);
    getNum(v, "min_step", p.min_step);
    getNum(v, "seed", p.seed);
    getNum(v, "impulse_density", p.impulse_density);
    getNum(v, "single_realization", p.single_realization);
    getNum(v, "isotropic_3D_sampling", p.isotropic_3d_sampling);
    getNum(v, "1D_sampling", p.sampling_1d);
    std::string scheme = "uni";
    if (const JValue *s = v.get("1D_sampling_scheme")) scheme = s->str;
    p.scheme_1d = (int32_t)stringToSamplingScheme1D(scheme);
    getNum(v, "1D_gradient_correlationXY", p.correlation_xy);
    getNum(v, "surf_vol_phase_separate", p.surf_vol_phase_separate);
    getNum(v, "surf_vol_phase_amp_thresh", p.surf_vol_phase_amp_thresh);
    if (const JValue *gp = v.get("gaussian_process")) {
        if (const JValue *m = gp->get("mean")) {
            readMean(*m, p.mean);
            if (const JValue *c = m->get("color")) readRamp(*c, p.mean_color);        // MeanFunction::fromJson, GPF.hpp:808-818
            if (const JValue *e = m->get("emission")) readRamp(*e, p.mean_emission);
        }
        if (const JValue *m = gp->get("mean_additional")) {
            p.has_mean_additional = 1;
            readMean(*m, p.mean_additional);
        }
        if (const JValue *c = gp->get("covariance")) {
            std::string type = "squared_exponential";
            if (const JValue *t = c->get("type")) type = t->str;
            if (type == "squared_exponential") {
                readSE(*c, p);
            } else if (type == "matern") {                         // MaternCovariance::fromJson, GPF.cpp:866-876
                p.kernel_type = GPIS_KERNEL_MATERN;
                getNum(*c, "sigma", p.sigma);
                getNum(*c, "v", p.matern_v);
                getNum(*c, "lengthScale", p.length_scale);
                getVec3(*c, "aniso", p.aniso);
                getNum(*c, "localScale", p.local_scale);
            } else if (type == "gabor_aniso" || type == "gabor_iso") {   // GPF.cpp:1086-1096, 1155-1162
                p.kernel_type = type == "gabor_aniso" ? GPIS_KERNEL_GABOR_ANISO : GPIS_KERNEL_GABOR_ISO;
                getNum(*c, "sigma", p.sigma);
                getNum(*c, "a_inv", p.gabor_a_inv);
                getNum(*c, "f_inv", p.gabor_f_inv);
                getVec3(*c, "omega", p.gabor_omega);
                getNum(*c, "localScale", p.local_scale);
            } else if (type == "proc_nonstationary") {           // GPF.hpp:2211-2217, GPF.cpp:1590-1606
                p.nonstationary = 1;
                getNum(*c, "multiResolutionGrid", p.multi_resolution_grid);
                if (const JValue *inner = c->get("cov")) readSE(*inner, p);
                if (const JValue *ls = c->get("ls")) {            // ProceduralNoiseVec, GPF.hpp:759-776
                    std::string noise = "bottom_top";
                    if (const JValue *n = ls->get("noise")) noise = n->str;
                    p.ls_ramp_type = noiseType(noise);
                    getNum(*ls, "min", p.ls_min);
                    getNum(*ls, "max", p.ls_max);
                    getNum(*ls, "start", p.ls_start);
                    getNum(*ls, "end", p.ls_end);
                    getNum(*ls, "min2", p.ls_min2);
                    getNum(*ls, "max2", p.ls_max2);
                    getNum(*ls, "start2", p.ls_start2);
                    getNum(*ls, "end2", p.ls_end2);
                }
                if (const JValue *var = c->get("var")) readRamp(*var, p.var);          // GPF.cpp:1593-1595
                if (const JValue *an = c->get("aniso")) readRamp(*an, p.aniso_field);  // GPF.cpp:1600-1602
            } else {
                throw std::runtime_error("Unsupported covariance type: '" + type + "'");
            }
        }
    }
}

void HipSparseConvNoiseMedium::prepareForRender(int device)
{
    teardownAfterRender();
    for (int c = 0; c < 3; ++c) {
        _sigmaA[c] = _params.sigma_a[c] * _params.density;
        _sigmaS[c] = _params.sigma_s[c] * _params.density;
        _sigmaT[c] = _sigmaA[c] + _sigmaS[c];
    }
    int st = gpis_create(&_params, device, &_handle);
    if (st != GPIS_OK) {
        _handle = nullptr;
        throw std::runtime_error(std::string("gpis_create failed: ") + gpis_last_error());
    }
}
void HipSparseConvNoiseMedium::teardownAfterRender()
{
    if (_handle) gpis_destroy(_handle);
    _handle = nullptr;
}
void HipSparseConvNoiseMedium::requireHandle() const
{
    if (!_handle) throw std::runtime_error("HipSparseConvNoiseMedium: prepareForRender() has not been called");
}
Vec3f HipSparseConvNoiseMedium::sigmaA(Vec3f) const { Vec3f r; r.x = _sigmaA[0]; r.y = _sigmaA[1]; r.z = _sigmaA[2]; return r; }
Vec3f HipSparseConvNoiseMedium::sigmaS(Vec3f) const { Vec3f r; r.x = _sigmaS[0]; r.y = _sigmaS[1]; r.z = _sigmaS[2]; return r; }
Vec3f HipSparseConvNoiseMedium::sigmaT(Vec3f) const { Vec3f r; r.x = _sigmaT[0]; r.y = _sigmaT[1]; r.z = _sigmaT[2]; return r; }

void HipSparseConvNoiseMedium::fillRay(const Ray &ray, const MediumState &state, float u, gpis_ray_in &r)
{
    std::memset(&r, 0, sizeof r);
    r.pos[0] = ray.pos.x; r.pos[1] = ray.pos.y; r.pos[2] = ray.pos.z;
    r.dir[0] = ray.dir.x; r.dir[1] = ray.dir.y; r.dir[2] = ray.dir.z;
    r.near_t = ray.nearT; r.far_t = ray.farT;
    r.pixel[0] = state.info.pixelSampleSegment[0]; r.pixel[1] = state.info.pixelSampleSegment[1];
    r.spp = state.info.pixelSampleSegment[2]; r.segment = state.info.pixelSampleSegment[3];
    r.scene_seed = state.info.sceneSeed; r.info_t = state.info.t;
    r.u_jitter = u;
    r.first_scatter = state.firstScatter ? 1u : 0u;
    r.bounce = state.bounce;
    r.last_val = state.lastVal;
    r.last_gp_id = state.lastGPId;
    r.last_aniso[0] = state.lastAniso.x; r.last_aniso[1] = state.lastAniso.y; r.last_aniso[2] = state.lastAniso.z;
}

// The MediumState / MediumSample writes of GPM.cpp:224-340.
void HipSparseConvNoiseMedium::applyResult(const Ray &ray, const gpis_seg_out &o, const gpis_cond_coeff &c, MediumState &state, MediumSample &sample) const
{
    sample.emission = Vec3f();
    if (state.bounce >= _params.max_bounces)       // GPM.cpp:235-237: returns false, nothing written
        return;
    if (ray.farT == 0.f) {                         // GPM.cpp:239-248: returns true before any state update
        sample.t = 0.f;
        sample.weight.x = sample.weight.y = sample.weight.z = 1.f;
        sample.pdf = 1.0f;
        sample.exited = true;
        sample.p.x = o.p[0]; sample.p.y = o.p[1]; sample.p.z = o.p[2];
        sample.phase = 0;
        sample.sparseConv1DSamplingScheme = SparseConv1DSamplingScheme::UNI;
        return;
    }
    sample.aniso.x = o.aniso[0]; sample.aniso.y = o.aniso[1]; sample.aniso.z = o.aniso[2];
    sample.exited = o.exited != 0;
    auto ctxt = std::make_shared<GPContextSparseConvNoise>();
    ctxt->coeff = c;
    state.gpContext = ctxt;
    state.lastAniso = sample.aniso;
    state.lastVal = o.last_val;
    state.lastGPId = o.gp_id;
    if (!o.ok) {
        state.firstScatter = false;
        return;
    }
    sample.t = o.sample_t;
    sample.continuedT = o.continued_t;
    sample.weight.x = o.weight[0]; sample.weight.y = o.weight[1]; sample.weight.z = o.weight[2];
    sample.continuedWeight.x = o.continued_weight[0]; sample.continuedWeight.y = o.continued_weight[1]; sample.continuedWeight.z = o.continued_weight[2];
    sample.pdf = 1.0f;
    sample.p.x = o.p[0]; sample.p.y = o.p[1]; sample.p.z = o.p[2];
    sample.sparseConv1DSamplingScheme = (SparseConv1DSamplingScheme)o.scheme;
    if (!o.exited && _params.mean_emission.enabled) {          // sample.emission = emission(ro + rd*t), GPM.cpp:317
        double dl = std::sqrt((double)ray.dir.x * ray.dir.x + (double)ray.dir.y * ray.dir.y + (double)ray.dir.z * ray.dir.z);
        double p3[3] = {ray.pos.x + ray.dir.x / dl * o.t, ray.pos.y + ray.dir.y / dl * o.t, ray.pos.z + ray.dir.z / dl * o.t};
        float e[3] = {0, 0, 0};
        if (gpis_mean_color_emission_host(_handle, 1, p3, nullptr, e) != GPIS_OK) throw std::runtime_error(std::string("gpis_mean_color_emission_host: ") + gpis_last_error());
        sample.emission.x = e[0]; sample.emission.y = e[1]; sample.emission.z = e[2];
    }
    const bool absorption = _sigmaS[0] == 0.f && _sigmaS[1] == 0.f && _sigmaS[2] == 0.f;
    if (!absorption)                           // the absorption-only branch (GPM.cpp:250-258) does not advance;
        state.advance();
    else if (o.weight[0] == 0.f)               // its transmittance() clears firstScatter on a hit only (GPM.cpp:371-381)
        state.firstScatter = false;
    sample.phase = state.lastGPId;             // index into _phaseFunctions (GPM.cpp:335)
    sample.gpId = state.lastGPId;
    sample.ctxt = state.gpContext.get();
    state.info.t += sample.t;
    sample.rayInfo = state.info;
}

bool HipSparseConvNoiseMedium::sampleDistance(PathSampleGenerator &sampler, const Ray &ray, MediumState &state, MediumSample &sample) const
{
    requireHandle();
    gpis_ray_in r;
    // the path consumes exactly one next1D() per intersectGP (SCNM.cpp:129) — and none when
    // sampleDistance returns before marching (bounce limit, maxT == 0)
    const bool marches = state.bounce < _params.max_bounces && ray.farT != 0.f;
    fillRay(ray, state, marches ? sampler.next1D() : 0.f, r);
    gpis_seg_out o;
    gpis_cond_coeff c;
    int st = gpis_sample_distance_host(_handle, 1, &r, &o, &c);
    if (st != GPIS_OK) throw std::runtime_error(std::string("gpis_sample_distance_host: ") + gpis_last_error());
    applyResult(ray, o, c, state, sample);
    return o.ok != 0;
}

Vec3f HipSparseConvNoiseMedium::transmittance(PathSampleGenerator &sampler, const Ray &ray, bool, bool, MediumState *state) const
{
    requireHandle();
    if (!state) throw std::runtime_error("transmittance: the GPIS media need a MediumState (TraceBase.cpp:79-85)");
    gpis_ray_in r;
    fillRay(ray, *state, sampler.next1D(), r);
    uint8_t vis = 0;
    int st = gpis_transmittance_host(_handle, 1, &r, &vis);
    if (st != GPIS_OK) throw std::runtime_error(std::string("gpis_transmittance_host: ") + gpis_last_error());
    if (!vis) state->firstScatter = false;   // cleared on a hit only (GPM.cpp:371-381); lastAniso is not returned: shadow rays work on a copy of the state
    Vec3f out;
    out.x = out.y = out.z = vis ? 1.f : 0.f;
    return out;
}

void HipSparseConvNoiseMedium::sampleDistanceBatch(size_t n, const float *u, const Ray *rays, MediumState *states, MediumSample *samples, uint8_t *ok) const
{
    requireHandle();
    std::vector<gpis_ray_in> in(n);
    std::vector<gpis_seg_out> out(n);
    std::vector<gpis_cond_coeff> co(n);
    for (size_t i = 0; i < n; ++i) fillRay(rays[i], states[i], u[i], in[i]);
    int st = gpis_sample_distance_host(_handle, n, in.data(), out.data(), co.data());
    if (st != GPIS_OK) throw std::runtime_error(std::string("gpis_sample_distance_host: ") + gpis_last_error());
    for (size_t i = 0; i < n; ++i) {
        applyResult(rays[i], out[i], co[i], states[i], samples[i]);
        if (ok) ok[i] = out[i].ok ? 1 : 0;
    }
}
void HipSparseConvNoiseMedium::transmittanceBatch(size_t n, const float *u, const Ray *rays, const MediumState *states, uint8_t *visible) const
{
    requireHandle();
    std::vector<gpis_ray_in> in(n);
    for (size_t i = 0; i < n; ++i) fillRay(rays[i], states[i], u[i], in[i]);
    int st = gpis_transmittance_host(_handle, n, in.data(), visible);
    if (st != GPIS_OK) throw std::runtime_error(std::string("gpis_transmittance_host: ") + gpis_last_error());
}

static void fillNee(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info,
                    const GPContextSparseConvNoise &ctxt, gpis_nee_query &q)
{
    std::memset(&q, 0, sizeof q);
    q.ray_dir[0] = rayDir.x; q.ray_dir[1] = rayDir.y; q.ray_dir[2] = rayDir.z;
    q.normal[0] = normal.x; q.normal[1] = normal.y; q.normal[2] = normal.z;
    q.p[0] = p.x; q.p[1] = p.y; q.p[2] = p.z;
    q.t_segment = tSegment; q.info_t = info.t;
    q.pixel[0] = info.pixelSampleSegment[0]; q.pixel[1] = info.pixelSampleSegment[1];
    q.spp = info.pixelSampleSegment[2]; q.segment = info.pixelSampleSegment[3];
    q.scene_seed = info.sceneSeed;
    q.coeff = ctxt.coeff;
}
float HipSparseConvNoiseMedium::neePDF(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, float tSegment, const RayInfo &info,
                                       const GPContextSparseConvNoise &ctxt) const
{
    requireHandle();
    gpis_nee_query q;
    fillNee(rayDir, normal, p, tSegment, info, ctxt, q);
    float pdf = 0.f;
    if (gpis_nee_pdf_host(_handle, 1, &q, &pdf) != GPIS_OK) throw std::runtime_error(std::string("gpis_nee_pdf_host: ") + gpis_last_error());
    return pdf;
}
Vec3f HipSparseConvNoiseMedium::neeGrad(const Vec3f &rayDir, const Vec3f &normal, const Vec3f &p, const RayInfo &info,
                                        const GPContextSparseConvNoise &ctxt) const
{
    requireHandle();
    gpis_nee_query q;
    fillNee(rayDir, normal, p, 0.f, info, ctxt, q);
    float g[3] = {0, 0, 0};
    if (gpis_nee_grad_host(_handle, 1, &q, g) != GPIS_OK) throw std::runtime_error(std::string("gpis_nee_grad_host: ") + gpis_last_error());
    Vec3f r; r.x = g[0]; r.y = g[1]; r.z = g[2];
    return r;
}

}   // namespace gpis_host
