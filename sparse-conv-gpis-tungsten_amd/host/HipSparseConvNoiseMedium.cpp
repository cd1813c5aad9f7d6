// HipSparseConvNoiseMedium.cpp — see the header.  Plain C++17 (g++), links libgpis_hip.so.
#include "HipSparseConvNoiseMedium.hpp"
#include "gpis_json.hpp"

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

// accessor of include/gpis_json.hpp over this file's parser (the integration adapter has the same over Tungsten's JsonPtr)
struct HostJson {
    using Node = const JValue *;
    static bool child(const Node &o, const char *key, Node &out)
    {
        const JValue *v = o->get(key);
        if (!v) return false;
        out = v;
        return true;
    }
    template <typename T> static void num(const Node &o, const char *key, T &dst)
    {
        if (const JValue *v = o->get(key)) {
            if (v->kind == JValue::Number) dst = (T)v->num;
            else if (v->kind == JValue::Bool) dst = (T)(v->b ? 1 : 0);
        }
    }
    static void flag(const Node &o, const char *key, int32_t &dst) { num(o, key, dst); }
    static void str(const Node &o, const char *key, std::string &dst)
    {
        if (const JValue *v = o->get(key)) dst = v->str;
    }
    // Vec3f fields accept a scalar or a 3-array, as Tungsten's JsonPtr does for Vec3f
    template <typename T> static void vec3(const Node &o, const char *key, T *dst)
    {
        if (const JValue *v = o->get(key)) {
            if (v->kind == JValue::Number) dst[0] = dst[1] = dst[2] = (T)v->num;
            else if (v->kind == JValue::Array && v->arr.size() == 3)
                for (int i = 0; i < 3; ++i) dst[i] = (T)v->arr[i].num;
        }
    }
    static void vec3f(const Node &o, const char *key, float *dst) { vec3(o, key, dst); }
    static void vec3d(const Node &o, const char *key, double *dst) { vec3(o, key, dst); }
    static void mat3f(const Node &o, const char *key, float *dst9)
    {
        if (const JValue *m = o->get(key))
            if (m->kind == JValue::Array && m->arr.size() == 9)
                for (int i = 0; i < 9; ++i) dst9[i] = (float)m->arr[i].num;
    }
    [[noreturn]] static void fail(const std::string &what) { throw std::runtime_error(what); }
};

}   // namespace

// ---------------------------------------------------------------------------------------------
GPCorrelationContext HipSparseConvNoiseMedium::stringToCorrelationContext(const std::string &name)
{
    return (GPCorrelationContext)gpis_json::correlationContext<HostJson>(name);
}
SparseConv1DSamplingScheme HipSparseConvNoiseMedium::stringToSamplingScheme1D(const std::string &name)
{
    return (SparseConv1DSamplingScheme)gpis_json::samplingScheme1D<HostJson>(name);
}

HipSparseConvNoiseMedium::HipSparseConvNoiseMedium() { gpis_default_params(&_params); }
HipSparseConvNoiseMedium::~HipSparseConvNoiseMedium() { teardownAfterRender(); }

void HipSparseConvNoiseMedium::fromJson(const std::string &json)
{
    JParser parser(json);
    JValue v = parser.parse();
    if (v.kind != JValue::Object) throw std::runtime_error("medium JSON must be an object");
    HostJson::num(&v, "max_bounces", _params.max_bounces);      // Medium::fromJson (Medium.cpp:29-38)
    gpis_json::readMedium<HostJson>(&v, _params);               // the key table shared with the integration adapter (include/gpis_json.hpp)
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
