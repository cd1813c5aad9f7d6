// TungstenJsonAccess.hpp — the accessor include/gpis_json.hpp needs, over Tungsten's JsonPtr: both bindings of this directory
// (HipSparseConvNoiseMedium, HipFunctionSpaceMedium) read their JSON through the one key table the stand-alone adapter uses too.
#pragma once
#include "io/JsonObject.hpp"
#include "io/JsonPtr.hpp"
#include "math/Vec.hpp"

#include <Eigen/Dense>
#include <string>

#include "gpis_json.hpp"

namespace Tungsten {

// accessor of include/gpis_json.hpp (the JSON key -> gpis_params table shared with the stand-alone adapter) over Tungsten's JsonPtr
struct TungstenJson {
    using Node = JsonPtr;
    static bool child(const Node &o, const char *key, Node &out)
    {
        if (auto c = o[key]) { out = c; return true; }
        return false;
    }
    template <typename T> static void num(const Node &o, const char *key, T &dst) { o.getField(key, dst); }
    static void flag(const Node &o, const char *key, int32_t &dst)
    {
        bool b = dst != 0;
        o.getField(key, b);
        dst = b ? 1 : 0;
    }
    static void str(const Node &o, const char *key, std::string &dst) { o.getField(key, dst); }
    static void vec3f(const Node &o, const char *key, float *dst)
    {
        Vec3f v(dst[0], dst[1], dst[2]);
        o.getField(key, v);
        for (int i = 0; i < 3; ++i) dst[i] = v[i];
    }
    static void vec3d(const Node &o, const char *key, double *dst)
    {
        Vec3d v(dst[0], dst[1], dst[2]);
        o.getField(key, v);
        for (int i = 0; i < 3; ++i) dst[i] = v[i];
    }
    static void mat3f(const Node &o, const char *key, float *dst9)
    {
        if (auto mtx = o[key]) {
            Eigen::Matrix3f a;
            mtx.get(a);
            for (int r = 0; r < 3; ++r)
                for (int col = 0; col < 3; ++col)
                    dst9[3*r + col] = a(r, col);
        }
    }
    [[noreturn]] static void fail(const std::string &what) { FAIL("hip gpis medium: %s", what); }
};

}   // namespace Tungsten
