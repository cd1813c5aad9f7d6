// ref_probe.cpp — TEST INFRASTRUCTURE.  A thin extern "C" harness around the pieces of the
// REAL reference that compile in this image from the sources where they lie under
// /root/reference (header-only primitives + sampling/Gaussian.cpp + the vendored Eigen).
// Nothing of the reference is copied here: this file only #includes its headers and
// forwards calls.  Built by oracle/Makefile into oracle/_ref/libgpis_ref.so (git-ignored).
//
// The full noise evaluator (SparseConvolutionNoise.cpp / GPFunctions.cpp) is NOT buildable
// in this image: GPFunctions.hpp:24 includes <boost/math/special_functions/bessel.hpp> and
// GPFunctions.cpp:22 <fftw3.h>; neither Boost nor FFTW is installed and stand-ins are not
// allowed.  So this probe pins exactly what it can reach:
//   * MathUtil::xxhash32 x4            (src/core/math/MathUtil.hpp:179-224)
//   * UniformSampler (PCG32)           (src/core/sampling/UniformSampler.hpp:41-75)
//   * BitManip::normalizedUint         (src/core/math/BitManip.hpp:47-50)
//   * MathUtil::Bernoulli              (src/core/math/MathUtil.hpp:226-228)
//   * TangentFrame (Duff ONB)          (src/core/math/TangentFrame.hpp:63-107)
//   * rand_normal_2 / sample_standard_normal (src/core/sampling/Gaussian.cpp:21-34, 87-103)
//   * Vec3f length/normalized/dot      (src/core/math/Vec.hpp:134-206)
//   * Fresnel::conductorReflectance    (src/core/bsdfs/Fresnel.hpp:102-123)
//   * SampleWarp::powerHeuristic, uniformSphericalCapPdf (src/core/sampling/SampleWarp.hpp:131-134, 189-192)
//   * the Eigen expression FORMS the path uses (evaluation order of 3-term sums, 3x3
//     inverse), evaluated with the reference's vendored Eigen under the reference's flags.
#include "math/MathUtil.hpp"
#include "math/Vec.hpp"
#include "math/BitManip.hpp"
#include "math/TangentFrame.hpp"
#include "math/Angle.hpp"
#include "sampling/UniformSampler.hpp"
#include "sampling/Gaussian.hpp"
#include "sampling/SampleWarp.hpp"
#include "bsdfs/Fresnel.hpp"

#include <Eigen/Dense>
#include <cstdint>
#include <cstring>

using namespace Tungsten;

static inline Eigen::Matrix3f loadM(const float *m)
{
    Eigen::Matrix3f M;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            M(r, c) = m[3 * r + c];
    return M;
}
static inline void storeM(const Eigen::Matrix3f &M, float *m)
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            m[3 * r + c] = M(r, c);
}
static inline Eigen::Vector3f loadV(const float *v)
{
    Eigen::Vector3f e;
    e << v[0], v[1], v[2];
    return e;
}

extern "C" {

uint32_t ref_xxhash32_1(uint32_t a) { return MathUtil::xxhash32(uint(a)); }
uint32_t ref_xxhash32_2(uint32_t a, uint32_t b) { return MathUtil::xxhash32(Vec2u(a, b)); }
uint32_t ref_xxhash32_3(uint32_t a, uint32_t b, uint32_t c) { return MathUtil::xxhash32(Vec3u(a, b, c)); }
uint32_t ref_xxhash32_4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return MathUtil::xxhash32(Vec4u(a, b, c, d)); }

// raw nextI() stream after UniformSampler::set_state(state)
void ref_pcg32_stream(uint64_t state, uint32_t count, uint32_t *out)
{
    UniformSampler s;
    s.set_state(state);
    for (uint32_t i = 0; i < count; ++i)
        out[i] = s.nextI();
}

float ref_normalized_uint(uint32_t i) { return BitManip::normalizedUint(i); }

float ref_bernoulli(float rv) { return MathUtil::Bernoulli(rv, -1.f, 1.f, 0.5f); }

// What cell3D draws per impulse (SCN.cpp:384-385): next3D() then next1D(), k times.
// out = k * {x, y, z, u_w}.  Pins the argument-evaluation order of next3D() under g++.
void ref_cell3d_draws(uint64_t state, uint32_t k, float *out)
{
    UniformSampler s;
    s.set_state(state);
    for (uint32_t i = 0; i < k; ++i) {
        Vec3f p = Vec3f(s.next3D());
        float w = s.next1D();
        out[4 * i + 0] = p.x();
        out[4 * i + 1] = p.y();
        out[4 * i + 2] = p.z();
        out[4 * i + 3] = w;
    }
}

// TangentFrame(n): out = tangent(3), bitangent(3), normal(3)
void ref_tangent_frame(const float *n, float *out)
{
    TangentFrame tf(Vec3f(n[0], n[1], n[2]));
    for (int i = 0; i < 3; ++i) {
        out[i] = tf.tangent[i];
        out[3 + i] = tf.bitangent[i];
        out[6 + i] = tf.normal[i];
    }
}
void ref_frame_to_local(const float *n, const float *p, float *out)
{
    TangentFrame tf(Vec3f(n[0], n[1], n[2]));
    Vec3f r = tf.toLocal(Vec3f(p[0], p[1], p[2]));
    out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
void ref_frame_to_global(const float *n, const float *p, float *out)
{
    TangentFrame tf(Vec3f(n[0], n[1], n[2]));
    Vec3f r = tf.toGlobal(Vec3f(p[0], p[1], p[2]));
    out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}

void ref_vec3_normalized(const float *v, float *out)
{
    Vec3f r = Vec3f(v[0], v[1], v[2]).normalized();
    out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
float ref_vec3_length_sq(const float *v) { return Vec3f(v[0], v[1], v[2]).lengthSq(); }
float ref_vec3_dot(const float *a, const float *b) { return Vec3f(a[0], a[1], a[2]).dot(Vec3f(b[0], b[1], b[2])); }
void ref_vec3d_normalized(const double *v, double *out)
{
    Vec3d r = Vec3d(v[0], v[1], v[2]).normalized();
    out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}
double ref_vec3d_length(const double *v) { return Vec3d(v[0], v[1], v[2]).length(); }

// sample_standard_normal(n, sampler) after set_state(state); `draws` consecutive calls of size 2
// (SCN.cpp:151-152, 163, 180).  out = draws*2 doubles.
void ref_sample_standard_normal2(uint64_t state, uint32_t draws, double *out)
{
    UniformSampler s;
    s.set_state(state);
    for (uint32_t d = 0; d < draws; ++d) {
        Eigen::VectorXd v = sample_standard_normal(2, s);
        out[2 * d] = v(0);
        out[2 * d + 1] = v(1);
    }
}
// `sample_standard_normal(2, sampler) / sqrt(2.f)` as written at SCN.cpp:152.
void ref_sample_xy_over_sqrt2(uint64_t state, double *out)
{
    UniformSampler s;
    s.set_state(state);
    Eigen::VectorXd xy = sample_standard_normal(2, s) / sqrt(2.f);
    out[0] = xy(0);
    out[1] = xy(1);
}

// ---- Eigen expression forms -------------------------------------------------------------
// dist2_ab (GPF.hpp:75-78): ab.transpose() * aniso * ab
float ref_eig_dist2_ab(const float *ab, const float *m)
{
    Eigen::Vector3f v = loadV(ab);
    Eigen::Matrix3f M = loadM(m);
    float r = v.transpose() * M * v;
    return r;
}
// to_eigen3f(ab).dot(invCovMtx.col(c))  (GPF.cpp:813-815)
float ref_eig_dot_col(const float *ab, const float *m, int c)
{
    Eigen::Vector3f v = loadV(ab);
    Eigen::Matrix3f M = loadM(m);
    return v.dot(M.col(c));
}
// to_vec3f(M * to_eigen3f(v) / s)  (GPF.cpp:632, 641)
void ref_eig_matvec_div(const float *m, const float *v, float s, float *out)
{
    Eigen::Matrix3f M = loadM(m);
    Eigen::Vector3f r = M * loadV(v) / s;
    out[0] = r(0); out[1] = r(1); out[2] = r(2);
}
// to_vec3f(M * to_eigen3f(v) * s)  (GPF.cpp:636, 639)
void ref_eig_matvec_mul(const float *m, const float *v, float s, float *out)
{
    Eigen::Matrix3f M = loadM(m);
    Eigen::Vector3f r = M * loadV(v) * s;
    out[0] = r(0); out[1] = r(1); out[2] = r(2);
}
// to_vec3f(M * to_eigen3f(v))  (SCN.cpp:491, 501)
void ref_eig_matvec(const float *m, const float *v, float *out)
{
    Eigen::Matrix3f M = loadM(m);
    Eigen::Vector3f r = M * loadV(v);
    out[0] = r(0); out[1] = r(1); out[2] = r(2);
}
void ref_eig_inverse3(const float *m, float *out)
{
    Eigen::Matrix3f M = loadM(m);
    Eigen::Matrix3f I = M.inverse();
    storeM(I, out);
}
// sparseConvNoiseOneOverSecondDerivative tail (GPF.cpp:736-738)
void ref_eig_second_deriv_inv(const float *m, float *out)
{
    Eigen::Matrix3f invCovMtx = loadM(m);
    invCovMtx *= 0.25f;
    Eigen::Matrix3f covKernelSecondGrad = -2.f * invCovMtx;
    Eigen::Matrix3f r = covKernelSecondGrad.inverse();
    storeM(r, out);
}
// amplitude * A * B  (SCN.cpp:618)
void ref_eig_scaled_matmul(float amplitude, const float *a, const float *b, float *out)
{
    Eigen::Matrix3f A = loadM(a), B = loadM(b);
    Eigen::Matrix3f r = amplitude * A * B;
    storeM(r, out);
}
// W.transpose() * W and its determinant (GPF.cpp:672-674)
void ref_eig_gram(const float *w, float *out, float *det)
{
    Eigen::Matrix3f W = loadM(w);
    Eigen::Matrix3f G = W.transpose() * W;
    storeM(G, out);
    *det = G.determinant();
}
// l_conv * aniso_mtx and its inverse (GPF.cpp:670-671)
void ref_eig_scale_and_inverse(float l_conv, const float *a, float *l2w, float *w2l)
{
    Eigen::Matrix3f A = loadM(a);
    Eigen::Matrix3f L = l_conv * A;
    Eigen::Matrix3f W = L.inverse();
    storeM(L, l2w);
    storeM(W, w2l);
}
// getInvCovMtx scalings on a general matrix (GPF.cpp:794-800): M /= sqr(g); [*0.5]; /= sqr(l); *= 0.5
void ref_eig_invcov_scale(const float *m, float globalScale, float localScale, int isCov, float *out)
{
    Eigen::Matrix3f M = loadM(m);
    M /= sqr(globalScale);
    if (isCov)
        M *= 0.5f;
    M /= sqr(localScale);
    M *= 0.5f;
    storeM(M, out);
}
float ref_pi_float(void) { return PI; }
float ref_conductor_reflectance(float eta, float k, float cosThetaI) { return Fresnel::conductorReflectance(eta, k, cosThetaI); }
float ref_power_heuristic(float pdf0, float pdf1) { return SampleWarp::powerHeuristic(pdf0, pdf1); }
float ref_uniform_spherical_cap_pdf(float cosThetaMax) { return SampleWarp::uniformSphericalCapPdf(cosThetaMax); }
float ref_inv_pi_float(void) { return INV_PI; }

} // extern "C"

// =========================================================================================
// Round 3: the function-space path's linear-algebra / sampling layer and the fbm noise,
// straight from the reference's own sources and its vendored Eigen (no stand-ins):
//   * MultivariateNormalDistribution ctor + sample()   (src/core/sampling/Gaussian.cpp:121-167, 179-232)
//     with sampling/UniformPathSampler.hpp as the PathSampleGenerator
//   * Eigen::SelfAdjointEigenSolver<MatrixXd> (compute and computeFromTridiagonal), Eigen::LLT<MatrixXd>
//     (src/thirdparty/eigen/Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h, Cholesky/LLT.h)
//   * the expression forms of pseudo_inverse (src/core/math/GaussianProcess.cpp:645-662; that TU itself
//     includes GPFunctions.hpp -> Boost and cannot be built, so only its Eigen expressions are evaluated here)
//   * fbm / simplex3d / random3 (src/core/math/SdfFunctions.cpp:199-296; linked from that TU)
// =========================================================================================
#include "sampling/UniformPathSampler.hpp"
#include "math/SdfFunctions.hpp"
#include <cfloat>

namespace Tungsten {
float simplex3d(Vec3f p);      // defined (external linkage) in math/SdfFunctions.cpp:228
Vec3f random3(Vec3f c);        // math/SdfFunctions.cpp:199
}

static inline Eigen::MatrixXd loadMd(int n, int m, const double *a)
{
    return Eigen::Map<const Eigen::MatrixXd>(a, n, m);      // column-major
}

extern "C" {

// SelfAdjointEigenSolver<MatrixXd>(A): eigenvectors (column-major) and eigenvalues; returns info()
int ref_fs_eigh(int n, const double *A, double *evecs, double *evals)
{
    Eigen::MatrixXd M = loadMd(n, n, A);
    Eigen::SelfAdjointEigenSolver<Eigen::MatrixXd> eigs(M);
    Eigen::Map<Eigen::MatrixXd>(evecs, n, n) = eigs.eigenvectors();
    Eigen::Map<Eigen::VectorXd>(evals, n) = eigs.eigenvalues();
    return (int)eigs.info();
}
// the implicit symmetric QR iteration alone, on a given tridiagonal matrix (eigenvector accumulation starts from I)
int ref_fs_eigh_tridiagonal(int n, const double *diag, const double *sub, double *evecs, double *evals)
{
    Eigen::VectorXd d = Eigen::Map<const Eigen::VectorXd>(diag, n);
    Eigen::VectorXd s = Eigen::Map<const Eigen::VectorXd>(sub, n - 1);
    Eigen::SelfAdjointEigenSolver<Eigen::MatrixXd> eigs;
    eigs.computeFromTridiagonal(d, s, Eigen::ComputeEigenvectors);
    Eigen::Map<Eigen::MatrixXd>(evecs, n, n) = eigs.eigenvectors();
    Eigen::Map<Eigen::VectorXd>(evals, n) = eigs.eigenvalues();
    return (int)eigs.info();
}
// Tridiagonalization of the solver's first phase: diag, sub-diagonal and Q of internal::tridiagonalization_inplace
// on the scaled lower triangle, as SelfAdjointEigenSolver::compute runs it (SelfAdjointEigenSolver.h:441-452)
void ref_fs_tridiagonalize(int n, const double *A, double *diag, double *sub, double *Q, double *scale_out)
{
    Eigen::MatrixXd mat = loadMd(n, n, A).triangularView<Eigen::Lower>();
    double scale = mat.cwiseAbs().maxCoeff();
    if (scale == 0.0) scale = 1.0;
    mat.triangularView<Eigen::Lower>() /= scale;
    Eigen::VectorXd d(n), s(n - 1);
    Eigen::VectorXd hCoeffs(n - 1);
    Eigen::VectorXd workspace(n);
    Eigen::internal::tridiagonalization_inplace(mat, d, s, hCoeffs, workspace, true);
    Eigen::Map<Eigen::VectorXd>(diag, n) = d;
    Eigen::Map<Eigen::VectorXd>(sub, n - 1) = s;
    Eigen::Map<Eigen::MatrixXd>(Q, n, n) = mat;
    *scale_out = scale;
}
// LLT<MatrixXd>(S.triangularView<Lower>()): matrixL (dense, column-major); returns info()
int ref_fs_llt(int n, const double *S, double *L)
{
    Eigen::MatrixXd M = loadMd(n, n, S);
    Eigen::LLT<Eigen::MatrixXd> chol(M.triangularView<Eigen::Lower>());
    Eigen::MatrixXd l = chol.matrixL();
    Eigen::Map<Eigen::MatrixXd>(L, n, n) = l;
    return (int)chol.info();
}
// MultivariateNormalDistribution(mean, cov).normTransform
void ref_mvn_norm_transform(int n, const double *mean, const double *cov, double *T)
{
    Eigen::VectorXd mu = Eigen::Map<const Eigen::VectorXd>(mean, n);
    Eigen::MatrixXd S = loadMd(n, n, cov);
    MultivariateNormalDistribution mvn(mu, S);
    Eigen::Map<Eigen::MatrixXd>(T, n, n) = mvn.normTransform;
}
// MultivariateNormalDistribution(mean, cov).sample(constraints, numConstraints, samples, UniformPathSampler(UniformSampler after set_state))
// constraints: numConstraints x {startIdx, endIdx} ints + {minV, maxV} floats.  out: n x samples column-major.
// state_out: the sampler's state afterwards (how many draws were consumed).
void ref_mvn_sample(int n, const double *mean, const double *cov, uint64_t state, int numConstraints, const int *cidx, const float *cminmax,
                    int samples, double *out, uint64_t *state_out)
{
    Eigen::VectorXd mu = Eigen::Map<const Eigen::VectorXd>(mean, n);
    Eigen::MatrixXd S = loadMd(n, n, cov);
    MultivariateNormalDistribution mvn(mu, S);
    UniformSampler us;
    us.set_state(state);
    UniformPathSampler ps(us);
    std::vector<Constraint> cs(numConstraints);
    for (int i = 0; i < numConstraints; ++i)
        cs[i] = Constraint{cidx[2 * i], cidx[2 * i + 1], cminmax[2 * i], cminmax[2 * i + 1]};
    Eigen::MatrixXd r = mvn.sample(cs.data(), numConstraints, samples, ps);
    Eigen::Map<Eigen::MatrixXd>(out, n, samples) = r;
    if (state_out) *state_out = ps.uniformGenerator().state();
}
// rand_truncated_normal(mean, sigma, a, sampler)  (Gaussian.cpp:57-85)
double ref_rand_truncated_normal(double mean, double sigma, double a, uint64_t state, uint64_t *state_out)
{
    UniformSampler us;
    us.set_state(state);
    UniformPathSampler ps(us);
    double r = rand_truncated_normal(mean, sigma, a, ps);
    if (state_out) *state_out = ps.uniformGenerator().state();
    return r;
}
// the Eigen expressions of pseudo_inverse (GaussianProcess.cpp:645-662), written as there
void ref_fs_pinv_forms(int n, const double *A, double *out)
{
    Eigen::MatrixXd a = loadMd(n, n, A);
    Eigen::SelfAdjointEigenSolver<Eigen::MatrixXd> eigs(a);
    double eps = 1e6 * DBL_EPSILON * eigs.eigenvalues().cwiseAbs().maxCoeff();
    Eigen::VectorXd v = eigs.eigenvalues();
    Eigen::VectorXd s_pinv = v.cwiseAbs().cwiseLessOrEqual(eps).select(0., v.cwiseInverse());
    Eigen::MatrixXd U = eigs.eigenvectors() * s_pinv.cwiseSqrt().asDiagonal();
    Eigen::MatrixXd r = U * U.transpose();
    Eigen::Map<Eigen::MatrixXd>(out, n, n) = r;
}
// create_mvn_cond's dense products, as written at GaussianProcess.cpp:722-735:
//   solved = (pinv * s12).transpose();  mean += solved * resid;  cov = s22 - solved * s12
void ref_fs_cond_forms(int nc, int n, const double *pinv, const double *s12, const double *s22, const double *resid, const double *mean,
                       double *mean_out, double *cov_out)
{
    Eigen::MatrixXd P = loadMd(nc, nc, pinv), S12 = loadMd(nc, n, s12), S22 = loadMd(n, n, s22);
    Eigen::VectorXd r = Eigen::Map<const Eigen::VectorXd>(resid, nc), mu = Eigen::Map<const Eigen::VectorXd>(mean, n);
    Eigen::MatrixXd solved = (P * S12).transpose();
    Eigen::VectorXd m2 = mu + (solved * r);
    Eigen::MatrixXd c2 = S22 - (solved * S12);
    Eigen::Map<Eigen::VectorXd>(mean_out, n) = m2;
    Eigen::Map<Eigen::MatrixXd>(cov_out, n, n) = c2;
}

double ref_fbm(const double *uv, int octaves) { return fbm(Vec3d(uv[0], uv[1], uv[2]), octaves); }
float ref_simplex3d(const float *p) { return simplex3d(Vec3f(p[0], p[1], p[2])); }
void ref_random3(const float *c, float *out)
{
    Vec3f r = random3(Vec3f(c[0], c[1], c[2]));
    out[0] = r.x(); out[1] = r.y(); out[2] = r.z();
}

} // extern "C"
