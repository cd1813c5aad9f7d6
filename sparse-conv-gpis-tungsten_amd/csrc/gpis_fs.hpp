// gpis_fs.hpp — the function-space comparison path: one wave per segment, four segments per CU.
//
//   FunctionSpaceGaussianProcessMedium::intersectGP / sampleGradient   media/FunctionSpaceGaussianProcessMedium.cpp:58-345
//   GPRealNodeValues::{makeIntersect, sampleGrad, applyMemory}          math/GaussianProcess.cpp:76-168
//   GaussianProcess::{mean_and_cov, cov_sym, cov, sample_start_value, sample_cond, create_mvn_cond}, pseudo_inverse
//                                                                       math/GaussianProcess.cpp:314-753
//   MultivariateNormalDistribution, rand_truncated_normal               sampling/Gaussian.cpp:57-232
//   Eigen::SelfAdjointEigenSolver / Eigen::LLT as the reference's Eigen implements them (vendored sources:
//   Eigenvalues/{SelfAdjointEigenSolver,Tridiagonalization}.h, Householder/*.h, Jacobi/Jacobi.h, Cholesky/LLT.h)
//
// One segment = (64 + c)^2 doubles of dense linear algebra (c <= 66 conditioning entries): covariance build, a symmetric
// eigen-decomposition for the pseudo-inverse, two matrix products, a Cholesky factorisation (or a second eigen-decomposition
// when it fails — the rule for squared-exponential covariances on a fine grid), 64 normal variates, in fp64.  90 % of the time
// is the eigen-solver's serial phases (Householder tridiagonalisation, implicit-QR bulge chase: ~650 cycles of dependent fp64
// division / square root per rotation), so the matrix being factorised and the solver's vectors sit in LDS (37 KB) and FOUR
// one-wave workgroups — one per SIMD — share a CU; the matrices that are only touched in bulk live in a per-workgroup slice of
// a global workspace (L2).  Lane = matrix row, inner sums run serially in ascending index order inside each lane, so every value
// is produced by the same sequence of IEEE operations as in the CPU restatement (exp / log / sin / cos excepted).  No MFMA:
// the products are 66 x 66 x 64 fp64 (< 10 % of the time) between data-dependent factorisations.
#pragma once
#include "gpis_device.hpp"

#pragma clang fp contract(off)

namespace gpis {

constexpr int kFsN = GPIS_FS_MAX_POINTS, kFsC = GPIS_FS_MAX_CTX;
constexpr int FS_NONE = 0, FS_FIRST = 1;

// The matrix being factorised and the vectors of the eigen-solver live in LDS (37 KB: four workgroups — one wave per SIMD — per
// CU); everything that is only touched in bulk (the other three matrices, point lists, sample vectors) lives in a per-workgroup
// slice of a global workspace that stays in L2.  The serial phases (tridiagonalisation, bulge chase: 90 % of the time) only see LDS.
struct FsLds {
    double B1[kFsC * kFsC];      // s11 -> its eigenvectors -> its pseudo-inverse; later the normal transform T
    double w[kFsC], sub[kFsC], hco[kFsC], hv[kFsC], tmp[kFsC];
};
struct FsGlob {
    double B2[kFsC * kFsC];      // scaled eigenvectors; later (pinv * s12)
    double B3[kFsC * kFsN];      // s12
    double B4[kFsN * kFsN];      // s22 -> conditional covariance
    double mean[kFsC], z[kFsC], vals[kFsC], cv[kFsC], resid[kFsC], ts[kFsN];
    double pts[kFsC][3], cpts[kFsC][3], dirs[3][3];
    int der[kFsC], cder[kFsC];
};
static_assert(sizeof(FsLds) <= 40 * 1024, "four function-space workgroups share the LDS of one CU");

#define FS_SYNC() __syncthreads()

// -DGPIS_FS_PROF (diagnostic builds only): wave cycles per phase, read back with gpis_fs_prof_read
#ifdef GPIS_FS_PROF
__device__ unsigned long long g_fs_prof[16];
#define FS_T0() const long long fs_t0_ = (long long)__builtin_readcyclecounter()
#define FS_T(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_fs_prof[i], (unsigned long long)((long long)__builtin_readcyclecounter() - fs_t0_)); } while (0)
#else
#define FS_T0() do { } while (0)
#define FS_T(i) do { } while (0)
#endif

GPIS_DEV double fs_dot(V3d a, V3d b) { double r = a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
GPIS_DEV V3d fs_v3(const double *p) { return V3d{p[0], p[1], p[2]}; }

// CovarianceFunction::operator()(Derivative, Derivative, ...), GPF.hpp:1145-1158, squared exponential in its GP form
// (GPF.hpp:86-89, 1602-1605; GPF.cpp:770-772); the derivatives are the closed forms of what autodiff returns (GPF.cpp:490-504)
GPIS_DEV double fs_cov(const DevModel &M, int da, int db, V3d a, V3d b, V3d dirA, V3d dirB)
{
    const V3d an{(double)M.fs_aniso[0], (double)M.fs_aniso[1], (double)M.fs_aniso[2]};
    const V3d d{b.x - a.x, b.y - a.y, b.z - a.z};
    const V3d ad{an.x * d.x, an.y * d.y, an.z * d.z};
    const double absq = fs_dot(d, ad);
    const float s2 = M.sigma_raw * M.sigma_raw, l2 = M.k_l * M.k_l;
    const double c = (double)s2 * exp_glibc(-absq / (double)(2 * l2));
    if (da == FS_NONE && db == FS_NONE)
        return c;
    if (da == FS_FIRST && db == FS_NONE)
        return c * (fs_dot(ad, dirA) / (double)l2);
    if (da == FS_NONE && db == FS_FIRST)
        return c * (-fs_dot(ad, dirB) / (double)l2);
    const V3d adB{an.x * dirB.x, an.y * dirB.y, an.z * dirB.z};
    return c * (fs_dot(dirA, adB) / (double)l2) - c * (fs_dot(ad, dirA) / (double)l2) * (fs_dot(ad, dirB) / (double)l2);
}
// MeanFunction::operator()(Derivative, p, d), GPF.hpp:829-836
GPIS_DEV double fs_mean(const DevModel &M, int deriv, V3d p, V3d dir)
{
    if (deriv == FS_NONE)
        return mean_eval(M, 0, p);
    return fs_dot(dir, mean_grad(M, 0, p));
}
GPIS_DEV double fs_wave_max(double v)
{
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// ---- the reduction orders of the reference's Eigen build (vendored Eigen 3.4.90, SSE2 packets of two doubles, no FMA).  A
// vectorised sum is not the ascending-index sum; these helpers add in the order Eigen's kernels add (restated and pinned bit for
// bit against the reference's Eigen in oracle/gpis_fs_oracle.inc: eig_redux_prod, eig_gemv_row_dot, eig_symv_lower,
// eig_gemm_dot, eig_llt), so the factorisations below produce the reference's bits, not merely its values.
// DenseBase::redux on an expression without direct access (squaredNorm, dot), Core/Redux.h:321-375 with alignedStart = 0
GPIS_DEV double fs_redux_prod(const double *a, const double *b, int n)
{
    const int n2 = (n / 2) * 2, n4 = (n / 4) * 4;
    double res;
    if (n2) {
        double p0a = a[0] * b[0], p0b = a[1] * b[1];
        if (n2 > 2) {
            double p1a = a[2] * b[2], p1b = a[3] * b[3];
            for (int i = 4; i < n4; i += 4) {
                p0a += a[i] * b[i]; p0b += a[i + 1] * b[i + 1];
                p1a += a[i + 2] * b[i + 2]; p1b += a[i + 3] * b[i + 3];
            }
            p0a += p1a; p0b += p1b;
            if (n2 > n4) { p0a += a[n4] * b[n4]; p0b += a[n4 + 1] * b[n4 + 1]; }
        }
        res = p0a + p0b;
        for (int i = n2; i < n; ++i) res += a[i] * b[i];
    } else {
        res = a[0] * b[0];
        for (int i = 1; i < n; ++i) res += a[i] * b[i];
    }
    return res;
}
// general_matrix_vector_product, RowMajor (GeneralMatrixVector.h:330-520): one packet accumulator over the pairs, its lanes, the tail
GPIS_DEV double fs_gemv_row_dot(const double *a, const double *b, int n)
{
    const int n2 = (n / 2) * 2;
    double c0 = 0.0, c1 = 0.0;
    for (int j = 0; j < n2; j += 2) { c0 = a[j] * b[j] + c0; c1 = a[j + 1] * b[j + 1] + c1; }
    double cc = c0 + c1;
    for (int j = n2; j < n; ++j) cc += a[j] * b[j];
    return cc;
}
// gebp_kernel<double, mr = 4, nr = 4>: entry (i, j) of an R x C product of depth K — one accumulator over ascending k, except in
// the "one packet" rows of the last half-filled row block and full 4-column panels, where even and odd k accumulate separately
// over the peeled depth (GeneralBlockPanelKernel.h:1360-1445).  a / b: the K factors with strides sa / sb.
GPIS_DEV double fs_gemm_dot(const double *a, int sa, const double *b, int sb, int K, int i, int j, int R, int C, bool kernel_call)
{
    const int r4 = (R / 4) * 4, r2 = r4 + ((R % 4) / 2) * 2, c4 = (C / 4) * 4;
    if (i >= r4 && i < r2 && j < c4 && (kernel_call || R + C + K >= 20)) {
        const int pk = (K / 8) * 8;
        double c = 0.0, d = 0.0;
        for (int k = 0; k < pk; k += 2) { c = a[k * sa] * b[k * sb] + c; d = a[(k + 1) * sa] * b[(k + 1) * sb] + d; }
        c = c + d;
        for (int k = pk; k < K; ++k) c = a[k * sa] * b[k * sb] + c;
        return c;
    }
    double acc = 0.0;
    #pragma unroll 8
    for (int k = 0; k < K; ++k) acc += a[k * sa] * b[k * sb];
    return acc;
}
// selfadjoint_matrix_vector_product<double, ColMajor, Lower> (SelfadjointMatrixVector.h:43-154), row r of res = alpha * B * v for
// the size x size symmetric matrix whose lower triangle is B (column-major, leading dimension lda), res zero before the call:
// two columns per pass, packet loops whose peeling depends on the 16-byte alignment of res (res_odd).  The kernel adds the terms
// of columns j < r to res[r] pair by pair, then the diagonal, and LAST alpha x (the column sum over i > r, itself split into a
// scalar part and two packet lanes); the last <= 9 columns go one at a time.
GPIS_DEV double fs_symv_row(const double *B, int lda, const double *v, int size, double alpha, int res_odd, int r)
{
    const int bound = (size - 8 > 0 ? size - 8 : 0) & ~1;
    const int aE = size - ((size - res_odd) & 1);
    double res = 0.0;
    const int npair = r < bound ? (r & ~1) : bound;
    for (int c = 0; c < npair; c += 2) {
        const double t0 = alpha * v[c], t1 = alpha * v[c + 1];
        const double a0 = B[c * lda + r], a1 = B[(c + 1) * lda + r];
        if (r >= c + 2 + res_odd && r < aE) res = a0 * t0 + (a1 * t1 + res);
        else res += a0 * t0 + a1 * t1;
    }
    const double *col = B + r * lda;
    if (r < bound) {
        const int jr = r & ~1;
        const double t0 = alpha * v[jr], t1 = alpha * v[jr + 1];
        if (r == jr) {
            res += col[r] * t0;
        } else {
            res += col[r] * t1;
            res += B[jr * lda + r] * t0;
        }
        const int aS = jr + 2 + res_odd;
        double t = 0.0, pa = 0.0, pb = 0.0;
        int i = r + 1;
        for (; i < aS; ++i) t += col[i] * v[i];
        for (; i < aE; i += 2) { pa = col[i] * v[i] + pa; pb = col[i + 1] * v[i + 1] + pb; }
        for (; i < size; ++i) t += col[i] * v[i];
        res += alpha * (t + (pa + pb));
    } else {
        for (int j = bound; j < r; ++j) res += B[j * lda + r] * (alpha * v[j]);
        res += col[r] * (alpha * v[r]);
        double t2 = 0.0;
        for (int i = r + 1; i < size; ++i) t2 += col[i] * v[i];
        res += alpha * t2;
    }
    return res;
}

// Eigen::SelfAdjointEigenSolver<MatrixXd>::compute on the n x n column-major matrix A (lower triangle read): on return the
// columns of A are the eigenvectors and L.w the eigenvalues, ascending.  Every lane runs the scalar recurrences on the same
// data; lane 0 stores them.
GPIS_DEV void fs_eigh(FsLds &L, int n, double *A, int lane)
{
#define E(i, j) A[(j) * n + (i)]
    if (n == 1) {
        if (lane == 0) { L.w[0] = E(0, 0); E(0, 0) = 1.0; }
        FS_SYNC();
        return;
    }
    double scale = 0.0;
    for (int i = lane; i < n; i += 64)
        for (int j = 0; j <= i; ++j) { const double v = fabs(E(i, j)); if (v > scale) scale = v; }
    scale = fs_wave_max(scale);
    if (scale == 0.0) scale = 1.0;
    for (int i = lane; i < n; i += 64)
        for (int j = 0; j <= i; ++j) E(i, j) /= scale;
    FS_SYNC();
    { FS_T0();
    for (int i = 0; i < n - 1; ++i) {                       // tridiagonalization_inplace, Tridiagonalization.h:352-383
        const int rs = n - i - 1;
        const double tailSq = rs == 1 ? 0.0 : fs_redux_prod(&E(i + 2, i), &E(i + 2, i), rs - 1);     // tail.squaredNorm()
        const double c0 = E(i + 1, i);
        double h, beta;
        FS_SYNC();
        if (tailSq <= 2.2250738585072014e-308) {            // makeHouseholder, Householder.h:70-99
            h = 0.0; beta = c0;
            for (int k = i + 2 + lane; k < n; k += 64) E(k, i) = 0.0;
        } else {
            beta = sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.0) beta = -beta;
            for (int k = i + 2 + lane; k < n; k += 64) E(k, i) = E(k, i) / (c0 - beta);
            h = (beta - c0) / beta;
        }
        if (lane == 0) E(i + 1, i) = 1.0;
        FS_SYNC();
        // hCoeffs.tail(rs) = selfadjointView<Lower>(bottomRight) * (h v): h is the kernel's alpha, &hCoeffs[i] its destination
        for (int r = lane; r < rs; r += 64)
            L.hv[r] = fs_symv_row(&E(i + 1, i + 1), n, &E(i + 1, i), rs, h, i & 1, r);
        FS_SYNC();
        const double dotp = fs_redux_prod(L.hv, &E(i + 1, i), rs);
        const double alpha = h * -0.5 * dotp;
        FS_SYNC();
        for (int r = lane; r < rs; r += 64) L.hv[r] += alpha * E(i + 1 + r, i);
        FS_SYNC();
        for (int r = lane; r < rs; r += 64) {
            const double hr = L.hv[r], vr = E(i + 1 + r, i);
            #pragma unroll 8
            for (int c = 0; c <= r; ++c) {
                const double uc = -E(i + 1 + c, i), wc = -L.hv[c];
                E(i + 1 + r, i + 1 + c) += uc * hr + wc * vr;
            }
        }
        FS_SYNC();
        if (lane == 0) { E(i + 1, i) = beta; L.hco[i] = h; }
        FS_SYNC();
    }
    FS_T(0); }
    for (int k = lane; k < n; k += 64) {
        L.w[k] = E(k, k);
        if (k < n - 1) L.sub[k] = E(k + 1, k);
    }
    FS_SYNC();
    { FS_T0();
    // Q = HouseholderSequence(mat, hCoeffs).setLength(n - 1).setShift(1), evaluated in place (HouseholderSequence.h:293-318)
    for (int r = lane; r < n; r += 64) {
        E(r, r) = 1.0;
        for (int k = r + 1; k < n; ++k) E(r, k) = 0.0;
    }
    FS_SYNC();
    for (int k = n - 2; k >= 0; --k) {
        const int cs = n - k - 1, o = k + 1;
        const double tau = L.hco[k];
        if (cs == 1) {
            if (lane == 0) E(o, o) *= 1.0 - tau;
        } else if (tau != 0.0) {                            // applyHouseholderOnTheLeft, Householder.h:119-137
            for (int c = lane; c < cs; c += 64)            // tmp = essential.adjoint() * bottom (row-major gemv); tmp += row(0)
                L.tmp[c] = fs_gemv_row_dot(&E(o + 1, o + c), &E(k + 2, k), cs - 1) + E(o, o + c);
            FS_SYNC();
            for (int c = lane; c < cs; c += 64) E(o, o + c) -= tau * L.tmp[c];
            for (int r = 1 + lane; r < cs; r += 64) {
                const double te = tau * E(k + 1 + r, k);
                #pragma unroll 8
                for (int c = 0; c < cs; ++c) E(o + r, o + c) -= te * L.tmp[c];
            }
        }
        FS_SYNC();
        for (int r = k + 1 + lane; r < n; r += 64) E(r, k) = 0.0;
        FS_SYNC();
    }
    FS_T(1); }
    {   // computeFromTridiagonal_impl (m_maxIterations = 30), SelfAdjointEigenSolver.h:504-579, tridiagonal_qr_step :841-900
        FS_T0();
        int end = n - 1, start = 0, iter = 0;
        const double considerAsZero = 2.2250738585072014e-308, precision_inv = 1.0 / 2.220446049250313e-16;
        while (end > 0) {
            for (int i = start + lane; i < end; i += 64) {
                const double si = L.sub[i];
                if (fabs(si) < considerAsZero) {
                    L.sub[i] = 0.0;
                } else {
                    const double ss = precision_inv * si;
                    if (ss * ss <= (fabs(L.w[i]) + fabs(L.w[i + 1]))) L.sub[i] = 0.0;
                }
            }
            FS_SYNC();
            while (end > 0 && L.sub[end - 1] == 0.0) end--;
            if (end <= 0) break;
            iter++;
            if (iter > 30 * n) break;
            start = end - 1;
            while (start > 0 && L.sub[start - 1] != 0.0) start--;
            const double td = (L.w[end - 1] - L.w[end]) * 0.5, e = L.sub[end - 1];
            double mu = L.w[end];
            if (td == 0.0) {
                mu -= fabs(e);
            } else if (e != 0.0) {
                const double e2 = e * e;
                // numext::hypot = positive_real_hypot(|td|, |e|), MathFunctionsImpl.h:216-232
                const double ax = fabs(td), ay = fabs(e), p = ax > ay ? ax : ay, qp = (ay < ax ? ay : ax) / p;
                const double hh = p * sqrt(1.0 + qp * qp);
                if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? hh : -hh)) / e);
                else mu -= e2 / (td + (td > 0.0 ? hh : -hh));
            }
            // The sweep is one dependent chain (bulge chasing): what sits on it decides the time of the whole path.  The scalar
            // recurrence runs in registers (w[k], sub[k], sub[k-1] are the previous rotation's results; w[k+1], sub[k+1] are still
            // untouched in LDS and are fetched ONE ROTATION AHEAD), and each lane carries the element of its Q row that the next
            // rotation needs (column k+1 after rotation k) instead of writing it and reading it back; column k+2 is fetched ahead
            // as well.  The chain is then the two divisions and the square root of makeGivens, not LDS round trips.
            double wk = L.w[start], sk = L.sub[start], skm1 = 0.0;
            double x = wk - mu, z = sk;
            const int ia = lane, ib = lane + 64;
            const bool ha = ia < n, hb = ib < n;
            double qa = ha ? E(ia, start) : 0.0, qb = hb ? E(ib, start) : 0.0;                 // column k of the lane's rows
            double na = ha ? E(ia, start + 1) : 0.0, nb = hb ? E(ib, start + 1) : 0.0;         // column k + 1
            double pw = L.w[start + 1], ps = start < end - 1 ? L.sub[start + 1] : 0.0;
            FS_SYNC();
            int k = start;
            for (; k < end && z != 0.0; ++k) {
                const bool more = k + 2 <= end;
                const double fa = (more && ha) ? E(ia, k + 2) : 0.0, fb = (more && hb) ? E(ib, k + 2) : 0.0;
                const double fw = more ? L.w[k + 2] : 0.0, fsub = (k + 2 <= end - 1) ? L.sub[k + 2] : 0.0;
                double gc, gs;                              // JacobiRotation::makeGivens(x, z), Jacobi.h:234-270
                if (x == 0.0) { gc = 0.0; gs = z < 0.0 ? 1.0 : -1.0; }
                else if (fabs(x) > fabs(z)) { const double t = z / x; double u = sqrt(1.0 + t * t); if (x < 0.0) u = -u; gc = 1.0 / u; gs = -t * gc; }
                else { const double t = x / z; double u = sqrt(1.0 + t * t); if (z < 0.0) u = -u; gs = -1.0 / u; gc = -t * gs; }
                const double wk1 = pw;
                const double sdk = gs * wk + gc * sk;
                const double dkp1 = gs * sk + gc * wk1;
                const double nwk = gc * (gc * wk - gs * sk) - gs * (gc * sk - gs * wk1);
                const double nwk1 = gs * sdk + gc * dkp1;
                const double nsk = gc * sdk - gs * dkp1;
                double nskm1 = skm1;
                if (k > start) nskm1 = gc * skm1 - gs * z;
                x = nsk;
                double nsk1 = 0.0;
                if (k < end - 1) { const double sk1 = ps; z = -gs * sk1; nsk1 = gc * sk1; }
                if (lane == 0) {
                    L.w[k] = nwk; L.w[k + 1] = nwk1; L.sub[k] = nsk;
                    if (k > start) L.sub[k - 1] = nskm1;
                    if (k < end - 1) L.sub[k + 1] = nsk1;
                }
                // q.applyOnTheRight(k, k + 1, rot): column k is final, column k + 1 stays in the register
                if (ha) { E(ia, k) = gc * qa - gs * na; qa = gs * qa + gc * na; na = fa; }
                if (hb) { E(ib, k) = gc * qb - gs * nb; qb = gs * qb + gc * nb; nb = fb; }
                wk = nwk1; sk = nsk1; skm1 = nsk;
                pw = fw; ps = fsub;
            }
            if (ha) E(ia, k) = qa;
            if (hb) E(ib, k) = qb;
            FS_SYNC();
        }
        FS_T(2);
        if (iter <= 30 * n) {
            FS_T0();
            for (int i = 0; i < n - 1; ++i) {
                int k = 0;
                for (int j = 1; j < n - i; ++j) if (L.w[i + j] < L.w[i + k]) k = j;
                FS_SYNC();
                if (k > 0) {
                    if (lane == 0) { const double t = L.w[i]; L.w[i] = L.w[k + i]; L.w[k + i] = t; }
                    for (int r = lane; r < n; r += 64) { const double q = E(r, i); E(r, i) = E(r, k + i); E(r, k + i) = q; }
                }
                FS_SYNC();
            }
            FS_T(3);
        }
    }
    for (int k = lane; k < n; k += 64) L.w[k] *= scale;
    FS_SYNC();
#undef E
}

// pseudo_inverse, GaussianProcess.cpp:645-662: A (n x n in B1) is replaced by its pseudo-inverse; B2 is scratch
GPIS_DEV void fs_pinv(FsLds &L, FsGlob &G, int n, int lane)
{
    double *A = L.B1, *U = G.B2;
    fs_eigh(L, n, A, lane);
    double mx = 0.0;
    for (int i = 0; i < n; ++i) { const double v = fabs(L.w[i]); if (v > mx) mx = v; }
    const double eps = 1e6 * 2.220446049250313e-16 * mx;
    FS_T0();
    for (int i = lane; i < n; i += 64)
        for (int k = 0; k < n; ++k) {
            const double wk = L.w[k];
            const double sp = fabs(wk) <= eps ? 0.0 : 1.0 / wk;
            U[k * n + i] = A[k * n + i] * sqrt(sp);
        }
    FS_SYNC();
    for (int i = lane; i < n; i += 64)                     // U * U.transpose(): a general matrix product
        for (int j = 0; j < n; ++j)
            A[j * n + i] = fs_gemm_dot(U + i, n, U + j, n, n, i, j, n, n, false);
    FS_SYNC();
    FS_T(4);
}

// normTransform of MultivariateNormalDistribution, Gaussian.cpp:121-167: S (n x n in B4) -> T (in B1)
GPIS_DEV void fs_norm_transform(FsLds &L, FsGlob &G, int n, int lane)
{
    const double *S = G.B4;
    double *T = L.B1;
    for (int i = lane; i < n; i += 64)
        for (int j = 0; j < n; ++j) T[j * n + i] = S[j * n + i];
    FS_SYNC();
    bool ok = true;
    FS_T0();
#define LL(i, j) T[(j) * n + (i)]
    // Eigen::LLT<MatrixXd, Lower> = llt_inplace (Cholesky/LLT.h:312-385): below 32 rows the unblocked loop, else blocks of 8
    // columns — unblocked diagonal block, triangular solve of the panel below it in small panels of 4 columns, rank-8 update of the
    // trailing matrix through gebp (oracle: eig_llt).  A pivot x <= 0 is the failure.  Lane = row.
    const int bsz = n < 32 ? n : 8;
    for (int k0 = 0; k0 < n && ok; k0 += bsz) {
        const int bs = n - k0 < bsz ? n - k0 : bsz, rs = n - k0 - bs, o = k0 + bs;
        for (int k = 0; k < bs; ++k) {                      // unblocked on rows / columns k0 .. k0 + bs - 1
            const int kk = k0 + k;
            double x = LL(kk, kk);
            if (k > 0) {
                double sq = LL(kk, k0) * LL(kk, k0);
                for (int j = 1; j < k; ++j) sq += LL(kk, k0 + j) * LL(kk, k0 + j);
                x -= sq;
            }
            if (x <= 0.0) { ok = false; break; }
            x = sqrt(x);
            FS_SYNC();
            if (lane == 0) LL(kk, kk) = x;
            for (int i = kk + 1 + lane; i < o; i += 64) {
                double v = LL(i, kk);
                if (k > 0) {                                // A21 -= A20 * A10.adjoint(): column-major gemv, alpha = -1
                    double c = 0.0;
                    for (int j = 0; j < k; ++j) c = LL(i, k0 + j) * LL(kk, k0 + j) + c;
                    v = c * -1.0 + v;
                }
                LL(i, kk) = v / x;
            }
            FS_SYNC();
        }
        if (!ok || rs <= 0) continue;
        for (int i = lane; i < rs; i += 64) {               // A21 := A21 * A11^-T, one row per lane
            for (int j2 = 0; j2 < bs; j2 += 4) {
                const int pw = bs - j2 < 4 ? bs - j2 : 4;
                if (j2 > 0)
                    for (int c = 0; c < pw; ++c) {
                        double acc = 0.0;
                        for (int q = 0; q < j2; ++q) acc = LL(o + i, k0 + q) * LL(k0 + j2 + c, k0 + q) + acc;
                        LL(o + i, k0 + j2 + c) = acc * -1.0 + LL(o + i, k0 + j2 + c);
                    }
                for (int c = 0; c < pw; ++c) {
                    const int j = k0 + j2 + c;
                    double v = LL(o + i, j);
                    for (int c3 = 0; c3 < c; ++c3) v -= LL(o + i, k0 + j2 + c3) * LL(j, k0 + j2 + c3);
                    LL(o + i, j) = v * (1.0 / LL(j, j));
                }
            }
        }
        FS_SYNC();
        for (int i = lane; i < rs; i += 64)                 // A22.selfadjointView<Lower>().rankUpdate(A21, -1)
            for (int j = 0; j <= i; ++j)
                LL(o + i, o + j) = fs_gemm_dot(&LL(o + i, k0), n, &LL(o + j, k0), n, bs, i, j, rs, rs, true) * -1.0 + LL(o + i, o + j);
        FS_SYNC();
    }
    FS_SYNC();
    FS_T(8);
    if (ok) {
        for (int i = lane; i < n; i += 64)
            for (int j = i + 1; j < n; ++j) LL(i, j) = 0.0;
        FS_SYNC();
        return;
    }
#undef LL
    for (int i = lane; i < n; i += 64)
        for (int j = 0; j < n; ++j) T[j * n + i] = S[j * n + i];
    FS_SYNC();
    fs_eigh(L, n, T, lane);
    for (int i = lane; i < n; i += 64)
        for (int k = 0; k < n; ++k) {
            const double wk = L.w[k];
            T[k * n + i] *= sqrt(wk > 0.0 ? wk : 0.0);
        }
    FS_SYNC();
}

GPIS_DEV double fs_next1d(Pcg32 &s) { return (double)normalized_uint(s.next_i()); }
// rand_truncated_normal, Gaussian.cpp:57-85 (the rejection loop is capped: a wave must terminate)
GPIS_DEV double fs_rand_truncated_normal(double mean, double sigma, double a, Pcg32 &s)
{
    double z1, z2;
    if (fabs(a - mean) < 0.000001) {
        rand_normal_2(s, z1, z2);
        return fabs(mean + sigma * z1);
    }
    if (a < mean) {
        for (int it = 0; it < (1 << 20); ++it) {
            rand_normal_2(s, z1, z2);
            const double x = mean + sigma * z1;
            if (x >= a) return x;
        }
        return a;
    }
    const double a_bar = (a - mean) / sigma;
    double x_bar = 0.0;
    for (int i = 0; i < 1000; i++) {
        const double u = fs_next1d(s);
        x_bar = sqrt(a_bar * a_bar - 2 * log_glibc(1 - u));
        const double v = fs_next1d(s);
        if (v < x_bar / a_bar) break;
    }
    return sigma * x_bar + mean;
}

// GaussianProcess::sample / sample_cond -> create_mvn_cond -> MultivariateNormalDistribution::sample (one sample), with
// G.pts / G.der (+ G.dirs when has_dirs) the n target entries and G.cpts / G.cder / G.cv the nc conditioning entries.
// Result in G.vals[0 .. n).  GaussianProcess.cpp:605-632, 664-753; Gaussian.cpp:179-232.
GPIS_DEV void fs_sample_cond(const DevModel &M, FsLds &L, FsGlob &G, int n, int nc, V3d deriv_dir, bool has_dirs, Pcg32 &s, int lane)
{
    double *S = G.B4, *s11 = L.B1, *s12 = G.B3, *sol = G.B2;
    auto dir_of = [&](int i) { return has_dirs ? fs_v3(G.dirs[i]) : deriv_dir; };
    { FS_T0();
    for (int i = lane; i < n; i += 64) {
        G.mean[i] = fs_mean(M, G.der[i], fs_v3(G.pts[i]), dir_of(i));
        for (int j = 0; j <= i; ++j) {
            const double c = fs_cov(M, G.der[i], G.der[j], fs_v3(G.pts[i]), fs_v3(G.pts[j]), dir_of(i), dir_of(j));
            S[j * n + i] = c;
            S[i * n + j] = c;
        }
    }
    FS_SYNC();
    FS_T(5); }
    if (nc > 0) {
        { FS_T0();
        for (int i = lane; i < nc; i += 64)
            for (int j = 0; j <= i; ++j) {
                const double c = fs_cov(M, G.cder[i], G.cder[j], fs_v3(G.cpts[i]), fs_v3(G.cpts[j]), deriv_dir, deriv_dir);
                s11[j * nc + i] = c;
                s11[i * nc + j] = c;
            }
        for (int j = lane; j < n; j += 64)
            for (int i = 0; i < nc; ++i)
                s12[j * nc + i] = fs_cov(M, G.cder[i], G.der[j], fs_v3(G.cpts[i]), fs_v3(G.pts[j]), deriv_dir, dir_of(j));
        FS_SYNC();
        FS_T(5); }
        fs_pinv(L, G, nc, lane);
        FS_T0();
        for (int i = lane; i < nc; i += 64)                // pinv * s12: nc x n, depth nc
            for (int j = 0; j < n; ++j)
                sol[j * nc + i] = fs_gemm_dot(s11 + i, nc, s12 + j * nc, 1, nc, i, j, nc, n, false);
        for (int k = lane; k < nc; k += 64) G.resid[k] = G.cv[k] - fs_mean(M, G.cder[k], fs_v3(G.cpts[k]), deriv_dir);
        FS_SYNC();
        FS_T(6);
        for (int i = lane; i < n; i += 64) {
            double acc = 0.0;
            for (int k = 0; k < nc; ++k) acc += sol[i * nc + k] * G.resid[k];
            G.mean[i] += acc;
            for (int j = 0; j < n; ++j)                    // solved * s12: n x n, depth nc
                S[j * n + i] -= fs_gemm_dot(sol + i * nc, 1, s12 + j * nc, 1, nc, i, j, n, n, false);
        }
        FS_SYNC();
        FS_T(7);
    }
    fs_norm_transform(L, G, n, lane);
    FS_T0();
    // the variates: every lane advances its copy of the sampler identically
    for (int i = 0; i < n / 2; i++) {
        double a, b;
        rand_normal_2(s, a, b);
        if (lane == 0) { G.z[i * 2] = a; G.z[i * 2 + 1] = b; }
    }
    if (n % 2) {
        double a, b;
        rand_normal_2(s, a, b);
        if (lane == 0) G.z[n - 1] = a;
    }
    FS_SYNC();
    const double *T = L.B1;
    for (int i = lane; i < n; i += 64) {
        double acc = 0.0;
        #pragma unroll 8
        for (int j = 0; j < n; ++j) acc += T[j * n + i] * G.z[j];
        G.vals[i] = G.mean[i] + acc;
    }
    FS_SYNC();
    FS_T(9);
}

// TangentFrameD(n), TangentFrame.hpp:28-37
GPIS_DEV void fs_frame(V3d n, V3d &tangent, V3d &bitangent)
{
    const double sign = n.z < 0 ? -1. : 1.;
    const double a = -1.0 / (sign + n.z);
    const double b = n.x * n.y * a;
    tangent = V3d{1.0 + sign * n.x * n.x * a, sign * b, -sign * n.x};
    bitangent = V3d{b, sign + n.y * n.y * a, -n.y};
}

// FunctionSpaceGaussianProcessMedium::intersectGP, FunctionSpace...cpp:58-282 (skip_space = 0).  The state record lives in
// global memory and is rewritten on return; wave-uniform result.
GPIS_DEV bool fs_intersect_gp(const DevModel &M, FsLds &L, FsGlob &G, Pcg32 &s, V3 pos, V3 dir, float ray_near, float ray_far, bool first_scatter,
                              bool firstIsectAlongRay, gpis_fs_state *st, int &last_gp_id, double &t_out, int lane)
{
    const int n = M.fs_n;
    const double tOffset = fs_next1d(s);
    const V3d ro = to_d(pos);
    V3d rd = to_d(dir);
    { const double inv = 1.0 / length_d(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
    const double nearT = (double)ray_near, farT = (double)ray_far;
    double maxRayDist = farT - nearT;
    double step = maxRayDist / n;
    if (M.fs_step > 0 && M.fs_step < step)
        step = M.fs_step;
    maxRayDist = step * n;
    const double maxT = nearT + maxRayDist;
    for (int i = lane; i < n; i += 64) {
        double r = (i - tOffset) / (n - 1);
        r = r < 0. ? 0. : (r > 1. ? 1. : r);
        double rt = lerp_d((double)ray_near + step * 0.1, (double)ray_near + maxRayDist, r);
        if (i == 0) rt = nearT + step * 0.1;
        else if (i == n - 1) rt = nearT + maxRayDist;
        G.ts[i] = rt;
        const V3d p = ray_at(ro, rd, rt);
        G.pts[i][0] = p.x; G.pts[i][1] = p.y; G.pts[i][2] = p.z;
        G.der[i] = FS_NONE;
    }
    const bool conditioned = !first_scatter && st->has_context;
    if (conditioned && st->n_points == 0)
        return false;
    int nc = 0;
    if (!conditioned) {
        // sample_start_value: drawn, then unused — numCondPts is 0 in the call (FunctionSpace...cpp:139-148)
        const V3d rp = to_d(pos + dir * ray_near);
        const V3d zero{0., 0., 0.};
        const double mu = fs_mean(M, FS_NONE, rp, zero), sigma = sqrt(fs_cov(M, FS_NONE, FS_NONE, rp, rp, zero, zero));
        (void)fs_rand_truncated_normal(mu, sigma, 0, s);
    } else {
        const int np = st->n_points, nv = st->n_values;
        const V3d last = fs_v3(st->points[np - 1]);
        const double sg_rd = fs_dot(fs_v3(st->sampled_grad), rd);
        const double v_last = st->is_intersect ? st->values[nv - 2] : st->values[nv - 1];      // applyMemory, GaussianProcess.cpp:134-168
        if (M.ctx == GPIS_CTX_RENEWAL || M.ctx == GPIS_CTX_RENEWAL_PLUS) {
            nc = M.ctx == GPIS_CTX_RENEWAL ? 1 : 2;
            if (lane < nc) {
                G.cpts[lane][0] = last.x; G.cpts[lane][1] = last.y; G.cpts[lane][2] = last.z;
                G.cder[lane] = lane == 0 ? FS_NONE : FS_FIRST;
                G.cv[lane] = lane == 0 ? v_last : sg_rd;
            }
        } else if (M.ctx == GPIS_CTX_GLOBAL) {
            nc = np;
            for (int i = lane; i < nc; i += 64) {
                G.cpts[i][0] = st->points[i][0]; G.cpts[i][1] = st->points[i][1]; G.cpts[i][2] = st->points[i][2];
                G.cder[i] = st->derivs[i];
                G.cv[i] = st->values[i];
            }
            FS_SYNC();
            if (!st->is_intersect) {
                if (lane == 0) { G.cpts[nc][0] = last.x; G.cpts[nc][1] = last.y; G.cpts[nc][2] = last.z; G.cder[nc] = FS_FIRST; }
                nc++;
            }
            FS_SYNC();
            if (lane == 0) G.cv[nc - 1] = sg_rd;
        }
    }
    FS_SYNC();
    fs_sample_cond(M, L, G, n, nc, rd, false, s, lane);
    // the march over the sampled values (every lane walks the same array)
    double prevV = G.vals[0];
    int sign0 = prevV < 0 ? -1 : 1;
    double prevT = G.ts[0];
    for (int p = 1; p < n; p++) {
        const double currV = G.vals[p], currT = G.ts[p];
        const int signc = currV < 0 ? -1 : 1;
        if (!first_scatter && firstIsectAlongRay && p == 1) {
            sign0 = signc;
        } else if (signc != sign0) {
            const double offsetT = prevV / (prevV - currV);
            const double t = lerp_d(prevT, currT, offsetT);
            const V3d ip = ray_at(ro, rd, t);
            FS_SYNC();
            for (int i = lane; i < p + 2; i += 64) {
                double v = G.vals[i];
                V3d q = fs_v3(G.pts[i]);
                int dv = FS_NONE;
                if (i == p) { v = lerp_d(prevV, currV, offsetT); q = ip; }                  // makeIntersect, GaussianProcess.cpp:76-87
                if (i == p + 1) { v = (prevV - currV) / (prevT - currT); q = ip; dv = FS_FIRST; }
                st->points[i][0] = q.x; st->points[i][1] = q.y; st->points[i][2] = q.z;
                st->derivs[i] = dv; st->values[i] = v;
            }
            if (lane == 0) { st->has_context = 1; st->is_intersect = 1; st->n_points = p + 2; st->n_values = p + 2; }
            last_gp_id = 0;
            if (M.surf_vol_phase_separate)
                last_gp_id = 1.f < M.surf_vol_phase_amp_thresh ? 0 : 1;
            t_out = t;
            __threadfence_block();
            FS_SYNC();
            return true;
        }
        prevV = currV;
        prevT = currT;
    }
    FS_SYNC();
    for (int i = lane; i < n; i += 64) {
        st->points[i][0] = G.pts[i][0]; st->points[i][1] = G.pts[i][1]; st->points[i][2] = G.pts[i][2];
        st->derivs[i] = FS_NONE; st->values[i] = G.vals[i];
    }
    if (lane == 0) { st->has_context = 1; st->is_intersect = 0; st->n_points = n; st->n_values = n; }
    last_gp_id = 0;
    t_out = maxT;
    __threadfence_block();
    FS_SYNC();
    return false;
}

// sampleGradient (ConditionedGaussian) -> GPRealNodeValues::sampleGrad, FunctionSpace...cpp:284-305, GaussianProcess.cpp:89-132
GPIS_DEV bool fs_sample_gradient(const DevModel &M, FsLds &L, FsGlob &G, Pcg32 &s, V3d rd, V3d ip, gpis_fs_state *st, V3d &grad, int lane)
{
    if (!st->has_context || st->n_points == 0)
        return false;
    V3d tangent, bitangent;
    fs_frame(rd, tangent, bitangent);
    const int nc = st->n_values;
    const bool isect = st->is_intersect != 0;
    const double slope = st->values[nc - 1];
    FS_SYNC();
    for (int i = lane; i < nc; i += 64) {
        G.cpts[i][0] = st->points[i][0]; G.cpts[i][1] = st->points[i][1]; G.cpts[i][2] = st->points[i][2];
        G.cder[i] = st->derivs[i];
        G.cv[i] = st->values[i];
    }
    if (lane < 3) {
        G.pts[lane][0] = ip.x; G.pts[lane][1] = ip.y; G.pts[lane][2] = ip.z;
        G.der[lane] = FS_FIRST;
        const V3d d = lane == 0 ? tangent : (lane == 1 ? bitangent : rd);
        G.dirs[lane][0] = d.x; G.dirs[lane][1] = d.y; G.dirs[lane][2] = d.z;
    }
    FS_SYNC();
    fs_sample_cond(M, L, G, isect ? 2 : 3, nc, rd, true, s, lane);
    const double g0 = G.vals[0], g1 = G.vals[1], g2 = isect ? slope : G.vals[2];
    grad.x = (tangent.x * g0 + bitangent.x * g1) + rd.x * g2;
    grad.y = (tangent.y * g0 + bitangent.y * g1) + rd.y * g2;
    grad.z = (tangent.z * g0 + bitangent.z * g1) + rd.z * g2;
    FS_SYNC();
    if (lane == 0) { st->sampled_grad[0] = grad.x; st->sampled_grad[1] = grad.y; st->sampled_grad[2] = grad.z; }
    __threadfence_block();
    FS_SYNC();
    return true;
}

struct FsState { bool first_scatter; int last_gp_id; V3d last_aniso; };

// GaussianProcessMedium::transmittance over the function-space intersectGP, GPM.cpp:343-393 -> exited (false = blocked or failed)
GPIS_DEV bool fs_transmittance_one(const DevModel &M, FsLds &L, FsGlob &G, Pcg32 &s, const gpis_ray_in *ray, gpis_fs_state *st, FsState &state, int lane)
{
    const V3 pos = v3(ray->pos[0], ray->pos[1], ray->pos[2]), dir = v3(ray->dir[0], ray->dir[1], ray->dir[2]);
    double startT = (double)ray->near_t;
    float farT = ray->far_t;
    if (!__builtin_isfinite(farT)) farT = (float)(startT + 2000);
    const float maxT = farT;
    double t = (double)maxT;
    const V3d ro = to_d(pos);
    V3d rd = to_d(dir);
    { const double inv = 1.0 / length_d(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
    bool exited, first = true;
    do {
        exited = !fs_intersect_gp(M, L, G, s, pos, dir, (float)startT, farT, state.first_scatter, first, st, state.last_gp_id, t, lane);
        first = false;
        if (t < (double)maxT) {
            V3d grad;
            if (!fs_sample_gradient(M, L, G, s, to_d(dir), ray_at(ro, rd, t), st, grad, lane))
                return false;
            state.last_aniso = grad;
            state.first_scatter = false;
            if (!__builtin_isfinite((grad.x + grad.y + grad.z) / 3.0))
                return false;
        }
        startT = t;
    } while (t < (double)maxT && exited);
    return exited;
}

template <bool WANT_SAMPLE>
__global__ void __launch_bounds__(64) k_fs_march(const DevModel *__restrict__ Mp, size_t n_rays, const gpis_ray_in *__restrict__ rays,
                                                 gpis_fs_state *__restrict__ states, gpis_seg_out *__restrict__ outs, uint8_t *__restrict__ visible,
                                                 FsGlob *__restrict__ workspace)
{
    __shared__ FsLds L;
    FsGlob &G = workspace[blockIdx.x];
    const DevModel &M = *Mp;
    const int lane = (int)threadIdx.x;
    for (size_t idx = blockIdx.x; idx < n_rays; idx += gridDim.x) {
        const gpis_ray_in *ray = rays + idx;
        gpis_fs_state *st = states + idx;
        Pcg32 s;
        s.state = st->sampler_state;
        FsState state;
        state.first_scatter = ray->first_scatter != 0;
        state.last_gp_id = ray->last_gp_id;
        state.last_aniso = V3d{ray->last_aniso[0], ray->last_aniso[1], ray->last_aniso[2]};
        FS_SYNC();
        FS_T0();
        if (!WANT_SAMPLE) {
            const bool vis = fs_transmittance_one(M, L, G, s, ray, st, state, lane);
            if (lane == 0) { visible[idx] = vis ? 1 : 0; st->sampler_state = s.state; }
            continue;
        }
        // GaussianProcessMedium::sampleDistance, GPM.cpp:221-341
        gpis_seg_out o{};
        const V3 pos = v3(ray->pos[0], ray->pos[1], ray->pos[2]), dir = v3(ray->dir[0], ray->dir[1], ray->dir[2]);
        double startT = (double)ray->near_t;
        float farT = ray->far_t;
        if (!__builtin_isfinite(farT)) farT = (float)(startT + 2000);
        const float maxT = farT;
        o.gp_id = state.last_gp_id;
        o.last_val = ray->last_val;
        V3d aniso = state.last_aniso;
        bool finished = false;
        if (ray->bounce >= M.max_bounces) {
            o.ok = 0;
            finished = true;
        } else if (maxT == 0.f) {
            o.sample_t = maxT;
            o.weight[0] = o.weight[1] = o.weight[2] = 1.f;
            o.exited = 1;
            const V3 pp = pos + dir * o.sample_t;
            o.p[0] = pp.x; o.p[1] = pp.y; o.p[2] = pp.z;
            o.scheme = GPIS_UNI;
            o.ok = 1;
            finished = true;
        } else if (M.absorption_only) {
            if (maxT == __builtin_huge_valf()) {
                o.ok = 0;
                finished = true;
            } else {
                o.sample_t = maxT;
                const bool vis = fs_transmittance_one(M, L, G, s, ray, st, state, lane);
                o.weight[0] = o.weight[1] = o.weight[2] = vis ? 1.f : 0.f;
                o.exited = 1;
                o.scheme = GPIS_UNI;
                aniso = state.last_aniso;
            }
        } else {
            double t = (double)maxT;
            const V3d ro = to_d(pos);
            V3d rd = to_d(dir);
            { const double inv = 1.0 / length_d(rd); rd.x *= inv; rd.y *= inv; rd.z *= inv; }
            bool exited, first = true;
            do {
                exited = !fs_intersect_gp(M, L, G, s, pos, dir, (float)startT, farT, state.first_scatter, first, st, state.last_gp_id, t, lane);
                first = false;
                if (t < (double)maxT) {
                    V3d grad;
                    if (!fs_sample_gradient(M, L, G, s, to_d(dir), ray_at(ro, rd, t), st, grad, lane)) {
                        o.t = t; o.exited = exited; o.ok = 0; o.gp_id = state.last_gp_id;
                        finished = true;
                        break;
                    }
                    aniso = grad;
                    state.last_aniso = aniso;
                    state.first_scatter = false;
                    if (!__builtin_isfinite((aniso.x + aniso.y + aniso.z) / 3.0)) {
                        aniso = V3d{1., 0., 0.};
                        o.t = t; o.exited = exited; o.ok = 0; o.gp_id = state.last_gp_id;
                        finished = true;
                        break;
                    }
                }
                startT = t;
            } while (t < (double)maxT && exited);
            if (!finished) {
                o.t = t;
                o.exited = exited;
                if (!exited) {
                    double d = aniso.x * (double)dir.x; d += aniso.y * (double)dir.y; d += aniso.z * (double)dir.z;
                    double l2 = 0.; l2 += aniso.x * aniso.x; l2 += aniso.y * aniso.y; l2 += aniso.z * aniso.z;
                    if (d > 0) {
                        o.gp_id = state.last_gp_id; o.ok = 0;
                        finished = true;
                    } else if (l2 < (double)0.0000001f) {
                        aniso = V3d{1., 0., 0.};
                        o.gp_id = state.last_gp_id; o.ok = 0;
                        finished = true;
                    } else {
                        const float col = M.color.enabled ? (float)ramp_eval(M.color, ray_at(ro, rd, t)) : 1.f;     // ramp noises (fs_check refuses a sandstone / rust colour here)
                        o.weight[0] = o.weight[1] = o.weight[2] = col;
                        o.continued_weight[0] = o.continued_weight[1] = o.continued_weight[2] = col;
                    }
                } else {
                    V3d grad = aniso;
                    (void)fs_sample_gradient(M, L, G, s, to_d(dir), ray_at(ro, rd, t), st, grad, lane);   // GPM.cpp:319
                    aniso = grad;
                    o.weight[0] = o.weight[1] = o.weight[2] = 1.f;
                    o.continued_weight[0] = o.continued_weight[1] = o.continued_weight[2] = 1.f;
                }
                if (!finished) {
                    const float ft = (float)t;
                    o.sample_t = ft < maxT ? ft : maxT;
                    o.continued_t = (float)t;
                    for (int c = 0; c < 3; ++c) {
                        o.weight[c] *= M.sigma_s_over_t[c];
                        o.continued_weight[c] *= M.sigma_s_over_t[c];
                    }
                    o.scheme = GPIS_UNI;
                }
            }
        }
        if (!finished) {
            const V3 pp = pos + dir * o.sample_t;
            o.p[0] = pp.x; o.p[1] = pp.y; o.p[2] = pp.z;
            o.gp_id = state.last_gp_id;
            o.ok = 1;
        }
        o.aniso[0] = aniso.x; o.aniso[1] = aniso.y; o.aniso[2] = aniso.z;
        if (lane == 0) { outs[idx] = o; st->sampler_state = s.state; }
        FS_T(10);
    }
}

// Test surface (gpis_fs_linalg_batch): the dense linear algebra of this path on caller-supplied matrices, one wave per matrix —
// op 0: SelfAdjointEigenSolver (eigenvectors -> out, eigenvalues -> evals), 1: MultivariateNormalDistribution's normTransform
// (LLT, else the eigen square root), 2: pseudo_inverse.  n x n column-major doubles; only +, -, *, / and sqrt run here, so the
// results equal the CPU restatement's — and, through it, the reference's Eigen — bit for bit.
GPIS_TU_KERNEL __global__ void __launch_bounds__(64) k_fs_linalg(int op, int n, size_t count, const double *__restrict__ in, double *__restrict__ out,
                                                                double *__restrict__ evals, FsGlob *__restrict__ workspace)
{
    __shared__ FsLds L;
    FsGlob &G = workspace[blockIdx.x];
    const int lane = (int)threadIdx.x;
    for (size_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
        const double *a = in + idx * (size_t)n * n;
        double *o = out + idx * (size_t)n * n;
        FS_SYNC();
        double *dst = op == 1 ? G.B4 : L.B1;
        for (int i = lane; i < n * n; i += 64) dst[i] = a[i];
        FS_SYNC();
        if (op == 0) fs_eigh(L, n, L.B1, lane);
        else if (op == 1) fs_norm_transform(L, G, n, lane);
        else fs_pinv(L, G, n, lane);
        FS_SYNC();
        for (int i = lane; i < n * n; i += 64) o[i] = L.B1[i];
        if (op == 0 && evals)
            for (int i = lane; i < n; i += 64) evals[idx * (size_t)n + i] = L.w[i];
    }
}

}   // namespace gpis
