// tu_libm.hip — test surface of gpis_libm.hpp: evaluates one of the libm replicas on an array, so that tests/test_gpu_libm.py can
// compare what the device computes with the host's libm bit for bit.
#include <hip/hip_runtime.h>
#include "gpis.h"
#include "gpis_libm.hpp"
#include "gpis_launch.hpp"

#pragma clang fp contract(off)

namespace gpis {

__global__ void __launch_bounds__(256) k_libm(int fn, size_t n, const double *x, const double *y, double *out, double *out2)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    switch (fn) {
    case GPIS_LIBM_EXP: out[i] = exp_glibc(a); break;
    case GPIS_LIBM_LOG: out[i] = log_glibc(a); break;
    case GPIS_LIBM_LOGF: out[i] = (double)logf_glibc((float)a); break;
    case GPIS_LIBM_SIN: out[i] = sin_glibc(a); break;
    case GPIS_LIBM_COS: out[i] = cos_glibc(a); break;
    case GPIS_LIBM_SINCOS: { double s, c; sincos_glibc(a, &s, &c); out[i] = s; out2[i] = c; break; }
    case GPIS_LIBM_POW: out[i] = pow_glibc(a, y[i]); break;
    case GPIS_LIBM_SINCOSF: { float s, c; sincosf_glibc((float)a, &s, &c); out[i] = (double)s; out2[i] = (double)c; break; }
    default: out[i] = 0.0;
    }
}

namespace launch {
void libm_eval(int fn, size_t n, const double *x, const double *y, double *out, double *out2, hipStream_t s)
{
    k_libm<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(fn, n, x, y, out, out2);
}
}   // namespace launch
}   // namespace gpis
