// gpis_lane.hpp — the lane-per-ray kernels: one lane = one ray / query, impulses generated on the fly
// (k_sample_distance<P>, k_transmittance<P>), their persistent refilling form (k_persist_march<P, WANT_SAMPLE>,
// gpis_persist.inc) and the single-query entries of the all-features instance.  Included by the tu_lane_* and
// tu_persist_* translation units; each instantiates the instances it launches (gpis_launch.hpp).
#pragma once
#include "gpis_device.hpp"

#pragma clang fp contract(off)

namespace gpis {

// ======================================================================================
// kernels — generic path: one lane = one ray / query, impulses generated on the fly
// ======================================================================================
constexpr int kBlock = 64;   // one wave per workgroup: rays are independent, small blocks balance the march
#ifndef GPIS_GENERIC_OCC
#define GPIS_GENERIC_OCC 4   // waves per SIMD the lane-per-ray march kernels are register-allocated for (C2 scene S: 1 wave 11.6, 2 → 20.4, 3 → 25.2, 4 → 26.6, 5 → 26.0 Msamples/s)
#endif

__device__ __forceinline__ void flush_counters(Counters *cnt, uint32_t n_eval, uint32_t n_seg)
{
    // one pair of 64-bit atomics per wave
    unsigned long long e = n_eval, s = n_seg;
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_down(e, off, 64);
        s += __shfl_down(s, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (e) atomicAdd(&cnt->n_eval, e);
        if (s) atomicAdd(&cnt->n_seg, s);
    }
}

// P = the path instance (gpis_device.hpp: generic / spec_1d / spec_3d / spec_3d_multires)
template <class P>
__global__ void __launch_bounds__(kBlock, GPIS_GENERIC_OCC) k_sample_distance(const DevModel *__restrict__ Mp, size_t n, const gpis_ray_in *__restrict__ rays,
                                                            gpis_seg_out *__restrict__ out, gpis_cond_coeff *__restrict__ coeff,
                                                            const uint8_t *__restrict__ mask, Counters *cnt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const DevModel &M = *Mp;
    Realization noise{};
    uint32_t nseg = 0;
    if (i < n && (!mask || mask[i])) {
        gpis_ray_in ray = rays[i];
        gpis_seg_out o;
        P::sample_distance(M, noise, ray, o);
        out[i] = o;
        if (coeff) {
            gpis_cond_coeff c = noise.c;
            c.n_evals = noise.n_eval;
            coeff[i] = c;
        }
        nseg = 1;
    }
    flush_counters(cnt, noise.n_eval, nseg);
}

template <class P>
__global__ void __launch_bounds__(kBlock, GPIS_GENERIC_OCC) k_transmittance(const DevModel *__restrict__ Mp, size_t n, const gpis_ray_in *__restrict__ rays,
                                                          uint8_t *__restrict__ visible, const uint8_t *__restrict__ mask, Counters *cnt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const DevModel &M = *Mp;
    Realization noise{};
    uint32_t nseg = 0;
    if (i < n) {
        if (!mask || mask[i]) {
            gpis_ray_in ray = rays[i];
            MediumState st;
            state_from_ray(ray, st);
            visible[i] = P::transmittance(M, noise, ray, st) ? 1 : 0;
            nseg = 1;
        } else {
            visible[i] = 0;
        }
    }
    flush_counters(cnt, noise.n_eval, nseg);
}

// persistent refilling form of the two kernels above (gpis_persist.inc): a fixed grid of waves pulls rays from
// a counter.  Q = slots of the per-lane LDS queue between the impulse generator and the kernel body.
#ifndef GPIS_PERSIST_Q
#define GPIS_PERSIST_Q 16
#endif
constexpr int kPersistQ = GPIS_PERSIST_Q;
constexpr unsigned kPersistSlots = 256;   // ring of ray counters: one per persistent launch in flight
#ifndef GPIS_PERSIST_OCC
#define GPIS_PERSIST_OCC 3   // waves per SIMD of the register allocation (LDS: 12.5 KB per wave -> 12 waves per CU)
#endif
template <class P, bool WANT_SAMPLE>
__global__ void __launch_bounds__(kBlock, GPIS_PERSIST_OCC) k_persist_march(const DevModel *__restrict__ Mp, PersistArgs a)
{
    __shared__ PersistQueue<kPersistQ> q;
    fast_lds_init(q);
    P::template march<WANT_SAMPLE, kPersistQ>(*Mp, a, q);
}

__device__ __forceinline__ RayInfo info_of(const gpis_query &q) { return RayInfo{q.pixel[0], q.pixel[1], q.spp, q.segment, q.scene_seed, q.info_t}; }
__device__ __forceinline__ RayInfo info_of(const gpis_nee_query &q) { return RayInfo{q.pixel[0], q.pixel[1], q.spp, q.segment, q.scene_seed, q.info_t}; }

GPIS_TU_KERNEL __global__ void __launch_bounds__(kBlock) k_eval_value(const DevModel *__restrict__ Mp, size_t n, const gpis_query *__restrict__ q,
                                                       float *__restrict__ value, int32_t *__restrict__ gp_id, Counters *cnt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    Realization r{};
    if (i < n) {
        gpis_query qq = q[i];
        r.c = qq.coeff;
        int id;
        value[i] = generic::evaluate_value(*Mp, r, v3(qq.p[0], qq.p[1], qq.p[2]), v3(qq.dir[0], qq.dir[1], qq.dir[2]), info_of(qq), id);
        if (gp_id) gp_id[i] = id;
    }
    flush_counters(cnt, r.n_eval, 0);
}
GPIS_TU_KERNEL __global__ void __launch_bounds__(kBlock) k_eval_gradient(const DevModel *__restrict__ Mp, size_t n, const gpis_query *__restrict__ q,
                                                          float *__restrict__ grad3, Counters *cnt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    Realization r{};
    if (i < n) {
        gpis_query qq = q[i];
        r.c = qq.coeff;
        V3 g = generic::evaluate_gradient(*Mp, r, v3(qq.p[0], qq.p[1], qq.p[2]), qq.t_segment, v3(qq.dir[0], qq.dir[1], qq.dir[2]), info_of(qq));
        grad3[3 * i] = g.x; grad3[3 * i + 1] = g.y; grad3[3 * i + 2] = g.z;
    }
    flush_counters(cnt, r.n_eval, 0);
}
GPIS_TU_KERNEL __global__ void __launch_bounds__(kBlock) k_conditioning(const DevModel *__restrict__ Mp, size_t n, const gpis_query *__restrict__ q,
                                                         const float *__restrict__ tv, const float *__restrict__ tg,
                                                         gpis_cond_coeff *__restrict__ co, Counters *cnt)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    Realization r{};
    if (i < n) {
        gpis_query qq = q[i];
        generic::conditioning(*Mp, r, v3(qq.p[0], qq.p[1], qq.p[2]), v3(qq.dir[0], qq.dir[1], qq.dir[2]), tv[i], v3(tg[3 * i], tg[3 * i + 1], tg[3 * i + 2]), info_of(qq));
        gpis_cond_coeff c = r.c;
        c.n_evals = r.n_eval;
        co[i] = c;
    }
    flush_counters(cnt, r.n_eval, 0);
}
// neePDF / neeGrad queries (the BSDF hooks and the NEE driver).  P = the path instance: the NEE driver's medium is the 1D-sampling
// one (config C2), which gets the specialised instance; everything else the all-features one.
template <class P>
__global__ void __launch_bounds__(kBlock) k_nee(const DevModel *__restrict__ Mp, size_t n, const gpis_nee_query *__restrict__ q,
                                                float *__restrict__ pdf, float *__restrict__ grad3, Counters *cnt,
                                                const uint8_t *__restrict__ mask = nullptr)
{
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    Realization r{};
    if (i < n && (!mask || mask[i])) {
        gpis_nee_query qq = q[i];
        r.c = qq.coeff;
        V3 rd = v3(qq.ray_dir[0], qq.ray_dir[1], qq.ray_dir[2]), nn = v3(qq.normal[0], qq.normal[1], qq.normal[2]), p = v3(qq.p[0], qq.p[1], qq.p[2]);
        if (pdf) pdf[i] = P::nee_pdf_of(*Mp, r, rd, nn, p, qq.t_segment, info_of(qq));
        if (grad3) {
            V3 g = P::nee_grad_of(*Mp, r, rd, nn, p, info_of(qq));
            grad3[3 * i] = g.x; grad3[3 * i + 1] = g.y; grad3[3 * i + 2] = g.z;
        }
    }
    flush_counters(cnt, r.n_eval, 0);
}

}   // namespace gpis
