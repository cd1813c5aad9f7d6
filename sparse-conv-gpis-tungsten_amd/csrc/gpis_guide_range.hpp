// gpis_guide_range.hpp — the guided march (gpis_guide.hpp) with IN-WAVE REFILL.
//
// In k_guided_sample_distance a wave owns 64 rays for its whole life: rays of one pixel end at different march
// times, masked-out rays (shadow segments of samples that missed) never start, and by the time the wave reaches its
// exact evaluations on average 30 of 64 lanes still hold a live ray — yet a wave-cooperative evaluation costs the
// same whether it serves 3 lanes or 64 (measured: 69 % of the kernel's cycles are exact rounds at 30/64 lanes).
//
// Here a wave owns a contiguous RANGE of the batch (range_len rays = a few neighbouring pixels in the tile driver's
// sample order) and a lane whose segment is finished takes the range's next ray: ballot + prefix count, no atomics
// (the range belongs to this wave alone).  Consecutive rays of a batch are neighbours in lattice space, so the
// refilled lanes park for their exact values in the same few lattice cells as the lanes already waiting there: the
// clusters the cooperative evaluator serves grow towards 64 lanes, masked-out rays cost nothing, and the guide
// steps of loop A run with a full wave too.  Every transition is guided_march's, ray by ray, so results are
// bit-identical.
//
// The end-of-segment gradient (GPM.cpp:283 / 319) leaves the march kernel: a segment's march result (t, hit,
// lastVal, gpId) is written as a PENDING record and k_guided_range_grad — one lane per ray in batch order, i.e.
// coherent waves — evaluates all gradients with full clusters and completes the gpis_seg_out records.  The march
// kernel no longer carries the gradient evaluator nor its live values.
#pragma once
#include "gpis_guide.hpp"

#pragma clang fp contract(off)

namespace gpis {

constexpr int G_IDLE = 9;                       // no ray left in the wave's range (extends GPhase)

#ifndef GPIS_RANGE_OCC
#define GPIS_RANGE_OCC 4
#endif
#ifndef GPIS_RANGE_OCC_TR
#define GPIS_RANGE_OCC_TR 5
#endif

struct RangeArgs {
    size_t n;
    const gpis_ray_in *rays;
    gpis_seg_out *out;            // sampleDistance
    gpis_cond_coeff *coeff;       // sampleDistance, optional
    uint8_t *visible;             // transmittance
    const uint8_t *mask;
    Counters *cnt;
    unsigned long long *guide_cnt;
    uint32_t range_len;           // rays per wave (multiple of 64)
};

template <bool WANT_SAMPLE, bool SMALLARG>
GPIS_DEV void guided_march_range(const DevModel &M, const FastTable &T, const GuideField &F, FastLds &lds, const RangeArgs &a)
{
    const int lane = (int)(threadIdx.x & 63);
    const size_t base = (size_t)blockIdx.x * a.range_len;
    const uint32_t range_n = (uint32_t)((a.n - base) < (size_t)a.range_len ? (a.n - base) : (size_t)a.range_len);
    uint32_t next = 0;                 // wave-uniform: first offset of the range not handed out yet

    // ---- per-lane ray state (guided_march's) ----
    uint32_t off = 0;                  // offset of the lane's ray in the range
    bool have_ray = false;
    V3 pos = v3(0.f, 0.f, 1.f), dir = v3(0.f, 0.f, 1.f);
    float nearT = 0.f, farT = 1.f, u_jitter = 0.f, step_size = 1.f;
    bool first_scatter = true;
    GuideRay gr{};
    int phase = G_DONE;                // every lane starts by asking for a ray
    bool early_ok = false, bounce_stop = false;
    double t = 0., t_prevpos = 0., a_lo = 0., intp = 0., t_test = 0., t_prev = 0.;
    float pf = 0.f, fc = 0.f;
    bool pf_valid = false;
    int sign0 = 1, step = 0, gp = 0;
    float last_val = 0.f;
    bool hit = false;
    uint32_t n_eval = 0, n_eval_ray = 0, n_guide = 0, n_seg = 0;

    auto ray_frame = [&]() {
        Frame c{};
        if (M.iso3d) {
            V3 d = dir;
            asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
            c = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, d, 1.0f)));
        }
        return c;
    };
    auto world_at = [&](double tq) { return to_f(ray_at(to_d(pos), to_d(dir), tq)); };
    auto begin_march = [&]() {
        t_prevpos = (double)nearT;
        t = (double)(nearT + step_size * u_jitter);
        phase = (t < (double)farT) ? G_MARCH : X_FINAL;
    };
    auto advance = [&]() {
        t_prevpos = t;
        t += (double)step_size;
        phase = (t < (double)farT) ? G_MARCH : X_FINAL;
    };
    auto begin_refine = [&]() {   // SCNM.cpp:143-146
        intp = (double)pf / ((double)pf - (double)fc);
        a_lo = t - (double)step_size;
        t_prev = lerp_d(a_lo, t, intp);
        t_test = t_prev;
        phase = X_REFINE;
    };
    auto refine_step = [&](int sign_test) {   // SCNM.cpp:147-160
        bool done = false;
        if (sign_test == sign0) {
            done = true;
        } else {
            intp *= 0.9;
            if (intp <= 0.01) {
                t_prev = t_test = 0;
                done = true;
            } else {
                t_prev = t_test;
                t_test = lerp_d(a_lo, t, intp);
            }
        }
        if (done) {
            t = t_prev;
            hit = true;
            last_val = 0.0f;
            phase = G_DONE;
        }
    };

    for (;;) {
        // ---- R: finished lanes write their march result and take the next ray of the range ----
        const unsigned long long done_mask = __ballot(phase == G_DONE);
        if (done_mask != 0ULL) {
            if (phase == G_DONE && have_ray) {
                const size_t i = base + off;
                n_seg++;
                if (WANT_SAMPLE) {
                    if (early_ok || bounce_stop) {
                        finish_sample_distance(M, a.rays + i, pos, dir, farT, early_ok, false, false, 0., 0.f, 0, v3(0.f, 0.f, 0.f), a.out + i);
                    } else {
                        gpis_seg_out *o = a.out + i;     // pending record: k_guided_range_grad completes it
                        o->t = t;
                        o->exited = hit ? 0 : 1;
                        o->last_val = last_val;
                        o->gp_id = gp;
                        o->ok = kSegPending;
                    }
                    if (a.coeff) {
                        gpis_cond_coeff c;
                        memset(&c, 0, sizeof c);
                        c.n_evals = n_eval_ray;
                        a.coeff[i] = c;
                    }
                } else {
                    a.visible[i] = hit ? 0 : 1;
                }
                have_ray = false;
            }
            if (phase == G_DONE) {
                const uint32_t mine = next + (uint32_t)__popcll(done_mask & ((1ULL << lane) - 1ULL));
                if (mine >= range_n) {
                    phase = G_IDLE;
                } else {
                    const size_t i = base + mine;
                    if (a.mask && !a.mask[i]) {
                        if (!WANT_SAMPLE) a.visible[i] = 0;
                        // stays G_DONE: asks again in the next pass
                    } else {
                        const gpis_ray_in *rp = a.rays + i;
                        off = mine;
                        have_ray = true;
                        pos = v3(rp->pos[0], rp->pos[1], rp->pos[2]);
                        dir = v3(rp->dir[0], rp->dir[1], rp->dir[2]);
                        nearT = rp->near_t; farT = rp->far_t; u_jitter = rp->u_jitter;
                        first_scatter = rp->first_scatter != 0;
                        if (!__builtin_isfinite(farT))
                            farT = (float)((double)nearT + 2000);
                        step_size = (farT - nearT) / (float)M.min_step;
                        if (M.step_size < step_size)
                            step_size = M.step_size;
                        {
                            const Frame coord = ray_frame();
                            gr = guide_ray(M, F, pos, dir, coord, nearT);
                        }
                        early_ok = false; bounce_stop = false;
                        t = (double)nearT; t_prevpos = (double)nearT;
                        pf = 0.f; fc = 0.f; pf_valid = false;
                        sign0 = 1; step = 0; gp = 0; last_val = 0.f; hit = false;
                        n_eval_ray = 0;
                        phase = G_INIT;
                        if (WANT_SAMPLE && rp->bounce >= M.max_bounces) {
                            bounce_stop = true;
                            phase = G_DONE;
                        } else if (WANT_SAMPLE && farT == 0.f) {
                            early_ok = true;
                            phase = G_DONE;
                        }
                    }
                }
            }
            next += (uint32_t)__popcll(done_mask);
            continue;
        }
        // ---- A: guide steps for every lane that can take one ----
        for (;;) {
            if (!WANT_SAMPLE && phase == X_FINAL) {
                hit = false;                // transmittance: the segment exits, lastVal is not part of the result
                phase = G_DONE;
            }
            const bool stepping = phase == G_INIT || phase == G_MARCH;
            if (__ballot(stepping) == 0ULL)
                break;
            if (stepping) {
                if (phase == G_INIT) {
                    const int s = guide_sign_at(M, F, gr, (double)nearT);
                    if (s != 0) {
                        n_guide++;
                        sign0 = s;
                        pf_valid = false;
                        begin_march();
                    } else {
                        phase = X_F0;
                    }
                } else {
                    const int s = guide_sign_at(M, F, gr, t);
                    const bool adopt = !first_scatter && step == 0;
                    if (s != 0 && (adopt || s == sign0)) {
                        n_guide++;
                        step++;
                        if (adopt) sign0 = s;
                        pf_valid = false;
                        advance();
                    } else {
                        phase = X_CUR;
                    }
                }
            }
        }
        if (__ballot(phase == G_DONE) != 0ULL)
            continue;           // refill before the exact round: the new rays may join its cluster
        // ---- B: exact evaluations for the parked lanes, one coherent cluster at a time ----
        const bool need = phase >= X_F0 && phase <= X_FINAL;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask == 0ULL)
            break;              // every lane is idle: the range is done
        const double tq = phase == X_F0 ? (double)nearT : (phase == X_PREV ? t_prevpos : (phase == X_REFINE ? t_test : (phase == X_FINAL ? (double)farT : t)));
        const V3 pq = world_at(tq);
        const Frame coord = ray_frame();
        const V3 ug = grid_point(M, F, pq, coord);
        const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
        const int lead = __builtin_ctzll(need_mask);
        const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
        const bool in_cluster = need && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
        int gp_new;
        float fv;
        const unsigned long long cl_mask = __ballot(in_cluster);
        uint32_t ne = 0;
        if (__popcll(cl_mask) <= kSoloMaxLanes) {
            fv = 0.f;
            gp_new = 0;
            for (unsigned long long mm = cl_mask; mm; mm &= mm - 1ULL) {
                const int src = __builtin_ctzll(mm);
                int gpx;
                const float v = solo_evaluate_value(M, T, lds, src, pq, coord, gpx, ne);
                if (lane == src) { fv = v; gp_new = gpx; }
            }
        } else {
            fv = coop_evaluate_value<SMALLARG>(M, T, lds, in_cluster, pq, coord, gp_new, ne, kGuideSplit && __popcll(cl_mask) <= 32);
        }
        n_eval += ne; n_eval_ray += ne;
        if (in_cluster) {
            gp = gp_new;
            const double f = (double)fv;
            if (phase == X_F0) {                       // SCNM.cpp:125-128
                sign0 = f < 0 ? -1 : 1;
                pf = fv;
                pf_valid = true;
                begin_march();
            } else if (phase == X_CUR) {               // SCNM.cpp:133-141, 172-173
                step++;
                const int signc = f < 0 ? -1 : 1;
                if (!first_scatter && step == 1) {
                    sign0 = signc;
                    pf = fv; pf_valid = true;
                    advance();
                } else if (signc != sign0) {
                    fc = fv;
                    if (pf_valid) begin_refine();
                    else phase = X_PREV;
                } else {
                    pf = fv; pf_valid = true;
                    advance();
                }
            } else if (phase == X_PREV) {
                pf = fv; pf_valid = true;
                begin_refine();
            } else if (phase == X_REFINE) {
                refine_step(f < 0 ? -1 : 1);
            } else {                                   // X_FINAL: lastVal at farT (SCNM.cpp:176-181)
                t = (double)farT;
                last_val = fv;
                hit = false;
                phase = G_DONE;
            }
        }
    }
    fast_flush_counters(a.cnt, n_eval, n_seg);
    unsigned long long gsum = n_guide;
    for (int o2 = 32; o2 > 0; o2 >>= 1) gsum += __shfl_down(gsum, o2, 64);
    if (lane == 0 && gsum) atomicAdd(a.guide_cnt, gsum);
}

template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_RANGE_OCC) k_guided_range_sd(const DevModel *__restrict__ Mp, FastTable T, GuideField F, RangeArgs a)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    guided_march_range<true, SMALLARG>(*Mp, T, F, lds, a);
}
template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_RANGE_OCC_TR) k_guided_range_tr(const DevModel *__restrict__ Mp, FastTable T, GuideField F, RangeArgs a)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    guided_march_range<false, SMALLARG>(*Mp, T, F, lds, a);
}

// One gradient evaluation per pending segment (GPM.cpp:283 on a hit, :319 on exit) and the MediumSample writes of
// GPM.cpp:291-340.  One lane per ray in batch order: the 64 rays of a wave are neighbours, clusters are full.
template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_FAST_OCC) k_guided_range_grad(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n,
                                                                                const gpis_ray_in *__restrict__ rays, gpis_seg_out *__restrict__ out,
                                                                                gpis_cond_coeff *__restrict__ coeff, const uint8_t *__restrict__ mask, Counters *cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    const size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    bool want = i < n && (!mask || mask[i]);
    if (want)
        want = out[i].ok == kSegPending;
    V3 pos = v3(0.f, 0.f, 1.f), dir = v3(0.f, 0.f, 1.f);
    float farT = 1.f;
    double t = 0.;
    bool hit = false;
    float last_val = 0.f;
    int gp = 0;
    if (want) {
        const gpis_ray_in *rp = rays + i;
        pos = v3(rp->pos[0], rp->pos[1], rp->pos[2]);
        dir = v3(rp->dir[0], rp->dir[1], rp->dir[2]);
        farT = rp->far_t;
        if (!__builtin_isfinite(farT))
            farT = (float)((double)rp->near_t + 2000);
        t = out[i].t;
        hit = out[i].exited == 0;
        last_val = out[i].last_val;
        gp = out[i].gp_id;
    }
    Frame coord{};
    if (M.iso3d)
        coord = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, dir, 1.0f)));
    V3d rdn = to_d(dir);
    { double inv = 1.0 / length_d(rdn); rdn.x *= inv; rdn.y *= inv; rdn.z *= inv; }
    const V3 pgq = to_f(ray_at(to_d(pos), rdn, t));
    const V3 ug = grid_point(M, F, pgq, coord);
    const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
    uint32_t n_eval = 0;
    bool pending = want;
    V3 g = v3(0.f, 0.f, 0.f);
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ULL)
            break;
        const int lead = __builtin_ctzll(pm);
        const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
        const bool in_cluster = pending && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
        const V3 gi = coop_evaluate_gradient<SMALLARG>(M, T, lds, in_cluster, pgq, coord, n_eval);
        if (in_cluster) {
            g = gi;
            pending = false;
        }
    }
    if (want) {
        finish_sample_distance(M, rays + i, pos, dir, farT, false, true, hit, t, last_val, gp, g, out + i);
        if (coeff)
            coeff[i].n_evals += 1u;
    }
    fast_flush_counters(cnt, n_eval, 0u);
}

}   // namespace gpis
