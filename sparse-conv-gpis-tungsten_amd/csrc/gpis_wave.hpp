// gpis_wave.hpp — the guided march as a WAVEFRONT of three small kernels instead of one resident
// state machine per lane (gpis_guide.hpp: guided_march).
//
// In the resident kernel a wave serves its own 64 rays: when some of them need an exact value the
// others wait, only ~30 of 64 lanes take part in a cooperative evaluation, and the march state that
// must stay live across the evaluator costs 300+ B/lane of scratch.  Here the march state lives in HBM
// (80 B per ray) and every iteration is
//     k_wave_step   per ray: consume the delivered value, take guide steps until the next exact value
//                   is needed (or the segment ends), emit a request keyed by the Morton code of its
//                   lattice cell;
//     radix sort    requests by key — finished rays carry the largest key, so the sorted prefix is also
//                   the next iteration's compacted active list;
//     k_wave_eval   64 consecutive requests = 64 queries in (nearly) the same lattice cell: the
//                   cooperative evaluator runs with every lane active and a minimal cell union.
// The arithmetic of every evaluation and every transition is the one of guided_march (same functions,
// same order per ray), so results are bit-identical; only the grouping of work changes.
#pragma once
#include "gpis_guide.hpp"

#pragma clang fp contract(off)

namespace gpis {

struct WaveState {              // 80 bytes per ray
    double t, t_prevpos, intp, t_test, t_prev;
    float pf, fc, last_val;
    float fv;                   // value delivered for the pending request (k_wave_eval); gradient x later
    float gy, gz;               // gradient y, z (k_wave_grad)
    int32_t step;
    int8_t phase, sign0;
    uint8_t flags;              // kWavePfValid | kWaveHit | kWaveEarlyOk
    int8_t gp, gp_new;
    uint8_t _pad[3];
};
static_assert(sizeof(WaveState) == 80, "WaveState layout");
constexpr uint8_t kWavePfValid = 1, kWaveHit = 2, kWaveEarlyOk = 4;

struct WaveRay {                // the per-ray constants every kernel re-derives from gpis_ray_in
    V3 pos, dir;
    float nearT, farT, u_jitter, step_size;
    bool first_scatter;
    int bounce;
};
GPIS_DEV WaveRay wave_ray(const DevModel &M, const gpis_ray_in &r)
{
    WaveRay w;
    w.pos = v3(r.pos[0], r.pos[1], r.pos[2]);
    w.dir = v3(r.dir[0], r.dir[1], r.dir[2]);
    w.nearT = r.near_t; w.farT = r.far_t; w.u_jitter = r.u_jitter;
    w.first_scatter = r.first_scatter != 0;
    w.bounce = r.bounce;
    if (!__builtin_isfinite(w.farT))
        w.farT = (float)((double)w.nearT + 2000);
    w.step_size = (w.farT - w.nearT) / (float)M.min_step;
    if (M.step_size < w.step_size)
        w.step_size = M.step_size;
    return w;
}
GPIS_DEV Frame wave_frame(const DevModel &M, V3 dir)
{
    Frame c{};
    if (M.iso3d)
        c = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, dir, 1.0f)));
    return c;
}
GPIS_DEV V3 wave_point(const WaveRay &w, double tq) { return to_f(ray_at(to_d(w.pos), to_d(w.dir), tq)); }
// position of the exact value a parked ray waits for
GPIS_DEV double wave_request_t(const WaveRay &w, const WaveState &s)
{
    return s.phase == X_F0 ? (double)w.nearT : (s.phase == X_PREV ? s.t_prevpos : (s.phase == X_REFINE ? s.t_test : (s.phase == X_FINAL ? (double)w.farT : s.t)));
}
GPIS_DEV uint32_t wave_spread3(uint32_t v)   // 10 bits -> every third bit
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
// Morton code of the lattice cell of grid point u (cells clamped to +-511); never 0xFFFFFFFF
GPIS_DEV uint32_t wave_key(V3 u)
{
    const int cx = (int)floorf(fminf(fmaxf(u.x, -511.f), 511.f)) + 512;
    const int cy = (int)floorf(fminf(fmaxf(u.y, -511.f), 511.f)) + 512;
    const int cz = (int)floorf(fminf(fmaxf(u.z, -511.f), 511.f)) + 512;
    return (wave_spread3((uint32_t)cx) << 2) | (wave_spread3((uint32_t)cy) << 1) | wave_spread3((uint32_t)cz);
}

// ---- the per-ray state machine (guided_march's transitions, on a WaveState) ------------------------
GPIS_DEV void wave_begin_march(const WaveRay &w, WaveState &s)      // SCNM.cpp:129
{
    s.t_prevpos = (double)w.nearT;
    s.t = (double)(w.nearT + w.step_size * w.u_jitter);
    s.phase = (s.t < (double)w.farT) ? G_MARCH : X_FINAL;
}
GPIS_DEV void wave_advance(const WaveRay &w, WaveState &s)
{
    s.t_prevpos = s.t;
    s.t += (double)w.step_size;
    s.phase = (s.t < (double)w.farT) ? G_MARCH : X_FINAL;
}
GPIS_DEV void wave_begin_refine(const WaveRay &w, WaveState &s)     // SCNM.cpp:143-146
{
    s.intp = (double)s.pf / ((double)s.pf - (double)s.fc);
    const double a_lo = s.t - (double)w.step_size;
    s.t_prev = lerp_d(a_lo, s.t, s.intp);
    s.t_test = s.t_prev;
    s.phase = X_REFINE;
}
GPIS_DEV void wave_init(const DevModel &M, const WaveRay &w, bool masked_out, bool want_sample, WaveState &s)
{
    memset(&s, 0, sizeof s);
    s.phase = G_INIT;
    if (masked_out)
        s.phase = G_DONE;
    else if (want_sample && w.bounce >= M.max_bounces)
        s.phase = G_DONE;
    else if (want_sample && w.farT == 0.f) {
        s.phase = G_DONE;
        s.flags |= kWaveEarlyOk;
    }
    s.t = (double)w.nearT;
    s.t_prevpos = (double)w.nearT;
    s.sign0 = 1;
}
// the transitions of guided_march's part B on a delivered exact value (fv, gp)
template <bool WANT_SAMPLE>
GPIS_DEV void wave_consume(const WaveRay &w, WaveState &s, float fv, int gp)
{
    const double f = (double)fv;
    s.gp = (int8_t)gp;
    if (s.phase == X_F0) {                       // SCNM.cpp:125-128
        s.sign0 = f < 0 ? -1 : 1;
        s.pf = fv; s.flags |= kWavePfValid;
        wave_begin_march(w, s);
    } else if (s.phase == X_CUR) {               // SCNM.cpp:133-141, 172-173
        s.step++;
        const int signc = f < 0 ? -1 : 1;
        if (!w.first_scatter && s.step == 1) {
            s.sign0 = (int8_t)signc;
            s.pf = fv; s.flags |= kWavePfValid;
            wave_advance(w, s);
        } else if (signc != s.sign0) {
            s.fc = fv;
            if (s.flags & kWavePfValid) wave_begin_refine(w, s);
            else s.phase = X_PREV;
        } else {
            s.pf = fv; s.flags |= kWavePfValid;
            wave_advance(w, s);
        }
    } else if (s.phase == X_PREV) {
        s.pf = fv; s.flags |= kWavePfValid;
        wave_begin_refine(w, s);
    } else if (s.phase == X_REFINE) {            // SCNM.cpp:147-160
        const int sign_test = f < 0 ? -1 : 1;
        bool done = false;
        if (sign_test == s.sign0) {
            done = true;
        } else {
            s.intp *= 0.9;
            if (s.intp <= 0.01) {
                s.t_prev = s.t_test = 0;
                done = true;
            } else {
                s.t_prev = s.t_test;
                s.t_test = lerp_d(s.t - (double)w.step_size, s.t, s.intp);
            }
        }
        if (done) {
            s.t = s.t_prev;
            s.flags |= kWaveHit;
            s.last_val = 0.0f;
            s.phase = WANT_SAMPLE ? G_GRAD : G_DONE;
        }
    } else if (s.phase == X_FINAL) {             // lastVal at farT (SCNM.cpp:176-181)
        s.t = (double)w.farT;
        s.last_val = fv;
        s.flags &= (uint8_t)~kWaveHit;
        s.phase = G_GRAD;
    }
}
// guide steps (guided_march's part A) until an exact value is needed or the segment ends
template <bool WANT_SAMPLE>
GPIS_DEV void wave_guide_steps(const DevModel &M, const GuideField &F, const WaveRay &w, WaveState &s, uint32_t &n_guide)
{
    if (!(s.phase == G_INIT || s.phase == G_MARCH || s.phase == X_FINAL))
        return;
    const GuideRay gr = guide_ray(M, F, w.pos, w.dir, wave_frame(M, w.dir), w.nearT);
    for (;;) {
        if (!WANT_SAMPLE && s.phase == X_FINAL) {
            s.flags &= (uint8_t)~kWaveHit;      // transmittance: the segment exits, lastVal is not part of the result
            s.phase = G_DONE;
        }
        if (s.phase == G_INIT) {
            const int sg = guide_sign_at(M, F, gr, (double)w.nearT);
            if (sg != 0) {
                n_guide++;
                s.sign0 = (int8_t)sg;
                s.flags &= (uint8_t)~kWavePfValid;
                wave_begin_march(w, s);
            } else {
                s.phase = X_F0;
            }
        } else if (s.phase == G_MARCH) {
            const int sg = guide_sign_at(M, F, gr, s.t);
            const bool adopt = !w.first_scatter && s.step == 0;   // SCNM.cpp:138-140
            if (sg != 0 && (adopt || sg == s.sign0)) {
                n_guide++;
                s.step++;
                if (adopt) s.sign0 = (int8_t)sg;
                s.flags &= (uint8_t)~kWavePfValid;
                wave_advance(w, s);
            } else {
                s.phase = X_CUR;
            }
        } else {
            break;
        }
    }
}

// ---- step: consume a delivered value, march on the guide, emit the next request -----------------
// slot j of the active list is ray i = active[j] (identity in the first iteration, which also
// initialises the state).  keys/vals[j] receive the request (or the largest key when the ray needs
// nothing more from the value evaluator).
template <bool WANT_SAMPLE>
__global__ void __launch_bounds__(256) k_wave_step(const DevModel *__restrict__ Mp, GuideField F, size_t n_active, const uint32_t *__restrict__ active,
                                                   int init, const gpis_ray_in *__restrict__ rays, const uint8_t *__restrict__ mask,
                                                   WaveState *__restrict__ state, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                   unsigned long long *__restrict__ req_counter, unsigned long long *__restrict__ guide_counter)
{
    const DevModel &M = *Mp;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_guide = 0;
    bool request = false;
    if (j < n_active) {
        const size_t i = active ? (size_t)active[j] : j;
        const WaveRay w = wave_ray(M, rays[i]);
        WaveState s;
        if (init) {
            wave_init(M, w, mask && !mask[i], WANT_SAMPLE, s);
        } else {
            s = state[i];
            if (s.phase >= X_F0 && s.phase <= X_FINAL)
                wave_consume<WANT_SAMPLE>(w, s, s.fv, (int)s.gp_new);
        }
        wave_guide_steps<WANT_SAMPLE>(M, F, w, s, n_guide);
        state[i] = s;
        uint32_t key = 0xFFFFFFFFu;
        if (s.phase >= X_F0 && s.phase <= X_FINAL) {
            key = wave_key(grid_point(M, F, wave_point(w, wave_request_t(w, s)), wave_frame(M, w.dir)));
            request = true;
        }
        keys[j] = key;
        vals[j] = (uint32_t)i;
    }
    // two atomics per wave
    const unsigned long long rq = __ballot(request);
    unsigned long long g = n_guide;
    for (int off = 32; off > 0; off >>= 1)
        g += __shfl_down(g, off, 64);
    if ((threadIdx.x & 63) == 0) {
        if (rq) atomicAdd(req_counter, (unsigned long long)__popcll(rq));
        if (g) atomicAdd(guide_counter, g);
    }
}

// ---- tail: when few rays are left, one wave finishes one ray ------------------------------------------
// The tail of the iteration (rays in long refinement chains, SCNM.cpp:147-160 shrinks up to 44 times) is
// latency, not throughput: a launch per value with a handful of waves.  Here every lane of a wave runs the
// same ray's state machine (uniform control flow, broadcast loads) and the 64 lanes share each exact
// value through the sideways evaluator.
template <bool WANT_SAMPLE>
__global__ void __launch_bounds__(kFastBlock) k_wave_tail(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n_active,
                                                         const uint32_t *__restrict__ active, const gpis_ray_in *__restrict__ rays,
                                                         WaveState *__restrict__ state, Counters *cnt, unsigned long long *__restrict__ guide_counter)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    const size_t i = (size_t)active[blockIdx.x];
    const WaveRay w = wave_ray(M, rays[i]);
    WaveState s = state[i];
    const Frame coord = wave_frame(M, w.dir);
    uint32_t n_eval = 0, n_guide = 0;
    // the value of the request the ray arrived with is already in s.fv
    if (s.phase >= X_F0 && s.phase <= X_FINAL)
        wave_consume<WANT_SAMPLE>(w, s, s.fv, (int)s.gp_new);
    for (;;) {
        wave_guide_steps<WANT_SAMPLE>(M, F, w, s, n_guide);
        if (!(s.phase >= X_F0 && s.phase <= X_FINAL))
            break;
        const V3 pq = wave_point(w, wave_request_t(w, s));
        int gp;
        const float fv = solo_evaluate_value<true>(M, T, lds, 0, pq, coord, gp, n_eval);
        const float fv0 = lane_f(fv, 0);
        wave_consume<WANT_SAMPLE>(w, s, fv0, __builtin_amdgcn_readfirstlane(gp));
    }
    if ((threadIdx.x & 63) == 0) {
        state[i] = s;
        if (n_eval) atomicAdd(&cnt->n_eval, (unsigned long long)n_eval);
        if (n_guide) atomicAdd(guide_counter, (unsigned long long)n_guide);
    }
}

// ---- eval: 64 consecutive sorted requests per wave -------------------------------------------------
template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_FAST_OCC) k_wave_eval(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n_req,
                                                                        const uint32_t *__restrict__ req, const gpis_ray_in *__restrict__ rays,
                                                                        WaveState *__restrict__ state, Counters *cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    const size_t k = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = k < n_req;
    const size_t i = valid ? (size_t)req[k] : 0;
    V3 pq = v3(0.f, 0.f, 0.f);
    Frame coord{};
    if (valid) {
        const WaveRay w = wave_ray(M, rays[i]);
        const WaveState s = state[i];
        pq = wave_point(w, wave_request_t(w, s));
        coord = wave_frame(M, w.dir);
    }
    const V3 ug = grid_point(M, F, pq, coord);
    const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
    uint32_t n_eval = 0;
    bool pending = valid;
    float fv = 0.f;
    int gp = 0;
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ULL)
            break;
        const int lead = __builtin_ctzll(pm);
        const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
        const bool in_cluster = pending && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
        const unsigned long long cl_mask = __ballot(in_cluster);
        if (__popcll(cl_mask) <= kSoloMaxLanes) {
            for (unsigned long long mm = cl_mask; mm; mm &= mm - 1ULL) {
                const int src = __builtin_ctzll(mm);
                int gpx;
                const float v = solo_evaluate_value<true>(M, T, lds, src, pq, coord, gpx, n_eval);
                if ((int)(threadIdx.x & 63) == src) { fv = v; gp = gpx; }
            }
        } else {
            int gpx;
            const float v = coop_evaluate_value<SMALLARG>(M, T, lds, in_cluster, pq, coord, gpx, n_eval);
            if (in_cluster) { fv = v; gp = gpx; }
        }
        if (in_cluster)
            pending = false;
    }
    if (valid) {
        state[i].fv = fv;
        state[i].gp_new = (int8_t)gp;
    }
    fast_flush_counters(cnt, n_eval, 0u);
}

// ---- gradient stage of sampleDistance (GPM.cpp:283 / 319) ------------------------------------------
GPIS_DEV V3 wave_grad_point(const WaveRay &w, double t)
{
    V3d rdn = to_d(w.dir);
    { double inv = 1.0 / length_d(rdn); rdn.x *= inv; rdn.y *= inv; rdn.z *= inv; }
    return to_f(ray_at(to_d(w.pos), rdn, t));
}
__global__ void __launch_bounds__(256) k_wave_grad_keys(const DevModel *__restrict__ Mp, GuideField F, size_t n, const gpis_ray_in *__restrict__ rays,
                                                        const WaveState *__restrict__ state, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                        unsigned long long *__restrict__ req_counter)
{
    const DevModel &M = *Mp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool want = false;
    if (i < n) {
        uint32_t key = 0xFFFFFFFFu;
        if (state[i].phase == G_GRAD) {
            const WaveRay w = wave_ray(M, rays[i]);
            key = wave_key(grid_point(M, F, wave_grad_point(w, state[i].t), wave_frame(M, w.dir)));
            want = true;
        }
        keys[i] = key;
        vals[i] = (uint32_t)i;
    }
    const unsigned long long rq = __ballot(want);
    if ((threadIdx.x & 63) == 0 && rq)
        atomicAdd(req_counter, (unsigned long long)__popcll(rq));
}
template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_FAST_OCC) k_wave_grad(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n_req,
                                                                        const uint32_t *__restrict__ req, const gpis_ray_in *__restrict__ rays,
                                                                        WaveState *__restrict__ state, Counters *cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    const size_t k = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = k < n_req;
    const size_t i = valid ? (size_t)req[k] : 0;
    V3 pq = v3(0.f, 0.f, 0.f);
    Frame coord{};
    if (valid) {
        const WaveRay w = wave_ray(M, rays[i]);
        pq = wave_grad_point(w, state[i].t);
        coord = wave_frame(M, w.dir);
    }
    const V3 ug = grid_point(M, F, pq, coord);
    const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
    uint32_t n_eval = 0;
    bool pending = valid;
    V3 g = v3(0.f, 0.f, 0.f);
    for (;;) {
        const unsigned long long pm = __ballot(pending);
        if (pm == 0ULL)
            break;
        const int lead = __builtin_ctzll(pm);
        const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
        const bool in_cluster = pending && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
        const V3 gi = coop_evaluate_gradient<SMALLARG>(M, T, lds, in_cluster, pq, coord, n_eval);
        if (in_cluster) {
            g = gi;
            pending = false;
        }
    }
    if (valid) {
        state[i].fv = g.x; state[i].gy = g.y; state[i].gz = g.z;
    }
    fast_flush_counters(cnt, n_eval, 0u);
}

// ---- results ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wave_finish_sd(const DevModel *__restrict__ Mp, size_t n, const gpis_ray_in *__restrict__ rays,
                                                        const uint8_t *__restrict__ mask, const WaveState *__restrict__ state,
                                                        gpis_seg_out *__restrict__ out, gpis_cond_coeff *__restrict__ coeff, Counters *cnt)
{
    const DevModel &M = *Mp;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    if (valid) {
        const WaveRay w = wave_ray(M, rays[i]);
        const WaveState s = state[i];
        const bool want_grad = s.phase == G_GRAD;
        finish_sample_distance(M, rays + i, w.pos, w.dir, w.farT, (s.flags & kWaveEarlyOk) != 0, want_grad, (s.flags & kWaveHit) != 0, s.t, s.last_val,
                               (int)s.gp, v3(s.fv, s.gy, s.gz), out + i);
        if (coeff) {
            gpis_cond_coeff c;
            memset(&c, 0, sizeof c);
            coeff[i] = c;
        }
    }
    fast_flush_counters(cnt, 0u, valid ? 1u : 0u);
}
__global__ void __launch_bounds__(256) k_wave_finish_tr(size_t n, const uint8_t *__restrict__ mask, const WaveState *__restrict__ state,
                                                        uint8_t *__restrict__ visible, Counters *cnt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    if (i < n)
        visible[i] = (valid && !(state[i].flags & kWaveHit)) ? 1 : 0;
    fast_flush_counters(cnt, 0u, valid ? 1u : 0u);
}

}   // namespace gpis
