// gpis_fast.hpp — wave-cooperative fast path for single-realization media (configs C0, C1).
//
// With single_realization the seed of every cell is the same for every ray (SCN.cpp:40-42), so the
// 64 rays of a wave that sit next to each other (consecutive spp of one pixel) need the SAME cells.
// Instead of every lane regenerating 27 x rho impulses per evaluation, the wave
//   1. reduces the lanes' grid positions to a bounding box (min/max over the wave),
//   2. walks the union of the lanes' 3x3x3 neighbourhoods in lexicographic (x, y, z) order — which
//      is every lane's own dx,dy,dz order (SCN.cpp:368-371), so each lane still accumulates its
//      cells in the reference's order,
//   3. generates each cell's impulse list ONCE, cooperatively: lane k jumps the cell's PCG32 stream
//      ahead by 4k draws (LCG jump-ahead, state_k = A[k]*s0 + C[k] mod 2^64) and produces impulse k
//      exactly as the sequential generator would (SCN.cpp:376-385); impulse k stays in lane k's
//      registers,
//   4. culls: lane k tests its impulse against the box of query points (sphere/box distance with
//      slack), a ballot gives the wave-uniform candidate mask,
//   5. consumes: for each candidate (scalar bit loop) the impulse is broadcast with v_readlane and
//      every lane applies the reference's unit-ball test and kernel in the reference's per-lane
//      summation order (SCN.cpp:383-392).
// A wave whose lanes are NOT close together (box wider than 2 cells on an axis) falls back to the
// per-lane generator for that evaluation: results never depend on coherence, only speed does.
//
// The march (SCNM.cpp:102-183) runs as a per-lane phase machine driven in lockstep: every
// iteration is one cooperative value evaluation for all lanes that still march / refine; the one
// gradient evaluation each segment ends with (GPM.cpp:283, 319) is deferred until no lane marches
// any more and done once for the whole wave.  During the march only the noise VALUE is summed (the
// reference computes and discards the gradient there: SCN.cpp:76, SCN.hpp:62).
#pragma once
#include "gpis_device.hpp"

#pragma clang fp contract(off)

namespace gpis {

// Cell table in HBM (served from L2 / Infinity Cache): the impulses of every cell (i, j, k) with
// -H <= i, j, k < H, as float4 (x, y, z, w=+-1) in generation order, `stride` slots per cell.  A wave
// reads a cell with ONE coalesced 512-B / 1-KiB load; cells outside the table are generated on the
// fly, so the table is a cache and never a correctness bound.
struct FastTable {
    float4 *cells;   // nullptr = every cell is generated per wave
    int half;        // H
    int stride;      // 32 or 64 slots per cell
    int enabled;
};

inline bool fast_supported(const DevModel &M)
{
    return M.single_realization && !M.sampling_1d && !M.nonstationary && !M.use_aniso_mtx && !M.absorption_only && !M.color.enabled && M.kernel_type == GPIS_KERNEL_SQUARED_EXPONENTIAL &&
           M.n_impulses >= 1 && M.n_impulses <= 64;
}
GPIS_TU_KERNEL __global__ void __launch_bounds__(64) k_fast_build_table(uint32_t seed, int half, int stride, float4 *__restrict__ cells)
{
    const int lane = (int)(threadIdx.x & 63);
    const int side = 2 * half;
    const size_t cell = blockIdx.x;
    const int ck = (int)(cell % side) - half, cj = (int)((cell / side) % side) - half, ci = (int)(cell / ((size_t)side * side)) - half;
    float px, py, pz, pw;
    gen_impulse((uint32_t)ci, (uint32_t)cj, (uint32_t)ck, seed, kJump4.A[lane], kJump4.C[lane], px, py, pz, pw);
    if (lane < stride)
        cells[cell * (size_t)stride + lane] = make_float4(px, py, pz, pw);
}
inline int fast_table_build(const DevModel &M, const DevModel *, FastTable *t)
{
    t->cells = nullptr;
    t->half = 0;
    t->stride = 0;
    t->enabled = fast_supported(M) && !getenv("GPIS_DISABLE_FAST");
    if (!t->enabled || getenv("GPIS_DISABLE_TABLE"))
        return GPIS_OK;
    int half = 16;   // covers |p| < 16 cells: scene S spans 14.2 cells (1.5 world units / cell size 0.106)
    if (const char *e = getenv("GPIS_TABLE_HALF_EXTENT")) half = atoi(e);
    if (half < 2) return GPIS_OK;
    if (half > 40) half = 40;
    const int stride = M.n_impulses <= 32 ? 32 : 64;
    const size_t ncell = (size_t)(2 * half) * (2 * half) * (2 * half);
    if (hipMalloc(&t->cells, ncell * stride * sizeof(float4)) != hipSuccess) {
        t->cells = nullptr;   // no table: fall back to per-wave generation
        (void)hipGetLastError();
        return GPIS_OK;
    }
    k_fast_build_table<0><<<(unsigned)ncell, 64>>>(M.seed, half, stride, t->cells);
    if (hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(t->cells);
        t->cells = nullptr;
        return GPIS_ERR_DEVICE;
    }
    t->half = half;
    t->stride = stride;
    return GPIS_OK;
}
inline void fast_table_free(FastTable *t)
{
    if (t->cells) (void)hipFree(t->cells);
    t->cells = nullptr;
    t->enabled = 0;
}

// ---- wave reductions (wave64) ----------------------------------------------------------------
GPIS_DEV float wave_min_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
GPIS_DEV float wave_max_f(float v)
{
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Diagnostic build only (-DGPIS_FAST_STATS): wave-level work counters of the cooperative loop.
// -DGPIS_DUP=k duplicates one part of the loop (result kept alive, outputs unchanged) so that
// t(dup k) - t(base) prices that part without changing control flow:
//   1 expf   2 whole body   3 unit-ball test   4 per-cell load + cull   5 per-evaluation prologue
#ifndef GPIS_DUP
#define GPIS_DUP 0
#endif
#define GPIS_KEEP(x) asm volatile("" ::"v"(x))
#ifdef GPIS_FAST_STATS
static __device__ unsigned long long g_fast_stats[32];   // one copy per translation unit (diagnostic builds)
#define FSTAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_fast_stats[i], (unsigned long long)(v)); } while (0)
#define FCLK() ((long long)__builtin_readcyclecounter())
#else
#define FSTAT(i, v) do { } while (0)
#define FCLK() 0LL
#endif

constexpr int kFastBlock = 64;          // one wave per workgroup
#ifndef GPIS_KSPLIT
#define GPIS_KSPLIT 1
#endif
#ifndef GPIS_KSPLIT_MAXLG
#define GPIS_KSPLIT_MAXLG 4
#endif
constexpr int kKSplitMaxLg = GPIS_KSPLIT_MAXLG;   // at most 2^this lanes per query
#ifndef GPIS_KSPLIT_MAXNA
#define GPIS_KSPLIT_MAXNA 16
#endif
constexpr int kKSplitMaxNa = GPIS_KSPLIT_MAXNA;   // clusters above this size keep the two-way split (one ds_bpermute per pass beats the LDS exchange at k = 2)
constexpr bool kKSplit = GPIS_KSPLIT != 0;   // clusters of at most 32 queries: k-way candidate split (coop_noise3d_ksplit) instead of the two-way one
#ifndef GPIS_FAST_OCC
#define GPIS_FAST_OCC 3                 // waves per SIMD the register allocator must leave room for
#endif
// cold path: every lane generates its own impulses
GPIS_DEV V4 noise3d_per_lane(const DevModel &M, V3 p, uint32_t seed, float R, float A0, float A1, float A2)
{
    float A[9] = {A0, 0.f, 0.f, 0.f, A1, 0.f, 0.f, 0.f, A2};
    return spec_3d::noise3d(M, p, seed, R, A, true);
}

// One cooperative noise3D (SCN.cpp:362-395) for the lanes with `active` set.
//   p      : the lane's query point in the space the grid lives in (world, or isotropic-ray space)
//   R      : kernelRadius (wave-uniform);  A0..A2: diagonal of getInvCovMtx (wave-uniform)
// Returns (value, gradient) — gradient only when GRAD.
// SMALLARG: max_a(A_a) * R^2 < 100 for this medium (DevModel::exp_arg_max, checked by the launcher): a lane
// that passes the unit-ball test has ab^T A ab < 100, far from expf's underflow threshold (103.97), and the
// value computed for a failing lane is discarded — the underflow select of expf is dead code there.
// SPLIT (value only, wave-uniform `split`): when at most 32 lanes hold a query, every query gets a HELPER lane from the
// idle half of the wave.  Candidates are consumed two per pass as before — cell = (cell + c(k0)) + c(k1) — but the
// owner evaluates c(k0) while its helper evaluates c(k1) on a copy of the same query point, and one ds_bpermute hands
// c(k1) over: the same values added in the same order, with half the kernel-body instructions per query
// (the exact rounds of the guided march run with ~30 of 64 lanes).
template <bool GRAD, bool SMALLARG = false>
GPIS_DEV V4 coop_noise3d(const DevModel &M, const FastTable &T, FastLds &lds, bool active, V3 p, uint32_t seed, float R, float A0, float A1, float A2,
                         bool split = false)
{
    const int lane = (int)(threadIdx.x & 63);
    V3 pg = p / R;
    V3 fl = v3(floorf(pg.x), floorf(pg.y), floorf(pg.z));
    V3 frac = pg - fl;
    int ci0 = (int)fl.x, cj0 = (int)fl.y, ck0 = (int)fl.z;
    bool helper = false;
    int partner = lane;
    if (!GRAD && split) {
        const unsigned long long am = __ballot(active);
        const int na = __popcll(am);
        const int rank_a = __popcll(am & ((1ULL << lane) - 1ULL)), rank_i = lane - rank_a;
        if (active) lds.owner_tab[rank_a] = (uint32_t)lane;
        else lds.idle_tab[rank_i] = (uint32_t)lane;
        helper = !active && rank_i < na;
        // one wave per workgroup and in-order LDS: only the compiler has to be told that lanes talk to each other here
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        partner = active ? (int)lds.idle_tab[rank_a] : (helper ? (int)lds.owner_tab[rank_i] : lane);
        // the helper works on a copy of its owner's query
        const float hx = __shfl(frac.x, partner, 64), hy = __shfl(frac.y, partner, 64), hz = __shfl(frac.z, partner, 64);
        const int hi = __shfl(ci0, partner, 64), hj = __shfl(cj0, partner, 64), hk = __shfl(ck0, partner, 64);
        if (helper) { frac = v3(hx, hy, hz); ci0 = hi; cj0 = hj; ck0 = hk; }
    }
    const bool act2 = active || helper;

    // bounding box of the active lanes' grid positions (inactive lanes contribute nothing)
    const float big = 3.0e38f;
    const float bx0 = uni_f(wave_min_f(active ? pg.x : big)), bx1 = uni_f(wave_max_f(active ? pg.x : -big));
    const float by0 = uni_f(wave_min_f(active ? pg.y : big)), by1 = uni_f(wave_max_f(active ? pg.y : -big));
    const float bz0 = uni_f(wave_min_f(active ? pg.z : big)), bz1 = uni_f(wave_max_f(active ? pg.z : -big));
    if (!(bx0 <= bx1))
        return v4(0.f, 0.f, 0.f, 0.f);   // no active lane in this wave
#if GPIS_DUP == 5
    {
        V3 pg2 = (p + v3(1e-3f, 1e-3f, 1e-3f)) / R;
        float a0 = uni_f(wave_min_f(active ? pg2.x : big)), a1 = uni_f(wave_max_f(active ? pg2.x : -big));
        float b0 = uni_f(wave_min_f(active ? pg2.y : big)), b1 = uni_f(wave_max_f(active ? pg2.y : -big));
        float c0 = uni_f(wave_min_f(active ? pg2.z : big)), c1 = uni_f(wave_max_f(active ? pg2.z : -big));
        GPIS_KEEP(a0 + a1 + b0 + b1 + c0 + c1 + floorf(pg2.x) + floorf(pg2.y) + floorf(pg2.z));
    }
#endif
    FSTAT(0, 1);
#ifdef GPIS_FAST_STATS
    {   // histogram of active lanes per cooperative evaluation: bins 1-2, 3-4, 5-8, 9-16, 17-32, 33-64
        const int na = __popcll(__ballot(active));
        const int bin = na <= 2 ? 0 : na <= 4 ? 1 : na <= 8 ? 2 : na <= 16 ? 3 : na <= 32 ? 4 : 5;
        FSTAT(9 + bin, 1);
        FSTAT(1, na);
    }
#endif
    // cells are addressed with int coordinates; keep far away from overflow and from float→int UB
    const float lim = 1.0e6f;
    bool sane = bx0 > -lim && bx1 < lim && by0 > -lim && by1 < lim && bz0 > -lim && bz1 < lim;
    int ilo = 0, ihi = 0, jlo = 0, jhi = 0, klo = 0, khi = 0;
    if (sane) {
        ilo = (int)floorf(bx0); ihi = (int)floorf(bx1);
        jlo = (int)floorf(by0); jhi = (int)floorf(by1);
        klo = (int)floorf(bz0); khi = (int)floorf(bz1);
    }
    const bool coherent = sane && (ihi - ilo) <= 1 && (jhi - jlo) <= 1 && (khi - klo) <= 1;
    if (!coherent) {
        // incoherent wave (or non-finite input): every lane generates its own impulses
        V4 r = v4(0.f, 0.f, 0.f, 0.f);
        FSTAT(8, 1);
        if (active)
            r = noise3d_per_lane(M, p, seed, R, A0, A1, A2);
        return r;
    }

    const uint32_t n = M.n_impulses;
    // the union of the lanes' 3x3x3 neighbourhoods, walked in lexicographic (x, y, z) order
    const int nj = jhi - jlo + 3, nk = khi - klo + 3;
    const int total = (ihi - ilo + 3) * nj * nk;          // 27 .. 64 cells
    const int H = T.half, S = T.stride;
    const unsigned side = 2u * (unsigned)H;
    const float4 *cells = T.cells;
    auto in_table = [&](int ci, int cj, int ck) {
        return cells && (unsigned)(ci + H) < side && (unsigned)(cj + H) < side && (unsigned)(ck + H) < side;
    };
    auto fetch = [&](int ci, int cj, int ck) {
        const size_t idx = (((size_t)(ci + H) * side + (size_t)(cj + H)) * side + (size_t)(ck + H)) * (size_t)S;
        return cells[idx + (size_t)(lane & (S - 1))];
    };
    int ci = ilo - 1, cj = jlo - 1, ck = klo - 1;
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_table(ci, cj, ck))
        nxt = fetch(ci, cj, ck);
    V4 sum = v4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < total; ++c) {
        const int ti = ci, tj = cj, tk = ck;             // this cell
        const float4 cur = nxt;
        // advance (ci, cj, ck) and prefetch the next cell's impulses while this one is consumed
        if (++ck > khi + 1) { ck = klo - 1; if (++cj > jhi + 1) { cj = jlo - 1; ++ci; } }
        if (c + 1 < total && in_table(ci, cj, ck))
            nxt = fetch(ci, cj, ck);
        const int di = ti - ci0, dj = tj - cj0, dk = tk - ck0;
        const bool mine = act2 && di >= -1 && di <= 1 && dj >= -1 && dj <= 1 && dk >= -1 && dk <= 1;
        FSTAT(2, 1);
        if (__ballot(mine) == 0ULL)
            continue;
        FSTAT(3, 1);
        float px, py, pz, pw;
        if (in_table(ti, tj, tk)) {
            px = cur.x; py = cur.y; pz = cur.z; pw = cur.w;
        } else {
            gen_impulse((uint32_t)ti, (uint32_t)tj, (uint32_t)tk, seed, kJump4.A[lane], kJump4.C[lane], px, py, pz, pw);
        }
        // cull: sphere (radius 1) vs box-of-queries distance, with slack for the differently rounded test
        const float qx0 = bx0 - (float)ti, qx1 = bx1 - (float)ti;
        const float qy0 = by0 - (float)tj, qy1 = by1 - (float)tj;
        const float qz0 = bz0 - (float)tk, qz1 = bz1 - (float)tk;
        float gx = fmaxf(fmaxf(qx0 - px, px - qx1), 0.f);
        float gy = fmaxf(fmaxf(qy0 - py, py - qy1), 0.f);
        float gz = fmaxf(fmaxf(qz0 - pz, pz - qz1), 0.f);
        bool cand_k = (uint32_t)lane < n && (gx * gx + gy * gy + gz * gz) < 1.0001f;
        unsigned long long cand = __ballot(cand_k);
#if GPIS_DUP == 4
        {
            float4 e = in_table(ti, tj, tk) ? fetch(ti, tj, tk ^ 1) : make_float4(px, py, pz, pw);
            float hx = fmaxf(fmaxf(qx0 - e.x, e.x - qx1), 0.f), hy = fmaxf(fmaxf(qy0 - e.y, e.y - qy1), 0.f), hz = fmaxf(fmaxf(qz0 - e.z, e.z - qz1), 0.f);
            unsigned long long c2 = __ballot((uint32_t)lane < n && (hx * hx + hy * hy + hz * hz) < 1.0001f);
            GPIS_KEEP((float)(unsigned)(c2 >> 7) + R * e.x + R * e.y + R * e.z);
        }
#endif
        if (cand == 0ULL)
            continue;
        FSTAT(4, 1);
        FSTAT(5, __popcll(cand));
        // --- consume (SCN.cpp:383-392), candidates in ascending k = the reference's order.  Impulse k
        //     lives in lane k's registers and is broadcast with v_readlane; every broadcast sits in
        //     wave-uniform control flow (a lane that is masked off may have had its registers spilled
        //     and not restored).  The body is branch-free: a lane whose unit-ball test fails adds +0,
        //     which is exact, so two candidates can be in flight at once (independent f64 chains).
        const float rx = R * px, ry = R * py, rz = R * pz;   // kernelRadius * p_i, once per impulse
        // lanes whose neighbourhood does not contain this cell get a far-away point: the unit-ball
        // test fails for them and their cell sum stays +0
        const V3 pc = mine ? frac - v3((float)di, (float)dj, (float)dk) : v3(8.f, 8.f, 8.f);
        const V3 Rp = R * pc;
#if GPIS_LDS_BCAST
        // The kernel is VALU-issue bound and the 7 v_readlane per candidate are VALU instructions; an LDS
        // read at a wave-uniform address is a broadcast that issues on the LDS pipe instead.  One wave per
        // workgroup, LDS operations of a wave execute in order: no barrier between the stores and the loads.
        lds.xyr[lane] = make_float4(px, py, rx, ry);
        lds.zw[lane] = make_float4(pz, rz, pw, 0.f);
        // the body in packed fp32 pairs; every element is the reference's own operation:
        //   lengthSq(pc - p_i) = (dx^2 + dy^2) + dz^2          (Vec.hpp:200-206)
        //   ab^T A ab          = tx*abx + (ty*aby + tz*abz)    (Eigen 3-term reduction), t = ab * diag(A)
        const float2v pcxy = float2v{pc.x, pc.y}, Rpxy = float2v{Rp.x, Rp.y}, pzRz = float2v{pc.z, Rp.z}, Axy = float2v{A0, A1};
        auto contribution = [&](int k) {
            const float4 qa = lds.xyr[k], qb = lds.zw[k];
            const float2v ixy = float2v{qa.x, qa.y}, rxy = float2v{qa.z, qa.w}, zrz = float2v{qb.x, qb.y};
            const float iw = qb.z;
            const float2v dxy = pcxy - ixy;             // (pc.x - x, pc.y - y)
            const float2v abxy = Rpxy - rxy;            // (ab.x, ab.y)
            const float2v dzab = pzRz - zrz;            // (pc.z - z, ab.z)
            const float2v d2 = dxy * dxy;
            const float2v txy = abxy * Axy;             // (t.x, t.y)
            const float2v tab = txy * abxy;
            const float tz = dzab.y * A2;
            const float2v zz = float2v{dzab.x, tz} * dzab;     // (dz^2, t.z * ab.z)
            const bool pass = ((d2.x + d2.y) + zz.x) < 1.0f;
            const float absq = tab.x + (tab.y + zz.y);
            const float f = expf_glibc_lds<true, SMALLARG>(lds, -absq);
            V4 r;
            r.v = pass ? iw * f : 0.f;
            if (GRAD) {
                r.gx = pass ? iw * (-2.f * txy.x * f) : 0.f;
                r.gy = pass ? iw * (-2.f * txy.y * f) : 0.f;
                r.gz = pass ? iw * (-2.f * tz * f) : 0.f;
            } else {
                r.gx = r.gy = r.gz = 0.f;
            }
            return r;
        };
#else
        auto contribution = [&](int k) {
            const float ix = lane_f(px, k), iy = lane_f(py, k), iz = lane_f(pz, k), iw = lane_f(pw, k);
            const float jx = lane_f(rx, k), jy = lane_f(ry, k), jz = lane_f(rz, k);
            const bool pass = length_sq(pc - v3(ix, iy, iz)) < 1.0f;
            const V3 ab = Rp - v3(jx, jy, jz);
            const V3 t = v3(ab.x * A0, ab.y * A1, ab.z * A2);
            const float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
            const float f = expf_glibc_lds(lds, -absq);
            V4 r;
            r.v = pass ? iw * f : 0.f;
            if (GRAD) {
                r.gx = pass ? iw * (-2.f * t.x * f) : 0.f;
                r.gy = pass ? iw * (-2.f * t.y * f) : 0.f;
                r.gz = pass ? iw * (-2.f * t.z * f) : 0.f;
            } else {
                r.gx = r.gy = r.gz = 0.f;
            }
            return r;
        };
#endif
        V4 cell = v4(0.f, 0.f, 0.f, 0.f);
        if (!GRAD && split) {
            while (cand) {
                const int k0 = __builtin_ctzll(cand);
                cand &= cand - 1ULL;
                int k1 = -1;
                if (cand) { k1 = __builtin_ctzll(cand); cand &= cand - 1ULL; }
                const float c = contribution((helper && k1 >= 0) ? k1 : k0).v;      // owner: c(k0); helper: c(k1)
                const float co = __shfl(c, partner, 64);
                if (k1 >= 0) cell.v = (cell.v + c) + co;                             // meaningful on the owners
                else cell.v = cell.v + c;
            }
            sum.v = sum.v + cell.v;
            continue;
        }
        while (cand) {
            const int k0 = __builtin_ctzll(cand);
            cand &= cand - 1ULL;
            if (cand) {
                const int k1 = __builtin_ctzll(cand);
                cand &= cand - 1ULL;
                const V4 c0 = contribution(k0), c1 = contribution(k1);
                if (GRAD) cell = (cell + c0) + c1;
                else cell.v = (cell.v + c0.v) + c1.v;
            } else {
                const V4 c0 = contribution(k0);
                if (GRAD) cell = cell + c0;
                else cell.v = cell.v + c0.v;
            }
        }
        if (GRAD) sum = sum + cell;
        else sum.v = sum.v + cell.v;
    }
    return sum;
}

// ---- k-way candidate split ------------------------------------------------------------------------------------------------
// The exact rounds of the guided march serve clusters of 1 ... 64 lanes (31 on average; 28 % of the rounds at most 16, 29 % at most
// 3).  A cooperative evaluation costs the same whatever the number of queries: every candidate impulse (~145) runs the kernel body
// on every lane.  Here the na <= 32 queries of a cluster are spread over the whole wave: k = 2, 4, 8 or 16 lanes per query (the
// owner + k-1 idle lanes working on copies of its query point).  Per lattice cell the candidates are compacted into LDS in ascending
// impulse order; in pass p the j-th lane of every group evaluates candidate p k + j, the k values of a group meet in LDS, and the
// owner adds them in order:  cell = (...((cell + c(pk)) + c(pk+1)) + ...) — the reference's own sequence (SCN.cpp:383-392), a
// missing candidate contributing +0, which is exact.  ceil(n / k) kernel bodies per lane instead of n.
// Measured on C1 (Msamples/s; 420.8 without it): used for every cluster <= 32 and instead of the sideways evaluator 416.8 / 403;
// next to the sideways evaluator (clusters <= 3) 426; and with the two-way split kept for 17 ... 32 queries (its exchange is one
// ds_bpermute) 428.9 — the default: sideways <= 3 queries, k = 16 / 8 / 4 for 4 / 5-8 / 9-16, two-way 17-32, unsplit above.
template <bool SMALLARG>
GPIS_DEV float coop_noise3d_ksplit(const DevModel &M, const FastTable &T, FastLds &lds, bool active, V3 p, uint32_t seed, float R, float A0, float A1, float A2)
{
    const int lane = (int)(threadIdx.x & 63);
    const V3 pg = p / R;
    const V3 fl = v3(floorf(pg.x), floorf(pg.y), floorf(pg.z));
    V3 frac = pg - fl;
    int ci0 = (int)fl.x, cj0 = (int)fl.y, ck0 = (int)fl.z;
    const unsigned long long am = __ballot(active);
    const int na = __popcll(am);
    if (na == 0)
        return 0.f;
    // bounding box of the queries (owners only)
    const float big = 3.0e38f;
    const float bx0 = uni_f(wave_min_f(active ? pg.x : big)), bx1 = uni_f(wave_max_f(active ? pg.x : -big));
    const float by0 = uni_f(wave_min_f(active ? pg.y : big)), by1 = uni_f(wave_max_f(active ? pg.y : -big));
    const float bz0 = uni_f(wave_min_f(active ? pg.z : big)), bz1 = uni_f(wave_max_f(active ? pg.z : -big));
    const float lim = 1.0e6f;
    const bool sane = bx0 > -lim && bx1 < lim && by0 > -lim && by1 < lim && bz0 > -lim && bz1 < lim;
    int ilo = 0, ihi = 0, jlo = 0, jhi = 0, klo = 0, khi = 0;
    if (sane) {
        ilo = (int)floorf(bx0); ihi = (int)floorf(bx1);
        jlo = (int)floorf(by0); jhi = (int)floorf(by1);
        klo = (int)floorf(bz0); khi = (int)floorf(bz1);
    }
    if (!(sane && (ihi - ilo) <= 1 && (jhi - jlo) <= 1 && (khi - klo) <= 1) || na > 32) {
        // incoherent cluster, non-finite input or nothing to split: every lane generates its own impulses
        float r = 0.f;
        if (active)
            r = noise3d_per_lane(M, p, seed, R, A0, A1, A2).v;
        return r;
    }
    // k = lanes per query: 2 (17..32 queries), 4 (9..16), 8 (5..8), 16 (1..4); G2 = 64 / k groups, the first na in use
    const int lgk0 = na > 16 ? 1 : na > 8 ? 2 : na > 4 ? 3 : 4;
    const int lgk = lgk0 < kKSplitMaxLg ? lgk0 : kKSplitMaxLg;
    const int k = 1 << lgk, lgG = 6 - lgk, G2 = 1 << lgG;
    const int rank_a = __popcll(am & ((1ULL << lane) - 1ULL)), rank_i = lane - rank_a;
    // owner: group rank_a, member 0.  idle lane of rank r: group r mod G2, member 1 + r / G2
    const int grp = active ? rank_a : (rank_i & (G2 - 1));
    const int mem = active ? 0 : 1 + (rank_i >> lgG);
    const bool helper = !active && grp < na && mem < k;
    const bool act2 = active || helper;
    if (active) lds.owner_tab[rank_a] = (uint32_t)lane;
    // one wave per workgroup and in-order LDS: only the compiler has to be told that lanes talk to each other here
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    {
        const int owner = helper ? (int)lds.owner_tab[grp] : lane;
        const float hx = __shfl(frac.x, owner, 64), hy = __shfl(frac.y, owner, 64), hz = __shfl(frac.z, owner, 64);
        const int hi = __shfl(ci0, owner, 64), hj = __shfl(cj0, owner, 64), hk = __shfl(ck0, owner, 64);
        if (helper) { frac = v3(hx, hy, hz); ci0 = hi; cj0 = hj; ck0 = hk; }
    }
    const int slot = (grp << lgk) + mem;                  // this lane's place in lds.exch (as floats)
    float *const exch = reinterpret_cast<float *>(lds.exch);

    const uint32_t n = M.n_impulses;
    const int nj = jhi - jlo + 3, nk = khi - klo + 3;
    const int total = (ihi - ilo + 3) * nj * nk;          // 27 .. 64 cells
    const int H = T.half, S = T.stride;
    const unsigned side = 2u * (unsigned)H;
    const float4 *cells = T.cells;
    auto in_table = [&](int ci, int cj, int ck) {
        return cells && (unsigned)(ci + H) < side && (unsigned)(cj + H) < side && (unsigned)(ck + H) < side;
    };
    auto fetch = [&](int ci, int cj, int ck) {
        const uint32_t idx = (((uint32_t)(ci + H) * side + (uint32_t)(cj + H)) * side + (uint32_t)(ck + H)) * (uint32_t)S;
        return cells[idx + (uint32_t)(lane & (S - 1))];
    };
    int ci = ilo - 1, cj = jlo - 1, ck = klo - 1;
    float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_table(ci, cj, ck))
        nxt = fetch(ci, cj, ck);
    float sum = 0.f;
    for (int c = 0; c < total; ++c) {
        const int ti = ci, tj = cj, tk = ck;             // this cell
        const float4 cur = nxt;
        if (++ck > khi + 1) { ck = klo - 1; if (++cj > jhi + 1) { cj = jlo - 1; ++ci; } }
        if (c + 1 < total && in_table(ci, cj, ck))
            nxt = fetch(ci, cj, ck);
        const int di = ti - ci0, dj = tj - cj0, dk = tk - ck0;
        const bool mine = act2 && di >= -1 && di <= 1 && dj >= -1 && dj <= 1 && dk >= -1 && dk <= 1;
        if (__ballot(mine) == 0ULL)
            continue;
        float px, py, pz, pw;
        if (in_table(ti, tj, tk)) {
            px = cur.x; py = cur.y; pz = cur.z; pw = cur.w;
        } else {
            gen_impulse((uint32_t)ti, (uint32_t)tj, (uint32_t)tk, seed, kJump4.A[lane], kJump4.C[lane], px, py, pz, pw);
        }
        // cull: sphere (radius 1) vs box-of-queries distance, with slack for the differently rounded test
        const float qx0 = bx0 - (float)ti, qx1 = bx1 - (float)ti;
        const float qy0 = by0 - (float)tj, qy1 = by1 - (float)tj;
        const float qz0 = bz0 - (float)tk, qz1 = bz1 - (float)tk;
        const float gx = fmaxf(fmaxf(qx0 - px, px - qx1), 0.f);
        const float gy = fmaxf(fmaxf(qy0 - py, py - qy1), 0.f);
        const float gz = fmaxf(fmaxf(qz0 - pz, pz - qz1), 0.f);
        const bool cand_k = (uint32_t)lane < n && (gx * gx + gy * gy + gz * gz) < 1.0001f;
        const unsigned long long cand = __ballot(cand_k);
        if (cand == 0ULL)
            continue;
        const int cnt = __popcll(cand);
        // the candidates, compacted in ascending impulse order, in the layout the packed body consumes
        if (cand_k) {
            const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cand >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cand, 0u));
            lds.xyr[r] = make_float4(px, py, R * px, R * py);
            lds.zw[r] = make_float4(pz, R * pz, pw, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // lanes whose neighbourhood does not contain this cell get a far-away point: the unit-ball test fails for them
        const V3 pc = mine ? frac - v3((float)di, (float)dj, (float)dk) : v3(8.f, 8.f, 8.f);
        const V3 Rp = R * pc;
        const float2v pcxy = float2v{pc.x, pc.y}, Rpxy = float2v{Rp.x, Rp.y}, pzRz = float2v{pc.z, Rp.z}, Axy = float2v{A0, A1};
        float cell = 0.f;
        for (int base = 0; base < cnt; base += k) {
            const int idx = base + mem;
            const bool valid = act2 && idx < cnt;
            const float4 qa = lds.xyr[idx & 63], qb = lds.zw[idx & 63];
            const float2v ixy = float2v{qa.x, qa.y}, rxy = float2v{qa.z, qa.w}, zrz = float2v{qb.x, qb.y};
            const float2v dxy = pcxy - ixy;             // (pc.x - x, pc.y - y)
            const float2v abxy = Rpxy - rxy;            // (ab.x, ab.y)
            const float2v dzab = pzRz - zrz;            // (pc.z - z, ab.z)
            const float2v d2 = dxy * dxy;
            const float2v txy = abxy * Axy;             // (t.x, t.y)
            const float2v tab = txy * abxy;
            const float tz = dzab.y * A2;
            const float2v zz = float2v{dzab.x, tz} * dzab;     // (dz^2, t.z * ab.z)
            const bool pass = valid && ((d2.x + d2.y) + zz.x) < 1.0f;
            const float absq = tab.x + (tab.y + zz.y);
            const float f = expf_glibc_lds<true, SMALLARG>(lds, -absq);
            if (act2)
                exch[slot] = pass ? qb.z * f : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            // the group's k values in candidate order (lanes of a group read the same addresses; only the owner's sum is used)
            if (lgk == 1) {
                const float2 v = *reinterpret_cast<const float2 *>(exch + (grp << 1));
                cell = (cell + v.x) + v.y;
            } else {
                for (int i = 0; i < k; i += 4) {
                    const float4 v = lds.exch[((grp << lgk) + i) >> 2];
                    cell = (((cell + v.x) + v.y) + v.z) + v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        sum = sum + cell;
    }
    return sum;
}

// noise3D VALUE for ONE query (the one held by lane `src`, src wave-uniform) with the wave turned
// sideways: lane k owns impulse k of the current cell and evaluates its kernel against the single query
// point; the cell sum is then accumulated over the passing lanes in ascending k (ChainSum: compacted into LDS
// and read back as broadcasts), which is the reference's order (SCN.cpp:383-392), and cells in dx,dy,dz order
// (SCN.cpp:368-371).
// ~2.4k instructions per query against ~10k for a cooperative evaluation: used when only a few lanes
// of a wave need an exact value.  The result is wave-uniform.
GPIS_DEV float solo_noise3d_value(const DevModel &M, const FastTable &T, FastLds &lds, int src, V3 p, uint32_t seed, float R, float A0, float A1, float A2)
{
    const int lane = (int)(threadIdx.x & 63);
    const V3 pg = p / R;
    const V3 fl = v3(floorf(pg.x), floorf(pg.y), floorf(pg.z));
    const V3 fr = pg - fl;
    // the query of lane `src`, broadcast
    const float fx = lane_f(fr.x, src), fy = lane_f(fr.y, src), fz = lane_f(fr.z, src);
    const float qx = lane_f(fl.x, src), qy = lane_f(fl.y, src), qz = lane_f(fl.z, src);
    if (!(fabsf(qx) < 1.0e6f && fabsf(qy) < 1.0e6f && fabsf(qz) < 1.0e6f)) {
        // outside the range where cells are addressed with ints: the per-lane generator handles it
        float r = 0.f;
        if (lane == src)
            r = noise3d_per_lane(M, p, seed, R, A0, A1, A2).v;
        return lane_f(r, src);
    }
    const int ci0 = (int)qx, cj0 = (int)qy, ck0 = (int)qz;
    const uint32_t n = M.n_impulses;
    const int H = T.half, S = T.stride;
    const unsigned side = 2u * (unsigned)H;
    ChainSum cs;
    for (int di = -1; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj)
            for (int dk = -1; dk <= 1; ++dk) {
                const int ci = ci0 + di, cj = cj0 + dj, ck = ck0 + dk;
                float px, py, pz, pw;
                if (T.cells && (unsigned)(ci + H) < side && (unsigned)(cj + H) < side && (unsigned)(ck + H) < side) {
                    const size_t idx = (((size_t)(ci + H) * side + (size_t)(cj + H)) * side + (size_t)(ck + H)) * (size_t)S;
                    const float4 v = T.cells[idx + (size_t)(lane & (S - 1))];
                    px = v.x; py = v.y; pz = v.z; pw = v.w;
                } else {
                    gen_impulse((uint32_t)ci, (uint32_t)cj, (uint32_t)ck, seed, kJump4.A[lane], kJump4.C[lane], px, py, pz, pw);
                }
                const V3 pc = v3(fx, fy, fz) - v3((float)di, (float)dj, (float)dk);
                const bool pass = (uint32_t)lane < n && length_sq(pc - v3(px, py, pz)) < 1.0f;
                const unsigned long long m = __ballot(pass);
                if (m == 0ULL)
                    continue;
                const V3 ab = R * pc - R * v3(px, py, pz);
                const V3 t = v3(ab.x * A0, ab.y * A1, ab.z * A2);
                const float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
                const float c = pw * expf_glibc_lds<true>(lds, -absq);      // absq >= 0: no overflow check
                chain_push(lds, cs, m, pass, c);
            }
    chain_flush(lds, cs);
    return cs.sum;
}

// The same evaluation with 64 / S cells per pass (two at rho <= 32): lane = (cell of the pass, impulse).
// Lanes are ordered (cell, k), so walking the ballot upwards visits the impulses in the reference's
// order, and each cell keeps its own partial sum (cell3D starts from 0, SCN.cpp:380).  Used by the
// wavefront kernels (gpis_wave.hpp), which have no march state to keep live; in the resident kernels its
// extra live values cost more in scratch traffic than the halved pass count returns.
GPIS_DEV float solo_noise3d_value_packed(const DevModel &M, const FastTable &T, FastLds &lds, int src, V3 p, uint32_t seed, float R, float A0, float A1, float A2)
{
    const int lane = (int)(threadIdx.x & 63);
    const V3 pg = p / R;
    const V3 fl = v3(floorf(pg.x), floorf(pg.y), floorf(pg.z));
    const V3 fr = pg - fl;
    const float fx = lane_f(fr.x, src), fy = lane_f(fr.y, src), fz = lane_f(fr.z, src);
    const float qx = lane_f(fl.x, src), qy = lane_f(fl.y, src), qz = lane_f(fl.z, src);
    if (!(fabsf(qx) < 1.0e6f && fabsf(qy) < 1.0e6f && fabsf(qz) < 1.0e6f)) {
        float r = 0.f;
        if (lane == src)
            r = noise3d_per_lane(M, p, seed, R, A0, A1, A2).v;
        return lane_f(r, src);
    }
    const int ci0 = (int)qx, cj0 = (int)qy, ck0 = (int)qz;
    const uint32_t n = M.n_impulses;
    const int H = T.half;
    const int S = n <= 32u ? 32 : 64;      // the table's slots per cell (fast_table_build); also the lane layout without a table
    const unsigned side = 2u * (unsigned)H;
    const int per_pass = 64 / S;
    const int sub = lane / S, k_imp = lane & (S - 1);
    const unsigned long long cell_bits = S == 64 ? ~0ULL : ((1ULL << S) - 1ULL);
    ChainSum cs;
    for (int c0 = 0; c0 < 27; c0 += per_pass) {
        const int c = c0 + sub;
        const int di = c / 9 - 1, dj = (c / 3) % 3 - 1, dk = c % 3 - 1;      // dx, dy, dz order (SCN.cpp:368-371)
        const int ci = ci0 + di, cj = cj0 + dj, ck = ck0 + dk;
        float px = 0.f, py = 0.f, pz = 0.f, pw = 0.f;
        const bool mine = c < 27 && (uint32_t)k_imp < n;
        if (mine) {
            if (T.cells && (unsigned)(ci + H) < side && (unsigned)(cj + H) < side && (unsigned)(ck + H) < side) {
                const size_t idx = (((size_t)(ci + H) * side + (size_t)(cj + H)) * side + (size_t)(ck + H)) * (size_t)S;
                const float4 v = T.cells[idx + (size_t)k_imp];
                px = v.x; py = v.y; pz = v.z; pw = v.w;
            } else {
                gen_impulse((uint32_t)ci, (uint32_t)cj, (uint32_t)ck, seed, kJump4.A[k_imp], kJump4.C[k_imp], px, py, pz, pw);
            }
        }
        const V3 pc = v3(fx, fy, fz) - v3((float)di, (float)dj, (float)dk);
        const bool pass = mine && length_sq(pc - v3(px, py, pz)) < 1.0f;
        const unsigned long long m = __ballot(pass);
        if (m == 0ULL)
            continue;
        const V3 ab = R * pc - R * v3(px, py, pz);
        const V3 t = v3(ab.x * A0, ab.y * A1, ab.z * A2);
        const float absq = sum3e(t.x * ab.x, t.y * ab.y, t.z * ab.z);
        const float cv = pw * expf_glibc_lds<true>(lds, -absq);     // absq >= 0: no overflow check
        for (int g = 0; g < per_pass; ++g) {
            const unsigned long long mg = m & (cell_bits << (g * S));
            if (mg == 0ULL)
                continue;                      // an empty cell adds +0: exact to skip
            chain_push(lds, cs, mg, pass && sub == g, cv);
        }
    }
    chain_flush(lds, cs);
    return cs.sum;
}

// evaluateNoise3D for the stationary single-realization case (SCN.cpp:101-116, 251-260, 291-320):
// per-lane transforms exactly as the generic path, the noise3D sum cooperatively.
// `coord` is the ray's isotropic-ray frame (SCN.cpp:296-297), constant along the segment and hoisted
// out of the march by the caller.
template <bool GRAD, bool SMALLARG = false>
GPIS_DEV V4 coop_eval_noise3d(const DevModel &M, const FastTable &T, FastLds &lds, bool active, V3 p, const Frame &coord, uint32_t &n_eval, bool split = false)
{
    if (active) n_eval++;
    if (!M.iso3d) {
        float R = M.radius_world;
        // getInvCovMtx(isCov=false, isIsotropic=false, globalScale=1, localScale=1): ((A / 1) / 1) * 0.5
        float A0 = M.invcov_world[0] / 1.f / 1.f * 0.5f, A1 = M.invcov_world[4] / 1.f / 1.f * 0.5f, A2 = M.invcov_world[8] / 1.f / 1.f * 0.5f;
        if (!GRAD && kKSplit && split && __popcll(__ballot(active)) <= kKSplitMaxNa)
            return v4(coop_noise3d_ksplit<SMALLARG>(M, T, lds, active, p, M.seed, R, A0, A1, A2) / M.norm3d_world, 0.f, 0.f, 0.f);
        V4 nz = coop_noise3d<GRAD, SMALLARG>(M, T, lds, active, p, M.seed, R, A0, A1, A2, split);
        return nz / M.norm3d_world;
    }
    V3 p_iso_ray = to_local(coord, spec_3d::cov_pos_w2l(M, p, 1.0f));
    if (!GRAD && kKSplit && split && __popcll(__ballot(active)) <= kKSplitMaxNa)
        return v4(coop_noise3d_ksplit<SMALLARG>(M, T, lds, active, p_iso_ray, M.seed, M.radius_iso, 0.5f, 0.5f, 0.5f) / M.norm3d_iso, 0.f, 0.f, 0.f);
    V4 nz = coop_noise3d<GRAD, SMALLARG>(M, T, lds, active, p_iso_ray, M.seed, M.radius_iso, 0.5f, 0.5f, 0.5f, split);
    if (GRAD) {
        V3 gw = spec_3d::cov_grad_l2w(M, to_global(coord, v3(nz.gx, nz.gy, nz.gz)), 1.0f);
        return v4(nz.v, gw.x, gw.y, gw.z) / M.norm3d_iso;
    }
    return v4(nz.v / M.norm3d_iso, 0.f, 0.f, 0.f);
}

// evaluateValue, SCN.cpp:73-89
template <bool SMALLARG = false>
GPIS_DEV float coop_evaluate_value(const DevModel &M, const FastTable &T, FastLds &lds, bool active, V3 p, const Frame &coord, int &gp_id, uint32_t &n_eval,
                                   bool split = false)
{
    float nv = coop_eval_noise3d<false, SMALLARG>(M, T, lds, active, p, coord, n_eval, split).v;
    double mean;
    int id;
    mean_weight_space(M, to_d(p), mean, id);
    gp_id = id;
    if (M.surf_vol_phase_separate)
        gp_id = (1.f < M.surf_vol_phase_amp_thresh) ? 0 : 1;
    return (float)((double)(M.sigma * nv) + mean);
}
// evaluateValue (SCN.cpp:73-89) for the single lane `src` through solo_noise3d_value; the value is
// delivered in lane src's return value (other lanes get an unspecified number).
template <bool PACKED = false>
GPIS_DEV float solo_evaluate_value(const DevModel &M, const FastTable &T, FastLds &lds, int src, V3 p, const Frame &coord, int &gp_id, uint32_t &n_eval)
{
    const int lane = (int)(threadIdx.x & 63);
    float nv;
    if (!M.iso3d) {
        float A0 = M.invcov_world[0] / 1.f / 1.f * 0.5f, A1 = M.invcov_world[4] / 1.f / 1.f * 0.5f, A2 = M.invcov_world[8] / 1.f / 1.f * 0.5f;
        nv = (PACKED ? solo_noise3d_value_packed(M, T, lds, src, p, M.seed, M.radius_world, A0, A1, A2)
                     : solo_noise3d_value(M, T, lds, src, p, M.seed, M.radius_world, A0, A1, A2)) / M.norm3d_world;
    } else {
        V3 p_iso_ray = to_local(coord, spec_3d::cov_pos_w2l(M, p, 1.0f));
        nv = (PACKED ? solo_noise3d_value_packed(M, T, lds, src, p_iso_ray, M.seed, M.radius_iso, 0.5f, 0.5f, 0.5f)
                     : solo_noise3d_value(M, T, lds, src, p_iso_ray, M.seed, M.radius_iso, 0.5f, 0.5f, 0.5f)) / M.norm3d_iso;
    }
    if (lane == src) n_eval++;
    double mean;
    int id;
    mean_weight_space(M, to_d(p), mean, id);
    gp_id = id;
    if (M.surf_vol_phase_separate)
        gp_id = (1.f < M.surf_vol_phase_amp_thresh) ? 0 : 1;
    return (float)((double)(M.sigma * nv) + mean);
}
// evaluateGradient, SCN.cpp:92-99
template <bool SMALLARG = false>
GPIS_DEV V3 coop_evaluate_gradient(const DevModel &M, const FastTable &T, FastLds &lds, bool active, V3 p, const Frame &coord, uint32_t &n_eval)
{
    V4 nz = coop_eval_noise3d<true, SMALLARG>(M, T, lds, active, p, coord, n_eval);
    double mean;
    int id;
    mean_weight_space(M, to_d(p), mean, id);
    V3 mg = to_f(mean_grad(M, id, to_d(p)));
    return M.sigma * v3(nz.gx, nz.gy, nz.gz) + mg;
}

enum Phase : int { PH_INIT = 0, PH_MARCH = 1, PH_REFINE = 2, PH_FINAL = 3, PH_GRAD = 4, PH_DONE = 5 };

// The MediumSample / MediumState writes of GPM.cpp:224-340 for one finished segment.
GPIS_DEV void finish_sample_distance(const DevModel &M, const gpis_ray_in *__restrict__ rayp, V3 pos, V3 dir, float farT, bool early_ok,
                                     bool want_grad, bool hit, double t, float last_val, int gp, V3 g, gpis_seg_out *out)
{
    const float maxT = farT;
    gpis_seg_out o;
    o.t = 0.;
    o.sample_t = 0.f; o.continued_t = 0.f;
    for (int c = 0; c < 3; ++c) { o.weight[c] = 0.f; o.continued_weight[c] = 0.f; o.p[c] = 0.f; }
    o.exited = 0; o.ok = 0; o.scheme = GPIS_UNI;
    o.gp_id = rayp->last_gp_id;
    o.last_val = rayp->last_val;
    o.aniso[0] = rayp->last_aniso[0]; o.aniso[1] = rayp->last_aniso[1]; o.aniso[2] = rayp->last_aniso[2];
    if (early_ok) {
        o.weight[0] = o.weight[1] = o.weight[2] = 1.f;
        o.exited = 1;
        V3 pp = pos + dir * o.sample_t;
        o.p[0] = pp.x; o.p[1] = pp.y; o.p[2] = pp.z;
        o.ok = 1;
        *out = o;
        return;
    }
    if (!want_grad) {   // bounce limit
        *out = o;
        return;
    }
    V3d aniso = to_d(g);
    o.t = t;
    o.exited = hit ? 0 : 1;
    o.last_val = last_val;
    o.gp_id = hit ? gp : rayp->last_gp_id;   // intersectGP stores GPId only on a hit (SCNM.cpp:162); an exit keeps state.lastGPId
    bool ok = true;
    if (hit) {
        double avg = (aniso.x + aniso.y + aniso.z) / 3.0;
        if (!__builtin_isfinite(avg)) {
            aniso = V3d{1., 0., 0.};
            ok = false;
        } else {
            double d = aniso.x * (double)dir.x; d += aniso.y * (double)dir.y; d += aniso.z * (double)dir.z;
            double l2 = 0.; l2 += aniso.x * aniso.x; l2 += aniso.y * aniso.y; l2 += aniso.z * aniso.z;
            if (d > 0) {
                ok = false;
            } else if (l2 < (double)0.0000001f) {
                aniso = V3d{1., 0., 0.};
                ok = false;
            }
        }
    }
    o.aniso[0] = aniso.x; o.aniso[1] = aniso.y; o.aniso[2] = aniso.z;
    if (ok) {
        float ft = (float)t;
        o.sample_t = ft < maxT ? ft : maxT;
        o.continued_t = ft;
        for (int c = 0; c < 3; ++c) {
            o.weight[c] = 1.f * M.sigma_s_over_t[c];
            o.continued_weight[c] = 1.f * M.sigma_s_over_t[c];
        }
        V3 pp = pos + dir * o.sample_t;
        o.p[0] = pp.x; o.p[1] = pp.y; o.p[2] = pp.z;
        o.ok = 1;
    }
    *out = o;
}

// The march of one wave of segments.  WANT_SAMPLE: sampleDistance (GPM.cpp:221-341) — otherwise
// transmittance (GPM.cpp:343-393), whose result does not depend on the end-of-segment gradient nor
// on lastVal, so those two evaluations are not performed.
// Register diet: only what the lockstep loop needs stays live (floats for pos/dir, widened on use
// exactly as `vec_conv<Vec3d>` does; pass-through fields are re-read from the ray at the end).
template <bool WANT_SAMPLE>
GPIS_DEV void fast_march(const DevModel &M, const FastTable &T, FastLds &lds, bool valid, const gpis_ray_in *__restrict__ rayp,
                         gpis_seg_out *out, bool &visible, uint32_t &n_eval)
{
    V3 pos = v3(0.f, 0.f, 1.f), dir = v3(0.f, 0.f, 1.f);
    float nearT = 0.f, farT = 1.f, u_jitter = 0.f;
    bool first_scatter = true;
    int bounce = 0;
    if (valid) {
        pos = v3(rayp->pos[0], rayp->pos[1], rayp->pos[2]);
        dir = v3(rayp->dir[0], rayp->dir[1], rayp->dir[2]);
        nearT = rayp->near_t; farT = rayp->far_t; u_jitter = rayp->u_jitter;
        first_scatter = rayp->first_scatter != 0;
        bounce = rayp->bounce;
    }
    if (!__builtin_isfinite(farT))
        farT = (float)((double)nearT + 2000);
    float step_size = (farT - nearT) / (float)M.min_step;
    if (M.step_size < step_size)
        step_size = M.step_size;
    Frame coord{};
    if (M.iso3d)
        coord = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, dir, 1.0f)));

    int phase = PH_INIT;
    bool early_ok = false;         // maxT == 0 shortcut
    if (!valid)
        phase = PH_DONE;
    else if (WANT_SAMPLE && bounce >= M.max_bounces)
        phase = PH_DONE;
    else if (WANT_SAMPLE && farT == 0.f) {
        phase = PH_DONE;
        early_ok = true;
    }
    // t: march position (PH_MARCH) / result; refinement bracket [a_lo, t] with factor intp
    double t = (double)nearT, a_lo = 0., intp = 0., t_test = 0., t_prev = 0.;
    float pf = 0.f;                // previous field value (a float widened on use, SCNM.cpp:128,172)
    int sign0 = 1, step = 0, gp = 0;
    float last_val = 0.f;
    bool hit = false;

    for (;;) {
        const bool want_val = phase <= PH_FINAL;
        if (__ballot(want_val) == 0ULL)
            break;
        const double tq = phase == PH_INIT ? (double)nearT : (phase == PH_REFINE ? t_test : (phase == PH_FINAL ? (double)farT : t));
        int gp_new;
        float fv = coop_evaluate_value(M, T, lds, want_val, to_f(ray_at(to_d(pos), to_d(dir), tq)), coord, gp_new, n_eval);
        if (!want_val)
            continue;
        gp = gp_new;
        const double f = (double)fv;
        if (phase == PH_INIT) {
            sign0 = f < 0 ? -1 : 1;
            pf = fv;
            t = (double)(nearT + step_size * u_jitter);
            phase = (t < (double)farT) ? PH_MARCH : PH_FINAL;
        } else if (phase == PH_MARCH) {
            step++;
            const int signc = f < 0 ? -1 : 1;
            if (!first_scatter && step == 1) {
                sign0 = signc;
                pf = fv;
                t += (double)step_size;
            } else if (signc != sign0) {
                intp = (double)pf / ((double)pf - f);
                a_lo = t - (double)step_size;
                t_prev = lerp_d(a_lo, t, intp);
                t_test = t_prev;
                phase = PH_REFINE;
            } else {
                pf = fv;
                t += (double)step_size;
            }
            if (phase == PH_MARCH && !(t < (double)farT))
                phase = PH_FINAL;
        } else if (phase == PH_REFINE) {
            const int sign_test = f < 0 ? -1 : 1;
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
                phase = WANT_SAMPLE ? PH_GRAD : PH_DONE;
            }
        } else {   // PH_FINAL: the lastVal evaluation at farT (SCNM.cpp:181)
            t = (double)farT;
            last_val = fv;
            hit = false;
            phase = WANT_SAMPLE ? PH_GRAD : PH_DONE;
        }
        if (!WANT_SAMPLE && phase == PH_FINAL) {
            // transmittance: the segment exits; lastVal is not part of the result
            t = (double)farT;
            hit = false;
            phase = PH_DONE;
        }
    }
    visible = valid && !hit;
    if (!WANT_SAMPLE)
        return;

    // one gradient evaluation per segment (GPM.cpp:283 on a hit, GPM.cpp:319 on exit), at
    // ro + normalize(rd) * t (GPM.cpp:262-263, 280)
    const bool want_grad = phase == PH_GRAD;
    V3 g = v3(0.f, 0.f, 0.f);
    if (__ballot(want_grad) != 0ULL) {
        V3d rdn = to_d(dir);
        { double inv = 1.0 / length_d(rdn); rdn.x *= inv; rdn.y *= inv; rdn.z *= inv; }
        g = coop_evaluate_gradient(M, T, lds, want_grad, to_f(ray_at(to_d(pos), rdn, t)), coord, n_eval);
    }
    if (!valid)
        return;

    finish_sample_distance(M, rayp, pos, dir, farT, early_ok, want_grad, hit, t, last_val, gp, g, out);
}

__device__ __forceinline__ void fast_flush_counters(Counters *cnt, uint32_t n_eval, uint32_t n_seg)
{
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

GPIS_TU_KERNEL __global__ void __launch_bounds__(kFastBlock, GPIS_FAST_OCC) k_fast_sample_distance(const DevModel *__restrict__ Mp, FastTable T, size_t n, const gpis_ray_in *__restrict__ rays,
                                                                    gpis_seg_out *__restrict__ out, gpis_cond_coeff *__restrict__ coeff,
                                                                    const uint8_t *__restrict__ mask, Counters *cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    uint32_t n_eval = 0;
    bool vis;
    fast_march<true>(*Mp, T, lds, valid, rays + (valid ? i : 0), out + (valid ? i : 0), vis, n_eval);
    if (valid && coeff) {
        gpis_cond_coeff c;
        memset(&c, 0, sizeof c);
        c.n_evals = n_eval;
        coeff[i] = c;
    }
    fast_flush_counters(cnt, n_eval, valid ? 1u : 0u);
}

GPIS_TU_KERNEL __global__ void __launch_bounds__(kFastBlock, GPIS_FAST_OCC) k_fast_transmittance(const DevModel *__restrict__ Mp, FastTable T, size_t n, const gpis_ray_in *__restrict__ rays,
                                                                  uint8_t *__restrict__ visible, const uint8_t *__restrict__ mask, Counters *cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    uint32_t n_eval = 0;
    bool vis = false;
    fast_march<false>(*Mp, T, lds, valid, rays + (valid ? i : 0), nullptr, vis, n_eval);
    if (i < n)
        visible[i] = (valid && vis) ? 1 : 0;
    fast_flush_counters(cnt, n_eval, valid ? 1u : 0u);
}

inline int fast_sample_distance(const DevModel *d_model, const FastTable *T, size_t n, const gpis_ray_in *rays, gpis_seg_out *out,
                                gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    unsigned grid = (unsigned)((n + kFastBlock - 1) / kFastBlock);
    k_fast_sample_distance<0><<<grid, kFastBlock, 0, s>>>(d_model, *T, n, rays, out, coeff, mask, cnt);
    return GPIS_OK;
}
inline int fast_transmittance(const DevModel *d_model, const FastTable *T, size_t n, const gpis_ray_in *rays, uint8_t *visible,
                              const uint8_t *mask, Counters *cnt, hipStream_t s)
{
    unsigned grid = (unsigned)((n + kFastBlock - 1) / kFastBlock);
    k_fast_transmittance<0><<<grid, kFastBlock, 0, s>>>(d_model, *T, n, rays, visible, mask, cnt);
    return GPIS_OK;
}

}   // namespace gpis
