// gpis_guide.hpp — certified guide field for single-realization media (this is what the 288 GB
// of HBM are for).
//
// With single_realization the noise3D sum (SCN.cpp:362-395) is ONE fixed function N(u) of the
// grid-space position u = p_grid (cell units), the same for every ray.  The march (SCNM.cpp:132-174)
// only needs the SIGN of sigma*N/norm + mean at almost every step; the value matters at the crossing
// step, the step before it, during refinement and at the segment end.  So:
//
//   V(g) = the lattice sum itself (with its unit-ball cut-off) sampled on a grid of spacing h = 1/ppc
//          cells:  V(g) = sum_i [|delta_i(g)| < 1] w_i exp(-q_i(g)),  q_i = sum_a alpha_a delta_a^2,
//          delta = g - (cell + p_i), alpha_a = A_a R^2;   G(u) = trilinear interpolation of V,
//   Err  = a RIGOROUS bound on |N_ref(u) - G(u)| for every u of a 4x4x4-point block, stored per block.
//
// A march step whose |sigma*G/norm + mean| exceeds sigma*Err/norm (+ rounding slack) has a certified
// sign and costs one 8-tap lookup instead of 27*rho kernel evaluations; all other steps run the exact
// cooperative evaluation, so every output is bit-identical to the exact path.
//
// Error budget for the cell C anchored at grid point g (c = its centre, m = half its diagonal, r_i =
// |delta_i(c)|, amin = min_a alpha_a).  Every impulse falls in exactly one class:
//   inside  r_i + m < 1 - 1e-5 : its indicator is 1 at u and at all 8 corners, so it contributes a smooth
//           term whose multilinear interpolation error is <= sum_a h^2/8 sup_C |d_aa w e^{-q}|, with
//           d_aa e^{-q} = (4 alpha_a^2 delta_a^2 - 2 alpha_a) e^{-q}  bounded by
//           max(4 alpha_a^2 (|delta_a(c)| + h/2)^2 - 2 alpha_a, 2 alpha_a) * exp(-amin max(r_i - m, 0)^2);
//   outside r_i - m >= 1 + 1e-5 : indicator 0 everywhere on C (also for the reference's fp32 test): nothing;
//   shell   otherwise: at u the reference adds either 0 or w e^{-q(u)}, the interpolant a convex
//           combination of 0 and w e^{-q(corner)}: they differ by at most sup_C e^{-q} <= exp(-amin max(r_i - m, 0)^2).
//   position: the march looks the guide up at w = a + t*b (double, GuideRay) while the exact evaluator derives its
//           position u in fp32; |u - w| <= 1e-4 cells per axis, so inside-class impulses add
//           1e-4 * sum_a 2 alpha_a (|delta_a(c)| + h/2) exp(-amin max(r_i - m, 0)^2)  (Lipschitz bound of w e^{-q});
//           the shell and outside classes are already taken over the cell expanded by that slack (mc, hh);
//   rounding: fp32 accumulation, v_exp_f32, the reference's own fp32 kernel arguments, the lerp: 5e-4
//           absolute + 1e-3 relative.
// Impulses farther than 1 + 1e-4 from the block's cells are "outside" for every cell of the block and
// are skipped without any error term.
#pragma once
#include "gpis_fast.hpp"

#pragma clang fp contract(off)

namespace gpis {

// Storage: the samples live in BRICKS of 16^3 grid points (+ one layer shared with the +neighbours: 17^3 floats, so the eight taps of
// a lookup never leave their brick), and only the bricks a march can need at level 1 are tabulated: a brick is needed when, over the
// world points that can map into it, |mean| can come below sigma * amax / norm, amax = an a-priori bound on |N| over the brick
// from a field 4x coarser (1/64 of the work).  Everywhere else level 0 (guide_sign_at) decides every step by the mean alone, and
// the block records say so: Err = +inf there, so level 1 can certify nothing and the exact evaluation takes over — the need
// analysis decides speed, never results.  Scene S: 8 % of the bricks (2.8 of 34 GB at 16:64).
constexpr int kBrick = 16;                         // grid points per brick edge = 4 blocks
constexpr uint32_t kBrickRow = 17, kBrickFloats = 17u * 17u * 17u;
constexpr uint32_t kNoBrick = 0xFFFFFFFFu;
struct GuideField {
    float *G;            // pool of tabulated bricks: slot * 17^3 + (lx * 17 + ly) * 17 + lz
    uint64_t *blk;       // (side/4)^3 block records, dense, 8 bytes = one load per march step: slot << 32 | fp16 Err << 16 | fp16 amax (both
                         // rounded UP; Err = the bound of the block, +inf where the brick is not tabulated; amax = a bound on |N| anywhere in
                         // the block's cells; slot = the block's brick in the pool, 0 where not tabulated: Err = +inf makes its taps irrelevant)
    int half;            // extent in cells: u in [-half, half)
    int ppc;             // grid points per cell (h = 1/ppc)
    int side;            // 2*half*ppc
    float alpha[3];      // A_a * R^2
    float R;             // kernelRadius of the grid space
    int enabled;
    uint32_t n_alloc, n_usable;   // bricks in the pool / bricks level 1 may use
};

// bounds are stored as fp16 rounded towards +inf (values >= 0 or +inf)
GPIS_DEV uint32_t guide_f2h_up(float x)
{
    _Float16 h = (_Float16)x;
    uint32_t bits = (uint32_t)__builtin_bit_cast(unsigned short, h);
    if ((float)h < x) bits += 1u;          // next fp16 up (the largest finite value steps to +inf)
    return bits;
}
GPIS_DEV float guide_h2f(uint32_t bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(bits & 0xFFFFu)); }
GPIS_DEV float guide_rec_amax(uint64_t r) { return guide_h2f((uint32_t)r); }
GPIS_DEV float guide_rec_err(uint64_t r) { return guide_h2f((uint32_t)r >> 16); }
GPIS_DEV uint32_t guide_rec_slot(uint64_t r) { return (uint32_t)(r >> 32); }
GPIS_DEV uint64_t guide_rec_make(uint32_t slot, uint32_t err16, uint32_t amax16) { return ((uint64_t)slot << 32) | (uint64_t)((err16 << 16) | (amax16 & 0xFFFFu)); }

#ifndef GPIS_GUIDE_OCC
// waves per SIMD the guided kernels are register-allocated for.  Guide lookups are latency-bound, so
// occupancy beats spill-free code here: measured on C1 960x540x64 (sd+tr ms, round 1) occ2 180 (no spills),
// occ3 149, occ4 137, occ5 146, occ6 154, occ8 183.  Round 2, after the path functions the cold fallback inlines were pinned to
// their 3D-stationary instance (scratch 320 -> 208 B): whole C1 frames, sd ms at 3 waves 282.5, 4 waves 244.7, 5 waves 232.1
#define GPIS_GUIDE_OCC 5
#endif
#ifndef GPIS_GUIDE_OCC_TR
// transmittance carries less state (no lastVal / gradient tail): whole C1 frames, sd / tr ms at 3 waves 336 / 135,
// 4 waves 315 / 122, 5 waves 325 / 117
#define GPIS_GUIDE_OCC_TR 5
#endif
#ifndef GPIS_SOLO_MAX
#define GPIS_SOLO_MAX 3
#endif
#ifndef GPIS_GUIDE_SPLIT
#define GPIS_GUIDE_SPLIT 1
#endif
#ifndef GPIS_SOLO_PACKED
#define GPIS_SOLO_PACKED 0
#endif
constexpr bool kSoloPacked = GPIS_SOLO_PACKED != 0;   // sideways evaluator with 64 / stride cells per pass (measured slower in the resident kernels: C1 403 vs 412 Msamples/s)
constexpr bool kGuideSplit = GPIS_GUIDE_SPLIT != 0;   // two-way candidate split for clusters of at most 32 lanes
constexpr int kSoloMaxLanes = GPIS_SOLO_MAX;   // clusters up to this size use the sideways evaluator (range / wavefront kernels)
#ifndef GPIS_GUIDE_SOLO_MAX
#define GPIS_GUIDE_SOLO_MAX GPIS_SOLO_MAX
#endif
constexpr int kGuideSoloMax = GPIS_GUIDE_SOLO_MAX;   // the same in the resident guided kernels (0: the k-way split serves the smallest clusters too — measured slower, 403 vs 429)
constexpr float kGuidePosEps = 1e-4f;         // bound on |u - w| per axis (cells): measured < 2e-5, see GuideRay
constexpr float kGuideCullRadius = 1.0001f;   // beyond this distance from the block's cells an impulse contributes exactly 0

// one wave = one 4x4x4 block of grid points.
// Cost: ~140 impulses lie within the cut-off of a block, and each of the 64 points meets each of them.  Two thirds of them are
// "inside" for the WHOLE block (distance from the block's centre + its half-diagonal < 1): for those a point only adds
// w exp(-q) to its sample (14 instructions), and their smooth error terms are computed once per block with lane = impulse from
// the block's centre and half-size — a bound on every cell's own term (|delta_a(c)| + h/2 <= |delta_a(cB)| + 2h,
// r_i(c) - m >= r_i(cB) - mB).  Only the impulses whose cut-off sphere passes near the block take the per-cell
// classification of the header comment.  (Before: every pair took it, 100 instructions with a correctly rounded sqrtf and four
// v_readlane; 2.5 s for the 34 GB field.)
// Work item = one 4x4x4 block of one ALLOCATED brick (64 per brick): `list[slot]` = the brick, `slot_of[brick]` = its slot or kNoBrick,
// `need[brick]` = level 1 may use it (then the block's Err is recorded; otherwise the block keeps the a-priori record of k_guide_need).
GPIS_TU_KERNEL __global__ void __launch_bounds__(64) k_guide_build(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t item_offset,
                                                                  const uint32_t *__restrict__ list, const uint32_t *__restrict__ slot_of,
                                                                  const uint8_t *__restrict__ need)
{
    __shared__ float4 stage[64];
    const DevModel &M = *Mp;
    const int lane = (int)(threadIdx.x & 63);
    const int bs = F.side / 4, nbk = F.side / kBrick;
    const size_t item = item_offset + blockIdx.x;
    const uint32_t slot = (uint32_t)(item >> 6), sub = (uint32_t)(item & 63);
    const uint32_t brick = list[slot];
    const int kz = (int)(brick % nbk), ky = (int)((brick / nbk) % nbk), kx = (int)(brick / ((uint32_t)nbk * nbk));
    const int bx = 4 * kx + (int)(sub >> 4), by = 4 * ky + (int)((sub >> 2) & 3), bz = 4 * kz + (int)(sub & 3);
    const int lz = lane & 3, ly = (lane >> 2) & 3, lx = lane >> 4;
    const int ix = 4 * bx + lx, iy = 4 * by + ly, iz = 4 * bz + lz;
    const float h = 1.0f / (float)F.ppc;
    // grid point in cell units; exact in fp32 (multiples of 1/32 below 2^5)
    const float ux = (float)ix * h - (float)F.half, uy = (float)iy * h - (float)F.half, uz = (float)iz * h - (float)F.half;
    // the block's box [lo, lo+3h] expanded by h (the cells anchored at its points)
    const float lox = (float)(4 * bx) * h - (float)F.half, loy = (float)(4 * by) * h - (float)F.half, loz = (float)(4 * bz) * h - (float)F.half;
    const int c0x = (int)floorf(lox), c0y = (int)floorf(loy), c0z = (int)floorf(loz);   // same lattice cell for the whole block
    const float ax = F.alpha[0], ay = F.alpha[1], az = F.alpha[2];
    const float amin = fminf(ax, fminf(ay, az));
    // bounds are taken around the CENTRE of the cell anchored at the grid point: every u of the cell is
    // within mc = h*sqrt(3)/2 (+ slack for a lookup position that is off by rounding) of it
    const float mc = h * 0.8660254f + 1e-4f, hh = 0.5f * h + 1e-4f;
    // the same for the whole block: centre of the box of its cells, half-diagonal, half-size
    const float cBx = lox + 2.f * h, cBy = loy + 2.f * h, cBz = loz + 2.f * h;
    const float mB = h * 3.4641017f + 1e-4f, hB = 2.f * h + 1e-4f;
    const uint32_t n = M.n_impulses;
    const int H = T.half, S = T.stride;
    const unsigned tside = 2u * (unsigned)H;
    float Ssum = 0.f, Tsum = 0.f, Esum = 0.f, Lsum = 0.f;      // this lane's grid point
    float EB = 0.f, LB = 0.f;                                   // this lane's impulses, block-level terms (summed over the wave at the end)
    for (int dx = -2; dx <= 2; ++dx)
        for (int dy = -2; dy <= 2; ++dy)
            for (int dz = -2; dz <= 2; ++dz) {
                const int cx = c0x + dx, cy = c0y + dy, cz = c0z + dz;
                // a whole lattice cell farther than the cut-off from the block's cells cannot matter
                const float bx0 = (float)cx - (lox + 4.f * h), bx1 = lox - (float)(cx + 1);
                const float by0 = (float)cy - (loy + 4.f * h), by1 = loy - (float)(cy + 1);
                const float bz0 = (float)cz - (loz + 4.f * h), bz1 = loz - (float)(cz + 1);
                const float ex = fmaxf(fmaxf(bx0, bx1), 0.f), ey = fmaxf(fmaxf(by0, by1), 0.f), ez = fmaxf(fmaxf(bz0, bz1), 0.f);
                if (ex * ex + ey * ey + ez * ez >= kGuideCullRadius * kGuideCullRadius)
                    continue;
                float px, py, pz, pw;
                if (T.cells && (unsigned)(cx + H) < tside && (unsigned)(cy + H) < tside && (unsigned)(cz + H) < tside) {
                    const uint32_t idx = (((uint32_t)(cx + H) * tside + (uint32_t)(cy + H)) * tside + (uint32_t)(cz + H)) * (uint32_t)S;
                    const float4 v = T.cells[idx + (uint32_t)(lane & (S - 1))];
                    px = v.x; py = v.y; pz = v.z; pw = v.w;
                } else {
                    gen_impulse((uint32_t)cx, (uint32_t)cy, (uint32_t)cz, M.seed, kJump4.A[lane], kJump4.C[lane], px, py, pz, pw);
                }
                // impulse position relative to the block's low corner, distance to the box of the block's cells
                const float rx = (float)cx + px - lox, ry = (float)cy + py - loy, rz = (float)cz + pz - loz;
                const float gx = fmaxf(fmaxf(-rx, rx - 4.f * h), 0.f), gy = fmaxf(fmaxf(-ry, ry - 4.f * h), 0.f), gz = fmaxf(fmaxf(-rz, rz - 4.f * h), 0.f);
                const bool near_k = (uint32_t)lane < n && (gx * gx + gy * gy + gz * gz) < kGuideCullRadius * kGuideCullRadius;
                unsigned long long cand = __ballot(near_k);
                if (cand == 0ULL)
                    continue;
                // lane = impulse: is it inside for the whole block?  (v_sqrt_f32: 1 ulp, covered by the 2e-5 / 1e-6 slacks)
                const float dBx = cBx - ((float)cx + px), dBy = cBy - ((float)cy + py), dBz = cBz - ((float)cz + pz);
                const float rB = __builtin_amdgcn_sqrtf(dBx * dBx + dBy * dBy + dBz * dBz);
                const bool inside_B = near_k && F.ppc >= 32 && (rB + mB < 1.f - 2e-5f);   // coarse grids have large blocks: the block-level terms would be loose, and their fields are small anyway
                if (inside_B) {
                    const float dmB = fmaxf(rB - mB - 1e-6f, 0.f);
                    const float e_dm = __expf(-amin * dmB * dmB);
                    const float adx = fabsf(dBx) + hB, ady = fabsf(dBy) + hB, adz = fabsf(dBz) + hB;
                    const float cxx = fmaxf(4.f * ax * ax * adx * adx - 2.f * ax, 2.f * ax);
                    const float cyy = fmaxf(4.f * ay * ay * ady * ady - 2.f * ay, 2.f * ay);
                    const float czz = fmaxf(4.f * az * az * adz * adz - 2.f * az, 2.f * az);
                    EB += (cxx + cyy + czz) * e_dm;
                    LB += 2.f * (ax * adx + ay * ady + az * adz) * e_dm;
                }
                const unsigned long long bin = __ballot(inside_B);
                // one wave per workgroup, in-order LDS: the fences only tell the compiler that lanes talk to each other here
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                stage[lane] = make_float4((float)cx + px, (float)cy + py, (float)cz + pz, pw);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                while (cand) {
                    const int k = __builtin_ctzll(cand);
                    cand &= cand - 1ULL;
                    const float4 q4 = stage[k];                       // wave-uniform address: a broadcast
                    // delta = g - (cell + p_i)
                    const float ddx = ux - q4.x, ddy = uy - q4.y, ddz = uz - q4.z;
                    const float q = ax * ddx * ddx + ay * ddy * ddy + az * ddz * ddz;
                    if ((bin >> k) & 1ULL) {                          // inside for every point and cell of the block
                        Ssum += q4.w * __expf(-q);
                        continue;
                    }
                    const float rg2 = ddx * ddx + ddy * ddy + ddz * ddz;
                    if (rg2 < 1.0f)
                        Ssum += q4.w * __expf(-q);
                    // the same impulse seen from the cell centre (delta_c = delta + h/2)
                    const float ccx = ddx + 0.5f * h, ccy = ddy + 0.5f * h, ccz = ddz + 0.5f * h;
                    const float rc = __builtin_amdgcn_sqrtf(ccx * ccx + ccy * ccy + ccz * ccz);
                    if (rc - mc >= 1.f + 1e-5f)
                        continue;                                      // outside on the whole cell
                    const float dm = fmaxf(rc - mc - 1e-6f, 0.f);
                    const float e_dm = __expf(-amin * dm * dm);
                    if (rc + mc < 1.f - 1e-5f) {                       // inside on the whole cell
                        const float adx = fabsf(ccx) + hh, ady = fabsf(ccy) + hh, adz = fabsf(ccz) + hh;
                        const float cxx = fmaxf(4.f * ax * ax * adx * adx - 2.f * ax, 2.f * ax);
                        const float cyy = fmaxf(4.f * ay * ay * ady * ady - 2.f * ay, 2.f * ay);
                        const float czz = fmaxf(4.f * az * az * adz * adz - 2.f * az, 2.f * az);
                        Esum += (cxx + cyy + czz) * e_dm;
                        // Lipschitz term: the exact evaluator's fp32 position u and the lookup position w differ by up to
                        // kPosEps cells per axis, and |d_a w e^{-q}| = 2 alpha_a |delta_a| e^{-q}
                        Lsum += 2.f * (ax * adx + ay * ady + az * adz) * e_dm;
                    } else {                                           // the cut-off sphere crosses the cell
                        Tsum += e_dm;
                    }
                }
            }
    for (int off = 32; off > 0; off >>= 1) {
        EB += __shfl_xor(EB, off, 64);
        LB += __shfl_xor(LB, off, 64);
    }
    float err = (h * h * 0.125f) * (Esum + EB) + Tsum + kGuidePosEps * (Lsum + LB);
    err = err * 1.001f + 5e-4f;
    // the sample goes to its own brick and, when it lies on a low face of the brick, into the shared layer (local index 16) of the
    // allocated bricks before it
    const uint32_t qx = (uint32_t)ix & 15u, qy = (uint32_t)iy & 15u, qz = (uint32_t)iz & 15u;
    F.G[(size_t)slot * kBrickFloats + (qx * kBrickRow + qy) * kBrickRow + qz] = Ssum;
    for (int m = 1; m < 8; ++m) {
        const int ox = m & 1, oy = (m >> 1) & 1, oz = m >> 2;
        if ((ox && (qx || !kx)) || (oy && (qy || !ky)) || (oz && (qz || !kz)))
            continue;
        const uint32_t nslot = slot_of[((uint32_t)(kx - ox) * nbk + (uint32_t)(ky - oy)) * nbk + (uint32_t)(kz - oz)];
        if (nslot != kNoBrick)
            F.G[(size_t)nslot * kBrickFloats + ((ox ? 16u : qx) * kBrickRow + (oy ? 16u : qy)) * kBrickRow + (oz ? 16u : qz)] = Ssum;
    }
    float emax = err;
    for (int off = 32; off > 0; off >>= 1) emax = fmaxf(emax, __shfl_xor(emax, off, 64));
    if (lane == 0 && need[brick]) {
        uint64_t &r = F.blk[((size_t)bx * bs + (size_t)by) * bs + (size_t)bz];
        r = guide_rec_make(guide_rec_slot(r), guide_f2h_up(emax), (uint32_t)r);
    }
}

// Second pass of the build: amax(block) = max |V| over the corners of every cell anchored in the block (5x5x5 grid points: the last
// layer belongs to the neighbouring blocks) + Err(block).  The interpolant of a cell is a convex combination of its corners, so
// |N_ref(u)| <= |G(u)| + Err <= amax for every u the march can look up in the block: when the mean alone exceeds sigma amax / norm
// the sign of the field is the sign of the mean, and the march skips the 8-tap lookup (guide_sign_at, level 0).
GPIS_TU_KERNEL __global__ void __launch_bounds__(256) k_guide_amax(GuideField F, size_t item_offset, size_t n_items, const uint32_t *__restrict__ list,
                                                                  const uint8_t *__restrict__ need)
{
    const size_t item = item_offset + (size_t)blockIdx.x * 256 + threadIdx.x;     // one 4^3 block of one allocated brick
    if (item >= n_items)
        return;
    const uint32_t slot = (uint32_t)(item >> 6), sub = (uint32_t)(item & 63);
    const uint32_t brick = list[slot];
    if (!need[brick])
        return;                                     // allocated only to complete its neighbours' shared layer: keeps its a-priori record
    const int bs = F.side / 4, nbk = F.side / kBrick;
    const int kz = (int)(brick % nbk), ky = (int)((brick / nbk) % nbk), kx = (int)(brick / ((uint32_t)nbk * nbk));
    const uint32_t sx = sub >> 4, sy = (sub >> 2) & 3, sz = sub & 3;
    const int bx = 4 * kx + (int)sx, by = 4 * ky + (int)sy, bz = 4 * kz + (int)sz;
    // the last layer of points of the field's last bricks does not exist (lookups stop at side - 2): clamp like the dense field did
    const uint32_t lx = kx == nbk - 1 ? 15u : 16u, ly = ky == nbk - 1 ? 15u : 16u, lz = kz == nbk - 1 ? 15u : 16u;
    const float *Gb = F.G + (size_t)slot * kBrickFloats;
    float m = 0.f;
    for (uint32_t dx = 0; dx <= 4; ++dx) {
        const uint32_t px = min(4 * sx + dx, lx);
        for (uint32_t dy = 0; dy <= 4; ++dy) {
            const float *row = Gb + (px * kBrickRow + min(4 * sy + dy, ly)) * kBrickRow;
            for (uint32_t dz = 0; dz <= 4; ++dz)
                m = fmaxf(m, fabsf(row[min(4 * sz + dz, lz)]));
        }
    }
    const size_t b = ((size_t)bx * bs + (size_t)by) * bs + (size_t)bz;
    const uint64_t r = F.blk[b];
    F.blk[b] = guide_rec_make(slot, (uint32_t)r >> 16, guide_f2h_up(m + guide_rec_err(r)));     // from here on level 1 may use the block
}

// Which bricks can level 1 be asked about?  One thread per brick: the block records of the brick get the a-priori pair (Err = +inf,
// amax = the coarse field's bound on |N| over the brick: its block of the same index covers exactly this brick's cells), and the brick
// is NEEDED when |mean| can come below sigma amax / norm somewhere in it.  The mean is a function of the world point; a brick is a box
// of the grid space: in world-space sampling a box of world points (p = R u), in isotropic-ray space — where every ray sees the lattice
// through its own rotation — the shell of world points with |W p| = R |u|, W = diag (no anisoMtx on this path).  Interval bounds per
// mean type; the CSG minimum of two means takes the minimum of the bounds.  `coarse` == nullptr: every brick is needed (small fields).
GPIS_TU_KERNEL __global__ void __launch_bounds__(256) k_guide_need(const DevModel *__restrict__ Mp, GuideField F, const uint64_t *__restrict__ coarse_blk, uint8_t *__restrict__ need)
{
    const uint32_t nbk = (uint32_t)F.side / kBrick, brick = blockIdx.x * 256u + threadIdx.x;
    if (brick >= nbk * nbk * nbk)
        return;
    const DevModel &M = *Mp;
    const uint32_t kz = brick % nbk, ky = (brick / nbk) % nbk, kx = brick / (nbk * nbk);
    const uint32_t bs = (uint32_t)F.side / 4;
    bool needed = true;
    float amax = __builtin_huge_valf();
    if (coarse_blk) {
        amax = guide_rec_amax(coarse_blk[brick]);
        const float h = 1.0f / (float)F.ppc, pad = 2e-3f;
        float lo[3], hi[3];                                       // the brick's cells in grid units, padded
        const uint32_t kk[3] = {kx, ky, kz};
        for (int a = 0; a < 3; ++a) {
            lo[a] = (float)(kBrick * kk[a]) * h - (float)F.half - pad;
            hi[a] = lo[a] + (float)kBrick * h + 2.f * pad;
        }
        float mlo = __builtin_huge_valf(), mhi = __builtin_huge_valf();      // range of min(mean, mean_additional)
        for (int w = 0; w < (M.has_mean_additional ? 2 : 1); ++w) {
            const gpis_mean &mu = M.mean[w];
            float vlo, vhi;
            if (mu.type == GPIS_MEAN_HOMOGENEOUS) {
                vlo = vhi = mu.offset;
            } else if (!M.iso3d) {
                // world space: p = R u, a box
                float dlo = 0.f, dhi = 0.f;
                if (mu.type == GPIS_MEAN_SPHERICAL) {
                    float n2 = 0.f, f2 = 0.f;
                    for (int a = 0; a < 3; ++a) {
                        const float c = (float)mu.center[a], pl = F.R * lo[a], ph = F.R * hi[a];
                        const float nearest = fmaxf(fmaxf(pl - c, c - ph), 0.f), farthest = fmaxf(fabsf(pl - c), fabsf(ph - c));
                        n2 += nearest * nearest; f2 += farthest * farthest;
                    }
                    dlo = sqrtf(n2) * 0.9999f; dhi = sqrtf(f2) * 1.0001f;
                    vlo = dlo - mu.radius; vhi = dhi - mu.radius;
                } else {
                    for (int a = 0; a < 3; ++a) {
                        const float c = (float)mu.center[a], l = (float)M.lin_dir[w][a], x0 = (F.R * lo[a] - c) * l, x1 = (F.R * hi[a] - c) * l;
                        dlo += fminf(x0, x1); dhi += fmaxf(x0, x1);
                    }
                    const float s0 = dlo * mu.scale, s1 = dhi * mu.scale;
                    vlo = fmaxf(fminf(s0, s1), mu.min); vhi = fmaxf(fmaxf(s0, s1), mu.min);
                }
            } else {
                // isotropic-ray space: |W p| = R |u| for some rotation of u: a shell of |p|
                float n2 = 0.f, f2 = 0.f, smin = __builtin_huge_valf(), smax = 0.f;
                for (int a = 0; a < 3; ++a) {
                    const float nearest = fmaxf(fmaxf(lo[a], -hi[a]), 0.f), farthest = fmaxf(fabsf(lo[a]), fabsf(hi[a]));
                    n2 += nearest * nearest; f2 += farthest * farthest;
                    const float sv = fabsf(GM(M.w2l, a, a));
                    smin = fminf(smin, sv); smax = fmaxf(smax, sv);
                }
                const float pmin = F.R * sqrtf(n2) / smax * 0.9999f, pmax = F.R * sqrtf(f2) / smin * 1.0001f;
                float cn = 0.f;
                for (int a = 0; a < 3; ++a) cn += (float)mu.center[a] * (float)mu.center[a];
                cn = sqrtf(cn) * 1.0001f;
                if (mu.type == GPIS_MEAN_SPHERICAL) {
                    vlo = fmaxf(fmaxf(pmin - cn, cn - pmax), 0.f) - mu.radius; vhi = pmax + cn - mu.radius;
                } else {
                    const float d = (pmax + cn) * 1.0001f, s = fabsf(mu.scale) * d;     // |direction| = 1
                    vlo = fmaxf(-s, mu.min); vhi = fmaxf(s, mu.min);
                }
            }
            mlo = fminf(mlo, vlo); mhi = fminf(mhi, vhi);
        }
        const float abs_lo = (mlo <= 0.f && mhi >= 0.f) ? 0.f : fminf(fabsf(mlo), fabsf(mhi));
        const float sn = M.sigma / (M.iso3d ? M.norm3d_iso : M.norm3d_world);
        needed = !(abs_lo > amax * sn * 1.001f + 1e-4f + 1e-5f * fmaxf(fabsf(mlo), fabsf(mhi)));
    }
    need[brick] = needed ? 1 : 0;
    // default records of the brick's 64 blocks: level 1 certifies nothing (Err = +inf), level 0 uses the a-priori bound
    for (uint32_t s = 0; s < 64; ++s) {
        const size_t b = ((size_t)(4 * kx + (s >> 4)) * bs + (size_t)(4 * ky + ((s >> 2) & 3))) * bs + (size_t)(4 * kz + (s & 3));
        F.blk[b] = guide_rec_make(0u, 0x7C00u, guide_f2h_up(amax));
    }
}
// Allocation: a brick gets samples when it is needed or completes the shared layer of a needed brick before it (one of the seven
// bricks at -1 along a subset of the axes is needed).  slot_of / list are the two directions of the mapping.
GPIS_TU_KERNEL __global__ void __launch_bounds__(256) k_guide_slots(GuideField F, const uint8_t *__restrict__ need, uint32_t *__restrict__ slot_of,
                                                                   uint32_t *__restrict__ list, uint32_t *__restrict__ counters)
{
    const uint32_t nbk = (uint32_t)F.side / kBrick, brick = blockIdx.x * 256u + threadIdx.x;
    if (brick >= nbk * nbk * nbk)
        return;
    const uint32_t kz = brick % nbk, ky = (brick / nbk) % nbk, kx = brick / (nbk * nbk);
    bool alloc = false;
    for (int m = 0; m < 8 && !alloc; ++m) {
        const uint32_t ox = m & 1, oy = (m >> 1) & 1, oz = m >> 2;
        if ((ox && !kx) || (oy && !ky) || (oz && !kz))
            continue;
        alloc = need[((kx - ox) * nbk + (ky - oy)) * nbk + (kz - oz)] != 0;
    }
    uint32_t slot = kNoBrick;
    if (alloc) {
        slot = atomicAdd(&counters[0], 1u);
        list[slot] = brick;
        if (need[brick]) atomicAdd(&counters[1], 1u);
    }
    slot_of[brick] = slot;
}
// G and the bound at index-space coordinates (tx, ty, tz) = (u + half) * ppc; false outside the
// tabulated volume
GPIS_DEV bool guide_lookup_index(const GuideField &F, float tx, float ty, float tz, float &g, float &err, float &amax);

// G(u), the bound Err for the cell containing u (+inf where the brick is not tabulated) and the block's bound amax on |N|; false when u
// is outside the field's volume
GPIS_DEV bool guide_lookup(const GuideField &F, V3 u, float &g, float &err, float &amax)
{
    const float s = (float)F.ppc, off = (float)F.half;
    return guide_lookup_index(F, (u.x + off) * s, (u.y + off) * s, (u.z + off) * s, g, err, amax);
}
GPIS_DEV bool guide_lookup(const GuideField &F, V3 u, float &g, float &err)
{
    float amax;
    return guide_lookup(F, u, g, err, amax);
}

GPIS_DEV bool guide_lookup_index(const GuideField &F, float tx, float ty, float tz, float &g, float &err, float &amax)
{
    typedef const float __attribute__((address_space(1))) *gfloat_p;   // the field is read through a struct in memory: keep the loads global_load
    const float fx0 = floorf(tx), fy0 = floorf(ty), fz0 = floorf(tz);
    // v_cvt_i32_f32 saturates and maps NaN to 0 (then the weights are NaN and no sign is certified), so one unsigned compare per
    // axis is the whole range check
    const uint32_t ix = (uint32_t)(int)fx0, iy = (uint32_t)(int)fy0, iz = (uint32_t)(int)fz0;
    const uint32_t side = (uint32_t)F.side, lim = side - 2u;
    if (!(ix <= lim && iy <= lim && iz <= lim))
        return false;
    const float wx = tx - fx0, wy = ty - fy0, wz = tz - fz0;
    // side <= 8192 (guide_build): ix*side + iy < 2^26 is a 24-bit multiply-add, the element index one 32x32->64 multiply-add
    // (as size_t arithmetic on ints this was 30 instructions of sign extensions and 64x64 multiplies per march step)
    // the block's record (bounds + the slot of its brick), then the eight taps inside the brick
    const uint32_t bs = side >> 2;
    const uint32_t brow = __umul24(ix >> 2, bs) + (iy >> 2);
    const uint64_t rec = ((const uint64_t __attribute__((address_space(1))) *)F.blk)[(uint64_t)brow * bs + (iz >> 2)];
    const uint32_t slot = guide_rec_slot(rec);
    gfloat_p p00 = (gfloat_p)F.G + ((uint64_t)slot * kBrickFloats + (((ix & 15u) * kBrickRow + (iy & 15u)) * kBrickRow + (iz & 15u)));
    gfloat_p p01 = p00 + kBrickRow, p10 = p00 + kBrickRow * kBrickRow, p11 = p10 + kBrickRow;
    const float a000 = p00[0], a001 = p00[1], a010 = p01[0], a011 = p01[1];
    const float a100 = p10[0], a101 = p10[1], a110 = p11[0], a111 = p11[1];
    // fused lerps: the guide value is not part of any result, its rounding is inside the stored bound's 5e-4 + 1e-3 relative
    const float c00 = __builtin_fmaf(a001 - a000, wz, a000), c01 = __builtin_fmaf(a011 - a010, wz, a010);
    const float c10 = __builtin_fmaf(a101 - a100, wz, a100), c11 = __builtin_fmaf(a111 - a110, wz, a110);
    const float c0 = __builtin_fmaf(c01 - c00, wy, c00), c1 = __builtin_fmaf(c11 - c10, wy, c10);
    g = __builtin_fmaf(c1 - c0, wx, c0);
    err = guide_rec_err(rec);         // +inf where the brick is not tabulated (g is then some other brick's value: never used against a finite bound)
    amax = guide_rec_amax(rec);
    return true;
}

// fp32 mean for the certification (the exact path evaluates it in double, GPF.hpp:903-945); `slack` returns a bound on
// its absolute error.  `perr` bounds the 1-norm distance between p and the position the exact evaluator uses.
GPIS_DEV float mean_approx(const DevModel &M, V3 p, float perr, float &slack)
{
    float best = 0.f;
    slack = 0.f;
    for (int w = 0; w < (M.has_mean_additional ? 2 : 1); ++w) {
        const gpis_mean &mu = M.mean[w];
        float v, s;
        if (mu.type == GPIS_MEAN_HOMOGENEOUS) {
            v = mu.offset; s = 0.f;
        } else if (mu.type == GPIS_MEAN_SPHERICAL) {
            const float dx = p.x - (float)mu.center[0], dy = p.y - (float)mu.center[1], dz = p.z - (float)mu.center[2];
            // v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf (16 instructions): its error is inside the 6e-7 r term
            const float r = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
            v = r - mu.radius;
            s = 6e-7f * (r + fabsf((float)mu.center[0]) + fabsf((float)mu.center[1]) + fabsf((float)mu.center[2]) + fabsf(mu.radius)) + 1e-7f + perr;   // |r(p) - r(p')| <= |p - p'|
        } else {
            const float dx = p.x - (float)mu.center[0], dy = p.y - (float)mu.center[1], dz = p.z - (float)mu.center[2];
            const float l0 = (float)M.lin_dir[w][0], l1 = (float)M.lin_dir[w][1], l2 = (float)M.lin_dir[w][2];
            const float d = dx * l0 + dy * l1 + dz * l2;
            v = fmaxf(d * mu.scale, mu.min);
            s = 1e-6f * (fabsf(d * mu.scale) + fabsf(mu.scale) * (fabsf(p.x) + fabsf(p.y) + fabsf(p.z) + 1.f)) + 1e-7f
                + 1.001f * fabsf(mu.scale) * fmaxf(fabsf(l0), fmaxf(fabsf(l1), fabsf(l2))) * perr;
        }
        if (w == 0 || v < best) best = v;          // CSG min (GaussianProcess.cpp:379-393): |min(a,b) - min(a',b')| <= max err
        slack = fmaxf(slack, s);
    }
    return best;
}

// ---- the guided march -------------------------------------------------------------------------
// Per lane (SCNM.cpp:102-183 restated around the certification):
//   G_INIT    sign0 = sign(f(nearT)) from the guide, else X_F0 (exact)
//   G_MARCH   one guide step per iteration: a certified step with the expected sign just advances;
//             anything else (uncertain, or a certified sign change) parks the lane for an exact
//             evaluation at t (X_CUR) and — when a crossing is confirmed while the previous step's
//             value is only known by sign — at the previous position (X_PREV), before the reference's
//             secant-and-shrink refinement (X_REFINE) runs on exact values only
//   X_FINAL   exact lastVal at farT (sampleDistance only), G_GRAD exact gradient, G_DONE
// Exact evaluations are cooperative (coop_evaluate_value); parked lanes are served in clusters whose
// grid cells span at most 2 per axis, so the evaluator always takes its coherent path.
constexpr int32_t kSegPending = 0x7FFFFFF0;     // gpis_seg_out::ok of a record that waits for its gradient (k_guided_range_grad)
enum GPhase : int { G_INIT = 0, G_MARCH = 1, X_F0 = 2, X_CUR = 3, X_PREV = 4, X_REFINE = 5, X_FINAL = 6, G_GRAD = 7, G_DONE = 8 };

GPIS_DEV V3 grid_point(const DevModel &M, const GuideField &F, V3 p, const Frame &coord)
{
    if (!M.iso3d)
        return p / F.R;
    return to_local(coord, spec_3d::cov_pos_w2l(M, p, 1.0f)) / F.R;
}

// certified sign of the field at world point p: +1 / -1, or 0 when the guide cannot decide
GPIS_DEV int guide_sign(const DevModel &M, const GuideField &F, V3 p, const Frame &coord)
{
    float g, err;
    if (!guide_lookup(F, grid_point(M, F, p, coord), g, err))
        return 0;
    const float norm = M.iso3d ? M.norm3d_iso : M.norm3d_world;
    float ms;
    const float mean = mean_approx(M, p, 0.f, ms);
    const float nv = M.sigma * (g / norm);
    const float fa = nv + mean;
    const float margin = M.sigma * (err / norm) * 1.0001f + ms + 2e-6f * (fabsf(nv) + fabsf(mean)) + 1e-7f;
    if (fa > margin) return 1;
    if (fa < -margin) return -1;
    return 0;
}

// The march visits p(t) = pos + t*dir, and the grid position is a LINEAR map of p (iso-ray space: rotation into the
// ray frame of the whitened position; world space: p / R), so the index-space coordinates of a step are a + t*b with
// a, b computed once per segment in double and kept in fp32; a step evaluates them with one fp32 fma each.
// Both are ANCHORED at the segment start t0 = nearT (a = index coordinates of pos + t0 dir, steps use t - t0): the fp32
// magnitudes |a| and |(t - t0) b| then stay of the order of the field's side whatever the distance of the ray origin from the
// medium (with a = coordinates of `pos` itself an origin 500 world units away put |a| ~ 3e5 index units, i.e. 1.2e-4 cells
// per rounding — outside the budget below).  A segment whose anchor is further than 200 lattice cells from the field's centre
// (nearT far from the medium's bounds) is not certified at all (a = -3e38: the range check of every step fails).
// Error of the lookup position against the position u the exact evaluator derives for the same t (the stored bound
// carries kGuidePosEps = 1e-4 cells for it, k_guide_build), in units of ppc index units = cells: |a| <= 200; inside the field
// |a + ts b| <= 2 half <= 128, hence |ts b| <= 328.  Roundings: a (2^-24 x 200), ts = (float)(t - t0) and b (2^-24 x 328
// each), the fma's result (2^-24 x 128): together 984 x 6e-8 = 5.9e-5 cells; u itself is a chain of ~8 fp32 roundings of
// |u| <= half cells (measured < 2e-5 at half = 16).  gpis_guide_raycheck walks real rays — camera, shadow, and bundles
// whose origins lie 15 (unanchored), 44, 500 and 2000 world units away — and checks every certified sign against the exact
// value.
// The mean is evaluated at pos0 + (float)(t - t0) * dir in fp32, pos0 = float(pos + t0 dir); `pe0 + pe1 |t - t0|` bounds
// its 1-norm distance from the exact evaluator's float(pos_d + t dir_d): (|ts - (float)ts| |dir_c| + the roundings of
// pos0_c, of the fma and of the exact position) summed over the components.
struct GuideRay {
    float ax, ay, az, bx, by, bz;    // index coordinates = a + (t - t0)*b
    V3 pos, dir;                     // the ray anchored at t0: pos = float(origin + t0 dir) (world space)
    float t0;                        // = nearT (the march already holds it: no extra register)
    float pe0, pe1;                  // position slack = pe0 + pe1 * |t - t0|
    float sn;                        // sigma / norm
};

GPIS_DEV GuideRay guide_ray(const DevModel &M, const GuideField &F, V3 pos, V3 dir, const Frame &coord, float nearT)
{
    auto lin = [&](V3d v, double &x, double &y, double &z) {
        double wx = v.x, wy = v.y, wz = v.z;
        if (M.iso3d) {
            const float *m = M.w2l;    // stationary media only: cov_xf_scale == 1 (cov_pos_w2l)
            const double lx = (double)GM(m, 0, 0) * wx + (double)GM(m, 0, 1) * wy + (double)GM(m, 0, 2) * wz;
            const double ly = (double)GM(m, 1, 0) * wx + (double)GM(m, 1, 1) * wy + (double)GM(m, 1, 2) * wz;
            const double lz = (double)GM(m, 2, 0) * wx + (double)GM(m, 2, 1) * wy + (double)GM(m, 2, 2) * wz;
            wx = (double)coord.tangent.x * lx + (double)coord.tangent.y * ly + (double)coord.tangent.z * lz;
            wy = (double)coord.bitangent.x * lx + (double)coord.bitangent.y * ly + (double)coord.bitangent.z * lz;
            wz = (double)coord.normal.x * lx + (double)coord.normal.y * ly + (double)coord.normal.z * lz;
        }
        const double s = (double)F.ppc / (double)F.R;
        x = wx * s; y = wy * s; z = wz * s;
    };
    double ax, ay, az, bx, by, bz;
    GuideRay g;
    g.t0 = nearT;
    const V3d pos0 = ray_at(to_d(pos), to_d(dir), (double)nearT);      // the segment's start, in double
    lin(pos0, ax, ay, az);
    lin(to_d(dir), bx, by, bz);
    const double offd = (double)F.half * (double)F.ppc;
    const float off = (float)offd;
    g.ax = (float)(ax + offd); g.ay = (float)(ay + offd); g.az = (float)(az + offd);
    const float far = 200.f * (float)F.ppc, mid = off;      // |anchor - field centre| <= 200 cells per axis
    if (!(fabsf(g.ax - mid) <= far && fabsf(g.ay - mid) <= far && fabsf(g.az - mid) <= far))
        g.ax = -3e38f;               // anchor far from the field: no step of this segment passes the range check
    g.bx = (float)bx; g.by = (float)by; g.bz = (float)bz;
    g.pos = to_f(pos0); g.dir = dir;
    const float d1 = fabsf(dir.x) + fabsf(dir.y) + fabsf(dir.z);
    g.pe0 = 2.5e-7f * (fabsf(g.pos.x) + fabsf(g.pos.y) + fabsf(g.pos.z));
    g.pe1 = 2.5e-7f * d1;
    g.sn = M.sigma / (M.iso3d ? M.norm3d_iso : M.norm3d_world);
    return g;
}

// certified sign at march parameter t (world point pos + t*dir): +1 / -1, or 0 when undecided.
// Level 0: the block's bound on |N| against the mean (one 8-byte load); level 1: the trilinear lookup.  Far from the surface
// — most steps of a march — level 0 decides.
GPIS_DEV int guide_sign_at(const DevModel &M, const GuideField &F, const GuideRay &gr, double t)
{
    typedef const float __attribute__((address_space(1))) *gfloat_p;
    const float tf = (float)(t - (double)gr.t0);
    const float tx = __builtin_fmaf(tf, gr.bx, gr.ax), ty = __builtin_fmaf(tf, gr.by, gr.ay), tz = __builtin_fmaf(tf, gr.bz, gr.az);
    const float fx0 = floorf(tx), fy0 = floorf(ty), fz0 = floorf(tz);
    // v_cvt_i32_f32 saturates and maps NaN to 0 (then the comparisons below fail and no sign is certified)
    const uint32_t ix = (uint32_t)(int)fx0, iy = (uint32_t)(int)fy0, iz = (uint32_t)(int)fz0;
    const uint32_t side = (uint32_t)F.side, lim = side - 2u;
    if (!(ix <= lim && iy <= lim && iz <= lim))
        return 0;
    const uint32_t bs = side >> 2;
    const uint32_t brow = __umul24(ix >> 2, bs) + (iy >> 2);
    const uint64_t rec = ((const uint64_t __attribute__((address_space(1))) *)__builtin_assume_aligned(F.blk, 8))[(uint64_t)brow * bs + (iz >> 2)];   // one 8-byte load
    const float e_amax = guide_rec_amax(rec);
    const V3 p = v3(__builtin_fmaf(tf, gr.dir.x, gr.pos.x), __builtin_fmaf(tf, gr.dir.y, gr.pos.y), __builtin_fmaf(tf, gr.dir.z, gr.pos.z));
    float ms;
    const float mean = mean_approx(M, p, __builtin_fmaf(fabsf(tf), gr.pe1, gr.pe0), ms);
    // level 0: |sigma N / norm| <= amax sn everywhere in the block
    if (fabsf(mean) > e_amax * gr.sn * 1.0001f + ms + 4e-6f * fabsf(mean) + 1e-7f)
        return mean > 0.f ? 1 : -1;
    // level 1
    if (((uint32_t)rec >> 16) == 0x7C00u)
        return 0;                    // Err = +inf: the brick is not tabulated, nothing to certify with (the exact evaluation decides)
    const float e_err = guide_rec_err(rec);
    const float wx = tx - fx0, wy = ty - fy0, wz = tz - fz0;
    // the eight taps inside the block's brick (17^3 floats: the +1 taps never leave it)
    const uint32_t slot = guide_rec_slot(rec);
    gfloat_p p00 = (gfloat_p)F.G + ((uint64_t)slot * kBrickFloats + (((ix & 15u) * kBrickRow + (iy & 15u)) * kBrickRow + (iz & 15u)));
    gfloat_p p01 = p00 + kBrickRow, p10 = p00 + kBrickRow * kBrickRow, p11 = p10 + kBrickRow;
    const float a000 = p00[0], a001 = p00[1], a010 = p01[0], a011 = p01[1];
    const float a100 = p10[0], a101 = p10[1], a110 = p11[0], a111 = p11[1];
    const float c00 = __builtin_fmaf(a001 - a000, wz, a000), c01 = __builtin_fmaf(a011 - a010, wz, a010);
    const float c10 = __builtin_fmaf(a101 - a100, wz, a100), c11 = __builtin_fmaf(a111 - a110, wz, a110);
    const float c0 = __builtin_fmaf(c01 - c00, wy, c00), c1 = __builtin_fmaf(c11 - c10, wy, c10);
    const float g = __builtin_fmaf(c1 - c0, wx, c0);
    const float nv = g * gr.sn;
    const float fa = nv + mean;
    // 4e-6 relative also covers sigma*(g/norm) vs g*(sigma/norm) and the product roundings
    const float margin = e_err * gr.sn * 1.0001f + ms + 4e-6f * (fabsf(nv) + fabsf(mean)) + 1e-7f;
    if (fa > margin) return 1;
    if (fa < -margin) return -1;
    return 0;
}

// DEFER_GRAD (sampleDistance only; GPIS_OPT_DEFER_GRAD): the segment's gradient evaluation is left to a second kernel — the record is
// written with ok = kSegPending and k_guided_range_grad completes it — so that the march kernel does not carry the gradient
// evaluator (the spill experiment of DESIGN.md 8).
template <bool WANT_SAMPLE, bool SMALLARG, bool DEFER_GRAD = false>
GPIS_DEV void guided_march(const DevModel &M, const FastTable &T, const GuideField &F, FastLds &lds, bool valid,
                           const gpis_ray_in *__restrict__ rayp, gpis_seg_out *out, bool &visible, uint32_t &n_eval, uint32_t &n_guide)
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
    // The ray's isotropic-ray frame (SCN.cpp:296-297) is needed by the exact evaluations only; it is
    // re-derived from `dir` there instead of being kept live across the march (9 VGPRs of a kernel
    // that spills).  The empty asm hides the loop invariance from LICM, which would hoist it back.
    auto ray_frame = [&]() {
        Frame c{};
        if (M.iso3d) {
            V3 d = dir;
            asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z));
            c = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, d, 1.0f)));
        }
        return c;
    };
    GuideRay gr;
    {
        const Frame coord = ray_frame();
        gr = guide_ray(M, F, pos, dir, coord, nearT);
    }
    int phase = G_INIT;
    bool early_ok = false;
    if (!valid)
        phase = G_DONE;
    else if (WANT_SAMPLE && bounce >= M.max_bounces)
        phase = G_DONE;
    else if (WANT_SAMPLE && farT == 0.f) {
        phase = G_DONE;
        early_ok = true;
    }
    double t = (double)nearT;          // position of the march step being decided
    double t_prevpos = (double)nearT;  // position the previous value `pf` belongs to
    double a_lo = 0., intp = 0., t_test = 0., t_prev = 0.;
    float pf = 0.f, fc = 0.f;
    bool pf_valid = false;             // pf is an exact value (otherwise only its sign is known)
    int sign0 = 1, step = 0, gp = 0;
    float last_val = 0.f;
    bool hit = false;

    auto world_at = [&](double tq) { return to_f(ray_at(to_d(pos), to_d(dir), tq)); };
    // after the evaluation at nearT: t = nearT + step*u (float arithmetic, SCNM.cpp:129)
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
    // one trip of the shrink loop (SCNM.cpp:147-160).  It consumes only the SIGN of f(t_test), but asking
    // the guide first does not pay: t_test lies inside the crossing step, i.e. inside the uncertainty band
    // (measured on C1: 4 033 of 86.9 M trips certified)
    auto refine_step = [&](int sign_test) {
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
            phase = WANT_SAMPLE ? G_GRAD : G_DONE;
        }
    };

    long long clk_a = 0, clk_b = 0;      // diagnostic builds only (GPIS_FAST_STATS)
    for (;;) {
        // ---- A: guide steps for every lane that can take one ----
        const long long clk_a0 = FCLK();
        for (;;) {
            if (!WANT_SAMPLE && phase == X_FINAL) {
                hit = false;                // transmittance: the segment exits, lastVal is not part of the result
                phase = G_DONE;
            }
            const bool stepping = phase == G_INIT || phase == G_MARCH;
            const unsigned long long step_mask = __ballot(stepping);
            if (step_mask == 0ULL)
                break;
            FSTAT(20, 1);
            FSTAT(21, __popcll(step_mask));
            if (stepping) {
                if (phase == G_INIT) {
                    const int s = guide_sign_at(M, F, gr, (double)nearT);
                    if (s != 0) {
                        n_guide++;          // counts certified steps: each stands for one evaluateValue of the reference
                        sign0 = s;
                        pf_valid = false;
                        begin_march();
                    } else {
                        phase = X_F0;
                    }
                } else {
                    const int s = guide_sign_at(M, F, gr, t);
                    const bool adopt = !first_scatter && step == 0;   // the reference's `step == 1` after step++ (SCNM.cpp:138-140)
                    if (s != 0 && (adopt || s == sign0)) {
                        n_guide++;
                        step++;
                        if (adopt) sign0 = s;
                        pf_valid = false;
                        advance();
                    } else {
                        phase = X_CUR;      // uncertain, or a certified sign change: exact values needed
                    }
                }
            }
        }
        clk_a += FCLK() - clk_a0;
        // ---- B: exact evaluations for the parked lanes, one coherent cluster at a time ----
        const long long clk_b0 = FCLK();
        const bool need = phase >= X_F0 && phase <= X_FINAL;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask == 0ULL)
            break;
        const double tq = phase == X_F0 ? (double)nearT : (phase == X_PREV ? t_prevpos : (phase == X_REFINE ? t_test : (phase == X_FINAL ? (double)farT : t)));
        const V3 pq = world_at(tq);
        const Frame coord = ray_frame();
        const V3 ug = grid_point(M, F, pq, coord);
        const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
        const int lead = __builtin_ctzll(need_mask);
        const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
        // the lead lane is always in its own cluster, so every round retires at least one request
        const bool in_cluster = need && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
        int gp_new;
        float fv;
        const unsigned long long cl_mask = __ballot(in_cluster);
        FSTAT(22, 1);
        FSTAT(23, __popcll(need_mask));
        if (__popcll(cl_mask) <= kGuideSoloMax) {
            // few requests: one sideways (lane = impulse) evaluation per requesting lane
            FSTAT(24, 1);
            FSTAT(25, __popcll(cl_mask));
            fv = 0.f;
            gp_new = 0;
            for (unsigned long long mm = cl_mask; mm; mm &= mm - 1ULL) {
                const int src = __builtin_ctzll(mm);
                int gpx;
                const float v = solo_evaluate_value<kSoloPacked>(M, T, lds, src, pq, coord, gpx, n_eval);
                if ((int)(threadIdx.x & 63) == src) { fv = v; gp_new = gpx; }
            }
        } else {
            // at most half a wave in the cluster: the idle half helps (two-way candidate split, gpis_fast.hpp)
            fv = coop_evaluate_value<SMALLARG>(M, T, lds, in_cluster, pq, coord, gp_new, n_eval, kGuideSplit && __popcll(cl_mask) <= 32);
        }
#ifdef GPIS_FAST_STATS
        for (int ph = X_F0; ph <= X_FINAL; ++ph) {
            const int served = __popcll(__ballot(in_cluster && phase == ph));
            FSTAT(26 + ph - X_F0, served);
        }
#endif
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
                phase = G_GRAD;
            }
        }
        clk_b += FCLK() - clk_b0;
    }
    FSTAT(16, clk_a);
    FSTAT(17, clk_b);
    FSTAT(18, clk_a + clk_b);
    visible = valid && !hit;
    if (!WANT_SAMPLE)
        return;

    // one gradient evaluation per segment (GPM.cpp:283 / 319), clustered like the value requests
    const long long clk_g0 = FCLK();
    const bool want_grad = phase == G_GRAD;
    V3 g = v3(0.f, 0.f, 0.f);
    if (DEFER_GRAD) {
        if (valid && want_grad) {
            out->t = t;
            out->exited = hit ? 0 : 1;
            out->last_val = last_val;
            out->gp_id = gp;
            out->ok = kSegPending;
            return;
        }
    } else {
        V3d rdn = to_d(dir);
        { double inv = 1.0 / length_d(rdn); rdn.x *= inv; rdn.y *= inv; rdn.z *= inv; }
        const V3 pgq = to_f(ray_at(to_d(pos), rdn, t));
        const Frame coord = ray_frame();
        const V3 ug = grid_point(M, F, pgq, coord);
        const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
        bool pending = want_grad;
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
    }
    FSTAT(19, FCLK() - clk_g0);
    if (!valid)
        return;
    finish_sample_distance(M, rayp, pos, dir, farT, early_ok, want_grad, hit, t, last_val, gp, g, out);
}

struct GuideCounters { unsigned long long n_guide; };

template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_GUIDE_OCC) k_guided_sample_distance(const DevModel *__restrict__ Mp, FastTable T, const GuideField *__restrict__ Fp, size_t n,
                                                                                     const gpis_ray_in *__restrict__ rays, gpis_seg_out *__restrict__ out,
                                                                                     gpis_cond_coeff *__restrict__ coeff, const uint8_t *__restrict__ mask,
                                                                                     Counters *cnt, unsigned long long *guide_cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    uint32_t n_eval = 0, n_guide = 0;
    bool vis;
    guided_march<true, SMALLARG>(*Mp, T, *Fp, lds, valid, rays + (valid ? i : 0), out + (valid ? i : 0), vis, n_eval, n_guide);
    if (valid && coeff) {
        gpis_cond_coeff c;
        memset(&c, 0, sizeof c);
        c.n_evals = n_eval;
        coeff[i] = c;
    }
    fast_flush_counters(cnt, n_eval, valid ? 1u : 0u);
    unsigned long long gsum = n_guide;
    for (int off = 32; off > 0; off >>= 1) gsum += __shfl_down(gsum, off, 64);
    if ((threadIdx.x & 63) == 0 && gsum) atomicAdd(guide_cnt, gsum);
}

// the same march with the gradient evaluation deferred to k_guided_range_grad (GPIS_OPT_DEFER_GRAD)
template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_GUIDE_OCC) k_guided_sample_distance_nograd(const DevModel *__restrict__ Mp, FastTable T, const GuideField *__restrict__ Fp, size_t n,
                                                                                            const gpis_ray_in *__restrict__ rays, gpis_seg_out *__restrict__ out,
                                                                                            gpis_cond_coeff *__restrict__ coeff, const uint8_t *__restrict__ mask,
                                                                                            Counters *cnt, unsigned long long *guide_cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    uint32_t n_eval = 0, n_guide = 0;
    bool vis;
    guided_march<true, SMALLARG, true>(*Mp, T, *Fp, lds, valid, rays + (valid ? i : 0), out + (valid ? i : 0), vis, n_eval, n_guide);
    if (valid && coeff) {
        gpis_cond_coeff c;
        memset(&c, 0, sizeof c);
        c.n_evals = n_eval;
        coeff[i] = c;
    }
    fast_flush_counters(cnt, n_eval, valid ? 1u : 0u);
    unsigned long long gsum = n_guide;
    for (int off = 32; off > 0; off >>= 1) gsum += __shfl_down(gsum, off, 64);
    if ((threadIdx.x & 63) == 0 && gsum) atomicAdd(guide_cnt, gsum);
}

template <bool SMALLARG>
__global__ void __launch_bounds__(kFastBlock, GPIS_GUIDE_OCC_TR) k_guided_transmittance(const DevModel *__restrict__ Mp, FastTable T, const GuideField *__restrict__ Fp, size_t n,
                                                                                   const gpis_ray_in *__restrict__ rays, uint8_t *__restrict__ visible,
                                                                                   const uint8_t *__restrict__ mask, Counters *cnt, unsigned long long *guide_cnt)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n && (!mask || mask[i]);
    uint32_t n_eval = 0, n_guide = 0;
    bool vis = false;
    guided_march<false, SMALLARG>(*Mp, T, *Fp, lds, valid, rays + (valid ? i : 0), nullptr, vis, n_eval, n_guide);
    if (i < n)
        visible[i] = (valid && vis) ? 1 : 0;
    fast_flush_counters(cnt, n_eval, valid ? 1u : 0u);
    unsigned long long gsum = n_guide;
    for (int off = 32; off > 0; off >>= 1) gsum += __shfl_down(gsum, off, 64);
    if ((threadIdx.x & 63) == 0 && gsum) atomicAdd(guide_cnt, gsum);
}

// Self-check (test surface): for n query points — taken in coherent groups of 64 — evaluates the exact
// noise3D sum and the guide, and counts violations of |N_ref - G| <= Err.  stats[0] = points checked,
// stats[1] = violations, stats[2] = bits of max(|N_ref - G| / Err), stats[3] = bits of the mean Err.
GPIS_TU_KERNEL __global__ void __launch_bounds__(kFastBlock) k_guide_selfcheck(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n,
                                                               const float *__restrict__ points3, unsigned long long *stats, float *ratio_max, float *err_sum)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n;
    V3 u = v3(0.f, 0.f, 0.f);
    if (valid) u = v3(points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]);
    // u is given in grid space already: evaluate the exact sum there
    float A0, A1, A2;
    if (!M.iso3d) { A0 = M.invcov_world[0] * 0.5f; A1 = M.invcov_world[4] * 0.5f; A2 = M.invcov_world[8] * 0.5f; }
    else { A0 = A1 = A2 = 0.5f; }
    const V3 p = F.R * u;
    const V4 ex = coop_noise3d<false>(M, T, lds, valid, p, M.seed, F.R, A0, A1, A2);
    float g, err, amax;
    const V3 pg = p / F.R;   // the grid point the exact evaluator used
    const bool ok = valid && guide_lookup(F, pg, g, err, amax);
    float ratio = 0.f;
    unsigned long long bad = 0, cnt = 0, tab = 0;
    float es = 0.f;
    if (ok) {
        // everywhere in the field: the block's bound amax on |N| (level 0 of the certificate); in tabulated bricks also |N - G| <= Err
        cnt = 1;
        bad = fabsf(ex.v) > amax ? 1 : 0;
        if (err < 3e38f) {
            tab = 1;
            ratio = fabsf(ex.v - g) / err;
            bad |= ratio > 1.0f ? 1 : 0;
            es = err;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        bad += __shfl_down(bad, off, 64);
        cnt += __shfl_down(cnt, off, 64);
        tab += __shfl_down(tab, off, 64);
        ratio = fmaxf(ratio, __shfl_down(ratio, off, 64));
        es += __shfl_down(es, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[0], cnt);
        atomicAdd(&stats[1], bad);
        atomicAdd(&stats[3], tab);
        atomicMax((unsigned int *)ratio_max, __float_as_uint(ratio));   // non-negative floats order like their bits
        atomicAdd(err_sum, es);
    }
}

// Test surface for the march's certificate: walks the first `steps` march positions of every ray
// (t = nearT + (k + u)*step, SCNM.cpp:129-132), and wherever the guide certifies a sign compares it
// with the sign of the exact evaluateValue at the same t.  stats[0] = certified steps, stats[1] =
// certified steps whose exact sign differs (must stay 0).
GPIS_TU_KERNEL __global__ void __launch_bounds__(kFastBlock) k_guide_raycheck(const DevModel *__restrict__ Mp, FastTable T, GuideField F, size_t n,
                                                              const gpis_ray_in *__restrict__ rays, uint32_t steps, unsigned long long *stats)
{
    __shared__ FastLds lds;
    fast_lds_init(lds);
    const DevModel &M = *Mp;
    size_t i = (size_t)blockIdx.x * kFastBlock + threadIdx.x;
    const bool valid = i < n;
    V3 pos = v3(0.f, 0.f, 1.f), dir = v3(0.f, 0.f, 1.f);
    float nearT = 0.f, farT = 1.f, u = 0.f;
    if (valid) {
        pos = v3(rays[i].pos[0], rays[i].pos[1], rays[i].pos[2]);
        dir = v3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        nearT = rays[i].near_t; farT = rays[i].far_t; u = rays[i].u_jitter;
    }
    float step_size = (farT - nearT) / (float)M.min_step;
    if (M.step_size < step_size)
        step_size = M.step_size;
    Frame coord{};
    if (M.iso3d)
        coord = frame_from_normal(normalized(spec_3d::cov_pos_w2l(M, dir, 1.0f)));
    const GuideRay gr = guide_ray(M, F, pos, dir, coord, nearT);
    double t = (double)(nearT + step_size * u);
    unsigned long long certified = 0, bad = 0;
    uint32_t n_eval = 0;
    for (uint32_t k = 0; k < steps; ++k) {
        const bool live = valid && t < (double)farT;
        const V3 p = to_f(ray_at(to_d(pos), to_d(dir), t));
        const int s = live ? guide_sign_at(M, F, gr, t) : 0;
        // exact values for the lanes that hold a certificate, clustered like the march does
        bool pending = s != 0;
        const V3 ug = grid_point(M, F, p, coord);
        const int cx = (int)floorf(ug.x), cy = (int)floorf(ug.y), cz = (int)floorf(ug.z);
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (pm == 0ULL)
                break;
            const int lead = __builtin_ctzll(pm);
            const int ax0 = __builtin_amdgcn_readlane(cx, lead), ay0 = __builtin_amdgcn_readlane(cy, lead), az0 = __builtin_amdgcn_readlane(cz, lead);
            const bool in_cluster = pending && cx >= ax0 && cx <= ax0 + 1 && cy >= ay0 && cy <= ay0 + 1 && cz >= az0 && cz <= az0 + 1;
            int gp;
            const float fv = coop_evaluate_value(M, T, lds, in_cluster, p, coord, gp, n_eval);
            if (in_cluster) {
                certified++;
                if ((((double)fv < 0) ? -1 : 1) != s)
                    bad++;
                pending = false;
            }
        }
        t += (double)step_size;
    }
    for (int off = 32; off > 0; off >>= 1) {
        certified += __shfl_down(certified, off, 64);
        bad += __shfl_down(bad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[0], certified);
        atomicAdd(&stats[1], bad);
    }
}

inline void guide_free(GuideField *F)
{
    if (F->G) (void)hipFree(F->G);
    if (F->blk) (void)hipFree(F->blk);
    F->G = nullptr; F->blk = nullptr; F->enabled = 0; F->n_alloc = 0; F->n_usable = 0;
}

// Builds the guide field for the grid space of medium M (world space or isotropic-ray space).  `sparse`: tabulate only the bricks
// level 1 of the certificate can be asked about (a 4x coarser field, built first, bounds |N| per brick); otherwise every brick.
inline int guide_build(const DevModel &M, const DevModel *d_model, const FastTable &T, int half, int ppc, GuideField *F, bool sparse = true)
{
    guide_free(F);
    if (!fast_supported(M) || half < 2 || half > 64 || (ppc != 8 && ppc != 16 && ppc != 32 && ppc != 64))
        return GPIS_ERR_UNSUPPORTED;
    F->half = half; F->ppc = ppc; F->side = 2 * half * ppc;
    if (!M.iso3d) {
        F->R = M.radius_world;
        for (int a = 0; a < 3; ++a) F->alpha[a] = (M.invcov_world[4 * a] * 0.5f) * F->R * F->R;
    } else {
        F->R = M.radius_iso;
        for (int a = 0; a < 3; ++a) F->alpha[a] = 0.5f * F->R * F->R;
    }
    const size_t nblk = ((size_t)F->side / 4) * (F->side / 4) * (F->side / 4);
    const uint32_t nbricks = (uint32_t)(nblk / 64);
    const unsigned bgrid = (nbricks + 255u) / 256u;
    uint8_t *need = nullptr;
    uint32_t *slot_of = nullptr, *list = nullptr, *counters = nullptr;
    GuideField C{};
    auto fail = [&](int code) {
        (void)hipGetLastError();
        if (need) (void)hipFree(need);
        if (slot_of) (void)hipFree(slot_of);
        if (list) (void)hipFree(list);
        if (counters) (void)hipFree(counters);
        guide_free(&C);
        guide_free(F);
        return code;
    };
    if (hipMalloc(&F->blk, nblk * sizeof(uint64_t)) != hipSuccess) { F->blk = nullptr; return fail(GPIS_ERR_DEVICE); }
    if (hipMalloc(&need, nbricks) != hipSuccess || hipMalloc(&slot_of, (size_t)nbricks * 4) != hipSuccess || hipMalloc(&list, (size_t)nbricks * 4) != hipSuccess ||
        hipMalloc(&counters, 8) != hipSuccess || hipMemset(counters, 0, 8) != hipSuccess)
        return fail(GPIS_ERR_DEVICE);
    // 1. which bricks does level 1 need?  (a-priori bound on |N| per brick from the field at ppc / 4: its 4-point block = this brick)
    if (sparse && ppc >= 32) {
        const int st = guide_build(M, d_model, T, half, ppc / 4, &C, false);
        if (st != GPIS_OK) return fail(st);
    }
    k_guide_need<0><<<bgrid, 256>>>(d_model, *F, C.blk, need);
    k_guide_slots<0><<<bgrid, 256>>>(*F, need, slot_of, list, counters);
    uint32_t cnt[2] = {0, 0};
    if (hipMemcpy(cnt, counters, 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(GPIS_ERR_DEVICE);
    guide_free(&C);
    F->n_alloc = cnt[0]; F->n_usable = cnt[1];
    // 2. the samples of the allocated bricks (at least one brick exists: the lookup table points unused bricks at slot 0)
    const size_t pool = (size_t)(cnt[0] ? cnt[0] : 1u) * kBrickFloats;
    if (hipMalloc(&F->G, pool * sizeof(float)) != hipSuccess) { F->G = nullptr; return fail(GPIS_ERR_DEVICE); }
    if (!cnt[0] && hipMemset(F->G, 0, pool * sizeof(float)) != hipSuccess) return fail(GPIS_ERR_DEVICE);
    const size_t n_items = (size_t)cnt[0] * 64, per_launch = (size_t)1 << 22;   // slabs of 4 Mi blocks
    for (size_t i0 = 0; i0 < n_items; i0 += per_launch)
        k_guide_build<0><<<(unsigned)(n_items - i0 < per_launch ? n_items - i0 : per_launch), 64>>>(d_model, T, *F, i0, list, slot_of, need);
    // 3. the blocks' bounds on |N| (reads the samples incl. the shared layers) and their bricks' slots
    for (size_t i0 = 0; i0 < n_items; i0 += (size_t)1 << 30)
        k_guide_amax<0><<<(unsigned)(((n_items - i0 < ((size_t)1 << 30) ? n_items - i0 : ((size_t)1 << 30)) + 255) / 256), 256>>>(*F, i0, n_items, list, need);
    if (hipDeviceSynchronize() != hipSuccess) return fail(GPIS_ERR_DEVICE);
    (void)hipFree(need); (void)hipFree(slot_of); (void)hipFree(list); (void)hipFree(counters);
    F->enabled = 1;
    return GPIS_OK;
}

}   // namespace gpis
