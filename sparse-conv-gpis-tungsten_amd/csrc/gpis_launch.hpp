// gpis_launch.hpp — host-side launch interface between the C-ABI translation unit (gpis_hip.hip) and the kernel
// translation units (tu_*.hip).  The library is built from several .hip files so that the big march kernels compile
// in parallel (one TU took 7 minutes); each kernel is defined in exactly one TU and reached through a plain function
// declared here (device pointers, PODs, a stream — no templates cross a TU boundary).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "gpis.h"

namespace gpis {
struct DevModel;
struct Counters;
struct FastTable;
struct GuideField;
struct PersistArgs;

// radix sort of (key, value) pairs (gpis_sort.hip); two-call convention: temp == nullptr only reports the scratch size
hipError_t sort_pairs_u32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                          size_t n, hipStream_t stream);

namespace launch {

// path instance a medium's flags select (gpis_device.hpp: spec_1d / spec_3d / spec_3d_multires / generic)
enum Inst : int { INST_1D = 0, INST_3D = 1, INST_3D_MULTIRES = 2, INST_GENERIC = 3 };

// ---- lane-per-ray march (tu_lane_spec.hip, tu_generic_sd.hip, tu_generic_tr.hip) --------------------------------------------
void lane_sample_distance(int inst, const DevModel *d_model, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                          const uint8_t *mask, Counters *cnt, hipStream_t s);
void lane_transmittance(int inst, const DevModel *d_model, size_t n, const gpis_ray_in *rays, uint8_t *visible, const uint8_t *mask,
                        Counters *cnt, hipStream_t s);
// ---- persistent refilling march (tu_persist_a.hip: 1D, 3D; tu_persist_b.hip: multi-resolution, generic) --------
int persist_blocks_per_cu(int inst, bool want_sample);        // resident one-wave workgroups per CU (occupancy query), <= 0 on failure
void persist_march(int inst, bool want_sample, unsigned grid, const DevModel *d_model, const PersistArgs &a, hipStream_t s);
// ---- single-query entries, all-features instance (tu_generic_eval.hip, tu_generic_cond.hip, tu_generic_nee.hip) ---------------------------------------
void eval_value(const DevModel *d_model, size_t n, const gpis_query *q, float *value, int32_t *gp_id, Counters *cnt, hipStream_t s);
void eval_gradient(const DevModel *d_model, size_t n, const gpis_query *q, float *grad3, Counters *cnt, hipStream_t s);
void conditioning(const DevModel *d_model, size_t n, const gpis_query *q, const float *tv, const float *tg, gpis_cond_coeff *co, Counters *cnt, hipStream_t s);
void nee(int inst, const DevModel *d_model, size_t n, const gpis_nee_query *q, float *pdf, float *grad3, Counters *cnt, const uint8_t *mask, hipStream_t s);   // inst: INST_1D or INST_GENERIC
// ---- wave-cooperative march for single-realization media (tu_fast.hip) ---------------------------------------
int fast_table_build(const DevModel &M, FastTable *t);
void fast_sample_distance(const DevModel *d_model, const FastTable &T, size_t n, const gpis_ray_in *rays, gpis_seg_out *out, gpis_cond_coeff *coeff,
                          const uint8_t *mask, Counters *cnt, hipStream_t s);
void fast_transmittance(const DevModel *d_model, const FastTable &T, size_t n, const gpis_ray_in *rays, uint8_t *visible, const uint8_t *mask,
                        Counters *cnt, hipStream_t s);
int fast_stats_read(unsigned long long *out32);                 // GPIS_FAST_STATS builds only; GPIS_ERR_UNSUPPORTED otherwise
// ---- certified guide field (tu_guide_build.hip: build + checks; tu_guided_sd.hip / tu_guided_tr.hip: the resident guided march) -----
int guide_build(const DevModel &M, const DevModel *d_model, const FastTable &T, int half, int ppc, GuideField *F, bool sparse);
void guide_selfcheck(const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const float *points3, unsigned long long *stats,
                     float *max_ratio, float *sum_bound, hipStream_t s);
void guide_raycheck(const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays, uint32_t steps,
                    unsigned long long *stats, hipStream_t s);
void guided_sample_distance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                            gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s);
void guided_sample_distance_nograd(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                                   gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s);
void range_grad(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays, gpis_seg_out *out,
                gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, hipStream_t s);      // completes the records a *_nograd / range march left pending
void guided_transmittance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField *d_guide, size_t n, const gpis_ray_in *rays,
                          uint8_t *visible, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s);
// ---- guided march with in-wave refill from a range of the batch (tu_range.hip; off by default) ------------------
void range_sample_distance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays,
                           gpis_seg_out *out, gpis_cond_coeff *coeff, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt,
                           uint32_t range_len, hipStream_t s);
void range_transmittance(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n, const gpis_ray_in *rays,
                         uint8_t *visible, const uint8_t *mask, Counters *cnt, unsigned long long *guide_cnt, uint32_t range_len, hipStream_t s);
// ---- wavefront form of the guided march (tu_wave.hip) -------------------------------------------------------------
struct WaveBufs {                      // per-call workspace carved by the caller
    void *state;                       // n x wave_state_bytes()
    uint32_t *k0, *v0, *k1, *v1;       // request keys / ray indices, unsorted and sorted
    unsigned long long *d_req;         // request counter
};
size_t wave_state_bytes();
void wave_step(bool want_sample, const DevModel *d_model, const GuideField &F, size_t n_active, const uint32_t *active, int init, const gpis_ray_in *rays,
               const uint8_t *mask, const WaveBufs &b, unsigned long long *guide_cnt, hipStream_t s);
void wave_eval(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_req, const uint32_t *sorted, const gpis_ray_in *rays,
               const WaveBufs &b, Counters *cnt, hipStream_t s);
void wave_tail(bool want_sample, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_active, const uint32_t *active,
               const gpis_ray_in *rays, const WaveBufs &b, Counters *cnt, unsigned long long *guide_cnt, hipStream_t s);
void wave_grad_keys(const DevModel *d_model, const GuideField &F, size_t n, const gpis_ray_in *rays, const WaveBufs &b, hipStream_t s);
void wave_grad(bool small_arg, const DevModel *d_model, const FastTable &T, const GuideField &F, size_t n_req, const uint32_t *sorted, const gpis_ray_in *rays,
               const WaveBufs &b, Counters *cnt, hipStream_t s);
void wave_finish_sd(const DevModel *d_model, size_t n, const gpis_ray_in *rays, const uint8_t *mask, const WaveBufs &b, gpis_seg_out *out,
                    gpis_cond_coeff *coeff, Counters *cnt, hipStream_t s);
void wave_finish_tr(size_t n, const uint8_t *mask, const WaveBufs &b, uint8_t *visible, Counters *cnt, hipStream_t s);
// ---- function-space comparison path (tu_fs.hip) -------------------------------------------------------------------
size_t fs_workspace_bytes_per_block();
void fs_march(bool want_sample, unsigned grid, const DevModel *d_model, size_t n, const gpis_ray_in *rays, gpis_fs_state *states, gpis_seg_out *out,
              uint8_t *visible, void *workspace, hipStream_t s);
int fs_prof_read(unsigned long long *out16, int reset);         // GPIS_FS_PROF builds only
void libm_eval(int fn, size_t n, const double *x, const double *y, double *out, double *out2, hipStream_t s);   // test surface (tu_libm.hip)
void fs_linalg(unsigned grid, int op, int n, size_t count, const double *in, double *out, double *evals, void *workspace, hipStream_t s);   // test surface
inline unsigned grid_of(size_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}   // namespace launch
}   // namespace gpis
