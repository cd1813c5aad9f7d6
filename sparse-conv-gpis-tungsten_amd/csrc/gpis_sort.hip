// gpis_sort.hip — stable LSD radix sort of (uint32 key, uint32 value) pairs, written for this library (until round 3 this unit
// wrapped hipcub::DeviceRadixSort).  The wavefront drivers use it to regroup the segments of a bounce so that the 64 lanes of a
// wave are neighbours in lattice space (keys = Morton codes of lattice cells, values = ray indices).
//
// Four passes of 8 bits.  A pass is three launches:
//   k_sort_count   one wave per tile of 2048 keys: digit histogram of the tile (LDS atomics) -> count[digit][tile]
//   k_sort_scan    exclusive prefix sum over count[] in (digit, tile) order = where each tile's run of each digit starts
//   k_sort_scatter one wave per tile: ranks every key among the keys of the same digit in the tile IN INPUT ORDER (wave-synchronous
//                  multi-split: 8 ballots give each lane the mask of its peers), reorders the tile through LDS so that equal
//                  digits are adjacent, and writes the runs out — consecutive lanes write consecutive addresses within a run.
// Traffic per pass: 4 n (count) + 8 n + 8 n bytes; 4 passes = 80 n bytes.  Buffers ping-pong in -> temp -> out -> temp -> out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace gpis {

namespace {

constexpr int kSortWave = 64;
constexpr int kSortItems = 32;                         // keys per lane
constexpr int kSortTile = kSortWave * kSortItems;      // 2048 keys per tile (one wave)
constexpr int kRadix = 256;

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift) { return (key >> shift) & 0xFFu; }

__global__ void __launch_bounds__(kSortWave) k_sort_count(const uint32_t *__restrict__ keys, size_t n, int shift, uint32_t n_tiles, uint32_t *__restrict__ count)
{
    __shared__ uint32_t h[kRadix];
    const int lane = (int)threadIdx.x;
    const size_t tile = blockIdx.x;
    for (int d = lane; d < kRadix; d += kSortWave) h[d] = 0u;
    __syncthreads();
    const size_t base = tile * (size_t)kSortTile;
    _Pragma("unroll") for (int r = 0; r < kSortItems; ++r) {
        const size_t i = base + (size_t)r * kSortWave + (size_t)lane;
        if (i < n) atomicAdd(&h[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    for (int d = lane; d < kRadix; d += kSortWave) count[(size_t)d * n_tiles + tile] = h[d];
}

// Exclusive scan of `m` counters in three steps (m = 256 * tiles, up to 2^24): per-block sums, a one-block scan of those, the
// per-block scan with its offset.
constexpr int kScanBlock = 256;
constexpr int kScanPer = 16;                           // counters per thread
constexpr int kScanChunk = kScanBlock * kScanPer;      // 4096 counters per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total)
{
    // 256 threads: wave scans by shuffles, the four wave totals through LDS
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    uint32_t off = 0u, tot = 0u;
    for (int w = 0; w < kScanBlock / 64; ++w) {
        const uint32_t s = lds[w];
        if (w < wave) off += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return off + x - v;
}

__global__ void __launch_bounds__(kScanBlock) k_sort_scan_sums(const uint32_t *__restrict__ count, size_t m, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t lds[8];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
    uint32_t s = 0u;
    for (int k = 0; k < kScanPer; ++k) { const size_t i = base + k; if (i < m) s += count[i]; }
    uint32_t total;
    (void)block_exclusive_scan(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kScanBlock) k_sort_scan_top(uint32_t *__restrict__ sums, uint32_t n_sums)
{
    // one block walks the per-block sums in chunks of 256 (n_sums <= 4096 for 2^24 counters)
    __shared__ uint32_t lds[8];
    uint32_t carry = 0u;
    for (uint32_t c = 0; c < n_sums; c += kScanBlock) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < n_sums ? sums[i] : 0u;
        uint32_t total;
        const uint32_t e = block_exclusive_scan(v, lds, total);
        if (i < n_sums) sums[i] = carry + e;
        carry += total;
    }
}
__global__ void __launch_bounds__(kScanBlock) k_sort_scan_apply(uint32_t *__restrict__ count, size_t m, const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t lds[8];
    const size_t base = (size_t)blockIdx.x * kScanChunk + (size_t)threadIdx.x * kScanPer;
    uint32_t v[kScanPer], s = 0u;
    for (int k = 0; k < kScanPer; ++k) { const size_t i = base + k; v[k] = i < m ? count[i] : 0u; s += v[k]; }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_exclusive_scan(s, lds, total);
    for (int k = 0; k < kScanPer; ++k) { const size_t i = base + k; if (i < m) count[i] = run; run += v[k]; }
}

__global__ void __launch_bounds__(kSortWave) k_sort_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, size_t n, int shift,
                                                            uint32_t n_tiles, const uint32_t *__restrict__ start)
{
    __shared__ uint32_t run[kRadix];                   // keys of this digit placed so far (in input order), then the digit's first slot in the tile
    __shared__ uint32_t first[kRadix];                 // the digit's first slot in the reordered tile
    __shared__ uint32_t skey[kSortTile], sval[kSortTile];
    const int lane = (int)threadIdx.x;
    const size_t tile = blockIdx.x;
    const size_t base = tile * (size_t)kSortTile;
    const uint32_t n_here = (uint32_t)((n - base) < (size_t)kSortTile ? (n - base) : (size_t)kSortTile);
    for (int d = lane; d < kRadix; d += kSortWave) run[d] = 0u;
    __syncthreads();
    // (1) rank of every key among the keys of the same digit in the tile, in input order
    uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    _Pragma("unroll") for (int r = 0; r < kSortItems; ++r) {
        const uint32_t j = (uint32_t)r * kSortWave + (uint32_t)lane;
        const bool live = j < n_here;
        key[r] = live ? keys_in[base + j] : 0xFFFFFFFFu;
        val[r] = live ? vals_in[base + j] : 0u;
        const uint32_t d = live ? digit_of(key[r], shift) : 0u;
        unsigned long long peers = __ballot(live);
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = live ? run[d] : 0u;
        rank[r] = before + (uint32_t)__popcll(peers & lt);
        __syncthreads();                               // every peer has read run[d] ...
        if (live && (peers & lt) == 0ULL) run[d] = before + (uint32_t)__popcll(peers);     // ... before the first peer advances it
        __syncthreads();
    }
    // (2) first slot of each digit in the reordered tile: exclusive scan of the tile's digit counts (256 counters, 4 per lane)
    {
        uint32_t c[4], s = 0u;
        for (int k = 0; k < 4; ++k) { c[k] = run[lane * 4 + k]; s += c[k]; }
        uint32_t x = s;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64);
            if (lane >= o) x += y;
        }
        uint32_t e = x - s;
        for (int k = 0; k < 4; ++k) { first[lane * 4 + k] = e; e += c[k]; }
    }
    __syncthreads();
    // (3) reorder through LDS
    _Pragma("unroll") for (int r = 0; r < kSortItems; ++r) {
        const uint32_t j = (uint32_t)r * kSortWave + (uint32_t)lane;
        if (j < n_here) {
            const uint32_t p = first[digit_of(key[r], shift)] + rank[r];
            skey[p] = key[r];
            sval[p] = val[r];
        }
    }
    __syncthreads();
    // (4) write the runs: slot p of the tile holds the (p - first[d])-th key of digit d of this tile
    _Pragma("unroll") for (int r = 0; r < kSortItems; ++r) {
        const uint32_t p = (uint32_t)r * kSortWave + (uint32_t)lane;
        if (p < n_here) {
            const uint32_t k = skey[p], d = digit_of(k, shift);
            const size_t dst = (size_t)start[(size_t)d * n_tiles + tile] + (size_t)(p - first[d]);
            keys_out[dst] = k;
            vals_out[dst] = sval[p];
        }
    }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

}   // namespace

// Two-call convention: temp == nullptr only reports the scratch size.  keys_in / vals_in are not modified; the sorted pairs end
// in keys_out / vals_out.  Stable.
hipError_t sort_pairs_u32(void *temp, size_t &temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *vals_in, uint32_t *vals_out, size_t n, hipStream_t stream)
{
    if (n > (size_t)0x7FFFFFFF)
        return hipErrorInvalidValue;
    const size_t n_tiles = (n + kSortTile - 1) / kSortTile;
    const size_t m = (size_t)kRadix * n_tiles;                     // counters per pass
    const size_t n_sums = (m + kScanChunk - 1) / kScanChunk;
    const size_t b_pairs = align256(n * sizeof(uint32_t));
    const size_t need = 2 * b_pairs + align256(m * sizeof(uint32_t)) + align256(n_sums * sizeof(uint32_t)) + 256;
    if (temp == nullptr) {
        temp_bytes = need;
        return hipSuccess;
    }
    if (temp_bytes < need)
        return hipErrorInvalidValue;
    if (n == 0)
        return hipSuccess;
    char *t = (char *)(((uintptr_t)temp + 255) & ~(uintptr_t)255);
    uint32_t *tk = (uint32_t *)t, *tv = (uint32_t *)(t + b_pairs);
    uint32_t *count = (uint32_t *)(t + 2 * b_pairs);
    uint32_t *sums = (uint32_t *)(t + 2 * b_pairs + align256(m * sizeof(uint32_t)));
    const uint32_t *src_k = keys_in, *src_v = vals_in;
    for (int pass = 0; pass < 4; ++pass) {
        uint32_t *dst_k = (pass & 1) ? keys_out : tk, *dst_v = (pass & 1) ? vals_out : tv;      // in -> temp -> out -> temp -> out
        const int shift = 8 * pass;
        k_sort_count<<<(unsigned)n_tiles, kSortWave, 0, stream>>>(src_k, n, shift, (uint32_t)n_tiles, count);
        k_sort_scan_sums<<<(unsigned)n_sums, kScanBlock, 0, stream>>>(count, m, sums);
        k_sort_scan_top<<<1, kScanBlock, 0, stream>>>(sums, (uint32_t)n_sums);
        k_sort_scan_apply<<<(unsigned)n_sums, kScanBlock, 0, stream>>>(count, m, sums);
        k_sort_scatter<<<(unsigned)n_tiles, kSortWave, 0, stream>>>(src_k, src_v, dst_k, dst_v, n, shift, (uint32_t)n_tiles, count);
        src_k = dst_k; src_v = dst_v;
    }
    return hipGetLastError();
}

}   // namespace gpis
