// lt_logtally.hip -- reduction of the walk's deposit log into the voxel grid.
//
// Why: with per-deposit global atomics the walk runs at the memory-side atomic unit's request rate
// (~16e9 requests/s) while its arithmetic alone would sustain 4x that.  MI355X has 288 GB of
// HBM at ~6 TB/s of streaming bandwidth, so the deposits are instead WRITTEN as a coalesced log
// (lt_kernels.hip: emit_deposit) and reduced here with streaming passes only:
//   (walk)       records per grid tile, LDS histogram (tile = 32 x 32 x 16 voxels = one LDS-sized block of the grid)
//   k_log_scan   exclusive prefix -> where every tile's records will live; level-1 / level-2 cursors
//   k_log_part   LSD-free two-level radix partition by tile id (pass 1: high digit, pass 2: low digit);
//                a workgroup sorts 4096 records by digit in LDS and writes each digit's run contiguously
//   k_log_reduce one workgroup per tile: LDS adds of all its records, then ONE plain read-add-write per
//                touched voxel (the tile has exactly one owner, so no global atomics at all)
// Integer (u64 fixed-point) tallies stay bit-identical to the atomic path; float tallies differ by summation
// order only, as they already do between two atomic runs.
#include <hip/hip_runtime.h>

#include "lt_internal.hpp"

namespace ltk {

namespace {

template <typename TV> __device__ __forceinline__ void lds_add(TV* p, TV v) { atomicAdd(p, v); }

// type the LDS tile accumulates in: f32 deposits are summed in f64 (ds_add_f64 sustains twice the rate of
// ds_add_f32 on the hot voxels here, and the f32 grid then sees one rounding per tile instead of one per deposit)
template <typename TV> struct AccT { typedef TV type; };
template <> struct AccT<float> { typedef double type; };

__device__ __forceinline__ unsigned tile_of(unsigned idx) { return idx >> kTileShift; }

// records of one tile handled by one reduce workgroup; a tile with more records is split over several workgroups,
// which bounds both the load imbalance and the same-address serialisation of the LDS adds on hot voxels
constexpr uint32_t kReduceSlice = 131072;

// Partition work item = kPartThreads lanes x kPerThread records held in registers.  16 records per lane (one log
// chunk per item, 114 VGPRs) gives the longest per-tile runs; 8 per lane (64 VGPRs) lets a partition workgroup sit
// beside three walk waves per SIMD when two jobs are in flight on the device (bench.py --inflight 2).
#ifndef LT_PART_PER_THREAD
#define LT_PART_PER_THREAD 16
#endif
constexpr int kPartThreads = 512;
constexpr int kPerThread = LT_PART_PER_THREAD;
constexpr uint32_t kPartItem = kPartThreads * kPerThread;       // records per work item
constexpr uint32_t kItemsPerChunk = kLogChunk / kPartItem;
static_assert(kLogChunk % kPartItem == 0, "a log chunk must be a whole number of partition items");

// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_log_hist(const uint32_t* __restrict__ log_idx, const uint32_t* __restrict__ fill,
                                                  uint32_t n_chunks, uint32_t* __restrict__ hist, uint32_t n_tiles)
{
    extern __shared__ uint32_t s_h[];
    for (uint32_t t = threadIdx.x; t < n_tiles; t += blockDim.x) s_h[t] = 0;
    __syncthreads();
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t n = fill[c];
        const uint32_t* src = log_idx + (size_t)c * kLogChunk;
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) atomicAdd(&s_h[tile_of(src[k])], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n_tiles; t += blockDim.x)
        if (s_h[t]) atomicAdd(&hist[t], s_h[t]);
}

// One workgroup of 1024 lanes; each lane owns a contiguous run of tiles (n_tiles <= 16384 -> <= 16 per lane), a
// two-level shuffle/LDS scan gives the exclusive prefixes of the record counts and of the reduce work items.
constexpr int kScanThreads = 1024;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < kScanThreads / 64 ? s_wave[lane] : 0, wi = w;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(wi, off, 64); if (lane >= off) wi += o; }
        if (lane < kScanThreads / 64) s_wave[lane] = wi - w;
        if (lane == kScanThreads / 64 - 1) *total = wi;
    }
    __syncthreads();
    const uint32_t r = s_wave[wave] + incl - v;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(kScanThreads) k_log_scan(const uint32_t* hist, uint32_t* tile_base, uint32_t* cursor1,
                                                          uint32_t* cursor2, uint32_t* items2, uint32_t* items_r,
                                                          uint32_t* totals, uint32_t n_tiles, uint32_t bits2)
{
    __shared__ uint32_t s_wave[kScanThreads / 64];
    __shared__ uint32_t s_tot[2];
    const uint32_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint32_t t0 = threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    uint32_t cnt = 0, rit = 0;
    for (uint32_t t = t0; t < t1; t++) { const uint32_t h = hist[t]; cnt += h; rit += (h + kReduceSlice - 1) / kReduceSlice; }
    uint32_t base = block_exclusive_scan(cnt, s_wave, &s_tot[0]);
    uint32_t rbase = block_exclusive_scan(rit, s_wave, &s_tot[1]);
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t h = hist[t];
        tile_base[t] = base; cursor2[t] = base; items_r[t] = rbase;
        base += h; rbase += (h + kReduceSlice - 1) / kReduceSlice;   // hot tiles get several reduce workgroups
    }
    __syncthreads();
    if (threadIdx.x == 0) { tile_base[n_tiles] = s_tot[0]; items_r[n_tiles] = s_tot[1]; totals[0] = s_tot[0]; totals[2] = s_tot[1]; }
    __threadfence();
    __syncthreads();
    // level-1 bins: few (<= 128 for 16384 tiles); tile_base of this block's own writes is visible after the fence
    const uint32_t nb1 = (n_tiles + (1u << bits2) - 1) >> bits2;
    uint32_t it = 0;
    for (uint32_t b = threadIdx.x; b < nb1; b += kScanThreads) {
        const uint32_t a0 = b << bits2, a1 = ((b + 1) << bits2) < n_tiles ? ((b + 1) << bits2) : n_tiles;
        const uint32_t lo = tile_base[a0], hi = a1 < n_tiles ? tile_base[a1] : s_tot[0];
        cursor1[b] = lo;
        it = (hi - lo + kPartItem - 1) / kPartItem;
        items2[b] = it;   // counts for now; prefixed below
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {   // nb1 <= 1024 when bits2 = 0 (single pass: pass-2 items unused), else <= 128: serial is fine
        uint32_t run = 0;
        for (uint32_t b = 0; b < nb1; b++) { const uint32_t c = items2[b]; items2[b] = run; run += c; }
        items2[nb1] = run; totals[1] = run;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Partition work item: <= 4096 records.  Registers hold the item (16 records per lane), LDS holds the histogram
// and the digit-sorted copy; each digit's run leaves as one contiguous, coalesced write.
constexpr int kMaxBins = 1024;

template <typename TV, int PASS>
__global__ void __launch_bounds__(kPartThreads) k_log_part(LogReduceParams L)
{
    __shared__ uint32_t s_hist[kMaxBins], s_off[kMaxBins + 1], s_gbase[kMaxBins];
    __shared__ uint32_t s_range[3], s_next;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];   // digit-sorted copy of the item (96 KiB at 12 B)
    TV* s_val = reinterpret_cast<TV*>(s_dyn);
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_dyn + (size_t)kPartItem * sizeof(TV));

    // Persistent workgroups pull work items from a device counter; the item count lives in device memory too
    // (chunks the walk claimed / items the scan derived), so the host never has to read anything back.
    uint32_t n_items = PASS == 1 ? L.chunks_used[0] : L.totals[1];
    if (PASS == 1) n_items = (n_items > L.cap_chunks ? L.cap_chunks : n_items) * kItemsPerChunk;
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(&L.work[PASS - 1], 1u);
    __syncthreads();
    const uint32_t item = s_next;
    if (item >= n_items) break;
    const uint32_t nb1 = (L.n_tiles + (1u << L.bits2) - 1) >> L.bits2;
    const uint32_t mask2 = (1u << L.bits2) - 1;
    const bool final_pass = PASS == 2 || L.bits2 == 0;
    const uint32_t* in_idx; const TV* in_val; uint32_t* out_idx; TV* out_val; uint32_t* cursor; uint32_t nb;
    uint32_t lo, n;
    if (PASS == 1) {
        in_idx = L.log_idx; in_val = reinterpret_cast<const TV*>(L.log_val);
        out_idx = L.tmp_idx; out_val = reinterpret_cast<TV*>(L.tmp_val);
        const uint32_t fill = L.log_fill[item / kItemsPerChunk], part = (item % kItemsPerChunk) * kPartItem;
        lo = item * kPartItem; n = fill > part ? (fill - part < kPartItem ? fill - part : kPartItem) : 0;
        cursor = L.cursor1; nb = nb1;
    } else {
        in_idx = L.tmp_idx; in_val = reinterpret_cast<const TV*>(L.tmp_val);
        out_idx = const_cast<uint32_t*>(L.log_idx); out_val = reinterpret_cast<TV*>(const_cast<void*>(L.log_val));
        if (threadIdx.x == 0) {  // which level-1 bin does this item belong to? (items2 is a prefix over bins)
            uint32_t a = 0, b = nb1;
            while (b - a > 1) { uint32_t m = (a + b) >> 1; if (L.items2[m] <= item) a = m; else b = m; }
            const uint32_t t0 = a << L.bits2, t1 = ((a + 1) << L.bits2) < L.n_tiles ? ((a + 1) << L.bits2) : L.n_tiles;
            const uint32_t rlo = L.tile_base[t0] + (item - L.items2[a]) * kPartItem, rhi = L.tile_base[t1];
            s_range[0] = rlo; s_range[1] = rhi - rlo < kPartItem ? rhi - rlo : kPartItem; s_range[2] = a;
        }
        __syncthreads();
        lo = s_range[0]; n = s_range[1];
        cursor = L.cursor2 + (s_range[2] << L.bits2); nb = 1u << L.bits2;
    }
    for (uint32_t d = threadIdx.x; d < nb; d += kPartThreads) s_hist[d] = 0;
    __syncthreads();

    uint32_t key[kPerThread], rank[kPerThread];
    TV val[kPerThread];
#pragma unroll
    for (int r = 0; r < kPerThread; r++) {
        const uint32_t k = threadIdx.x + r * kPartThreads;
        if (k < n) { key[r] = in_idx[lo + k]; val[r] = in_val[lo + k]; }
    }
#pragma unroll
    for (int r = 0; r < kPerThread; r++) {
        const uint32_t k = threadIdx.x + r * kPartThreads;
        if (k < n) {
            const uint32_t t = tile_of(key[r]);
            const uint32_t d = PASS == 1 ? (t >> L.bits2) : (t & mask2);
            rank[r] = atomicAdd(&s_hist[d], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix over <= 512 bins by one wave: 8 bins per lane + wave scan
        const uint32_t per = (nb + 63) / 64;
        uint32_t sum = 0;
        for (uint32_t q = 0; q < per; q++) { const uint32_t d = threadIdx.x * per + q; if (d < nb) sum += s_hist[d]; }
        uint32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(incl, off, 64); if ((int)threadIdx.x >= off) incl += o; }
        uint32_t run = incl - sum;
        for (uint32_t q = 0; q < per; q++) { const uint32_t d = threadIdx.x * per + q; if (d < nb) { s_off[d] = run; run += s_hist[d]; } }
    }
    for (uint32_t d = threadIdx.x; d < nb; d += kPartThreads)
        if (s_hist[d]) s_gbase[d] = atomicAdd(&cursor[d], s_hist[d]);   // one returning atomic per non-empty digit
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kPerThread; r++) {
        const uint32_t k = threadIdx.x + r * kPartThreads;
        if (k < n) {
            const uint32_t t = tile_of(key[r]);
            const uint32_t d = PASS == 1 ? (t >> L.bits2) : (t & mask2);
            const uint32_t p = s_off[d] + rank[r];
            s_key[p] = key[r]; s_val[p] = val[r];
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < n; p += kPartThreads) {
        const uint32_t kk = s_key[p];
        const uint32_t t = tile_of(kk);
        const uint32_t d = PASS == 1 ? (t >> L.bits2) : (t & mask2);
        const uint32_t dst = s_gbase[d] + (p - s_off[d]);
        // the last pass leaves a tile's records together, so only the 14-bit position inside the tile is kept: 2 bytes
        // instead of 4 written here and read by the reduce
        if (final_pass) reinterpret_cast<uint16_t*>(out_idx)[dst] = (uint16_t)(kk & (kTileSize - 1));
        else out_idx[dst] = kk;
        out_val[dst] = s_val[p];
    }
    __syncthreads();   // LDS is reused by the next item
  }
}

// ---------------------------------------------------------------------------------------------------------
constexpr int kReduceThreads = 512;

template <typename TV>
__global__ void __launch_bounds__(kReduceThreads) k_log_reduce(LogReduceParams L)
{
    typedef typename AccT<TV>::type AT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    AT* s_tile = reinterpret_cast<AT*>(s_raw);
    __shared__ uint32_t s_item[3], s_next;
    const uint32_t n_items = L.totals[2];
  for (;;) {
    if (threadIdx.x == 0) s_next = atomicAdd(&L.work[2], 1u);
    __syncthreads();
    const uint32_t item = s_next;
    if (item >= n_items) break;
    if (threadIdx.x == 0) {   // tile of this work item: items_r is a prefix over tiles
        uint32_t a = 0, b = L.n_tiles;
        while (b - a > 1) { uint32_t m = (a + b) >> 1; if (L.items_r[m] <= item) a = m; else b = m; }
        const uint32_t lo = L.tile_base[a] + (item - L.items_r[a]) * kReduceSlice, end = L.tile_base[a + 1];
        s_item[0] = a; s_item[1] = lo; s_item[2] = end - lo < kReduceSlice ? end : lo + kReduceSlice;
    }
    for (uint32_t v = threadIdx.x; v < kTileSize; v += kReduceThreads) s_tile[v] = 0;
    __syncthreads();
    const uint32_t t = s_item[0], lo = s_item[1], hi = s_item[2];
    const bool shared_tile = L.items_r[t + 1] - L.items_r[t] > 1;
    const uint16_t* idx = reinterpret_cast<const uint16_t*>(L.log_idx);   // in-tile positions, see k_log_part's last pass
    const TV* val = reinterpret_cast<const TV*>(L.log_val);
    // four independent records in flight per lane (the tile pins the workgroup at 8 waves per CU, so memory-level
    // parallelism has to come from the instruction stream)
    uint32_t k = lo + threadIdx.x;
    for (; k + 3 * kReduceThreads < hi; k += 4 * kReduceThreads) {
        const uint32_t i0 = idx[k], i1 = idx[k + kReduceThreads], i2 = idx[k + 2 * kReduceThreads], i3 = idx[k + 3 * kReduceThreads];
        const TV v0 = val[k], v1 = val[k + kReduceThreads], v2 = val[k + 2 * kReduceThreads], v3 = val[k + 3 * kReduceThreads];
        lds_add(&s_tile[i0 & (kTileSize - 1)], (AT)v0); lds_add(&s_tile[i1 & (kTileSize - 1)], (AT)v1);
        lds_add(&s_tile[i2 & (kTileSize - 1)], (AT)v2); lds_add(&s_tile[i3 & (kTileSize - 1)], (AT)v3);
    }
    for (; k < hi; k += kReduceThreads) lds_add(&s_tile[idx[k] & (kTileSize - 1)], (AT)val[k]);
    __syncthreads();
    TV* grid = reinterpret_cast<TV*>(L.grid);
    const uint32_t tx = t % L.ntx, ty = (t / L.ntx) % L.nty, tz = t / (L.ntx * L.nty);
    for (uint32_t v = threadIdx.x; v < kTileSize; v += kReduceThreads) {
        const TV a = (TV)s_tile[v];
        const uint32_t vx = (tx << kTileBX) | (v & 31u), vy = (ty << kTileBY) | ((v >> 5) & 31u), vz = (tz << kTileBZ) | (v >> 10);
        if (a != 0 && vx < L.nx && vy < L.ny && vz < L.nz) {
            TV* dst = &grid[((size_t)vz * L.ny + vy) * L.nx + vx];
            if (shared_tile) __hip_atomic_fetch_add(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst += a;   // exclusive owner of this tile: plain read-add-write
        }
    }
    __syncthreads();   // the LDS tile is re-zeroed for the next item
  }
}

}  // namespace

hipError_t launch_log_hist(const LogReduceParams& L, hipStream_t s)
{
    if (L.n_chunks == 0) return hipSuccess;
    const unsigned blocks = L.n_chunks < 2048 ? L.n_chunks : 2048;
    hipLaunchKernelGGL(k_log_hist, dim3(blocks), dim3(256), L.n_tiles * sizeof(uint32_t), s, L.log_idx, L.log_fill,
                       L.n_chunks, L.hist, L.n_tiles);
    return hipGetLastError();
}

hipError_t launch_log_scan(const LogReduceParams& L, hipStream_t s)
{
    hipLaunchKernelGGL(k_log_scan, dim3(1), dim3(kScanThreads), 0, s, L.hist, L.tile_base, L.cursor1, L.cursor2, L.items2, L.items_r, L.totals,
                       L.n_tiles, L.bits2);
    return hipGetLastError();
}

// persistent grids: as many workgroups as are resident at once (occupancy query x CUs)
static unsigned persistent_blocks(const void* fn, int threads, size_t lds)
{
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    return (unsigned)(per_cu * cus);
}

template <typename TV, int PASS> static hipError_t launch_part_t(const LogReduceParams& L, hipStream_t s)
{
    const size_t lds = (size_t)kPartItem * (sizeof(TV) + sizeof(uint32_t));
    const void* fn = reinterpret_cast<const void*>(&k_log_part<TV, PASS>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    static unsigned blocks = 0;   // per instantiation
    if (!blocks) blocks = persistent_blocks(fn, kPartThreads, lds);
    hipLaunchKernelGGL((k_log_part<TV, PASS>), dim3(blocks), dim3(kPartThreads), lds, s, L);
    return hipGetLastError();
}
template <int PASS> static hipError_t launch_part(const LogReduceParams& L, hipStream_t s)
{
    if (L.tally == LT_TALLY_F32) return launch_part_t<float, PASS>(L, s);
    if (L.tally == LT_TALLY_F64) return launch_part_t<double, PASS>(L, s);
    return launch_part_t<unsigned long long, PASS>(L, s);
}
hipError_t launch_log_part1(const LogReduceParams& L, hipStream_t s) { return launch_part<1>(L, s); }
hipError_t launch_log_part2(const LogReduceParams& L, hipStream_t s) { return launch_part<2>(L, s); }

template <typename TV> static hipError_t launch_reduce_t(const LogReduceParams& L, hipStream_t s)
{
    const size_t lds = (size_t)kTileSize * sizeof(typename AccT<TV>::type);
    const void* fn = reinterpret_cast<const void*>(&k_log_reduce<TV>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    static unsigned blocks = 0;
    if (!blocks) blocks = persistent_blocks(fn, kReduceThreads, lds);
    hipLaunchKernelGGL(k_log_reduce<TV>, dim3(blocks), dim3(kReduceThreads), lds, s, L);
    return hipGetLastError();
}
hipError_t launch_log_reduce(const LogReduceParams& L, hipStream_t s)
{
    if (L.tally == LT_TALLY_F32) return launch_reduce_t<float>(L, s);
    if (L.tally == LT_TALLY_F64) return launch_reduce_t<double>(L, s);
    return launch_reduce_t<unsigned long long>(L, s);
}

}  // namespace ltk
