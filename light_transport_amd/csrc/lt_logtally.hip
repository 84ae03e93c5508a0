// lt_logtally.hip -- reduction of the walk's deposit log into the voxel grid.
//
// Why: with per-deposit global atomics the walk runs at the memory-side atomic unit's request rate
// (~16e9 requests/s) while its arithmetic alone would sustain 4x that.  MI355X has 288 GB of
// HBM at ~6 TB/s of streaming bandwidth, so the deposits are instead WRITTEN as a coalesced log
// (lt_kernels.hip: emit_deposit) and reduced here with streaming passes only:
//   (walk)            records per partition bin, LDS histogram (<= 4 KiB per workgroup)
//   k_log_scan_*      exclusive prefixes -> where every bin / tile's records will live, work-item tables
//   k_log_part        counting sort of 4096-record work items by tile id in LDS; every digit's run leaves as one
//                     contiguous write.  Grids of <= 1024 tiles: one pass straight to tiles.  Larger grids: pass 1 by
//                     the high digit, k_log_count2 (tile histogram of pass 1's output), pass 2 by the low digit
//   k_log_reduce      one workgroup per tile slice: LDS adds of all its records, then one read-add-write per touched
//                     voxel (a tile with a single slice has exactly one owner: no global atomics at all)
// Tile = 32 x 32 x 16 voxels = one LDS-sized block of the grid.  Integer (u64 fixed-point) tallies stay bit-identical
// to the atomic path; float tallies differ by summation order only, as they already do between two atomic runs.
//
// Memory access shape: every streaming load is 16 bytes per lane (a lane owns 4 consecutive record indices and the
// 4 values that go with them; the reduce reads 8 packed 2-byte positions and 8 values per lane and keeps two such
// groups in flight), regions start at multiples of 8 records so those loads are aligned, and two partition workgroups
// (2 x 8 waves, 64 VGPRs) share a CU so that one's LDS sort runs under the other's loads and stores.
#include <hip/hip_runtime.h>

#include <atomic>

#include "lt_internal.hpp"

namespace ltk {

namespace {

template <typename TV> __device__ __forceinline__ void lds_add(TV* p, TV v) { atomicAdd(p, v); }

// type the LDS tile accumulates in: f32 deposits are summed in f64 (ds_add_f64 sustains twice the rate of
// ds_add_f32 on the hot voxels here, and the f32 grid then sees one rounding per tile instead of one per deposit)
template <typename TV> struct AccT { typedef TV type; };
template <> struct AccT<float> { typedef double type; };

__device__ __forceinline__ unsigned tile_of(unsigned idx) { return idx >> kTileShift; }

// four consecutive values: 32 bytes (two dwordx4 loads) for the 8-byte tallies, 16 bytes for f32
template <typename TV> struct alignas(16) Quad { TV v[4]; };

// Partition work item = kPartThreads lanes x kPerThread records held in registers.  Half a log chunk per item:
// 4096 records x 12 B = 48 KiB of LDS for the digit-sorted copy + 8 KiB of tables, so two workgroups fit a CU.
constexpr int kPartThreads = 512;
constexpr int kPerThread = 8;
constexpr uint32_t kPartItem = kPartThreads * kPerThread;       // records per work item
constexpr uint32_t kItemsPerChunk = kLogChunk / kPartItem;
static_assert(kLogChunk % kPartItem == 0, "a log chunk must be a whole number of partition items");
static_assert(kPerThread % 4 == 0, "a lane loads its records four at a time");
constexpr int kMaxBins = 1024;          // digits per pass
constexpr uint32_t kMaxBits2 = 7;       // two-pass form: tiles per level-1 bin <= 128 (k_log_count2's LDS histogram)
constexpr uint32_t kCountGroup = 32;    // pass-2 items whose tiles one k_log_count2 workgroup counts before it flushes

// ---------------------------------------------------------------------------------------------------------
// scans: one workgroup of 1024 lanes, shuffle + LDS block scan
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

__device__ __forceinline__ uint32_t round_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

// Per tile: where its records will live (starts padded to kTileAlign), the cursors of the final partition pass
// (one per group; G = 1 when the histogram comes from k_log_count2), and the prefix of reduce work items (a tile
// with more than `slice` records is split over several workgroups, which bounds both the load imbalance and the
// same-address serialisation of the LDS adds on hot voxels).
__global__ void __launch_bounds__(kScanThreads) k_log_scan_tiles(const uint32_t* hist, uint32_t G, uint32_t* tile_base,
                                                                uint32_t* tile_cnt, uint32_t* cursor, uint32_t* items_r,
                                                                uint32_t* meta, unsigned long long* job, uint32_t n_tiles,
                                                                const uint16_t* dmap, const uint32_t* dmeta,
                                                                const uint32_t* bin_base, const uint32_t* bin_cnt)
{
    __shared__ uint32_t s_wave[kScanThreads / 64];
    __shared__ uint32_t s_tot[3];
    const uint32_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint32_t t0 = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    // hot-tile form: a hot tile's records already sit, final, in its own bin of pass 1's output
    const uint32_t H = dmap ? dmeta[0] : 0u;
    uint32_t cnt = 0, pad = 0;
    for (uint32_t t = t0; t < t1; t++) {
        uint32_t h = 0;
        const uint32_t d = dmap ? dmap[t] : 0u;
        if (d < H) h = bin_cnt[d];
        else { for (uint32_t g = 0; g < G; g++) h += hist[t * G + g]; pad += round_up(h, kTileAlign); }
        tile_cnt[t] = h; cnt += h;
    }
    uint32_t base = block_exclusive_scan(pad, s_wave, &s_tot[0]);
    (void)block_exclusive_scan(cnt, s_wave, &s_tot[1]);
    const uint32_t total = s_tot[1];
    // records per reduce work item: ~4096 items per launch keep 256 CUs level to a few per cent while the per-item
    // costs (zeroing and flushing a 128 KiB LDS tile) stay small against the slice's own traffic
    uint32_t slice = round_up(total / 4096u + 1u, 8192u);
    slice = slice < 65536u ? 65536u : (slice > (1u << 20) ? (1u << 20) : slice);
    uint32_t rit = 0;
    for (uint32_t t = t0; t < t1; t++) rit += (tile_cnt[t] + slice - 1) / slice;
    uint32_t rbase = block_exclusive_scan(rit, s_wave, &s_tot[2]);
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t h = tile_cnt[t];
        const uint32_t d = dmap ? dmap[t] : 0u;
        items_r[t] = rbase;
        if (d < H) tile_base[t] = bin_base[d];
        else {
            tile_base[t] = base;
            uint32_t run = base;
            for (uint32_t g = 0; g < G; g++) { cursor[t * G + g] = run; run += hist[t * G + g]; }
            base += round_up(h, kTileAlign);
        }
        rbase += (h + slice - 1) / slice;
    }
    if (threadIdx.x == 0) {
        tile_base[n_tiles] = s_tot[0]; items_r[n_tiles] = s_tot[2];
        meta[LM_RECORDS] = total; meta[LM_ITEMS_R] = s_tot[2]; meta[LM_SLICE] = slice;
        if (job) { atomicAdd(&job[0], (unsigned long long)total); atomicAdd(&job[1], (unsigned long long)meta[LM_OVERFLOW]); }
    }
}

// Two-pass form, per digit of pass 1 (the level-1 bins; in the hot-tile form H hot tiles come first): where pass 1 puts
// it (starts padded to kBinAlign), one cursor per group, the prefix of pass-2 work items and of k_log_count2 work
// items (none for a hot tile: it is final after pass 1).
__global__ void __launch_bounds__(kScanThreads) k_log_scan_bins(const uint32_t* hist1, uint32_t* bin_base, uint32_t* bin_cnt,
                                                               uint32_t* cursor1, uint32_t* items2, uint32_t* items_c,
                                                               uint32_t* meta, uint32_t nb1, const uint32_t* dmeta)
{
    __shared__ uint32_t s_wave[kScanThreads / 64];
    __shared__ uint32_t s_tot[3];
    const uint32_t H = dmeta ? dmeta[0] : 0u, nd = H + nb1;      // <= kScanThreads (k_log_plan bounds H)
    const uint32_t b = threadIdx.x;
    uint32_t h = 0;
    if (b < nd) for (uint32_t g = 0; g < kLogGroups; g++) h += hist1[b * kLogGroups + g];
    const uint32_t it = b < H ? 0u : (h + kPartItem - 1) / kPartItem, ic = (it + kCountGroup - 1) / kCountGroup;
    const uint32_t base = block_exclusive_scan(round_up(h, kBinAlign), s_wave, &s_tot[0]);
    const uint32_t ibase = block_exclusive_scan(it, s_wave, &s_tot[1]);
    const uint32_t cbase = block_exclusive_scan(ic, s_wave, &s_tot[2]);
    if (b < nd) {
        bin_base[b] = base; bin_cnt[b] = h; items2[b] = ibase; items_c[b] = cbase;
        uint32_t run = base;
        for (uint32_t g = 0; g < kLogGroups; g++) { cursor1[b * kLogGroups + g] = run; run += hist1[b * kLogGroups + g]; }
    }
    if (b == 0) {
        bin_base[nd] = s_tot[0]; items2[nd] = s_tot[1]; items_c[nd] = s_tot[2];
        meta[LM_ITEMS2] = s_tot[1]; meta[LM_ITEMS_C] = s_tot[2];
    }
}

// largest a in [0, n) with prefix[a] <= x (prefix ascending, prefix[0] = 0)
__device__ __forceinline__ uint32_t upper_slot(const uint32_t* prefix, uint32_t n, uint32_t x)
{
    uint32_t a = 0, b = n;
    while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (prefix[m] <= x) a = m; else b = m; }
    return a;
}

// ---------------------------------------------------------------------------------------------------------
// Two-pass form: descriptor of every pass-2 work unit (up to kCountGroup x 4096 consecutive records of ONE level-1
// bin in pass 1's output: where they start, how many, which pass-1 digit), so that k_log_count2 and pass 2 read one record per
// unit -- a unit ahead -- instead of searching the bin prefix with dependent loads while 511 lanes wait.
__global__ void __launch_bounds__(256) k_log_items2(LogReduceParams L)
{
    const uint32_t nd = (L.dmap ? L.dmeta[0] : 0u) + ((L.n_tiles + (1u << L.bits2) - 1) >> L.bits2);
    const uint32_t n_units = L.meta[LM_ITEMS_C];
    for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < n_units; u += gridDim.x * blockDim.x) {
        const uint32_t a = upper_slot(L.items_c, nd, u);      // digit of pass 1 (hot tiles have no units)
        const uint32_t done = (u - L.items_c[a]) * (kCountGroup * kPartItem), left = L.bin_cnt[a] - done;
        reinterpret_cast<uint4*>(L.itab)[u] =
            make_uint4(L.bin_base[a] + done, left < kCountGroup * kPartItem ? left : kCountGroup * kPartItem, a, 0u);
    }
}

// chunks of the log that may hold records: the groups claim interleaved chunk indices (group + kLogGroups * n)
__device__ __forceinline__ uint32_t claimed_chunks(const LogReduceParams& L)
{
    uint32_t n = 0;
#pragma unroll
    for (uint32_t g = 0; g < kLogGroups; g++) {
        const uint32_t c = L.meta[LM_NEXT + g];
        const uint32_t top = c ? g + kLogGroups * (c - 1) + 1 : 0;
        n = top > n ? top : n;
    }
    return n < L.cap_chunks ? n : L.cap_chunks;
}

// ---------------------------------------------------------------------------------------------------------
// Partition work item: <= 4096 records.  Registers hold the item (8 records per lane, loaded 16 bytes at a time), LDS
// holds the digit histogram / offsets and the digit-sorted copy; each digit's run leaves as one contiguous write.
// Work is strided over the grid in units of one log chunk (pass 1) / one k_log_items2 unit (pass 2): no shared work
// counter (one word sustains only ~90 returning atomics per microsecond, less than the item rate of this kernel),
// and a workgroup only ever serves ONE cursor group (see kLogGroups).
// Software pipeline: the next item's loads are issued as soon as this item's records sit in LDS (their registers are
// free then), so they fly during the write-out; the cursor atomics are issued before the digit scan and collected
// after the LDS scatter.
// Two register budgets.  CORUN (64 VGPRs, the f64 build spills 28 B per lane): two partition workgroups fit a CU beside a
// walk that runs at two workgroups per CU -- the overlapped regimes (lanes inside a launch, jobs in flight).  ALONE (80
// VGPRs, no scratch): nothing shares the CU; C2's partition 10.59 -> 9.38 ms.  Measured both ways (profiles/r03a_*.log):
// the ALONE build beside a walk costs C5's two-pass partition its second workgroup per CU -- two_jobs 61.3-63.1 -> 65.7 ms,
// one call 60.4-60.7 -> 64.5 ms per share -- so it is used only where LogReduceParams::alone says nothing co-runs.
#ifndef LT_PART_WAVES_CORUN
#define LT_PART_WAVES_CORUN 8
#endif
#ifndef LT_PART_WAVES_ALONE
#define LT_PART_WAVES_ALONE 6
#endif
template <typename TV, int PASS, bool HOT, bool ALONE>
__global__ void __launch_bounds__(kPartThreads, ALONE ? LT_PART_WAVES_ALONE : LT_PART_WAVES_CORUN) k_log_part(LogReduceParams L)
{
#if LT_RED_PRIO      // (compile option: the log reduction's waves issue ahead of a walk's -- they need 8 % of its VALU work and sit on the critical path of their job)
    __builtin_amdgcn_s_setprio(LT_RED_PRIO);
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    TV* s_val = reinterpret_cast<TV*>(s_dyn);                                                   // [kPartItem]
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_dyn + (size_t)kPartItem * sizeof(TV));      // [kPartItem]
    uint32_t* s_a = s_key + kPartItem;        // [kMaxBins + 2] digit counts, then exclusive offsets
    uint32_t* s_b = s_a + kMaxBins + 2;       // [kMaxBins]     global base of the digit's run minus its offset
    const uint16_t* s_dmap = reinterpret_cast<const uint16_t*>(s_b + kMaxBins);   // [n_tiles], HOT only
    __shared__ uint32_t s_wsum[kPartThreads / 64];

    const uint32_t nb1 = L.bits2 ? (L.n_tiles + (1u << L.bits2) - 1) >> L.bits2 : L.n_tiles;
    const uint32_t mask2 = (1u << L.bits2) - 1;
    const uint32_t H = L.dmap ? L.dmeta[0] : 0u;          // hot tiles: pass-1 digits 0 .. H-1, final after pass 1
    const uint32_t nb = PASS == 1 ? H + nb1 : (1u << L.bits2);
    const bool final_pass = PASS == 2 || L.bits2 == 0;
    if (HOT) {
        uint32_t* w = reinterpret_cast<uint32_t*>(s_b + kMaxBins);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(L.dmap);      // (the map is padded to a whole word)
        for (uint32_t i = threadIdx.x; i < (L.n_tiles + 1) / 2; i += kPartThreads) w[i] = src[i];
        __syncthreads();
    }
    // the amount of work lives in device memory (chunks the walk claimed / units the scan derived), so the host never
    // has to read anything back between the kernels of a batch
    const uint32_t n_units = PASS == 1 ? claimed_chunks(L) : L.meta[LM_ITEMS_C];
    constexpr uint32_t kSub = PASS == 1 ? kItemsPerChunk : kCountGroup;     // 4096-record items per unit
    const uint32_t* in_idx = PASS == 1 ? L.log_idx : L.tmp_idx;
    const TV* in_val = reinterpret_cast<const TV*>(PASS == 1 ? L.log_val : L.tmp_val);
    uint32_t* out_idx = PASS == 1 ? L.tmp_idx : const_cast<uint32_t*>(L.log_idx);
    TV* out_val = reinterpret_cast<TV*>(PASS == 1 ? L.tmp_val : const_cast<void*>(L.log_val));
    auto digit = [&](uint32_t k_) -> uint32_t { const uint32_t t_ = tile_of(k_); return PASS == 1 ? (HOT ? (uint32_t)s_dmap[t_] : (t_ >> L.bits2)) : (t_ & mask2); };

    // unit descriptor (start, records, level-1 bin): pass 1 -- a chunk and its fill; pass 2 -- k_log_items2's table
    auto describe = [&](uint32_t u) -> uint4 {
        if (PASS == 1) return make_uint4(u * kLogChunk, L.log_fill[u], 0u, 0u);
        return reinterpret_cast<const uint4*>(L.itab)[u];
    };
    uint32_t key[kPerThread], ret2[kPerThread / 2];      // ranks inside the digit (< 4096): two per register
    TV val[kPerThread];
    // lane owns records k = g * 2048 + tid * 4 + j: 16 B of indices and 16 / 32 B of values per load group.  Groups are
    // read whole and unconditionally: the bytes behind an item's end are mapped (slack behind every buffer), and
    // records at k >= n are masked out below.  Indices are requested an item ahead; values only after the ranking (they
    // are first needed at the LDS scatter), which keeps the kernel within 64 VGPRs: four waves per SIMD beside a walk.
    auto load_keys = [&](uint32_t lo_, uint32_t tid_) {
#pragma unroll
        for (int g = 0; g < kPerThread / 4; g++) {
            const uint4 q = *reinterpret_cast<const uint4*>(in_idx + lo_ + (uint32_t)g * (kPartThreads * 4) + tid_ * 4);
            key[4 * g] = q.x; key[4 * g + 1] = q.y; key[4 * g + 2] = q.z; key[4 * g + 3] = q.w;
        }
    };
    auto load_vals = [&](uint32_t lo_, uint32_t tid_) {
#pragma unroll
        for (int g = 0; g < kPerThread / 4; g++) {
            const Quad<TV> v = *reinterpret_cast<const Quad<TV>*>(in_val + lo_ + (uint32_t)g * (kPartThreads * 4) + tid_ * 4);
#pragma unroll
            for (int j = 0; j < 4; j++) val[4 * g + j] = v.v[j];
        }
    };

    uint32_t unit = blockIdx.x, sub = 0;
    if (unit >= n_units) return;
    uint4 ds = describe(unit);
    uint4 dsn = unit + gridDim.x < n_units ? describe(unit + gridDim.x) : make_uint4(0u, 0u, 0u, 0u);
    load_keys(ds.x, threadIdx.x);
  for (;;) {
    // The lane's id passes through an empty asm once per item: everything derived from it (its slice of the item, LDS and
    // global addresses, the scan's lane masks) is then recomputed per item -- a handful of integer instructions against 4096
    // records -- instead of being hoisted in front of the loop as loop invariants, where the 64-VGPR build had to spill them
    // (28 B of scratch per lane and a dozen wave masks parked in VGPR lanes).
    uint32_t tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
    // this item: records [ds.x + sub * 4096, ...) of the unit; the one after it
    const uint32_t off = sub * kPartItem;
    const uint32_t n = ds.y > off ? (ds.y - off < kPartItem ? ds.y - off : kPartItem) : 0;
    const bool more_here = sub + 1 < kSub && ds.y > off + kPartItem;
    const bool have_next = more_here || unit + gridDim.x < n_units;
    const uint32_t next_lo = more_here ? ds.x + off + kPartItem : dsn.x;
    uint32_t* cursor;
    uint32_t cstride;
    if (PASS == 1) { cursor = L.cursor1 + (unit & (kLogGroups - 1)); cstride = kLogGroups; }        // the cursors of the chunk's group
    else { cursor = L.cursor2 + ((size_t)((ds.z - H) << L.bits2) * kLogGroups2 + (unit & (kLogGroups2 - 1))); cstride = kLogGroups2; }
    if (n != 0) {
    for (uint32_t d = tid; d < nb + 2; d += kPartThreads) s_a[d] = 0;
    __syncthreads();
    // ---- rank inside the digit: one returning LDS atomic per record.  (Counting the lanes that share the first
    //      lane's digit with a ballot and adding once for all of them removes the same-address serialisation of the
    //      hot digits but costs ~150 instructions per record slot: measured slower, alone and beside a walk.)
#pragma unroll
    for (int r = 0; r < kPerThread; r++) {
        const uint32_t k = (uint32_t)(r >> 2) * (kPartThreads * 4) + tid * 4 + (r & 3);
        const uint32_t rk = k < n ? atomicAdd(&s_a[digit(key[r])], 1u) : 0u;
        if (r & 1) ret2[r >> 1] |= rk << 16; else ret2[r >> 1] = rk;
    }
    load_vals(ds.x + off, tid);     // in flight during the scan
    __syncthreads();
    // ---- space for every non-empty digit: one returning global atomic each, issued before the scan so that its
    //      latency runs under the scan and the LDS scatter
    uint32_t c0 = 0, c1 = 0, g0 = 0, g1 = 0, incl;
    {
        const uint32_t d = 2 * tid;
        if (d < nb) { const uint2 cc = *reinterpret_cast<const uint2*>(&s_a[d]); c0 = cc.x; c1 = d + 1 < nb ? cc.y : 0; }
        if (c0) g0 = atomicAdd(&cursor[d * cstride], c0);
        if (c1) g1 = atomicAdd(&cursor[(d + 1) * cstride], c1);
        // ---- exclusive prefix over the digits, in place: two digits per lane, wave scan, wave totals through LDS
        incl = c0 + c1;
#pragma unroll
        for (int off2 = 1; off2 < 64; off2 <<= 1) { const uint32_t o = __shfl_up(incl, off2, 64); if (lane >= off2) incl += o; }
        if (lane == 63) s_wsum[wave] = incl;
    }
    __syncthreads();
    uint32_t ex = 0;
    {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < kPartThreads / 64; w++) before += w < wave ? s_wsum[w] : 0;
        ex = before + incl - (c0 + c1);
        const uint32_t d = 2 * tid;
        if (d < nb) *reinterpret_cast<uint2*>(&s_a[d]) = make_uint2(ex, ex + c0);
    }
    __syncthreads();
    // ---- digit-sorted copy in LDS
#pragma unroll
    for (int r = 0; r < kPerThread; r++) {
        const uint32_t k = (uint32_t)(r >> 2) * (kPartThreads * 4) + tid * 4 + (r & 3);
        if (k < n) {
            const uint32_t p = s_a[digit(key[r])] + ((r & 1) ? ret2[r >> 1] >> 16 : (ret2[r >> 1] & 0xffffu));
            s_key[p] = key[r]; s_val[p] = val[r];
        }
    }
    // ---- the registers are free: request the next item now, it arrives during the write-out
    if (have_next) load_keys(next_lo, tid);
    {
        const uint32_t d = 2 * tid;
        if (c0) s_b[d] = g0 - ex;
        if (c1) s_b[d + 1] = g1 - (ex + c0);
    }
    __syncthreads();
    // ---- write out: consecutive sorted positions of one digit are consecutive in memory
#pragma unroll
    for (int i = 0; i < kPerThread; i++) {
        const uint32_t p = tid + (uint32_t)i * kPartThreads;
        if (p < n) {
            const uint32_t kk = s_key[p];
            const uint32_t dg = digit(kk);
#if defined(LT_PART_DEBUG)          /* timing experiments only (results are wrong): the sorted item goes back where it came from */
            const uint32_t dst = ds.x + off + p;
#else
            const uint32_t dst = s_b[dg] + p;
#endif
            // the last pass leaves a tile's records together, so only the 14-bit position inside the tile is kept:
            // 2 bytes instead of 4 written here and read by the reduce.  (Hot tiles come first in pass 1's output, so
            // their 2-byte positions [0, 2 D) never meet the 4-byte indices of the bins behind them [4 D, ...).)
            if (final_pass || (HOT && dg < H)) reinterpret_cast<uint16_t*>(out_idx)[dst] = (uint16_t)(kk & (kTileSize - 1));
            else out_idx[dst] = kk;
            out_val[dst] = s_val[p];
        }
    }
    __syncthreads();   // LDS is reused by the next item
    } else if (have_next) load_keys(next_lo, tid);     // an empty item (a chunk nobody claimed, a short last chunk)
    if (!have_next) break;
    if (more_here) sub++;
    else {
        unit += gridDim.x; sub = 0; ds = dsn;
        if (unit + gridDim.x < n_units) dsn = describe(unit + gridDim.x);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same counting sort for ONE workgroup per CU (one-pass grids; lt_set_tuning "part_lds").  k_log_part keeps an item in
// registers and needs two workgroups per CU (2 x 8 waves, 2 x 128 VGPRs per SIMD lane) so that one's loads fly under the other's
// LDS work -- at one workgroup per CU it is latency-bound (10 -> 25-34 ms), which is all the register quarter beside a walk
// train offers it.  Here the NEXT item arrives by LDS-DMA (global_load_lds_dwordx4: no registers, the image is lane-linear,
// i.e. a plain copy of the item) while this one is ranked, scanned and scattered: its 32 KiB of values into a second staging
// buffer from the top of the item, its 16 KiB of indices into the ONE index buffer as soon as every wave holds this item's
// indices in registers (after the ranking).  LDS: 16 + 2 x 32 KiB staging + 48 KiB sorted copy + 4 KiB of tables = 136 KiB,
// which leaves three slab walk workgroups their 4.6 KiB each.  One vmcnt wait per item: where the cursor atomics' results are
// needed -- the DMA was issued before them and vmcnt counts in order, so it has landed too; the item's own stores drain under
// the next item's ranking.  Barriers are raw s_barrier with an lgkmcnt-only wait.  Measured (C2, 1.81e9 records): 10.5 ms
// against k_log_part's 10.1-10.4 on an idle device (both ~3.8 TB/s: the bound is the scattered write-out, not latency -- 1024
// lanes per workgroup changed nothing, and the per-phase clocks of tools/part_phases.py are flat).  THREADS = 256: one wave per
// SIMD, items of 2048 records (8 per lane as at 512), 68 KiB -- the build that fits beside FOUR walk waves of <= 112 VGPRs
// (14.3 ms alone).  No jobs-in-flight regime gains from either (DESIGN "Overlap": a walk loses about what the reduction beside
// it takes, whatever registers it leaves) -- hence a knob, not the default.
#ifdef LT_PART_PROF      // measurement builds only (tools/build_variant.sh): clock sums per phase of k_log_part_lds, first wave of every workgroup
__device__ unsigned long long g_part_phase[8];
#define LT_PP_DECL unsigned long long pp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pp_t = __builtin_readcyclecounter()
#define LT_PP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); pp_[i] += t_ - pp_t; pp_t = t_; } while (0)
#define LT_PP_FLUSH do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_part_phase[i_], pp_[i_]); } while (0)
#else
#define LT_PP_DECL
#define LT_PP(i)
#define LT_PP_FLUSH
#endif
#define LT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
constexpr uint32_t lds_part_item(int threads) { return threads >= 512 ? kPartItem : (uint32_t)threads * 8u; }
template <typename TV, int THREADS>
__global__ void __launch_bounds__(THREADS, 8) k_log_part_lds(LogReduceParams L)
{
#if LT_RED_PRIO      // (compile option: the log reduction's waves issue ahead of a walk's -- they need 8 % of its VALU work and sit on the critical path of their job)
    __builtin_amdgcn_s_setprio(LT_RED_PRIO);
#endif
    constexpr uint32_t kItem = lds_part_item(THREADS), kItems = kLogChunk / kItem;      // records per item (8 per lane); items per log chunk
    constexpr int kPerThread = (int)kItem / THREADS;      // (shadows the register-staged kernel's: 8 at 512 lanes, 4 at 1024)
    constexpr uint32_t kPartThreads = THREADS;
    constexpr int DPL = kMaxBins / THREADS > 2 ? kMaxBins / THREADS : 2;      // digits a lane owns in the scan
    static_assert(kPerThread % 4 == 0 && DPL * THREADS >= kMaxBins, "a lane reads its records four at a time; the lanes own all digits");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t* sK = reinterpret_cast<uint32_t*>(s_dyn);                                                           // [kItem] staged indices (ONE buffer: in registers after the ranking's first read)
    TV* sV = reinterpret_cast<TV*>(s_dyn + (size_t)kItem * sizeof(uint32_t));                                 // [2][kItem] staged values
    TV* oV = sV + 2 * (size_t)kItem;                                                                          // [kItem] digit-sorted values
    uint32_t* oK = reinterpret_cast<uint32_t*>(oV + kItem);                                                   // [kItem] digit-sorted indices
    uint32_t* s_a = oK + kItem;                                                                               // [kMaxBins + 2] counts -> offsets -> global bases
    __shared__ uint32_t s_wsum[kPartThreads / 64];
    const uint32_t nb = L.n_tiles;
    const uint32_t n_units = claimed_chunks(L);
    const uint32_t* in_idx = L.log_idx;
    const TV* in_val = reinterpret_cast<const TV*>(L.log_val);
    uint32_t* out_idx = L.tmp_idx;
    TV* out_val = reinterpret_cast<TV*>(L.tmp_val);

    // the items of this workgroup, in order: (first record, records) of every non-empty 4096-record half of its chunks.  Chunk
    // fills come through the SCALAR cache (s_load: lgkmcnt): a vector load's result could only be awaited with vmcnt, and a
    // vmcnt wait at the top of an item would also wait for the previous item's stores, which nothing else has to.
    auto fill_of = [&](uint32_t u) -> uint32_t {
        if (u >= n_units) return 0u;
        uint32_t v;
        const uint32_t* p_ = L.log_fill + u;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p_) : "memory");
        return v;
    };
    uint32_t unit = blockIdx.x, sub = 0, fill = fill_of(unit);
    auto settle = [&]() -> bool {      // move (unit, sub) to the next non-empty item at or after the current position
        for (;;) {
            if (unit >= n_units) return false;
            if (sub < kItems && fill > sub * kItem) return true;
            unit += gridDim.x; sub = 0; fill = fill_of(unit);
        }
    };
    // One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS[dst + lane * 16] (M0 = dst, wave-uniform).
    // Written as asm so that the compiler's wait-count pass does not know about it: through the builtin it drains the DMA
    // (s_waitcnt vmcnt(0)) in front of the first LDS read that follows, i.e. at once.  The waits are placed by hand below;
    // the compiler's own counted waits can only become stricter by operations it does not see.
    auto glds16 = [](const void* gsrc, uint32_t lds_dst) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    };
    auto lds_addr = [](const void* p_) -> uint32_t {
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(const __attribute__((address_space(3))) void*)p_);
    };
    auto stage_keys = [&](uint32_t lo) {               // LDS-DMA of the indices of the item that starts at record lo
        const uint32_t tid_ = threadIdx.x, w = tid_ >> 6, l = tid_ & 63u;
        constexpr uint32_t nk = kItem * sizeof(uint32_t) / (THREADS * 16);      // instructions per wave
#pragma unroll
        for (uint32_t j = 0; j < nk; j++)
            glds16(in_idx + lo + (w * nk + j) * 256 + l * 4, lds_addr(sK + (w * nk + j) * 256));
    };
    auto stage_vals = [&](uint32_t lo, uint32_t buf) {      // ... and of its values, into staging buffer buf
        const uint32_t tid_ = threadIdx.x, w = tid_ >> 6, l = tid_ & 63u;
        constexpr uint32_t per = 16 / sizeof(TV), nj = kItem * sizeof(TV) / (THREADS * 16);      // values per lane per instruction; instructions per wave
#pragma unroll
        for (uint32_t j = 0; j < nj; j++)
            glds16(in_val + lo + (w * nj + j) * (64 * per) + l * per, lds_addr(sV + buf * kItem + (w * nj + j) * (64 * per)));
    };
    if (!settle()) return;
#if LT_PART_DROP == 9      // (measurement builds: no partition at all)
    return;
#endif
    uint32_t buf = 0;
    stage_keys(unit * kLogChunk + sub * kItem);
    stage_vals(unit * kLogChunk + sub * kItem, buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LT_PP_DECL;
    for (;;) {
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));       // (lane-derived addresses and masks recomputed per item, not hoisted: see k_log_part)
        const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
        const uint32_t cur_unit = unit, n = fill - sub * kItem < kItem ? fill - sub * kItem : kItem;
        sub++;
        const bool have_next = settle();
        uint32_t* cursor = L.cursor1 + (cur_unit & (kLogGroups - 1));
        const uint32_t next_lo = unit * kLogChunk + sub * kItem;       // (meaningful if have_next)
        const TV* vb = sV + buf * kItem;
        // This item's DMA has landed: the first item's was awaited in front of the loop, every other one was issued BEFORE the
        // previous item's cursor atomics, whose results were awaited there (vmcnt counts in order).  No vmcnt wait here: the
        // previous item's stores drain under this item's ranking.
        LT_LDS_BARRIER();                                     // every wave is done with the sorted copy and the other staging buffer
        LT_PP(0);
#if LT_PART_DROP != 3
        if (have_next) stage_vals(next_lo, buf ^ 1u);
#endif
        for (uint32_t d = tid; d < nb + 2; d += kPartThreads) s_a[d] = 0;
        LT_LDS_BARRIER();
        LT_PP(1);
        // ---- rank inside the digit
        uint32_t key[kPerThread], ret2[kPerThread / 2];
#pragma unroll
        for (int g = 0; g < kPerThread / 4; g++) {
            const uint4 q = *reinterpret_cast<const uint4*>(sK + (uint32_t)g * (kPartThreads * 4) + tid * 4);
            key[4 * g] = q.x; key[4 * g + 1] = q.y; key[4 * g + 2] = q.z; key[4 * g + 3] = q.w;
        }
#pragma unroll
        for (int r = 0; r < kPerThread; r++) {
            const uint32_t k = (uint32_t)(r >> 2) * (kPartThreads * 4) + tid * 4 + (r & 3);
#if LT_PART_DROP == 1      // (measurement builds: which of the partition's resources slows a walk beside it -- results are garbage)
            const uint32_t rk = 0u;
#else
            const uint32_t rk = k < n ? atomicAdd(&s_a[tile_of(key[r])], 1u) : 0u;
#endif
            if (r & 1) ret2[r >> 1] |= rk << 16; else ret2[r >> 1] = rk;
        }
        LT_LDS_BARRIER();
        LT_PP(2);
#if LT_PART_DROP != 3
        if (have_next) stage_keys(next_lo);      // every wave holds its indices in registers: the one index buffer takes the next item's
#endif
        // ---- space for every non-empty digit (one returning global atomic each), exclusive prefix over the digits
        uint32_t c[DPL], g[DPL], incl, sum = 0;
        {
            const uint32_t d0 = DPL * tid;
#pragma unroll
            for (int i = 0; i < DPL; i++) { c[i] = d0 + i < nb ? s_a[d0 + i] : 0u; g[i] = 0; }
#pragma unroll
            for (int i = 0; i < DPL; i++) { if (c[i]) g[i] = atomicAdd(&cursor[(d0 + i) * kLogGroups], c[i]); sum += c[i]; }
            incl = sum;
#pragma unroll
            for (int off2 = 1; off2 < 64; off2 <<= 1) { const uint32_t o = __shfl_up(incl, off2, 64); if (lane >= off2) incl += o; }
            if (lane == 63) s_wsum[wave] = incl;
        }
        LT_LDS_BARRIER();
        uint32_t ex = 0;
        {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < (int)(kPartThreads / 64); w++) before += w < wave ? s_wsum[w] : 0;
            ex = before + incl - sum;
            const uint32_t d0 = DPL * tid;
            uint32_t run = ex;
#pragma unroll
            for (int i = 0; i < DPL; i++) { if (d0 + i < nb) s_a[d0 + i] = run; run += c[i]; }
        }
        LT_LDS_BARRIER();
        LT_PP(3);
        // ---- digit-sorted copy in LDS (values straight from the staging buffer)
#pragma unroll
        for (int g = 0; g < kPerThread / 4; g++) {
            const Quad<TV> v = *reinterpret_cast<const Quad<TV>*>(vb + (uint32_t)g * (kPartThreads * 4) + tid * 4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int r = 4 * g + j;
                const uint32_t k = (uint32_t)g * (kPartThreads * 4) + tid * 4 + (uint32_t)j;
                if (k < n) {
                    const uint32_t p = s_a[tile_of(key[r])] + ((r & 1) ? ret2[r >> 1] >> 16 : (ret2[r >> 1] & 0xffffu));
#if LT_PART_DROP == 4
                    asm volatile("" :: "v"(p), "v"(v.v[j]));
#else
                    oK[p] = key[r]; oV[p] = v.v[j];
#endif
                }
            }
        }
        LT_PP(4);
        // (the bases are "used" on every path: were their only use under if (c0) / if (c1), the path "atomic issued, use skipped"
        //  would exist for the compiler and it would wait for them -- vmcnt(0), the DMA with them -- before the next item's atomics)
#pragma unroll
        for (int i = 0; i < DPL; i++) asm volatile("" :: "v"(g[i]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // ... and with them the next item's DMA, issued before them
        LT_LDS_BARRIER();       // every lane has read its digits' offsets: the global bases take their place
        {
            const uint32_t d0 = DPL * tid;
            uint32_t run = ex;
#pragma unroll
            for (int i = 0; i < DPL; i++) { if (c[i]) s_a[d0 + i] = g[i] - run; run += c[i]; }
        }
        LT_LDS_BARRIER();
        LT_PP(5);
        // ---- write out: consecutive sorted positions of one digit are consecutive in memory; the final pass keeps the
        //      14-bit position inside the tile only (2 bytes)
#pragma unroll
        for (int i = 0; i < kPerThread; i++) {
            const uint32_t p = tid + (uint32_t)i * kPartThreads;
            if (p < n) {
                const uint32_t kk = oK[p];
                const uint32_t dst = s_a[tile_of(kk)] + p;
#if LT_PART_DROP == 2
                asm volatile("" :: "v"(dst), "v"(oV[p]));
#else
                reinterpret_cast<uint16_t*>(out_idx)[dst] = (uint16_t)(kk & (kTileSize - 1));
                out_val[dst] = oV[p];
#endif
            }
        }
        LT_PP(6);
        if (!have_next) break;
        buf ^= 1u;
        // (the barrier at the top of the next item separates these LDS reads from the next item's writes)
    }
    LT_PP_FLUSH;
}

// ---------------------------------------------------------------------------------------------------------
// Two-pass form: records per tile, counted from pass 1's output (indices only).  A workgroup takes up to kCountGroup
// consecutive pass-2 items of ONE level-1 bin, counts their tiles in an LDS histogram of <= 128 entries and flushes it
// with one global atomic per non-empty tile.
__global__ void __launch_bounds__(kPartThreads) k_log_count2(LogReduceParams L)
{
    __shared__ uint32_t s_cnt[1u << kMaxBits2];
    const uint32_t mask2 = (1u << L.bits2) - 1;
    const uint32_t n_units = L.meta[LM_ITEMS_C];
    const uint32_t H = L.dmap ? L.dmeta[0] : 0u;
    const int lane = threadIdx.x & 63;
  for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    const uint4 ds = reinterpret_cast<const uint4*>(L.itab)[unit];
    if (threadIdx.x < (1u << kMaxBits2)) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lo = ds.x, n = ds.y, bin = ds.z - H;
    for (uint32_t k = threadIdx.x * 4; k < n; k += kPartThreads * 4) {    // lo is a multiple of 4: aligned 16-byte loads
        const uint4 q = *reinterpret_cast<const uint4*>(L.tmp_idx + lo + k);
        const uint32_t kk[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t d = tile_of(kk[j]) & mask2;
            const bool valid = k + j < n;
            // same aggregation as the partition's ranking: the first lane's tile is added once for all its lanes
            const unsigned long long act = __ballot(valid);
            if (act != 0ull) {
                const int l0 = __ffsll((long long)act) - 1;
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, l0);
                const unsigned long long m0 = __ballot(valid && d == d0);
                if (lane == l0) atomicAdd(&s_cnt[d0], (uint32_t)__popcll(m0));
                else if (valid && d != d0) atomicAdd(&s_cnt[d], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < (1u << L.bits2)) {
        const uint32_t t = (bin << L.bits2) + threadIdx.x, v = s_cnt[threadIdx.x];
        if (v && t < L.n_tiles)       // the unit's group: the cursors pass 2 will use for this unit (see kLogGroups)
            __hip_atomic_fetch_add(&L.hist[t * kLogGroups2 + (unit & (kLogGroups2 - 1))], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Hot-tile form: records per pass-1 digit and group, counted from the log (indices only: 4 of a record's 12 bytes).
// A workgroup serves the chunks of ONE group (the grid is a multiple of kLogGroups) and flushes its LDS histogram once.
__global__ void __launch_bounds__(kPartThreads) k_log_count1(LogReduceParams L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_dyn);                       // [kMaxBins]
    uint32_t* s_map32 = s_cnt + kMaxBins;
    const uint16_t* s_dmap = reinterpret_cast<const uint16_t*>(s_map32);       // [n_tiles]
    const uint32_t nb1 = (L.n_tiles + (1u << L.bits2) - 1) >> L.bits2;
    const uint32_t nd = L.dmeta[0] + nb1;
    const uint32_t n_units = claimed_chunks(L);
    for (uint32_t d = threadIdx.x; d < (uint32_t)kMaxBins; d += kPartThreads) s_cnt[d] = 0;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(L.dmap);
    for (uint32_t i = threadIdx.x; i < (L.n_tiles + 1) / 2; i += kPartThreads) s_map32[i] = src[i];
    __syncthreads();
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const uint32_t lo = unit * kLogChunk, n = L.log_fill[unit];
        for (uint32_t k = threadIdx.x * 4; k < n; k += kPartThreads * 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(L.log_idx + lo + k);
            const uint32_t kk[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) if (k + j < n) atomicAdd(&s_cnt[s_dmap[tile_of(kk[j])]], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nd; d += kPartThreads)
        if (s_cnt[d]) __hip_atomic_fetch_add(&L.hist1[d * kLogGroups + (blockIdx.x & (kLogGroups - 1))], s_cnt[d], __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
}

// Hot-tile map from a tile histogram (the tile counts of an earlier batch of the same scene -- normally the pilot
// batch): the tiles with >= tau records are hot, tau the smallest threshold that leaves at most max_hot of them.
// One workgroup; the histogram is a few thousand words.
__global__ void __launch_bounds__(kScanThreads) k_log_plan(const uint32_t* cnt, uint32_t n_tiles, uint32_t bits2, uint32_t max_hot,
                                                          uint16_t* dmap, uint32_t* dmeta)
{
    __shared__ uint32_t s_wave[kScanThreads / 64];
    __shared__ uint32_t s_tot, s_n;
    const uint32_t per = (n_tiles + kScanThreads - 1) / kScanThreads;
    const uint32_t t0 = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    uint32_t a = 1u, b = 0xffffffffu;         // invariant: #{cnt >= b} <= max_hot
    while (a < b) {
        const uint32_t m = a + ((b - a) >> 1);
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        uint32_t mine = 0;
        for (uint32_t t = t0; t < t1; t++) mine += cnt[t] >= m ? 1u : 0u;
        if (mine) atomicAdd(&s_n, mine);
        __syncthreads();
        const uint32_t n = s_n;
        __syncthreads();
        if (n <= max_hot) b = m; else a = m + 1u;
    }
    const uint32_t tau = a;
    uint32_t mine = 0;
    for (uint32_t t = t0; t < t1; t++) mine += cnt[t] >= tau ? 1u : 0u;
    uint32_t rank = block_exclusive_scan(mine, s_wave, &s_tot);
    const uint32_t H = s_tot;
    for (uint32_t t = t0; t < t1; t++) {
        if (cnt[t] >= tau) dmap[t] = (uint16_t)rank++;
        else dmap[t] = (uint16_t)(H + (t >> bits2));
    }
    if (threadIdx.x == 0) { dmeta[0] = H; dmeta[1] = tau; if (n_tiles & 1u) dmap[n_tiles] = 0; }
}

// ---------------------------------------------------------------------------------------------------------
constexpr int kReduceThreads = 512;
#ifndef LT_REDUCE_WAVES
#define LT_REDUCE_WAVES 8     /* <= 64 VGPRs: the reduce workgroup (2 waves per SIMD) fits the 128 registers three walk waves leave */
#endif

template <typename TV, typename AT>
__device__ __forceinline__ void add8(AT* s_tile, const uint4 q, const Quad<TV>& a, const Quad<TV>& b)
{
    const uint32_t m = kTileSize - 1;
    lds_add(&s_tile[q.x & m], (AT)a.v[0]); lds_add(&s_tile[(q.x >> 16) & m], (AT)a.v[1]);
    lds_add(&s_tile[q.y & m], (AT)a.v[2]); lds_add(&s_tile[(q.y >> 16) & m], (AT)a.v[3]);
    lds_add(&s_tile[q.z & m], (AT)b.v[0]); lds_add(&s_tile[(q.z >> 16) & m], (AT)b.v[1]);
    lds_add(&s_tile[q.w & m], (AT)b.v[2]); lds_add(&s_tile[(q.w >> 16) & m], (AT)b.v[3]);
}

template <typename TV, int THREADS>
__global__ void __launch_bounds__(THREADS, LT_REDUCE_WAVES) k_log_reduce(LogReduceParams L)
{
    constexpr uint32_t kReduceThreads = THREADS;      // (512; 256 = one wave per SIMD, for the register quarter beside four 112-VGPR walk waves)
#if LT_RED_PRIO      // (compile option: the log reduction's waves issue ahead of a walk's -- they need 8 % of its VALU work and sit on the critical path of their job)
    __builtin_amdgcn_s_setprio(LT_RED_PRIO);
#endif
    typedef typename AccT<TV>::type AT;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    AT* s_tile = reinterpret_cast<AT*>(s_raw);
    __shared__ uint32_t s_item[2][6];     // [tile, lo, hi, slices of the tile, item, hot] of the current and the next work item
    const uint32_t n_items = L.meta[LM_ITEMS_R], slice = L.meta[LM_SLICE];
    // work items differ in size (a slice holds up to `slice` records, most tiles far fewer): they are pulled from a
    // device counter (a few thousand per launch).  Lane 0 claims and describes the NEXT item while the workgroup adds
    // the current one: the tile search (dependent loads) is off the critical path.
    auto claim = [&](uint32_t* out) {
        const uint32_t item = atomicAdd(&L.meta[LM_WORK], 1u);
        out[4] = item;
        if (item < n_items) {
            const uint32_t a = upper_slot(L.items_r, L.n_tiles, item);
            const uint32_t lo = L.tile_base[a] + (item - L.items_r[a]) * slice, end = L.tile_base[a] + L.tile_cnt[a];
            out[0] = a; out[1] = lo; out[2] = end - lo < slice ? end : lo + slice;
            out[3] = L.items_r[a + 1] - L.items_r[a];
            out[5] = L.dmap && L.dmap[a] < L.dmeta[0];      // hot tile: final in pass 1's output (the tmp buffers)
        }
    };
    if (threadIdx.x == 0) claim(s_item[0]);
    for (uint32_t v = threadIdx.x; v < kTileSize; v += kReduceThreads) s_tile[v] = 0;
    __syncthreads();
  for (int cur = 0;; cur ^= 1) {
    if (s_item[cur][4] >= n_items) break;
    if (threadIdx.x == 0) claim(s_item[cur ^ 1]);
    const uint32_t t = s_item[cur][0], lo = s_item[cur][1], hi = s_item[cur][2];
    const bool shared_tile = L.flush_atomic || s_item[cur][3] > 1;
    const bool hot = __builtin_amdgcn_readfirstlane((int)s_item[cur][5]) != 0;     // (scalar: the buffer choice costs no vector register)
    const uint16_t* idx = reinterpret_cast<const uint16_t*>(hot ? L.tmp_idx : L.log_idx);   // in-tile positions, see k_log_part's last pass
    const TV* val = reinterpret_cast<const TV*>(hot ? L.tmp_val : L.log_val);
    // `lo` is a multiple of 8 records (tile starts are padded, slices are multiples of 8192): a lane takes 8
    // consecutive records with one 16-byte load of positions and 2 (f32) / 4 (f64, u64) 16-byte loads of values, and
    // keeps two such groups in flight -- the tile pins the workgroup at 8 waves per CU, so the memory-level
    // parallelism (~80 KB per CU) has to come from the instruction stream.
    const uint32_t n8 = (hi - lo) >> 3;
    const uint4* idx8 = reinterpret_cast<const uint4*>(idx + lo);
    const Quad<TV>* val4 = reinterpret_cast<const Quad<TV>*>(val + lo);
    uint32_t g = threadIdx.x;
    for (; g + kReduceThreads < n8; g += 2 * kReduceThreads) {
        const uint4 q0 = idx8[g], q1 = idx8[g + kReduceThreads];
        const Quad<TV> a0 = val4[2 * g], b0 = val4[2 * g + 1];
        const Quad<TV> a1 = val4[2 * (g + kReduceThreads)], b1 = val4[2 * (g + kReduceThreads) + 1];
        add8<TV, AT>(s_tile, q0, a0, b0);
        add8<TV, AT>(s_tile, q1, a1, b1);
    }
    if (g < n8) { const uint4 q0 = idx8[g]; const Quad<TV> a0 = val4[2 * g], b0 = val4[2 * g + 1]; add8<TV, AT>(s_tile, q0, a0, b0); }
    {
        const uint32_t k = lo + (n8 << 3) + threadIdx.x;
        if (k < hi) lds_add(&s_tile[idx[k] & (kTileSize - 1)], (AT)val[k]);
    }
    __syncthreads();
    TV* grid = reinterpret_cast<TV*>(L.grid);
    const uint32_t tx = t % L.ntx, ty = (t / L.ntx) % L.nty, tz = t / (L.ntx * L.nty);
    for (uint32_t v = threadIdx.x; v < kTileSize; v += kReduceThreads) {
        const TV a = (TV)s_tile[v];
        s_tile[v] = 0;                                   // ready for the next item
        const uint32_t vx = (tx << kTileBX) | (v & 31u), vy = (ty << kTileBY) | ((v >> 5) & 31u), vz = (tz << kTileBZ) | (v >> 10);
        if (a != 0 && vx < L.nx && vy < L.ny && vz < L.nz) {
            TV* dst = &grid[((size_t)vz * L.ny + vy) * L.nx + vx];
            if (shared_tile) __hip_atomic_fetch_add(dst, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst += a;   // exclusive owner of this tile: plain read-add-write
        }
    }
    __syncthreads();   // the LDS tile is clean again, s_item[cur ^ 1] is complete
  }
}

}  // namespace

uint32_t log_part_item() { return kPartItem; }
uint32_t log_max_digits() { return (uint32_t)kMaxBins; }

hipError_t launch_log_scan_tiles(const LogReduceParams& L, hipStream_t s)
{
    // one-pass form: the only partition pass is the final one and runs on cursor1
    hipLaunchKernelGGL(k_log_scan_tiles, dim3(1), dim3(kScanThreads), 0, s, L.hist, L.bits2 ? kLogGroups2 : kLogGroups, L.tile_base, L.tile_cnt,
                       L.bits2 ? L.cursor2 : L.cursor1, L.items_r, L.meta, L.job, L.n_tiles, L.bits2 ? L.dmap : nullptr, L.dmeta,
                       L.bin_base, L.bin_cnt);
    return hipGetLastError();
}

hipError_t launch_log_scan_bins(const LogReduceParams& L, hipStream_t s)
{
    const uint32_t nb1 = (L.n_tiles + (1u << L.bits2) - 1) >> L.bits2;
    if (L.bits2 == 0 || L.bits2 > kMaxBits2 || nb1 > (uint32_t)kMaxBins) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_log_scan_bins, dim3(1), dim3(kScanThreads), 0, s, L.hist1, L.bin_base, L.bin_cnt, L.cursor1, L.items2,
                       L.items_c, L.meta, nb1, L.dmap ? L.dmeta : nullptr);
    return hipGetLastError();
}

hipError_t launch_log_plan(const uint32_t* tile_cnt, uint32_t n_tiles, uint32_t bits2, uint32_t max_hot, uint16_t* dmap,
                           uint32_t* dmeta, hipStream_t s)
{
    const uint32_t nb1 = (n_tiles + (1u << bits2) - 1) >> bits2;
    if (bits2 == 0 || n_tiles > kMaxHotTiles || nb1 + max_hot > (uint32_t)kMaxBins) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_log_plan, dim3(1), dim3(kScanThreads), 0, s, tile_cnt, n_tiles, bits2, max_hot, dmap, dmeta);
    return hipGetLastError();
}

// persistent grids: as many workgroups as are resident at once (occupancy query x CUs); the result is cached per
// kernel (a benign race: every thread computes the same number)
constexpr int kMaxDevices = 16;      // occupancy of a kernel is cached per device: a process may drive GPUs that differ (lt_create takes a device id)
struct BlockCache { std::atomic<unsigned> per_dev[kMaxDevices]; };
static unsigned persistent_blocks(BlockCache& cache, const void* fn, int threads, size_t lds, int max_per_cu = 0)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    std::atomic<unsigned>* slot = dev < kMaxDevices ? &cache.per_dev[dev] : nullptr;      // (beyond that: ask every time)
    unsigned b = slot ? slot->load(std::memory_order_relaxed) : 0u;
    if (b) return b;
    int per_cu = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    b = (unsigned)(per_cu * cus);
    if (slot) slot->store(b, std::memory_order_relaxed);
    return b;
}

// the hot-tile map of k_log_part<.., 1, true> / k_log_count1 in LDS: one size for every grid, so that the occupancy
// (and the cached persistent grid) does not depend on the scene
constexpr size_t kMapLds = (size_t)kMaxHotTiles * sizeof(uint16_t) + 4;

template <typename TV, int PASS, bool HOT, bool ALONE> static hipError_t launch_part_ta(const LogReduceParams& L, hipStream_t s)
{
    const size_t lds = (size_t)kPartItem * (sizeof(TV) + sizeof(uint32_t)) + (size_t)(2 * kMaxBins + 2) * sizeof(uint32_t) +
                       (HOT ? kMapLds : 0);
    const void* fn = reinterpret_cast<const void*>(&k_log_part<TV, PASS, HOT, ALONE>);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    static BlockCache blocks{};   // per instantiation
    hipLaunchKernelGGL((k_log_part<TV, PASS, HOT, ALONE>), dim3(persistent_blocks(blocks, fn, kPartThreads, lds)), dim3(kPartThreads), lds, s, L);
    return hipGetLastError();
}
template <typename TV, int PASS, bool HOT> static hipError_t launch_part_t(const LogReduceParams& L, hipStream_t s)
{
    return L.alone ? launch_part_ta<TV, PASS, HOT, true>(L, s) : launch_part_ta<TV, PASS, HOT, false>(L, s);
}
template <int PASS, bool HOT> static hipError_t launch_part(const LogReduceParams& L, hipStream_t s)
{
    if (L.tally == LT_TALLY_F32) return launch_part_t<float, PASS, HOT>(L, s);
    if (L.tally == LT_TALLY_F64) return launch_part_t<double, PASS, HOT>(L, s);
    return launch_part_t<unsigned long long, PASS, HOT>(L, s);
}
template <typename TV, int THREADS> static hipError_t launch_part_lds_t(const LogReduceParams& L, hipStream_t s)
{
    const size_t lds = (size_t)lds_part_item(THREADS) * (3 * sizeof(TV) + 2 * sizeof(uint32_t)) + (size_t)(kMaxBins + 2) * sizeof(uint32_t);
    const void* fn = reinterpret_cast<const void*>(&k_log_part_lds<TV, THREADS>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    static BlockCache blocks{};
    unsigned grid = persistent_blocks(blocks, fn, THREADS, lds, 1) / kLogGroups * kLogGroups;      // one per CU; a workgroup serves one cursor group
    if (grid < kLogGroups) grid = kLogGroups;
    hipLaunchKernelGGL((k_log_part_lds<TV, THREADS>), dim3(grid), dim3(THREADS), lds, s, L);
    return hipGetLastError();
}
template <int THREADS> static hipError_t launch_part_lds(const LogReduceParams& L, hipStream_t s)
{
    if (L.tally == LT_TALLY_F32) return launch_part_lds_t<float, THREADS>(L, s);
    if (L.tally == LT_TALLY_F64) return launch_part_lds_t<double, THREADS>(L, s);
    return launch_part_lds_t<unsigned long long, THREADS>(L, s);
}

hipError_t launch_log_part1(const LogReduceParams& L, hipStream_t s)
{
    if (L.lds_part) {      // lt_set_tuning "part_lds": bit 0 = where it applies (one-pass grids), bit 1 = or fail (tests: proves the route), bit 2 = 1024 lanes
        if (!L.dmap && L.bits2 == 0 && L.n_tiles <= (uint32_t)kMaxBins)
            return (L.lds_part & 4) ? launch_part_lds<256>(L, s) : launch_part_lds<512>(L, s);
        if (L.lds_part & 2) return hipErrorInvalidValue;
    }
    if (L.dmap) {
        if (L.bits2 == 0 || L.n_tiles > kMaxHotTiles) return hipErrorInvalidValue;
        return launch_part<1, true>(L, s);
    }
    return launch_part<1, false>(L, s);
}
hipError_t launch_log_count1(const LogReduceParams& L, hipStream_t s)
{
    if (!L.dmap || L.bits2 == 0 || L.n_tiles > kMaxHotTiles) return hipErrorInvalidValue;
    const size_t lds = (size_t)kMaxBins * sizeof(uint32_t) + kMapLds;
    const void* fn = reinterpret_cast<const void*>(&k_log_count1);
    static BlockCache blocks{};
    const unsigned b = persistent_blocks(blocks, fn, kPartThreads, lds) / kLogGroups * kLogGroups;   // a workgroup serves one group
    hipLaunchKernelGGL(k_log_count1, dim3(b < kLogGroups ? kLogGroups : b), dim3(kPartThreads), lds, s, L);
    return hipGetLastError();
}
hipError_t launch_log_part2(const LogReduceParams& L, hipStream_t s)
{
    // work units of pass 2, then the tiles of pass 1's output are counted per unit group: the scan of those counts
    // places pass 2's output
    hipLaunchKernelGGL(k_log_items2, dim3(256), dim3(256), 0, s, L);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    static BlockCache blocks{};
    const void* fn = reinterpret_cast<const void*>(&k_log_count2);
    hipLaunchKernelGGL(k_log_count2, dim3(persistent_blocks(blocks, fn, kPartThreads, 0)), dim3(kPartThreads), 0, s, L);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = launch_log_scan_tiles(L, s)) != hipSuccess) return e;
    return launch_part<2, false>(L, s);
}

template <typename TV, int THREADS> static hipError_t launch_reduce_t(const LogReduceParams& L, hipStream_t s)
{
    const size_t lds = (size_t)kTileSize * sizeof(typename AccT<TV>::type);
    const void* fn = reinterpret_cast<const void*>(&k_log_reduce<TV, THREADS>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    static BlockCache blocks{};
    hipLaunchKernelGGL((k_log_reduce<TV, THREADS>), dim3(persistent_blocks(blocks, fn, THREADS, lds, 1)), dim3(THREADS), lds, s, L);
    return hipGetLastError();
}
template <int THREADS> static hipError_t launch_reduce_n(const LogReduceParams& L, hipStream_t s)
{
    if (L.tally == LT_TALLY_F32) return launch_reduce_t<float, THREADS>(L, s);
    if (L.tally == LT_TALLY_F64) return launch_reduce_t<double, THREADS>(L, s);
    return launch_reduce_t<unsigned long long, THREADS>(L, s);
}
hipError_t launch_log_reduce(const LogReduceParams& L, hipStream_t s)
{
    return (L.lds_part & 4) ? launch_reduce_n<256>(L, s) : launch_reduce_n<kReduceThreads>(L, s);
}

}  // namespace ltk

#ifdef LT_PART_PROF
extern "C" int lt_diag_part_phases(unsigned long long* out8, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(ltk::g_part_phase), 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    static const unsigned long long zero[8] = {};
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(ltk::g_part_phase), zero, sizeof(zero)) != hipSuccess) return 3;
    return 0;
}
#endif
