// regions.hip -- connected-component labelling of the decoded image and
// regionprops-style aggregation, for gfx950 (MI355X).
//
// Replaces the per-barcode skimage.measure.label + regionprops loops of
// merlin/util/decoding.py:165-241 (extract_barcodes_with_index), :323-334
// (extract_refactors) and :359-369 (_extract_backgrounds): ONE multi-label
// union-find pass labels every barcode's regions at once, and one segmented
// pass over the assigned pixels (a few % of the image) produces every region
// row.  All of it is HBM-bound integer work on sparse data; nothing here is
// shaped like a GEMM.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include "common.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ CCL ---

__device__ __forceinline__ int ld_label(const int32_t *L, int i)
{
    // A relaxed load that the compiler may not cache in a register.  A stale
    // value (another XCD linked the node meanwhile) is harmless: labels only
    // ever decrease and every link is made by an atomicMin whose return value
    // is checked, so a stale "root" just costs another round in unite().
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int find_root(const int32_t *L, int x)
{
    int p = ld_label(L, x);
    while (p != x) { x = p; p = ld_label(L, x); }
    return x;
}

// Union by smallest index: the root of a component is always its first pixel
// in raster order, which is what orders skimage's labels.
__device__ __forceinline__ void unite(int32_t *L, int a, int b)
{
    for (;;) {
        a = find_root(L, a);
        b = find_root(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;            // a was linked meanwhile: keep uniting its parent
    }
}

// ---- tile-local labelling -------------------------------------------------
// A 64 x 32 tile per workgroup, labelled completely inside LDS: lane = column,
// wave w owns rows 8w .. 8w+7.  Every pixel of a horizontal run first points
// at the run's first pixel (one ballot per row, no atomics); a run is then
// united with the runs it touches in the row above only where it FIRST touches
// them (with N unless W and NW match -- the W pixel made that link; without N,
// with NW unless W matches, and with NE), by atomicMin on LDS labels.  The
// tile's components leave with the global raster index of their first pixel
// in the tile as label; ccl_border then makes the few links that cross a tile
// edge in global memory.  (One global union-find over all pixels, the previous
// design, spent 0.28 ms per 7-z FOV waiting on dependent L2 round trips.)
constexpr int CT_W = 64, CT_H = 32, CT_ROWS = CT_H / (NT / 64);

__device__ __forceinline__ int lds_find(const int *lab, int x)
{
    int p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != x) {
        x = p;
        p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return x;
}

__device__ __forceinline__ void lds_unite(int *lab, int a, int b)
{
    for (;;) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(lab + a, b);
        if (old == a) return;
        a = old;
    }
}

constexpr int CT_LINKS = 192;       // links a wave gathers before it makes them (>= 128)

// the lanes of one wave exchange data through LDS: order the LDS traffic only
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ void __launch_bounds__(NT)
ccl_tile(const int16_t *__restrict__ dec, int32_t *__restrict__ L, int H, int W)
{
    __shared__ __align__(16) int16_t s_code[CT_H][CT_W];
    __shared__ int s_lab[CT_H * CT_W];
    __shared__ int2 s_links[NT / 64][CT_LINKS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * CT_W + lane;
    const int y0 = blockIdx.y * CT_H + wave * CT_ROWS;
    const int64_t plane = (int64_t)blockIdx.z * H * W;

    int code[CT_ROWS];
    bool cont[CT_ROWS];       // same barcode as the pixel to the left, in this tile
    // the tile's barcodes: one 16-byte load (8 pixels) per thread where the
    // image allows it, instead of eight 2-byte loads
    const bool wide = !(W & 7) && !(reinterpret_cast<uintptr_t>(dec) & 15) && blockIdx.x * CT_W + CT_W <= W &&
                      blockIdx.y * CT_H + CT_H <= H;
    if (wide) {
        const int ty = threadIdx.x >> 3, tx = (threadIdx.x & 7) * 8;
        const int64_t src = plane + (int64_t)(blockIdx.y * CT_H + ty) * W + blockIdx.x * CT_W + tx;
        *reinterpret_cast<int4 *>(&s_code[ty][tx]) = *reinterpret_cast<const int4 *>(dec + src);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < CT_ROWS; ++r) {
        const int y = y0 + r, ly = wave * CT_ROWS + r;
        if (wide) {
            code[r] = (int)s_code[ly][lane];
        } else {
            code[r] = x < W && y < H ? (int)dec[plane + (int64_t)y * W + x] : -1;
            s_code[ly][lane] = (int16_t)code[r];
        }
        const int prev = __shfl_up(code[r], 1);
        cont[r] = code[r] >= 0 && lane > 0 && prev == code[r];
        const unsigned long long breaks = __ballot(!cont[r]);       // bit 0 is always set
        const unsigned long long upto = breaks & (~0ull >> (63 - lane));
        const int start = 63 - __clzll((long long)upto);
        s_lab[ly * CT_W + lane] = code[r] >= 0 ? ly * CT_W + start : -1;
    }
    __syncthreads();
    // The links of the wave's eight rows are first gathered into a list (ballot
    // + prefix count, no atomics) and then made 64 at a time, one per lane:
    // made row by row, each row cost the wave the latency of its slowest
    // lane's find / atomicMin chain, eight times over.
    int n_links = 0;                                   // wave-uniform
    int2 *links = s_links[wave];
    auto make_links = [&]() {
        wave_lds_sync();
        for (int q = lane; q < n_links; q += 64) lds_unite(s_lab, links[q].x, links[q].y);
        wave_lds_sync();
        n_links = 0;
    };
    auto append = [&](bool want, int a, int b) {
        const unsigned long long m = __ballot(want);
        const int cnt = __popcll(m);
        if (n_links + cnt > CT_LINKS) make_links();
        if (want) links[n_links + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(a, b);
        n_links += cnt;
    };
#pragma unroll
    for (int r = 0; r < CT_ROWS; ++r) {
        const int ly = wave * CT_ROWS + r;
        const bool live = code[r] >= 0 && ly > 0;
        const int i = ly * CT_W + lane;
        const int c = code[r];
        const int up = live ? ly - 1 : 0;              // row 0 is never read as a link target
        const bool n = live && s_code[up][lane] == c;
        const bool nw = live && lane > 0 && s_code[up][lane - 1] == c;
        const bool ne = live && lane < CT_W - 1 && s_code[up][lane + 1] == c;
        // with N unless W and NW match (the W pixel made that link); without N,
        // with NW unless W matches, and with NE
        const bool link_n = n && !(cont[r] && nw);
        const bool link_nw = !n && nw && !cont[r];
        append(link_n || link_nw, i, link_n ? i - CT_W : i - CT_W - 1);
        append(!n && ne, i, i - CT_W + 1);
    }
    make_links();
    __syncthreads();
    const int base = (int)plane + blockIdx.y * CT_H * W + blockIdx.x * CT_W;
    if (wide && !(reinterpret_cast<uintptr_t>(L) & 15)) {
        // four labels (16 bytes) per store
#pragma unroll
        for (int k = 0; k < CT_H * CT_W / 4 / NT; ++k) {
            const int q = k * NT + threadIdx.x;
            const int ly = q / (CT_W / 4), lx = (q % (CT_W / 4)) * 4;
            int4 v;
            int *e = reinterpret_cast<int *>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = -1;
                if (s_code[ly][lx + j] >= 0) {
                    const int root = lds_find(s_lab, ly * CT_W + lx + j);
                    e[j] = base + (root / CT_W) * W + (root % CT_W);
                }
            }
            *reinterpret_cast<int4 *>(L + plane + (int64_t)(blockIdx.y * CT_H + ly) * W + blockIdx.x * CT_W + lx) = v;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < CT_ROWS; ++r) {
        const int y = y0 + r, ly = wave * CT_ROWS + r;
        if (x >= W || y >= H) continue;
        int out = -1;
        if (code[r] >= 0) {
            const int root = lds_find(s_lab, ly * CT_W + lane);
            out = base + (root / CT_W) * W + (root % CT_W);
        }
        L[plane + (int64_t)y * W + x] = out;
    }
}

// the backward links of one pixel (W at a tile's left edge, NW, N, NE), with
// the same first-touch rule as ccl_tile, in global memory
__device__ __forceinline__ void merge_pixel(const int16_t *__restrict__ dec, int32_t *L, int x,
                                            int y, int z, int H, int W)
{
    const int i = (z * H + y) * W + x;
    const int16_t code = dec[i];
    if (code < 0) return;
    const bool w = x > 0 && dec[i - 1] == code;
    const bool n = y > 0 && dec[i - W] == code;
    const bool nw = y > 0 && x > 0 && dec[i - W - 1] == code;
    const bool ne = y > 0 && x + 1 < W && dec[i - W + 1] == code;
    if (w && x % CT_W == 0) unite(L, i, i - 1);
    if (n) {
        if (!(w && nw)) unite(L, i, i - W);
    } else {
        if (nw && !w) unite(L, i, i - W - 1);
        if (ne) unite(L, i, i - W + 1);
    }
}

// Links across tile edges, one thread per edge pixel of plane blockIdx.y: first
// the pixels of the tiles' top rows (y = 32 k, k >= 1), then the pixels of the
// tiles' left and right columns (x = 64 k and 64 k + 63).  A link that stays
// inside a tile is found already made and costs two label loads.  (A flat index:
// the earlier 3-D grid launched 31 808 workgroups per 7-z FOV, most of them empty.)
__global__ void __launch_bounds__(NT)
ccl_border(const int16_t *__restrict__ dec, int32_t *L, int H, int W, int n_edge_rows,
           int n_edge_cols)
{
    const int z = blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int64_t row_px = (int64_t)n_edge_rows * W;
    if (idx < row_px) {
        const int k = (int)(idx / W);
        merge_pixel(dec, L, (int)(idx - (int64_t)k * W), (k + 1) * CT_H, z, H, W);
        return;
    }
    const int64_t j = idx - row_px;
    if (j >= (int64_t)n_edge_cols * H) return;
    const int col = (int)(j % n_edge_cols), y = (int)(j / n_edge_cols);
    const int x = (col >> 1) * CT_W + (col & 1 ? CT_W - 1 : 0);
    // (a tile's top row belongs to the edge rows above, except image row 0)
    if (x < W && (y % CT_H != 0 || y == 0)) merge_pixel(dec, L, x, y, z, H, W);
}

// 26-connectivity: links to the nine neighbours in the plane below
__global__ void __launch_bounds__(NT)
ccl_merge_z(const int16_t *__restrict__ dec, int32_t *L, int H, int W)
{
    const int x = blockIdx.x * NT + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, z = blockIdx.z + 1;
    const int i = (z * H + y) * W + x;
    const int16_t code = dec[i];
    if (code < 0) return;
    const int base = i - W * H;
    for (int dy = -1; dy <= 1; ++dy) {
        if (y + dy < 0 || y + dy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            if (x + dx < 0 || x + dx >= W) continue;
            const int q = base + dy * W + dx;
            if (dec[q] == code) unite(L, i, q);
        }
    }
}

__global__ void __launch_bounds__(NT)
ccl_flatten(int32_t *L, int n, unsigned long long *counts)
{
    // grid-stride over groups of four labels; each thread keeps private counts
    // and the block issues ONE atomic pair at the very end, into one of 64 slot
    // pairs (same-address atomics serialise at ~17 ns each: with every block
    // adding to the same two words the kernel took 90 us at 2048 blocks, 122 at
    // 4096 and 210 at 8192; slotted it takes 57 / 54 / 51 us)
    __shared__ int s_fg[NT / 64], s_roots[NT / 64];
    int fg = 0, roots = 0;
    const int n4 = n >> 2;
    int4 *L4 = reinterpret_cast<int4 *>(L);
    for (int q = blockIdx.x * NT + threadIdx.x; q < n4; q += gridDim.x * NT) {
        int4 v = L4[q];
        if ((v.x & v.y & v.z & v.w) < 0) continue;      // four unassigned pixels
        int *e = reinterpret_cast<int *>(&v);
        bool changed = false;
        // the four first hops are independent loads; most labels are final
        // already and need no second one
        int hop[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hop[k] = e[k] >= 0 ? L[e[k]] : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = e[k];
            if (p >= 0) {
                int r = p;
                int t = hop[k];
                while (t != r) { r = t; t = L[r]; }
                changed |= r != p;
                e[k] = r;
                ++fg;
                roots += r == 4 * q + k;
            }
        }
        // most components lie inside one tile and left ccl_tile with their final label
        if (changed) L4[q] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {      // tail
        const int i = (n4 << 2) + threadIdx.x;
        const int p = L[i];
        if (p >= 0) {
            int r = p;
            int t = L[r];
            while (t != r) { r = t; t = L[r]; }
            L[i] = r;
            ++fg;
            roots += r == i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        fg += __shfl_down(fg, off);
        roots += __shfl_down(roots, off);
    }
    if ((threadIdx.x & 63) == 0) { s_fg[threadIdx.x >> 6] = fg; s_roots[threadIdx.x >> 6] = roots; }
    __syncthreads();
    if (threadIdx.x == 0) {
        fg = roots = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { fg += s_fg[w]; roots += s_roots[w]; }
        // partial counts go to one of 64 slot pairs behind the two results
        unsigned long long *slot = counts + 2 + 2 * (blockIdx.x % MF_LABEL_COUNT_SLOTS);
        if (fg) atomicAdd(slot + 0, (unsigned long long)fg);
        if (roots) atomicAdd(slot + 1, (unsigned long long)roots);
    }
}

__global__ void __launch_bounds__(64) ccl_counts_reduce(unsigned long long *counts)
{
    static_assert(MF_LABEL_COUNT_SLOTS == 64, "one lane per slot");
    const int lane = threadIdx.x;
    unsigned long long fg = counts[2 + 2 * lane], roots = counts[2 + 2 * lane + 1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        fg += __shfl_down(fg, off);
        roots += __shfl_down(roots, off);
    }
    if (lane == 0) { counts[0] = fg; counts[1] = roots; }
}

// --------------------------------------------------------- region table ---

// Order-preserving compaction in three fully parallel steps (count per
// 1024-item chunk, one-block scan of the chunk counts, emit): no serial
// look-back chain, which is what a device-wide select spends its time on at
// these sizes (measured: 128 us for 29 M items, 118 us for 7 M).
constexpr int CHUNK = 4 * NT;      // items per block: 4 consecutive per thread

__device__ __forceinline__ int block_reduce_sum(int v)
{
    __shared__ int s_part[NT / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += s_part[w];
    return t;
}

// exclusive prefix of v over the block's threads
__device__ __forceinline__ int block_exclusive_scan(int v)
{
    __shared__ int s_wave[NT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int j = 0; j < NT / 64; ++j) base += j < w ? s_wave[j] : 0;
    return base + inc - v;
}

// in-place exclusive scan of the chunk counts by ONE block; thread t owns a
// contiguous slice of `per` counts (a multiple of 4, read as independent
// 16-byte loads: the kernel is pure latency)
constexpr int SCAN_NT = 1024;
// (block b scans the b-th of gridDim.x arrays laid out one after the other, each
// padded to a multiple of 4 entries: `stride`; total[b] receives its sum)
__global__ void __launch_bounds__(SCAN_NT)
chunk_scan(int32_t *counts, int n_chunks, unsigned long long *total, int stride = 0)
{
    __shared__ int s_wave[SCAN_NT / 64];
    counts += (size_t)blockIdx.x * stride;
    if (total) total += blockIdx.x;
    const int per = ((n_chunks + SCAN_NT - 1) / SCAN_NT + 3) & ~3;
    const int lo = min(n_chunks, (int)threadIdx.x * per), hi = min(n_chunks, lo + per);
    const int hi4 = lo + ((hi - lo) & ~3);
    int sum = 0;
#pragma unroll 8
    for (int i = lo; i < hi4; i += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(counts + i);
        sum += v.x + v.y + v.z + v.w;
    }
    for (int i = hi4; i < hi; ++i) sum += counts[i];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    int base = 0;
    for (int j = 0; j < w; ++j) base += s_wave[j];
    int run = base + inc - sum;
#pragma unroll 8
    for (int i = lo; i < hi4; i += 4) {
        int4 v = *reinterpret_cast<const int4 *>(counts + i);
        const int a = run, b = a + v.x, c = b + v.y, d = c + v.z;
        run = d + v.w;
        *reinterpret_cast<int4 *>(counts + i) = make_int4(a, b, c, d);
    }
    for (int i = hi4; i < hi; ++i) {
        const int c = counts[i];
        counts[i] = run;
        run += c;
    }
    // the last non-empty slice ends at the grand total
    if (total && hi == n_chunks && lo < n_chunks) *total = (unsigned long long)run;
}

struct StatsArgs {
    const int32_t *pix;      // grouped by region in row order, raster order inside a region
    const int32_t *start;    // [n_regions + 1] where row r's pixels begin in pix
    int64_t tr_p;
    int64_t n_pix, n_reg;
    const int16_t *dec;
    const float *mag;
    const void *dist;
    int dist_f64;
    const float *traces;
    int64_t tr_z, tr_b;
    int B, Z, H, W, conn3d;
    double *rec;
    float *inten, *mean_tm, *min_tm;
};

// 16 lanes per region.  The region's pixels are taken 16 at a time: every lane
// fetches one pixel's index, magnitude and distance, works out where its trace
// lives and parks that in LDS; then all 16 lanes walk the 16 staged pixels in
// raster order, lane `sub` accumulating bits sub and sub+16, so the float sums
// run in the reference's sequential order while the memory latency and the
// index arithmetic are paid once per 16 pixels.
//
// The kernel is bound by VALU issue (float64 instructions hold a SIMD four
// clocks), so the walk carries as little as it can:
//  * the two float64 running sums (magnitude, distance) are ONE add per step:
//    the staged pixel holds {(double)m, d} side by side, even lanes add the
//    first, odd lanes the second -- each still the sequential sum;
//  * max magnitude / min distance are order-free: every lane folds its own
//    pixel and the 16 lanes are reduced once at the end;
//  * the centroid products (double)coord * (m / sum m) are formed by the lane
//    that owns the pixel, and lanes 0..2 add z, y, x sequentially (one add
//    per step instead of three conversions, three products, three adds);
//  * (tried and dropped: a pixel's 64 bytes of traces as four 16-byte loads
//    by four lanes, turned through LDS -- 4 instead of 16 load instructions
//    per chunk: 211 us against 202; processing the regions in root order
//    instead of emission order: 196.  What is left is a 64-byte-granular
//    gather of 0.55 GB at 2.7 TB/s.)
//  * entries past the end of a short chunk hold +0.0, which changes no sum
//    (the sums start at +0.0, so they are never -0.0), and the walk stops at
//    the longest chunk of the wave's four groups (a scalar branch).
constexpr int RSUB = 16;

struct StageGroup {                 // 448 bytes per 16-lane group
    union {
        struct {
            int64_t toff[RSUB];     // element offset of the pixel's trace (bit 0)
            double val[RSUB][2];    // {(double) magnitude, distance}
            float m[RSUB];          // magnitude (trace x magnitude statistics)
        } p1;
        struct {
            double prod[RSUB][3];   // z, y, x times m / sum(m)
            double out[4];          // msum broadcast; centroid hand-over
        } p2;
    };
};

__device__ __forceinline__ void group_sync()
{
    // the 16 lanes of a group sit in one wave: order the LDS traffic only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// longest chunk among the wave's active groups (wave-uniform): the ballot of
// sub < cnt is a low-bit mask per 16-lane field
__device__ __forceinline__ int wave_longest(int sub, int cnt)
{
    const unsigned long long b = __ballot(sub < cnt);
    const unsigned f = (unsigned)(b | b >> 16 | b >> 32 | b >> 48) & 0xffffu;
    return __builtin_amdgcn_readfirstlane(32 - __builtin_clz(f | 1u)) ;
}

// WIDE: more than 16 bits per codeword (a lane then also carries bit sub + 16)
// TM:   also the mean / min of trace x magnitude (OptimizeIteration)
template <bool WIDE, bool TM>
__global__ void __launch_bounds__(NT)
region_stats(const StatsArgs a)
{
    __shared__ StageGroup stage[NT / RSUB];
    int64_t r = ((int64_t)blockIdx.x * NT + threadIdx.x) / RSUB;
    const int sub = threadIdx.x & (RSUB - 1);
    StageGroup &st = stage[threadIdx.x / RSUB];
    const bool valid = r < a.n_reg;
    if (!valid) r = a.n_reg - 1;          // keep whole groups alive
    const int i0 = a.start[r], i1 = a.start[r + 1];
    const int n = i1 - i0;
    const int plane = a.H * a.W;
    const int first = a.pix[i0];
    const int zfirst = first / plane;

    // pass 1: magnitude sum / max, distance sum / min; per-bit sums
    double acc = 0.0;                     // even lanes: sum m, odd lanes: sum d
    double dmin = INFINITY;
    float mmax = -INFINITY;
    float s0 = 0.0f, stm0 = 0.0f, mn0 = INFINITY;     // bit sub
    float s1 = 0.0f, stm1 = 0.0f, mn1 = INFINITY;     // bit sub + 16
    const bool has0 = sub < a.B, has1 = WIDE && sub + RSUB < a.B;
    const int64_t boff0 = (int64_t)sub * a.tr_b, boff1 = (int64_t)(sub + RSUB) * a.tr_b;
    const int par = sub & 1;
    // The dependent loads run ahead of the walk: a chunk's pixel index is
    // fetched two chunks early and its magnitude and distance one chunk early
    // (beside the previous chunk's trace loads); the first two chunks'
    // (pixel, magnitude) stay in registers for the centroid pass.
    auto load_dist = [&](int p) -> double {
        return a.dist_f64 ? reinterpret_cast<const double *>(a.dist)[p]
                          : (double)reinterpret_cast<const float *>(a.dist)[p];
    };
    int p0 = first, p1 = first;
    float m0 = 0.0f, m1 = 0.0f;
    int pixA = i0 + sub < i1 ? a.pix[i0 + sub] : first;                    // chunk 0
    int pixB = i0 + RSUB + sub < i1 ? a.pix[i0 + RSUB + sub] : first;      // chunk 1
    float magA = a.mag[pixA];
    double distA = load_dist(pixA);
    int chunk = 0;
    for (int base = i0; base < i1; base += RSUB, ++chunk) {
        const int cnt = min(RSUB, i1 - base);
        const int kmax = wave_longest(sub, cnt);
        const int curp = pixA;
        const float curm = magA;
        const double curd = distA;
        if (base + RSUB + sub < i1) {          // the next chunk's pixel is known by now
            pixA = pixB;
            magA = a.mag[pixA];
            distA = load_dist(pixA);
        }
        if (base + 2 * RSUB + sub < i1) pixB = a.pix[base + 2 * RSUB + sub];
        if (chunk == 0) { p0 = curp; m0 = curm; }
        else if (chunk == 1) { p1 = curp; m1 = curm; }
        if (sub < cnt) {
            // a 2-D region lies in one plane; only 3-D needs the division
            const int zz = a.conn3d ? curp / plane : zfirst;
            st.p1.toff[sub] = (int64_t)zz * a.tr_z + (int64_t)(curp - zz * plane) * a.tr_p;
            st.p1.val[sub][0] = (double)curm;
            st.p1.val[sub][1] = curd;
            if constexpr (TM) st.p1.m[sub] = curm;
            mmax = fmaxf(mmax, curm);
            dmin = fmin(dmin, curd);
        } else {
            st.p1.val[sub][0] = 0.0;
            st.p1.val[sub][1] = 0.0;
        }
        group_sync();
        // issue this lane's trace loads for the whole chunk first
        float t0[RSUB], t1[RSUB];
#pragma unroll
        for (int k = 0; k < RSUB; ++k) {
            t0[k] = 0.0f;
            t1[k] = 0.0f;
            if (k < kmax) {                                   // scalar branch
                if (k < cnt) {
                    const int64_t off = st.p1.toff[k];
                    if (has0) t0[k] = a.traces[off + boff0];
                    if constexpr (WIDE) if (has1) t1[k] = a.traces[off + boff1];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < RSUB; ++k) {
            if (k < kmax) {                                   // scalar branch
                acc += st.p1.val[k][par];
                s0 = s0 + t0[k];
                if constexpr (WIDE) s1 = s1 + t1[k];
                if constexpr (TM) {
                    const float m = st.p1.m[k < cnt ? k : 0];
                    const float tm0 = t0[k] * m;
                    stm0 = stm0 + tm0;
                    mn0 = fminf(mn0, k < cnt ? tm0 : INFINITY);
                    if constexpr (WIDE) {
                        const float tm1 = t1[k] * m;
                        stm1 = stm1 + tm1;
                        mn1 = fminf(mn1, k < cnt ? tm1 : INFINITY);
                    }
                }
            }
        }
        group_sync();      // the next chunk (or pass 2) overwrites the stage
    }
    // fold the order-free extremes over the group's 16 lanes
#pragma unroll
    for (int d = 1; d < RSUB; d <<= 1) {
        mmax = fmaxf(mmax, __shfl_xor(mmax, d, RSUB));
        dmin = fmin(dmin, __shfl_xor(dmin, d, RSUB));
    }
    // sum m sits in the even lanes, sum d in the odd ones
    if (sub < 2) st.p2.out[sub] = acc;
    group_sync();
    const double msum = st.p2.out[0], dsum = st.p2.out[1];
    group_sync();

    // pass 2: magnitude-weighted centroid, lanes 0..2 = z, y, x
    double cacc = 0.0;
    const int csel = min(sub, 2);
    chunk = 0;
    for (int base = i0; base < i1; base += RSUB, ++chunk) {
        const int cnt = min(RSUB, i1 - base);
        const int kmax = wave_longest(sub, cnt);
        if (sub < cnt) {
            int myp;
            float mym;
            if (chunk == 0) { myp = p0; mym = m0; }
            else if (chunk == 1) { myp = p1; mym = m1; }
            else {                     // beyond 32 pixels: fetched again
                myp = a.pix[base + sub];
                mym = a.mag[myp];
            }
            const int zz = a.conn3d ? myp / plane : zfirst;
            const int rem = myp - zz * plane;
            const int y = rem / a.W;
            const int x = rem - y * a.W;
            const double w = (double)mym / msum;
            st.p2.prod[sub][0] = (double)zz * w;
            st.p2.prod[sub][1] = (double)y * w;
            st.p2.prod[sub][2] = (double)x * w;
        } else {
            st.p2.prod[sub][0] = 0.0;
            st.p2.prod[sub][1] = 0.0;
            st.p2.prod[sub][2] = 0.0;
        }
        group_sync();
#pragma unroll
        for (int k = 0; k < RSUB; ++k)
            if (k < kmax) cacc += st.p2.prod[k][csel];        // scalar branch
        group_sync();
    }
    if (sub < 3) st.p2.out[sub] = cacc;
    group_sync();
    if (!valid) return;
    double cz = st.p2.out[0], cy = st.p2.out[1], cx = st.p2.out[2];
    if (n == 1) {
        cz = zfirst; cy = (first % plane) / a.W; cx = first % a.W;
    }
    double mean_int = msum / n, mean_d = n > 1 ? dsum / n : dsum;
    if (a.conn3d) mean_int = (double)(float)mean_int;        // regionprops: f32
    if (a.conn3d && !a.dist_f64) mean_d = (double)(float)mean_d;
    if (!a.conn3d) cz = zfirst;

    if (sub == 0) {
        double *row = a.rec + r * MF_REC_COLS;
        row[0] = (double)a.dec[first];
        row[1] = (double)zfirst;
        row[2] = (double)n;
        row[3] = mean_int;
        row[4] = (double)mmax;
        row[5] = mean_d;
        row[6] = dmin;
        row[7] = cz;
        row[8] = cy;
        row[9] = cx;
        row[10] = (double)first;
        row[11] = 0.0;      // reserved (the index inside the group: mf_region_frame)
    }
    // per-bit means: sequential float32 sums in raster order, then / n
    // (np.mean(list of f32 rows, axis 0), decoding.py:197, 220, 330)
    if (has0) {
        a.inten[r * a.B + sub] = n > 1 ? s0 / (float)n : s0;
        if constexpr (TM) {
            if (a.mean_tm) a.mean_tm[r * a.B + sub] = stm0 / (float)n;
            if (a.min_tm) a.min_tm[r * a.B + sub] = mn0;
        }
    }
    if (has1) {
        a.inten[r * a.B + sub + RSUB] = n > 1 ? s1 / (float)n : s1;
        if constexpr (TM) {
            if (a.mean_tm) a.mean_tm[r * a.B + sub + RSUB] = stm1 / (float)n;
            if (a.min_tm) a.min_tm[r * a.B + sub + RSUB] = mn1;
        }
    }
}

// ---- horizontal runs ---------------------------------------------------------
// A run is a maximal horizontal sequence of pixels with the same barcode >= 0.
// Horizontally adjacent equal barcodes are always connected, so a run lies in
// ONE region, a region is a set of runs, and its first pixel in raster order
// (the root the labelling left in L) is the first pixel of one of its runs.
// The table is therefore prepared on runs (one per ~5 assigned pixels), not on
// pixels: the runs come out of the barcode image in raster order (count / scan /
// emit over 1024-pixel chunks: starts, ends and region roots in one go), are
// keyed with their region's place in the reference's row order and sorted
// (stable: a region's runs stay in raster order), and are then expanded back
// into the pixel list region_stats walks.

// A thread owns RUN_PXT = 16 consecutive pixels (two 16-byte loads of the
// barcode image; a block 4096 pixels).  Flags: bit k = run start, bit 16 + k =
// run end, `roots` bit k = region root (a run start whose label is its own
// index); lab[k] = the label of a run start.  32-bit arithmetic: the volume
// has fewer than 2^31 pixels.
constexpr int RUN_PXT = 16;
constexpr int RUN_CHUNK = RUN_PXT * NT;

__device__ __forceinline__ unsigned run_flags(const int16_t *__restrict__ dec,
                                              const int32_t *__restrict__ L, uint32_t i0,
                                              uint32_t n, uint32_t W, int32_t (&lab)[RUN_PXT],
                                              unsigned &roots)
{
    roots = 0u;
    if (i0 >= n) return 0u;
    int c[RUN_PXT + 2];             // c[0] = left neighbour, c[17] = right neighbour
    if (i0 + RUN_PXT <= n) {
        const uint4 a = *reinterpret_cast<const uint4 *>(dec + i0);
        const uint4 b = *reinterpret_cast<const uint4 *>(dec + i0 + 8);
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if ((a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w & 0x80008000u) == 0x80008000u)
            return 0u;              // sixteen unassigned pixels
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            c[1 + 2 * k] = (int)(int16_t)(w[k] & 0xffffu);
            c[2 + 2 * k] = (int)(int16_t)(w[k] >> 16);
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < RUN_PXT; ++k) c[1 + k] = i0 + k < n ? (int)dec[i0 + k] : -1;
    }
    // the neighbours matter only beside an assigned pixel
    c[0] = c[1] >= 0 && i0 > 0 ? (int)dec[i0 - 1] : -1;
    c[RUN_PXT + 1] = c[RUN_PXT] >= 0 && i0 + RUN_PXT < n ? (int)dec[i0 + RUN_PXT] : -1;
    uint32_t x = i0 % W;
    unsigned f = 0;
#pragma unroll
    for (int k = 0; k < RUN_PXT; ++k) {
        if (c[1 + k] >= 0) {
            if (x == 0 || c[k] != c[1 + k]) f |= 1u << k;
            if (x == W - 1 || c[2 + k] != c[1 + k]) f |= 0x10000u << k;
        }
        if (++x == W) x = 0;
    }
    // the labels of the run starts: all loads first, then the root test
#pragma unroll
    for (int k = 0; k < RUN_PXT; ++k) lab[k] = f >> k & 1u ? L[i0 + k] : -1;
#pragma unroll
    for (int k = 0; k < RUN_PXT; ++k)
        if (lab[k] == (int32_t)(i0 + k)) roots |= 1u << k;
    return f;
}

// three counts per chunk packed into one 64-bit word (21 bits each)
__device__ __forceinline__ unsigned long long pack3(unsigned f, unsigned roots)
{
    return (unsigned long long)__popc(f & 0xffffu) | (unsigned long long)__popc(f >> 16) << 21 |
           (unsigned long long)__popc(roots) << 42;
}

__global__ void __launch_bounds__(NT)
run_count(const int16_t *__restrict__ dec, const int32_t *__restrict__ L, int64_t n, int W,
          int n_chunks, int32_t *__restrict__ counts)
{
    __shared__ unsigned long long s_part[NT / 64];
    const uint32_t i0 = blockIdx.x * (uint32_t)RUN_CHUNK + (uint32_t)RUN_PXT * threadIdx.x;
    int32_t lab[RUN_PXT];
    unsigned roots;
    const unsigned f = run_flags(dec, L, i0, (uint32_t)n, (uint32_t)W, lab, roots);
    unsigned long long v = pack3(f, roots);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) t += s_part[w];
        const int cs = (n_chunks + 3) & ~3;       // arrays padded for chunk_scan's 16-byte loads
        counts[blockIdx.x] = (int32_t)(t & 0x1fffffu);
        counts[cs + blockIdx.x] = (int32_t)(t >> 21 & 0x1fffffu);
        counts[2 * cs + blockIdx.x] = (int32_t)(t >> 42);
    }
}

// Runs in raster order: first pixel, last pixel and region root of run i; the
// regions' roots in raster (= label) order with the key of the reference's row
// order: z plane (2-D mode), then barcode.
__global__ void __launch_bounds__(NT)
run_emit(const int16_t *__restrict__ dec, const int32_t *__restrict__ L, int64_t n, int W,
         int n_chunks, const int32_t *__restrict__ offsets, int64_t cap_runs, int64_t cap_regions,
         int plane_px, int conn3d, uint32_t key_mul, int32_t *__restrict__ run_start,
         int32_t *__restrict__ run_end,
         int32_t *__restrict__ run_root, uint32_t *__restrict__ reg_root,
         uint32_t *__restrict__ reg_key)
{
    __shared__ unsigned long long s_wave[NT / 64];
    const uint32_t i0 = blockIdx.x * (uint32_t)RUN_CHUNK + (uint32_t)RUN_PXT * threadIdx.x;
    int32_t lab[RUN_PXT];
    unsigned roots;
    const unsigned f = run_flags(dec, L, i0, (uint32_t)n, (uint32_t)W, lab, roots);
    const unsigned long long mine = pack3(f, roots);
    // exclusive prefix of the packed counts over the block's threads
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    unsigned long long base = 0;
#pragma unroll
    for (int j = 0; j < NT / 64; ++j) base += j < w ? s_wave[j] : 0ull;
    const unsigned long long ex = base + inc - mine;
    if (!f) return;
    const int cs = (n_chunks + 3) & ~3;           // the count arrays' stride
    uint32_t ps = (uint32_t)offsets[blockIdx.x] + (uint32_t)(ex & 0x1fffffu);
    uint32_t pe = (uint32_t)offsets[cs + blockIdx.x] + (uint32_t)(ex >> 21 & 0x1fffffu);
    uint32_t pr = (uint32_t)offsets[2 * cs + blockIdx.x] + (uint32_t)(ex >> 42);
#pragma unroll
    for (int k = 0; k < RUN_PXT; ++k) {
        const uint32_t p = i0 + k;
        if (f >> k & 1u) {
            if (ps < cap_runs) {
                run_start[ps] = (int32_t)p;
                run_root[ps] = lab[k];
            }
            ++ps;
            if (roots >> k & 1u) {
                if (pr < cap_regions) {
                    reg_root[pr] = p;
                    const uint32_t grp = conn3d ? 0u : p / (uint32_t)plane_px;
                    reg_key[pr] = grp * key_mul + (uint32_t)dec[p];
                }
                ++pr;
            }
        }
        if (f >> (16 + k) & 1u) {
            if (pe < cap_runs) run_end[pe] = (int32_t)p;
            ++pe;
        }
    }
}

// ---- stable least-significant-digit radix sort -------------------------------
// Keys of `bits` significant bits, digits of up to 10 bits, one pass per digit:
// a histogram of every 4096-item tile, an exclusive scan of the [tile][digit]
// counts down the tiles, and a scatter that ranks a tile's items among the
// items of equal digit in index order (wave w owns items [1024 w, 1024 w + 1024)
// of the tile, taken 64 consecutive items at a time; equal digits inside a step
// are found with one ballot per digit bit).  Sizes here are 2e5 - 2e6 items:
// what matters is the number of passes and launches, not the last GB/s.
constexpr int RS_TILE = 4096;
constexpr int RS_IPT = RS_TILE / NT;          // 16
constexpr int RS_MAX_BINS = 1024;
constexpr int RS_SEGS = 16;                   // row segments of the count scan

__device__ __forceinline__ int64_t rs_count(const unsigned long long *n_dev, int64_t n_host)
{
    if (!n_dev) return n_host;
    const int64_t v = (int64_t)*n_dev;
    return v < n_host ? v : n_host;
}

__global__ void __launch_bounds__(NT)
rs_hist(const uint32_t *__restrict__ keys, const unsigned long long *n_dev, int64_t n_host,
        int shift, int bins, int32_t *__restrict__ cnt)
{
    __shared__ int hist[RS_MAX_BINS];
    const int64_t n = rs_count(n_dev, n_host);
    const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
    if (t0 >= n) return;
    for (int b = threadIdx.x; b < bins; b += NT) hist[b] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const int64_t i = t0 + k * NT + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (uint32_t)(bins - 1)], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += NT) cnt[(size_t)blockIdx.x * bins + b] = hist[b];
}

// exclusive scan of cnt[tile][digit] down the tiles, per digit; totals[digit]
// = items with that digit.  A block owns 64 digits; its 16 waves each take a
// contiguous segment of the tiles.
__global__ void __launch_bounds__(64 * RS_SEGS)
rs_scan(int32_t *__restrict__ cnt, const unsigned long long *n_dev, int64_t n_host, int bins,
        int32_t *__restrict__ totals)
{
    __shared__ int seg_sum[RS_SEGS][64];
    const int64_t n = rs_count(n_dev, n_host);
    const int rows = (int)((n + RS_TILE - 1) / RS_TILE);
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + lane;
    const bool live = bin < bins;
    const int per = (rows + RS_SEGS - 1) / RS_SEGS;
    const int lo = min(rows, seg * per), hi = min(rows, lo + per);
    int sum = 0;
    if (live)
        for (int r = lo; r < hi; ++r) sum += cnt[(size_t)r * bins + bin];
    seg_sum[seg][lane] = sum;
    __syncthreads();
    int run = 0, total = 0;
#pragma unroll
    for (int s = 0; s < RS_SEGS; ++s) {
        const int v = seg_sum[s][lane];
        if (s < seg) run += v;
        total += v;
    }
    if (live) {
        for (int r = lo; r < hi; ++r) {
            const size_t at = (size_t)r * bins + bin;
            const int v = cnt[at];
            cnt[at] = run;
            run += v;
        }
        if (seg == 0) totals[bin] = total;
    }
}

template <typename V, int NBITS>
__global__ void __launch_bounds__(NT)
rs_scatter(const uint32_t *__restrict__ keys_in, const V *__restrict__ vals_in,
           const unsigned long long *n_dev, int64_t n_host, int shift,
           const int32_t *__restrict__ cnt, const int32_t *__restrict__ totals,
           uint32_t *__restrict__ keys_out, V *__restrict__ vals_out)
{
    __shared__ int cw[NT / 64][RS_MAX_BINS];      // per wave: running count, then position
    __shared__ int s_wave[NT / 64];
    const int64_t n = rs_count(n_dev, n_host);
    const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
    if (t0 >= n) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int bins = 1 << NBITS;
    const uint32_t mask = (uint32_t)(bins - 1);
#pragma unroll
    for (int ww = 0; ww < NT / 64; ++ww)
        for (int b = threadIdx.x; b < bins; b += NT) cw[ww][b] = 0;
    __syncthreads();

    uint32_t key[RS_IPT];
    V val[RS_IPT];
    int rank[RS_IPT];
    const int64_t w0 = t0 + (int64_t)w * (RS_TILE / (NT / 64));
    // every load of the tile is issued before the first one is used
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const int64_t i = w0 + k * 64 + lane;
        key[k] = i < n ? keys_in[i] : 0u;
        val[k] = i < n ? vals_in[i] : V(0);
    }
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const int64_t i = w0 + k * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (key[k] >> shift) & mask;
        // lanes of this step with the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < NBITS; ++b) {
            const unsigned long long vote = __ballot((d >> b) & 1u);
            peers &= (d >> b) & 1u ? vote : ~vote;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        int r = 0;
        if (valid) {
            r = cw[w][d] + __popcll(below);
            rank[k] = r;
        } else {
            rank[k] = 0;
        }
        group_sync();                       // every lane has read the count
        if (valid && below == 0ull) cw[w][d] = r + __popcll(peers);   // the group's first lane
        group_sync();
    }
    __syncthreads();
    // digit bases: items of smaller digits, plus this digit's items in earlier
    // tiles, plus this digit's items in the tile's earlier waves
    {
        // exclusive scan of totals over the digits: thread t owns 4 consecutive
        const int per = RS_MAX_BINS / NT;             // 4
        int tv[RS_MAX_BINS / NT];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            const int b = threadIdx.x * per + j;
            tv[j] = b < bins ? totals[b] : 0;
            sum += tv[j];
        }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int j = 0; j < NT / 64; ++j) base += j < w ? s_wave[j] : 0;
        int run = base + inc - sum;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            const int b = threadIdx.x * per + j;
            if (b < bins) {
                int pos = run + cnt[(size_t)blockIdx.x * bins + b];
#pragma unroll
                for (int ww = 0; ww < NT / 64; ++ww) {
                    const int c = cw[ww][b];
                    cw[ww][b] = pos;
                    pos += c;
                }
            }
            run += tv[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_IPT; ++k) {
        const int64_t i = w0 + k * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[k] >> shift) & mask;
            const int64_t dst = (int64_t)cw[w][d] + rank[k];
            keys_out[dst] = key[k];
            vals_out[dst] = val[k];
        }
    }
}

struct RadixPlan {
    int passes, digit_bits[4];
};

inline RadixPlan radix_plan(int bits)
{
    RadixPlan p;
    if (bits < 1) bits = 1;
    p.passes = (bits + 9) / 10;
    const int per = (bits + p.passes - 1) / p.passes;
    int left = bits;
    for (int k = 0; k < p.passes; ++k) {
        p.digit_bits[k] = left < per ? left : per;
        left -= p.digit_bits[k];
    }
    return p;
}

// Sorts (keys, vals) of n items (n = min(*n_dev, n_host) when n_dev is given;
// the grid is sized for n_host).  Returns 0 when the result is in the `a`
// buffers, 1 when it is in the `b` buffers.
template <typename V>
int radix_sort(uint32_t *keys_a, uint32_t *keys_b, V *vals_a, V *vals_b,
               const unsigned long long *n_dev, int64_t n_host, int bits, int32_t *cnt,
               int32_t *totals, hipStream_t s)
{
    const RadixPlan p = radix_plan(bits);
    const unsigned tiles = (unsigned)((n_host + RS_TILE - 1) / RS_TILE);
    if (tiles == 0) return 0;
    int shift = 0, where = 0;
    for (int k = 0; k < p.passes; ++k) {
        const int bins = 1 << p.digit_bits[k];
        const uint32_t *ki = where ? keys_b : keys_a;
        uint32_t *ko = where ? keys_a : keys_b;
        const V *vi = where ? vals_b : vals_a;
        V *vo = where ? vals_a : vals_b;
        hipLaunchKernelGGL(rs_hist, dim3(tiles), dim3(NT), 0, s, ki, n_dev, n_host, shift, bins, cnt);
        hipLaunchKernelGGL(rs_scan, dim3((unsigned)((bins + 63) / 64)), dim3(64 * RS_SEGS), 0, s, cnt,
                           n_dev, n_host, bins, totals);
        switch (p.digit_bits[k]) {
#define MF_RS_CASE(NB)                                                                             \
    case NB:                                                                                       \
        hipLaunchKernelGGL((rs_scatter<V, NB>), dim3(tiles), dim3(NT), 0, s, ki, vi, n_dev, n_host, \
                           shift, cnt, totals, ko, vo);                                            \
        break;
            MF_RS_CASE(1) MF_RS_CASE(2) MF_RS_CASE(3) MF_RS_CASE(4) MF_RS_CASE(5)
            MF_RS_CASE(6) MF_RS_CASE(7) MF_RS_CASE(8) MF_RS_CASE(9) MF_RS_CASE(10)
#undef MF_RS_CASE
        }
        shift += p.digit_bits[k];
        where ^= 1;
    }
    return where;
}

// place of every region in the reference's row order, written at its root
__global__ void __launch_bounds__(NT)
region_ranks(const uint32_t *__restrict__ roots_sorted, int64_t n_reg, int64_t n_pix,
             int32_t *__restrict__ aux, int32_t *__restrict__ reg_start)
{
    const int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (e == 0) reg_start[n_reg] = (int32_t)n_pix;
    if (e < n_reg) aux[roots_sorted[e]] = (int32_t)e;
}

// sort key (the region's place) and payload (first pixel | length << 32) of a run
__global__ void __launch_bounds__(NT)
run_keys(const int32_t *__restrict__ run_start, const int32_t *__restrict__ run_end,
         const int32_t *__restrict__ run_root, const int32_t *__restrict__ aux,
         const unsigned long long *n_dev, int64_t cap, uint32_t *__restrict__ keys,
         unsigned long long *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= rs_count(n_dev, cap)) return;
    const int32_t s = run_start[i];
    keys[i] = (uint32_t)aux[run_root[i]];
    vals[i] = (unsigned long long)(uint32_t)s |
              (unsigned long long)(uint32_t)(run_end[i] - s + 1) << 32;
}

// pixels per 1024-run chunk of the sorted runs
__global__ void __launch_bounds__(NT)
run_len_count(const unsigned long long *__restrict__ vals, const unsigned long long *n_dev,
              int64_t cap, int32_t *__restrict__ counts)
{
    const int64_t n = rs_count(n_dev, cap);
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) sum += (int)(vals[i0 + k] >> 32);
    const int total = block_reduce_sum(sum);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// the sorted runs spelled out as pixels, and where every region's pixels begin
__global__ void __launch_bounds__(NT)
run_expand(const uint32_t *__restrict__ keys, const unsigned long long *__restrict__ vals,
           const unsigned long long *n_dev, int64_t cap, const int32_t *__restrict__ offsets,
           int64_t n_pix, int32_t *__restrict__ pix, int32_t *__restrict__ reg_start)
{
    const int64_t n = rs_count(n_dev, cap);
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    unsigned long long v[4];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = i0 + k < n ? vals[i0 + k] : 0ull;
        sum += (int)(v[k] >> 32);
    }
    int64_t pos = (int64_t)offsets[blockIdx.x] + block_exclusive_scan(sum);
    uint32_t prev = i0 > 0 && i0 < n ? keys[i0 - 1] : 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k >= n) break;
        const uint32_t key = keys[i0 + k];
        if (i0 + k == 0 || key != prev) reg_start[key] = (int32_t)pos;
        prev = key;
        const int32_t first = (int32_t)(uint32_t)v[k];
        const int len = (int)(v[k] >> 32);
        for (int t = 0; t < len; ++t)
            if (pos + t < n_pix) pix[pos + t] = first + t;
        pos += len;
    }
}

// workspace carve-up shared by the size query and the call
struct Plan {
    size_t run_start, run_end, run_root, keys_a, keys_b, vals_a, vals_b, reg_root_a, reg_root_b,
        reg_key_a, reg_key_b, reg_start, aux, chunks, totals, sort_cnt, sort_totals, total;
    int vol_chunks, run_chunks;
    int region_key_bits, rank_bits, key_mul;
};

inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

int make_plan(int Z, int H, int W, int64_t n_pix, int64_t n_reg, Plan &pl, int n_codewords = 0,
              int conn3d = 0)
{
    const int64_t vol = (int64_t)Z * H * W;
    // region key = plane * key_mul + barcode (one group in 3-D mode); with the
    // codebook's size known the keys are dense: 7 planes x 140 codewords sort
    // in ONE 10-bit pass instead of two
    pl.key_mul = n_codewords > 0 ? n_codewords : 32768;
    const int64_t n_keys = (int64_t)(conn3d ? 1 : Z) * pl.key_mul;
    int kb = 1;
    while (((int64_t)1 << kb) < n_keys) ++kb;
    pl.region_key_bits = kb;
    if (pl.region_key_bits > 30)
        return mf::fail(MF_EUNSUPPORTED, "region key needs %d bits", pl.region_key_bits);
    int rb = 1;
    while (((int64_t)1 << rb) < n_reg) ++rb;
    pl.rank_bits = rb;

    // a run has at least one pixel: n_pix bounds the number of runs
    const size_t np = (size_t)(n_pix > 0 ? n_pix : 1), nr = (size_t)(n_reg > 0 ? n_reg : 1);
    pl.vol_chunks = (int)((vol + RUN_CHUNK - 1) / RUN_CHUNK);
    pl.run_chunks = (int)((np + CHUNK - 1) / CHUNK);
    const size_t tiles = (np + RS_TILE - 1) / RS_TILE;

    size_t off = 0;
    pl.run_start = off; off += align_up(np * 4);      // later: the pixel list
    pl.run_end = off; off += align_up(np * 4);
    pl.run_root = off; off += align_up(np * 4);
    pl.keys_a = off; off += align_up(np * 4);
    pl.keys_b = off; off += align_up(np * 4);
    pl.vals_a = off; off += align_up(np * 8);
    pl.vals_b = off; off += align_up(np * 8);
    pl.reg_root_a = off; off += align_up(nr * 4);
    pl.reg_root_b = off; off += align_up(nr * 4);
    pl.reg_key_a = off; off += align_up(nr * 4);
    pl.reg_key_b = off; off += align_up(nr * 4);
    pl.reg_start = off; off += align_up((nr + 1) * 4);
    pl.aux = off; off += align_up((size_t)vol * 4);
    const size_t n_chunk_ints = (size_t)3 * (pl.vol_chunks + 4) > (size_t)pl.run_chunks
                                    ? (size_t)3 * (pl.vol_chunks + 4) : (size_t)pl.run_chunks;
    pl.chunks = off; off += align_up(n_chunk_ints * 4);
    pl.totals = off; off += align_up(4 * 8);
    pl.sort_cnt = off; off += align_up(tiles * RS_MAX_BINS * 4);
    pl.sort_totals = off; off += align_up(RS_MAX_BINS * 4);
    pl.total = off;
    return MF_OK;
}

// ------------------------------------------------- refactor sums (K5) ---
// extract_refactors / _extract_backgrounds (decoding.py:321-334, 355-370) reduce
// the regions of every barcode to one row of per-bit sums.  One workgroup per
// barcode walks that barcode's rows of the region table (grouped by plane, then
// barcode, in label order: the reference's loop order), eight rows at a time,
// 32 lanes = the bits of one row; float64 throughout like the reference.
struct RefactorArgs {
    const double *rec;
    const float *stat;        // mean_tm (refactors) or min_tm (backgrounds)
    int64_t n_reg;
    int B, N, plane, Z, normalize;
    double area_thr;
    double offset[MF_MAX_BITS];
    double *sums, *seen;
};

// first row whose (plane, barcode) key is >= (p, b)
__device__ __forceinline__ int64_t refactor_lower_bound(const double *rec, int64_t n, double p,
                                                        double b)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const double rp = rec[mid * MF_REC_COLS + 1], rb = rec[mid * MF_REC_COLS + 0];
        if (rp < p || (rp == p && rb < b)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(NT)
refactor_sums(const RefactorArgs a)
{
    constexpr int GROUPS = NT / 32;
    __shared__ double s_part[GROUPS][32];
    __shared__ int s_cnt[GROUPS];
    __shared__ int64_t s_range[2];
    const int code = blockIdx.x;
    const int bit = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const bool has = bit < a.B;
    const double off = has ? a.offset[bit] : 0.0;
    const int p_lo = a.plane < 0 ? 0 : a.plane, p_hi = a.plane < 0 ? a.Z : a.plane + 1;

    // pass 1: how many of the barcode's regions are large enough
    int mine = 0;
    for (int p = p_lo; p < p_hi; ++p) {
        if (threadIdx.x == 0) {
            s_range[0] = refactor_lower_bound(a.rec, a.n_reg, (double)p, (double)code);
            s_range[1] = refactor_lower_bound(a.rec, a.n_reg, (double)p, (double)code + 1.0);
        }
        __syncthreads();
        const int64_t lo = s_range[0], hi = s_range[1];
        for (int64_t r = lo + threadIdx.x; r < hi; r += NT)
            mine += a.rec[r * MF_REC_COLS + 2] >= a.area_thr;
        __syncthreads();
    }
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 32);
    if (bit == 0) s_cnt[grp] = mine;
    __syncthreads();
    int seen = 0;
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) seen += s_cnt[g];
    __syncthreads();

    // pass 2: the rows in order, group g taking every GROUPS-th one
    double acc = 0.0;
    for (int p = p_lo; p < p_hi; ++p) {
        if (threadIdx.x == 0) {
            s_range[0] = refactor_lower_bound(a.rec, a.n_reg, (double)p, (double)code);
            s_range[1] = refactor_lower_bound(a.rec, a.n_reg, (double)p, (double)code + 1.0);
        }
        __syncthreads();
        const int64_t lo = s_range[0], hi = s_range[1];
        for (int64_t r = lo + grp; r < hi; r += GROUPS) {
            if (!(a.rec[r * MF_REC_COLS + 2] >= a.area_thr)) continue;     // uniform per group
            double v = has ? (double)a.stat[r * a.B + bit] - off : 0.0;
            if (a.normalize) {
                // np.linalg.norm of the row, then the row's share of the mean
                double sq = v * v;
#pragma unroll
                for (int d = 16; d > 0; d >>= 1) sq += __shfl_xor(sq, d, 32);
                v = v / sqrt(sq) / (double)seen;
            }
            acc += v;
        }
        __syncthreads();
    }
    s_part[grp][bit] = acc;
    __syncthreads();
    if (grp == 0 && has) {
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) t += s_part[g][bit];
        a.sums[(int64_t)code * a.B + bit] = t;
    }
    if (threadIdx.x == 0) a.seen[code] = (double)seen;
}

}  // namespace

extern "C" int mf_refactor_sums(const double *rec, const float *stat, int64_t n_regions, int B, int N,
                                int Z, int plane, double area_threshold, const double *h_offset,
                                int normalize, double *sums, double *seen, void *stream)
{
    if (!sums || !seen || (n_regions > 0 && (!rec || !stat)))
        return mf::fail(MF_EINVAL, "null pointer argument");
    if (B < 1 || B > MF_MAX_BITS || N < 1 || N > 32767 || Z < 1 || plane >= Z || n_regions < 0)
        return mf::fail(MF_EINVAL, "bad sizes B=%d N=%d Z=%d plane=%d", B, N, Z, plane);
    RefactorArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rec = rec; a.stat = stat; a.n_reg = n_regions;
    a.B = B; a.N = N; a.plane = plane; a.Z = Z; a.normalize = normalize != 0;
    a.area_thr = area_threshold;
    for (int b = 0; b < B; ++b) a.offset[b] = h_offset ? h_offset[b] : 0.0;
    a.sums = sums; a.seen = seen;
    hipLaunchKernelGGL(refactor_sums, dim3((unsigned)N), dim3(NT), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

namespace {

// ------------------------------------------------------------ frame rows ---

struct KeepRow {
    const double *rec;
    double lo, x_hi, y_hi, min_area;
    __device__ bool operator()(int64_t r) const
    {
        const double *row = rec + r * MF_REC_COLS;
        const double x = row[9], y = row[8];
        return x > lo && x < x_hi && y > lo && y < y_hi && row[2] >= min_area;
    }
};

// Order-preserving selection of the kept rows with the same count / scan /
// emit compaction the pixel lists use (no device-wide look-back chain).
__device__ __forceinline__ unsigned keep_flags(const KeepRow &keep, int64_t i0, int64_t n)
{
    unsigned f = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n && keep(i0 + k)) f |= 1u << k;
    return f;
}

__global__ void __launch_bounds__(NT)
keep_count(const KeepRow keep, int64_t n, int32_t *__restrict__ counts)
{
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    const int total = block_reduce_sum(__popc(keep_flags(keep, i0, n)));
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(NT)
keep_emit(const KeepRow keep, int64_t n, const int32_t *__restrict__ offsets,
          int32_t *__restrict__ sel)
{
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    const unsigned f = keep_flags(keep, i0, n);
    int pos = offsets[blockIdx.x] + block_exclusive_scan(__popc(f));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (f >> k & 1u) sel[pos++] = (int32_t)(i0 + k);
}

// Position of a row inside its (z plane, barcode) group -- the DataFrame
// index the reference ends up with, since it builds one frame per plane and
// barcode (decoding.py:246-252, decode.py:225-230).  Rows are sorted by that
// key: a binary search for the group's first row.  2-D key = (column 1,
// column 0), 3-D key = column 0.
__device__ __forceinline__ int64_t group_index(const double *__restrict__ rec, int64_t r,
                                               int conn3d)
{
    const double kb = rec[r * MF_REC_COLS], kz = conn3d ? 0.0 : rec[r * MF_REC_COLS + 1];
    int64_t lo = 0, hi = r;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        const double mb = rec[mid * MF_REC_COLS], mz = conn3d ? 0.0 : rec[mid * MF_REC_COLS + 1];
        if (mz < kz || (mz == kz && mb < kb)) lo = mid + 1; else hi = mid;
    }
    return r - lo;
}

// column-major gather of the kept rows; `count` was written by the scan.
constexpr int GATHER_NT = 1024;

__global__ void __launch_bounds__(GATHER_NT)
frame_gather(const double *__restrict__ rec, const float *__restrict__ inten,
             const int32_t *__restrict__ sel,
             const unsigned long long *__restrict__ count,
             int64_t n_reg, int B, int conn3d, double *__restrict__ out64,
             float *__restrict__ out32)
{
    const int64_t m = (int64_t)*count;
    const int zcol = conn3d ? 7 : 1;
    const int cols[MF_FRAME_COLS - 1] = {0, 3, 4, 2, 5, 6, 9, 8, zcol};
    for (int64_t i = (int64_t)blockIdx.x * GATHER_NT + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * GATHER_NT) {
        const int64_t r = sel[i];
        const double *row = rec + r * MF_REC_COLS;
#pragma unroll
        for (int c = 0; c < MF_FRAME_COLS - 1; ++c) out64[(int64_t)c * n_reg + i] = row[cols[c]];
        out64[(int64_t)(MF_FRAME_COLS - 1) * n_reg + i] = (double)group_index(rec, r, conn3d);
        for (int b = 0; b < B; ++b) out32[(int64_t)b * n_reg + i] = inten[r * B + b];
    }
}

// The same rows in the barcode database's column types
// (merlin/util/barcodedb.py:46-65, the astype of write_barcodes :264-266):
// uint16 barcode_id / area, float32 statistics and coordinates, int32 index.
// float64 -> float32 rounds to nearest like numpy's astype; area wraps modulo
// 2^16 like the C cast numpy makes.
struct TypedOut {
    uint16_t *u16;      // [2][n]  barcode_id, area
    float *f32;         // [7][n]  mean_intensity, max_intensity, mean_distance,
                        //         min_distance, x, y, z
    int32_t *index;     // [n]     position inside the (plane, barcode) frame
    double *xy64;       // [2][n]  x, y in float64 (for the global aligner) or null
};

__global__ void __launch_bounds__(GATHER_NT)
frame_gather_typed(const double *__restrict__ rec, const float *__restrict__ inten,
                   const int32_t *__restrict__ sel,
                   const unsigned long long *__restrict__ count, int64_t n_reg, int B,
                   int conn3d, TypedOut o, float *__restrict__ out32)
{
    const int64_t m = (int64_t)*count;
    const int zcol = conn3d ? 7 : 1;
    const int cols[7] = {3, 4, 5, 6, 9, 8, zcol};
    for (int64_t i = (int64_t)blockIdx.x * GATHER_NT + threadIdx.x; i < m;
         i += (int64_t)gridDim.x * GATHER_NT) {
        const int64_t r = sel[i];
        const double *row = rec + r * MF_REC_COLS;
        o.u16[i] = (uint16_t)(int64_t)row[0];
        o.u16[n_reg + i] = (uint16_t)(int64_t)row[2];
#pragma unroll
        for (int c = 0; c < 7; ++c) o.f32[(int64_t)c * n_reg + i] = (float)row[cols[c]];
        o.index[i] = (int32_t)group_index(rec, r, conn3d);
        if (o.xy64) {
            o.xy64[i] = row[9];
            o.xy64[n_reg + i] = row[8];
        }
        for (int b = 0; b < B; ++b) out32[(int64_t)b * n_reg + i] = inten[r * B + b];
    }
}

// device-side address of a buffer the kernels may write: device memory or
// page-locked host memory that is mapped into the device's address space
template <typename T>
int device_view(T *p, T **out, const char *what)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess || !at.devicePointer) {
        (void)hipGetLastError();
        return mf::fail(MF_EINVAL, "%s is neither device memory nor mapped pinned host memory", what);
    }
    *out = reinterpret_cast<T *>(at.devicePointer);
    return MF_OK;
}

// scratch of mf_region_frame: the kept rows' indexes and the chunk counts
inline size_t frame_scratch_bytes(int64_t n_regions)
{
    const size_t nr = (size_t)(n_regions > 0 ? n_regions : 1);
    return align_up(nr * 4) + align_up(((nr + CHUNK - 1) / CHUNK) * 4);
}

// shared by mf_region_frame (scratch = the caller's workspace) and
// mf_region_frame_static (scratch = this GPU's copy of the static array)
int region_frame_run(const double *rec, const float *inten, int64_t n_regions, int B, int conn3d,
                     double crop_width, double crop_shape0, double crop_shape1,
                     double minimum_area, double *out64, float *out32, int64_t *count,
                     unsigned char *scratch, hipStream_t s, const TypedOut *typed = nullptr)
{
    int64_t *d_count = nullptr;
    int rc = device_view(count, &d_count, "count");
    if (rc) return rc;
    if (n_regions == 0) {
        MF_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(int64_t), s));
        return MF_OK;
    }
    if (!rec || !inten || (!out64 && !typed) || !out32)
        return mf::fail(MF_EINVAL, "null pointer argument");
    if (B < 1) return mf::fail(MF_EINVAL, "bad sizes");
    double *d64 = nullptr;
    float *d32 = nullptr;
    TypedOut t{nullptr, nullptr, nullptr, nullptr};
    if ((rc = device_view(out32, &d32, "out32"))) return rc;
    if (typed) {
        if (!typed->u16 || !typed->f32 || !typed->index)
            return mf::fail(MF_EINVAL, "null pointer argument");
        if ((rc = device_view(typed->u16, &t.u16, "out_u16")) ||
            (rc = device_view(typed->f32, &t.f32, "out_f32")) ||
            (rc = device_view(typed->index, &t.index, "out_index")))
            return rc;
        if (typed->xy64 && (rc = device_view(typed->xy64, &t.xy64, "out_xy64"))) return rc;
    } else if ((rc = device_view(out64, &d64, "out64"))) {
        return rc;
    }
    int32_t *sel = reinterpret_cast<int32_t *>(scratch);
    int32_t *chunks = reinterpret_cast<int32_t *>(scratch + align_up((size_t)n_regions * 4));
    const int n_chunks = (int)((n_regions + CHUNK - 1) / CHUNK);
    // exclusive crop on the centroid with the reference's quirk: x is compared
    // with shape[0] and y with shape[1] (decoding.py:244)
    const KeepRow keep{rec, crop_width, crop_shape0 - crop_width, crop_shape1 - crop_width,
                       minimum_area};
    hipLaunchKernelGGL(keep_count, dim3((unsigned)n_chunks), dim3(NT), 0, s, keep, n_regions, chunks);
    hipLaunchKernelGGL(chunk_scan, dim3(1), dim3(SCAN_NT), 0, s, chunks, n_chunks,
                       reinterpret_cast<unsigned long long *>(d_count));
    hipLaunchKernelGGL(keep_emit, dim3((unsigned)n_chunks), dim3(NT), 0, s, keep, n_regions, chunks,
                       sel);
    // rows written over PCIe into mapped host memory: few workgroups, so the
    // decode kernel of the next FOV keeps its CUs; into device memory: a full grid
    unsigned g = (unsigned)((n_regions + GATHER_NT - 1) / GATHER_NT);
    hipPointerAttribute_t at;
    const bool to_device = hipPointerGetAttributes(&at, out32) == hipSuccess &&
                           at.type == hipMemoryTypeDevice;
    const unsigned g_max = to_device ? 512u : 4u;
    if (g > g_max) g = g_max;
    if (typed)
        hipLaunchKernelGGL(frame_gather_typed, dim3(g), dim3(GATHER_NT), 0, s, rec, inten, sel,
                           reinterpret_cast<const unsigned long long *>(d_count), n_regions, B,
                           conn3d, t, d32);
    else
        hipLaunchKernelGGL(frame_gather, dim3(g), dim3(GATHER_NT), 0, s, rec, inten, sel,
                           reinterpret_cast<const unsigned long long *>(d_count), n_regions, B,
                           conn3d, d64, d32);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

}  // namespace

extern "C" int64_t mf_region_frame_workspace_bytes(int64_t n_regions)
{
    if (n_regions < 0 || n_regions > INT32_MAX) return mf::fail(MF_EINVAL, "bad region count");
    return (int64_t)frame_scratch_bytes(n_regions);
}

extern "C" int mf_region_frame(const double *rec, const float *inten, int64_t n_regions, int B,
                               int conn3d, double crop_width, double crop_shape0,
                               double crop_shape1, double minimum_area, double *out64,
                               float *out32, int64_t *count, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    if (!count) return mf::fail(MF_EINVAL, "null pointer argument");
    if (n_regions < 0 || n_regions > INT32_MAX) return mf::fail(MF_EINVAL, "bad region count");
    if (n_regions > 0) {
        if (!workspace) return mf::fail(MF_EINVAL, "null pointer argument");
        const int64_t need = (int64_t)frame_scratch_bytes(n_regions);
        if (workspace_bytes < need)
            return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes",
                            (long long)workspace_bytes, (long long)need);
    }
    return region_frame_run(rec, inten, n_regions, B, conn3d, crop_width, crop_shape0, crop_shape1,
                            minimum_area, out64, out32, count,
                            reinterpret_cast<unsigned char *>(workspace),
                            reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mf_region_frame_typed(const double *rec, const float *inten, int64_t n_regions,
                                     int B, int conn3d, double crop_width, double crop_shape0,
                                     double crop_shape1, double minimum_area, uint16_t *out_u16,
                                     float *out_f32, int32_t *out_index, float *out_inten,
                                     double *out_xy64, int64_t *count, void *workspace,
                                     int64_t workspace_bytes, void *stream)
{
    if (!count) return mf::fail(MF_EINVAL, "null pointer argument");
    if (n_regions < 0 || n_regions > INT32_MAX) return mf::fail(MF_EINVAL, "bad region count");
    if (n_regions > 0) {
        if (!workspace) return mf::fail(MF_EINVAL, "null pointer argument");
        const int64_t need = (int64_t)frame_scratch_bytes(n_regions);
        if (workspace_bytes < need)
            return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes",
                            (long long)workspace_bytes, (long long)need);
    }
    const TypedOut typed{out_u16, out_f32, out_index, out_xy64};
    return region_frame_run(rec, inten, n_regions, B, conn3d, crop_width, crop_shape0, crop_shape1,
                            minimum_area, nullptr, out_inten, count,
                            reinterpret_cast<unsigned char *>(workspace),
                            reinterpret_cast<hipStream_t>(stream), &typed);
}

// ---- allocation-free variant (no counterpart in the reference) -------------
// Scratch of mf_region_frame_static: one copy per GPU, held until the process
// ends.  `scratch` is a host-side stand-in for the array, never a device
// address: host code only uses the per-device pointer hipGetSymbolAddress
// returns.
constexpr size_t FRAME_STATIC_BYTES =
    (((size_t)MF_FRAME_STATIC_MAX_ROWS * 4 + 255) & ~size_t(255)) +
    ((((size_t)MF_FRAME_STATIC_MAX_ROWS + CHUNK - 1) / CHUNK * 4 + 255) & ~size_t(255));
static __device__ __align__(256) unsigned char scratch[FRAME_STATIC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
unsigned char *g_scratch_ptr[MAX_DEVICES];     // this process's address of each GPU's copy
std::mutex g_scratch_mutex;

// Address of the current device's copy of `scratch`, resolved on the first
// call on that device.
int static_scratch(unsigned char **out, int *device)
{
    int dev = -1;
    MF_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES)
        return mf::fail(MF_EUNSUPPORTED, "device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        MF_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return mf::fail(MF_EHIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = reinterpret_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[dev];
    *device = dev;
    return MF_OK;
}
}  // namespace

extern "C" int mf_region_frame_static(const double *rec, const float *inten, int64_t n_regions,
                                      int B, int conn3d, double crop_width, double crop_shape0,
                                      double crop_shape1, double minimum_area, double *out64,
                                      float *out32, int64_t *count, void *stream)
{
    if (!count) return mf::fail(MF_EINVAL, "null pointer argument");
    if (n_regions < 0) return mf::fail(MF_EINVAL, "bad region count");
    if (n_regions > MF_FRAME_STATIC_MAX_ROWS)
        return mf::fail(MF_EWORKSPACE,
                        "mf_region_frame_static holds at most %d rows, got %lld: use "
                        "mf_region_frame with a workspace",
                        MF_FRAME_STATIC_MAX_ROWS, (long long)n_regions);
    unsigned char *ptr = nullptr;
    int dev = -1;
    int rc = static_scratch(&ptr, &dev);
    if (rc) return rc;
    if (rec) {
        // the scratch used is the CURRENT device's: the table must live there
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, rec) != hipSuccess) {
            (void)hipGetLastError();
            return mf::fail(MF_EINVAL, "rec is not device memory");
        }
        if (at.type == hipMemoryTypeDevice && at.device != dev)
            return mf::fail(MF_EINVAL, "rec lives on device %d but device %d is current", at.device,
                            dev);
    }
    return region_frame_run(rec, inten, n_regions, B, conn3d, crop_width, crop_shape0, crop_shape1,
                            minimum_area, out64, out32, count, ptr,
                            reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mf_label_components(const int16_t *decoded, int Z, int H, int W, int conn3d,
                                   int32_t *labels, int64_t *counts, void *stream)
{
    if (!decoded || !labels || !counts) return mf::fail(MF_EINVAL, "null pointer argument");
    if (Z < 1 || H < 1 || W < 1) return mf::fail(MF_EINVAL, "bad shape Z=%d H=%d W=%d", Z, H, W);
    const int64_t vol = (int64_t)Z * H * W;
    if (vol > INT32_MAX) return mf::fail(MF_EUNSUPPORTED, "volume exceeds 2^31 pixels");
    if (reinterpret_cast<uintptr_t>(labels) & 15)
        return mf::fail(MF_EINVAL, "labels must be 16-byte aligned");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n = (int)vol;
    const unsigned grid = (unsigned)((n + NT - 1) / NT);
    if (H > 65535 || Z > 65535) return mf::fail(MF_EUNSUPPORTED, "H and Z must be <= 65535");
    MF_HIP_CHECK(hipMemsetAsync(counts, 0, MF_LABEL_COUNTS_LEN * sizeof(int64_t), s));
    const int tiles_x = (W + CT_W - 1) / CT_W, tiles_y = (H + CT_H - 1) / CT_H;
    hipLaunchKernelGGL(ccl_tile, dim3((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)Z), dim3(NT), 0,
                       s, decoded, labels, H, W);
    {
        const int n_edge_rows = tiles_y - 1, n_edge_cols = 2 * tiles_x;
        const int64_t items = (int64_t)n_edge_rows * W + (int64_t)n_edge_cols * H;
        hipLaunchKernelGGL(ccl_border, dim3((unsigned)((items + NT - 1) / NT), (unsigned)Z), dim3(NT), 0, s,
                           decoded, labels, H, W, n_edge_rows, n_edge_cols);
    }
    if (conn3d && Z > 1)
        hipLaunchKernelGGL(ccl_merge_z,
                           dim3((unsigned)((W + NT - 1) / NT), (unsigned)H, (unsigned)(Z - 1)),
                           dim3(NT), 0, s, decoded, labels, H, W);
#ifndef MF_FLATTEN_BLOCKS
#define MF_FLATTEN_BLOCKS 8192u
#endif
    const unsigned fgrid = grid < MF_FLATTEN_BLOCKS ? grid : MF_FLATTEN_BLOCKS;
    hipLaunchKernelGGL(ccl_flatten, dim3(fgrid), dim3(NT), 0, s, labels, n,
                       reinterpret_cast<unsigned long long *>(counts));
    hipLaunchKernelGGL(ccl_counts_reduce, dim3(1), dim3(64), 0, s,
                       reinterpret_cast<unsigned long long *>(counts));
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

extern "C" int64_t mf_region_workspace_bytes(int Z, int H, int W, int64_t n_pixels,
                                             int64_t n_regions)
{
    Plan pl;
    const int rc = make_plan(Z, H, W, n_pixels, n_regions, pl);
    if (rc) return rc;
    return (int64_t)pl.total;
}

extern "C" int mf_region_table(const int16_t *decoded, const float *magnitudes,
                               const void *distances, int dist_dtype, const float *traces,
                               int64_t tr_z_stride, int64_t tr_bit_stride,
                               int64_t tr_pix_stride, const int32_t *labels, int B,
                               int n_codewords, int Z, int H, int W, int conn3d,
                               int64_t n_pixels, int64_t n_regions, double *rec, float *inten,
                               float *mean_tm, float *min_tm, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    if (n_regions == 0 || n_pixels == 0) return MF_OK;
    if (!decoded || !magnitudes || !distances || !traces || !labels || !rec || !inten || !workspace)
        return mf::fail(MF_EINVAL, "null pointer argument");
    if (B < 1 || Z < 1 || H < 1 || W < 1 || n_pixels < n_regions || n_regions < 0)
        return mf::fail(MF_EINVAL, "bad sizes");
    if (n_codewords < 0 || n_codewords > 32768)
        return mf::fail(MF_EINVAL, "codeword count %d outside 0..32768", n_codewords);
    if (dist_dtype != MF_DTYPE_F32 && dist_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "distance dtype %d not f32/f64", dist_dtype);
    const int64_t vol = (int64_t)Z * H * W;
    if (vol > INT32_MAX || n_pixels > vol) return mf::fail(MF_EINVAL, "bad pixel count");
    if (reinterpret_cast<uintptr_t>(decoded) & 15)
        return mf::fail(MF_EINVAL, "decoded must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(labels) & 15)
        return mf::fail(MF_EINVAL, "labels must be 16-byte aligned");
    Plan pl;
    int rc = make_plan(Z, H, W, n_pixels, n_regions, pl, n_codewords, conn3d);
    if (rc) return rc;
    if ((int64_t)pl.total > workspace_bytes)
        return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes", (long long)workspace_bytes,
                        (long long)pl.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    int32_t *run_start = reinterpret_cast<int32_t *>(ws + pl.run_start);
    int32_t *run_end = reinterpret_cast<int32_t *>(ws + pl.run_end);
    int32_t *run_root = reinterpret_cast<int32_t *>(ws + pl.run_root);
    uint32_t *keys_a = reinterpret_cast<uint32_t *>(ws + pl.keys_a);
    uint32_t *keys_b = reinterpret_cast<uint32_t *>(ws + pl.keys_b);
    unsigned long long *vals_a = reinterpret_cast<unsigned long long *>(ws + pl.vals_a);
    unsigned long long *vals_b = reinterpret_cast<unsigned long long *>(ws + pl.vals_b);
    uint32_t *reg_root_a = reinterpret_cast<uint32_t *>(ws + pl.reg_root_a);
    uint32_t *reg_root_b = reinterpret_cast<uint32_t *>(ws + pl.reg_root_b);
    uint32_t *reg_key_a = reinterpret_cast<uint32_t *>(ws + pl.reg_key_a);
    uint32_t *reg_key_b = reinterpret_cast<uint32_t *>(ws + pl.reg_key_b);
    int32_t *reg_start = reinterpret_cast<int32_t *>(ws + pl.reg_start);
    int32_t *aux = reinterpret_cast<int32_t *>(ws + pl.aux);
    int32_t *chunks = reinterpret_cast<int32_t *>(ws + pl.chunks);
    unsigned long long *totals = reinterpret_cast<unsigned long long *>(ws + pl.totals);
    int32_t *sort_cnt = reinterpret_cast<int32_t *>(ws + pl.sort_cnt);
    int32_t *sort_totals = reinterpret_cast<int32_t *>(ws + pl.sort_totals);
    const unsigned long long *n_runs = totals;        // device-side count of the runs

    // 1. horizontal runs in raster order (first pixel, last pixel, region root)
    //    and the regions' roots in label order with their (plane, barcode) key
    hipLaunchKernelGGL(run_count, dim3(pl.vol_chunks), dim3(NT), 0, s, decoded, labels, vol, W,
                       pl.vol_chunks, chunks);
    hipLaunchKernelGGL(chunk_scan, dim3(3), dim3(SCAN_NT), 0, s, chunks, pl.vol_chunks, totals,
                       (pl.vol_chunks + 3) & ~3);
    hipLaunchKernelGGL(run_emit, dim3(pl.vol_chunks), dim3(NT), 0, s, decoded, labels, vol, W,
                       pl.vol_chunks, chunks, n_pixels, n_regions, H * W, conn3d, (uint32_t)pl.key_mul,
                       run_start, run_end,
                       run_root, reg_root_a, reg_key_a);
    // 2. the reference's row order of the regions: (plane, barcode), label order
    //    inside (the sort is stable); every region's place goes to its root
    const int rw = radix_sort<uint32_t>(reg_key_a, reg_key_b, reg_root_a, reg_root_b, nullptr,
                                        n_regions, pl.region_key_bits, sort_cnt, sort_totals, s);
    hipLaunchKernelGGL(region_ranks, dim3((unsigned)((n_regions + NT - 1) / NT)), dim3(NT), 0, s,
                       rw ? reg_root_b : reg_root_a, n_regions, n_pixels, aux, reg_start);
    // 3. runs keyed by their region's place, sorted (a region's runs stay in
    //    raster order), spelled out as pixels
    hipLaunchKernelGGL(run_keys, dim3((unsigned)((n_pixels + NT - 1) / NT)), dim3(NT), 0, s, run_start,
                       run_end, run_root, aux, n_runs, n_pixels, keys_a, vals_a);
    const int pw = radix_sort<unsigned long long>(keys_a, keys_b, vals_a, vals_b, n_runs, n_pixels,
                                                  pl.rank_bits, sort_cnt, sort_totals, s);
    const uint32_t *skeys = pw ? keys_b : keys_a;
    const unsigned long long *svals = pw ? vals_b : vals_a;
    int32_t *pix = run_start;                         // the runs' own arrays are free now
    hipLaunchKernelGGL(run_len_count, dim3(pl.run_chunks), dim3(NT), 0, s, svals, n_runs, n_pixels,
                       chunks);
    hipLaunchKernelGGL(chunk_scan, dim3(1), dim3(SCAN_NT), 0, s, chunks, pl.run_chunks,
                       (unsigned long long *)nullptr);
    hipLaunchKernelGGL(run_expand, dim3(pl.run_chunks), dim3(NT), 0, s, skeys, svals, n_runs, n_pixels,
                       chunks, n_pixels, pix, reg_start);
    // 5. per-region rows
    StatsArgs a;
    a.pix = pix; a.start = reg_start; a.n_pix = n_pixels; a.n_reg = n_regions;
    a.dec = decoded; a.mag = magnitudes; a.dist = distances;
    a.dist_f64 = dist_dtype == MF_DTYPE_F64;
    a.traces = traces; a.tr_z = tr_z_stride; a.tr_b = tr_bit_stride; a.tr_p = tr_pix_stride;
    a.B = B; a.Z = Z; a.H = H; a.W = W; a.conn3d = conn3d;
    a.rec = rec; a.inten = inten; a.mean_tm = mean_tm; a.min_tm = min_tm;
    const unsigned gr = (unsigned)((n_regions * RSUB + NT - 1) / NT);
    const bool tm = mean_tm || min_tm;
    if (B > RSUB) {
        if (tm) hipLaunchKernelGGL((region_stats<true, true>), dim3(gr), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((region_stats<true, false>), dim3(gr), dim3(NT), 0, s, a);
    } else {
        if (tm) hipLaunchKernelGGL((region_stats<false, true>), dim3(gr), dim3(NT), 0, s, a);
        else hipLaunchKernelGGL((region_stats<false, false>), dim3(gr), dim3(NT), 0, s, a);
    }
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}
