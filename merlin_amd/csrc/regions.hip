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
#include <rocprim/device/device_radix_sort.hpp>
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

__global__ void __launch_bounds__(NT)
ccl_tile(const int16_t *__restrict__ dec, int32_t *__restrict__ L, int H, int W)
{
    __shared__ __align__(16) int16_t s_code[CT_H][CT_W];
    __shared__ int s_lab[CT_H * CT_W];
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
#pragma unroll
    for (int r = 0; r < CT_ROWS; ++r) {
        const int ly = wave * CT_ROWS + r;
        if (code[r] < 0 || ly == 0) continue;
        const int i = ly * CT_W + lane;
        const int c = code[r];
        const bool n = s_code[ly - 1][lane] == c;
        const bool nw = lane > 0 && s_code[ly - 1][lane - 1] == c;
        const bool ne = lane < CT_W - 1 && s_code[ly - 1][lane + 1] == c;
        if (n) {
            if (!(cont[r] && nw)) lds_unite(s_lab, i, i - CT_W);
        } else {
            if (nw && !cont[r]) lds_unite(s_lab, i, i - CT_W - 1);
            if (ne) lds_unite(s_lab, i, i - CT_W + 1);
        }
    }
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

// flags of the four items a thread owns: bit k set iff item i0 + k is kept
__device__ __forceinline__ unsigned assigned_flags(const int16_t *dec, int64_t i0, int64_t n)
{
    unsigned f = 0;
    if (i0 + 3 < n) {
        // i0 is a multiple of 4: one aligned 8-byte load
        const ushort4 v = *reinterpret_cast<const ushort4 *>(dec + i0);
        f = (unsigned)(!(v.x >> 15)) | (unsigned)(!(v.y >> 15)) << 1 |
            (unsigned)(!(v.z >> 15)) << 2 | (unsigned)(!(v.w >> 15)) << 3;
    } else {
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n && dec[i0 + k] >= 0) f |= 1u << k;
    }
    return f;
}

__device__ __forceinline__ unsigned boundary_flags(const uint32_t *keys, int64_t i0, int64_t n)
{
    unsigned f = 0;
    uint32_t prev = i0 > 0 && i0 < n ? keys[i0 - 1] : 0u;
    for (int k = 0; k < 4; ++k) {
        if (i0 + k >= n) break;
        const uint32_t cur = keys[i0 + k];
        if (i0 + k == 0 || cur != prev) f |= 1u << k;
        prev = cur;
    }
    return f;
}

template <bool BOUNDARY>
__global__ void __launch_bounds__(NT)
chunk_count(const int16_t *__restrict__ dec, const uint32_t *__restrict__ keys, int64_t n,
            int32_t *__restrict__ counts)
{
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    const unsigned f = BOUNDARY ? boundary_flags(keys, i0, n) : assigned_flags(dec, i0, n);
    const int total = block_reduce_sum(__popc(f));
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

// in-place exclusive scan of the chunk counts by ONE block; thread t owns a
// contiguous slice of `per` counts (a multiple of 4, read as independent
// 16-byte loads: the kernel is pure latency)
constexpr int SCAN_NT = 1024;
__global__ void __launch_bounds__(SCAN_NT)
chunk_scan(int32_t *counts, int n_chunks, unsigned long long *total)
{
    __shared__ int s_wave[SCAN_NT / 64];
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

// assigned pixels in raster order with their component root as sort key
__global__ void __launch_bounds__(NT)
emit_assigned(const int16_t *__restrict__ dec, const int32_t *__restrict__ L, int64_t n,
              const int32_t *__restrict__ offsets, int64_t cap, int32_t *__restrict__ pix,
              uint32_t *__restrict__ keys)
{
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    // a pixel is assigned iff its label is >= 0 (mf_label_components): the four
    // labels come as one 16-byte load and the barcodes need not be read again
    int lab[4] = {-1, -1, -1, -1};
    if (i0 + 3 < n) {
        const int4 v = *reinterpret_cast<const int4 *>(L + i0);
        lab[0] = v.x; lab[1] = v.y; lab[2] = v.z; lab[3] = v.w;
    } else {
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n) lab[k] = L[i0 + k];
    }
    (void)dec;
    unsigned f = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) f |= (unsigned)(lab[k] >= 0) << k;
    int pos = offsets[blockIdx.x] + block_exclusive_scan(__popc(f));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (f >> k & 1u) {
            if (pos < cap) {           // cap: the caller's n_pixels sized the lists
                pix[pos] = (int32_t)(i0 + k);
                keys[pos] = (uint32_t)lab[k];
            }
            ++pos;
        }
}

// first entry of every region in the root-sorted pixel list
__global__ void __launch_bounds__(NT)
emit_boundaries(const uint32_t *__restrict__ keys, int64_t n,
                const int32_t *__restrict__ offsets, int64_t cap, int32_t *__restrict__ starts)
{
    const int64_t i0 = (int64_t)blockIdx.x * CHUNK + 4 * threadIdx.x;
    const unsigned f = boundary_flags(keys, i0, n);
    int pos = offsets[blockIdx.x] + block_exclusive_scan(__popc(f));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (f >> k & 1u) {
            if (pos < cap) starts[pos] = (int32_t)(i0 + k);
            ++pos;
        }
}

// Regions arrive ordered by root.  The reference emits them by z plane (2-D
// mode), then barcode, then label order (= root order): key = plane * 32768 +
// barcode, sorted stably.  In 2-D mode the plane is root / plane_px.
__global__ void __launch_bounds__(NT)
region_keys(const uint32_t *__restrict__ roots_sorted, const int32_t *__restrict__ offsets,
            const int16_t *__restrict__ dec, int64_t n_reg, int64_t n_pix, int plane_px,
            int conn3d, uint32_t *__restrict__ keys, int64_t *__restrict__ span)
{
    const int64_t j = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (j >= n_reg) return;
    const int32_t lo = offsets[j];
    const int64_t hi = j + 1 < n_reg ? offsets[j + 1] : n_pix;
    const uint32_t root = roots_sorted[lo];
    const uint32_t grp = conn3d ? 0u : root / (uint32_t)plane_px;
    keys[j] = grp * 32768u + (uint32_t)dec[root];
    // the sort carries the region's span in the pixel list along, so the
    // statistics kernel starts from one load instead of a chain of three
    span[j] = (int64_t)lo | (hi << 32);
}

struct StatsArgs {
    const int32_t *pix;      // sorted by root, raster order inside a region
    const int64_t *span;     // [n_regions] first | one-past-last << 32 of row r's pixels
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
    const int64_t sp_ = a.span[r];
    const int i0 = (int)(sp_ & 0xffffffffll), i1 = (int)(sp_ >> 32);
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

// Onesweep with 9-bit digits: a pass has a fixed cost of tens of us here
// almost regardless of the element count, so the number of passes is what
// matters -- 3 instead of 4 for the <= 27-bit pixel keys, 2 instead of 3 for
// the <= 18-bit region keys.  (rocPRIM's default for ~2e5 elements would be a
// block sort plus ~18 merge launches: the 4096 limit forces onesweep.)
#ifndef MF_RADIX_BITS
#define MF_RADIX_BITS 9
#endif
// Sort blocks of 512 threads x 12 items.  Per pass over the 6.9 M (root, pixel)
// pairs of a 7-z FOV with the GPU to itself: 512 x 8: 73 us, 512 x 12: 65,
// 512 x 16: 71, 256 x 16: 103, 1024 x 6: 55, 1024 x 8: 51, 1024 x 10 / 12: 72-74
// (dev/time_region_variants.sh).  What counts is the FOV pipeline, where the sort
// runs beside the next FOV's decode kernel and a block has to fit into the LDS
// and wave slots that kernel leaves free: 512 x 8: 1.873 ms per FOV, 512 x 10:
// 1.848, 512 x 12: 1.845, 256 x 16: 1.90, 256 x 24: 1.87, and 2.00-2.06 for every
// shape with 64 KB of LDS per block (1024 x 6 / 8, 512 x 16) (dev/pipedepth.py).
#ifndef MF_SORT_BLOCK
#define MF_SORT_BLOCK (1 << MF_RADIX_BITS)
#endif
#ifndef MF_SORT_IPT
#define MF_SORT_IPT 12
#endif
using Onesweep9 = rocprim::radix_sort_onesweep_config<rocprim::kernel_config<MF_SORT_BLOCK, MF_SORT_IPT>,
                                                      rocprim::kernel_config<MF_SORT_BLOCK, MF_SORT_IPT>,
                                                      MF_RADIX_BITS,
                                                      rocprim::block_radix_rank_algorithm::match>;
using PixelSortConfig =
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, Onesweep9, 4096>;
using RegionSortConfig = PixelSortConfig;

// workspace carve-up shared by the size query and the call
struct Plan {
    size_t keys_a, keys_b, pix_a, pix_b, offsets, rkeys_a, rkeys_b, ridx_a, ridx_b, chunks, temp,
        temp_bytes, total;
    int vol_chunks, pix_chunks;
    int root_bits, region_key_bits;
};

inline size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

int make_plan(int Z, int H, int W, int64_t n_pix, int64_t n_reg, Plan &pl)
{
    const int64_t vol = (int64_t)Z * H * W;
    int rb = 1;
    while (((int64_t)1 << rb) < vol) ++rb;
    pl.root_bits = rb;
    int zb = 0;
    while ((1 << zb) < Z) ++zb;
    pl.region_key_bits = 15 + zb;      // barcode < 32768
    if (pl.region_key_bits > 32)
        return mf::fail(MF_EUNSUPPORTED, "region key needs %d bits", pl.region_key_bits);

    const size_t np = (size_t)(n_pix > 0 ? n_pix : 1), nr = (size_t)(n_reg > 0 ? n_reg : 1);
    size_t t2 = 0, t4 = 0;
    // size queries (null temp storage): no work is enqueued
    hipError_t e;
    e = rocprim::radix_sort_pairs<PixelSortConfig>(nullptr, t2, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (int32_t *)nullptr, (int32_t *)nullptr, np, 0u,
                                  (unsigned)pl.root_bits);
    if (e != hipSuccess) return mf::fail(MF_EHIP, "rocprim::radix_sort size query: %s", hipGetErrorString(e));
    e = rocprim::radix_sort_pairs<RegionSortConfig>(nullptr, t4, (uint32_t *)nullptr,
                                                    (uint32_t *)nullptr, (int64_t *)nullptr,
                                                    (int64_t *)nullptr, nr, 0u,
                                                    (unsigned)pl.region_key_bits);
    if (e != hipSuccess) return mf::fail(MF_EHIP, "rocprim::radix_sort size query: %s", hipGetErrorString(e));
    pl.temp_bytes = t2 > t4 ? t2 : t4;
    pl.vol_chunks = (int)((vol + CHUNK - 1) / CHUNK);
    pl.pix_chunks = (int)((np + CHUNK - 1) / CHUNK);

    size_t off = 0;
    pl.keys_a = off; off += align_up(np * 4);
    pl.keys_b = off; off += align_up(np * 4);
    pl.pix_a = off; off += align_up(np * 4);
    pl.pix_b = off; off += align_up(np * 4);
    pl.offsets = off; off += align_up((nr + 1) * 4);
    pl.rkeys_a = off; off += align_up(nr * 4);
    pl.rkeys_b = off; off += align_up(nr * 4);
    pl.ridx_a = off; off += align_up(nr * 8);
    pl.ridx_b = off; off += align_up(nr * 8);
    pl.chunks = off; off += align_up((size_t)pl.vol_chunks * 4);
    pl.temp = off; off += align_up(pl.temp_bytes);
    pl.total = off;
    return MF_OK;
}

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
                               int64_t tr_pix_stride, const int32_t *labels, int B, int Z, int H,
                               int W, int conn3d,
                               int64_t n_pixels, int64_t n_regions, double *rec, float *inten,
                               float *mean_tm, float *min_tm, void *workspace,
                               int64_t workspace_bytes, void *stream)
{
    if (n_regions == 0 || n_pixels == 0) return MF_OK;
    if (!decoded || !magnitudes || !distances || !traces || !labels || !rec || !inten || !workspace)
        return mf::fail(MF_EINVAL, "null pointer argument");
    if (B < 1 || Z < 1 || H < 1 || W < 1 || n_pixels < n_regions || n_regions < 0)
        return mf::fail(MF_EINVAL, "bad sizes");
    if (dist_dtype != MF_DTYPE_F32 && dist_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "distance dtype %d not f32/f64", dist_dtype);
    const int64_t vol = (int64_t)Z * H * W;
    if (vol > INT32_MAX || n_pixels > vol) return mf::fail(MF_EINVAL, "bad pixel count");
    if (reinterpret_cast<uintptr_t>(decoded) & 7)
        return mf::fail(MF_EINVAL, "decoded must be 8-byte aligned");
    if (reinterpret_cast<uintptr_t>(labels) & 15)
        return mf::fail(MF_EINVAL, "labels must be 16-byte aligned");
    Plan pl;
    int rc = make_plan(Z, H, W, n_pixels, n_regions, pl);
    if (rc) return rc;
    if ((int64_t)pl.total > workspace_bytes)
        return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes", (long long)workspace_bytes,
                        (long long)pl.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned char *ws = reinterpret_cast<unsigned char *>(workspace);
    uint32_t *keys_a = reinterpret_cast<uint32_t *>(ws + pl.keys_a);
    uint32_t *keys_b = reinterpret_cast<uint32_t *>(ws + pl.keys_b);
    int32_t *pix_a = reinterpret_cast<int32_t *>(ws + pl.pix_a);
    int32_t *pix_b = reinterpret_cast<int32_t *>(ws + pl.pix_b);
    int32_t *offsets = reinterpret_cast<int32_t *>(ws + pl.offsets);
    uint32_t *rkeys_a = reinterpret_cast<uint32_t *>(ws + pl.rkeys_a);
    uint32_t *rkeys_b = reinterpret_cast<uint32_t *>(ws + pl.rkeys_b);
    int64_t *ridx_a = reinterpret_cast<int64_t *>(ws + pl.ridx_a);
    int64_t *ridx_b = reinterpret_cast<int64_t *>(ws + pl.ridx_b);
    int32_t *chunks = reinterpret_cast<int32_t *>(ws + pl.chunks);
    void *temp = ws + pl.temp;
    size_t tb = pl.temp_bytes;

    // 1. assigned pixels in raster order, keyed by their component root
    hipLaunchKernelGGL(chunk_count<false>, dim3(pl.vol_chunks), dim3(NT), 0, s, decoded,
                       (const uint32_t *)nullptr, vol, chunks);
    hipLaunchKernelGGL(chunk_scan, dim3(1), dim3(SCAN_NT), 0, s, chunks, pl.vol_chunks,
                       (unsigned long long *)nullptr);
    hipLaunchKernelGGL(emit_assigned, dim3(pl.vol_chunks), dim3(NT), 0, s, decoded, labels, vol,
                       chunks, n_pixels, pix_a, keys_a);
    // 2. group them by root; the stable sort keeps raster order inside a
    //    region and leaves the regions in root (= label) order
    MF_HIP_CHECK(rocprim::radix_sort_pairs<PixelSortConfig>(
        temp, tb, keys_a, keys_b, pix_a, pix_b, (size_t)n_pixels, 0u, (unsigned)pl.root_bits, s));
    // 3. region starts
    hipLaunchKernelGGL(chunk_count<true>, dim3(pl.pix_chunks), dim3(NT), 0, s,
                       (const int16_t *)nullptr, keys_b, n_pixels, chunks);
    hipLaunchKernelGGL(chunk_scan, dim3(1), dim3(SCAN_NT), 0, s, chunks, pl.pix_chunks,
                       (unsigned long long *)nullptr);
    hipLaunchKernelGGL(emit_boundaries, dim3(pl.pix_chunks), dim3(NT), 0, s, keys_b, n_pixels,
                       chunks, n_regions, offsets);
    // 4. emission order of the regions: (plane, barcode), label order inside
    {
        const unsigned gq = (unsigned)((n_regions + NT - 1) / NT);
        hipLaunchKernelGGL(region_keys, dim3(gq), dim3(NT), 0, s, keys_b, offsets, decoded,
                           n_regions, n_pixels, H * W, conn3d, rkeys_a, ridx_a);
    }
    tb = pl.temp_bytes;
    MF_HIP_CHECK(rocprim::radix_sort_pairs<RegionSortConfig>(
        temp, tb, rkeys_a, rkeys_b, ridx_a, ridx_b, (size_t)n_regions, 0u,
        (unsigned)pl.region_key_bits, s));
    // 5. per-region rows
    StatsArgs a;
    a.pix = pix_b; a.span = ridx_b; a.n_pix = n_pixels; a.n_reg = n_regions;
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
