// preprocess.hip -- high-pass filter + Lucy-Richardson deconvolution for gfx950
// (MI355X): the stage that produces the decoder's uint16 stack.
//
// Replaces merlin/util/imagefilters.py:9-28 (high_pass_filter),
// merlin/util/deconvolve.py:70-133 (deconvolve_lucyrichardson, odd windows) and
// merlin/analysis/preprocess.py:114-140 (_preprocess_image + pixel histogram).
//
// The deconvolution is float64 like the reference and bound by memory traffic:
// per iteration and pixel it reads J1, J2 (twice), the previous difference T1
// and the blurred ratio, and writes the ratio, the new J1 and T1 (82 B/px with
// a uint16 observed image).  Two launches per iteration:
//
//   lr_reblur : Y = max(J1 + lambda (J1 - J2), 0) is rebuilt on the fly for a
//               64 x 32 tile plus halo, blurred through LDS (row pass left to
//               right, column pass centre-out: OpenCV's RowFilter /
//               SymmColumnFilter order) and turned into the ratio
//               wI / max(reblurred, eps) + eps.
//   lr_update : blurs the ratio the same way, J1' = Y * blurred, T1' = J1' - Y,
//               and reduces sum(T1' T1) and sum(T1 T1) for the next lambda in a
//               fixed order: per thread, per wave, per tile, then the LAST
//               tile of the image to finish adds the tile partials in tile
//               order (no float atomics, run-to-run identical).
//
// Images are processed a few at a time (`batch`) so that one batch's working
// set (34 B/px: 143 MB for a 2048 x 2048 image) stays inside the 256 MiB
// Infinity Cache from one iteration to the next instead of streaming from HBM.
//
// Built with -ffp-contract=off: every product and sum below rounds on its own,
// as the non-FMA scalar path of OpenCV and numpy do.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "common.h"

namespace {

constexpr int TW = 64;          // tile width = wave width
constexpr int TH = 32;          // tile height: 4 waves x 8 rows
constexpr int NT = 256;
constexpr int ROWS_PER_WAVE = TH / 4;
constexpr int HIST_BINS = MF_HIST_BINS;
constexpr double EPS = DBL_EPSILON;      // np.finfo(float).eps, deconvolve.py:89

struct Taps {
    double k[2 * MF_PP_MAX_RADIUS + 1];
    uint32_t kfix[2 * MF_PP_MAX_RADIUS + 1];
};

// blockIdx.x -> tile so that the blocks of one XCD (blockIdx.x % 8 under
// round-robin placement) own one contiguous band of tiles: the halo rows of
// neighbouring tiles are then found in that XCD's L2.
__device__ __forceinline__ int xcd_tile(int tiles)
{
    const int per_xcd = (tiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    return (blockIdx.x >> 3) < per_xcd && t < tiles ? t : -1;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ double load_wi(const uint16_t *p, int64_t i) { return (double)p[i]; }
__device__ __forceinline__ double load_wi(const double *p, int64_t i) { return p[i]; }

// Y of the current iteration (deconvolve.py:112-114): cv2.subtract,
// cv2.addWeighted(J1, 1, Y, lambda, 0), clip at 0
__device__ __forceinline__ double make_y(double j1, double j2, double lam)
{
    const double d = j1 - j2;
    double y = j1 * 1.0 + d * lam;
    y = y + 0.0;
    return y < 0.0 ? 0.0 : y;      // np.clip keeps NaN
}

// Separable blur of one tile.  `src(gy, gx)` yields the (border-replicated)
// input; the result of this thread's ROWS_PER_WAVE vertically adjacent pixels
// (column x0 + lane, rows y0 + wave*8 ..) is left in out[].
template <int R, typename Src>
__device__ __forceinline__ void blur_tile(const Taps &tp, int x0, int y0, int H, int W,
                                          double *sIn, double *sH, Src src,
                                          double (&out)[ROWS_PER_WAVE])
{
    constexpr int IW = TW + 2 * R, IH = TH + 2 * R;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < IW * IH; e += NT) {
        const int r = e / IW, c = e - r * IW;
        sIn[e] = src(clampi(y0 - R + r, 0, H - 1), clampi(x0 - R + c, 0, W - 1));
    }
    __syncthreads();
    // row pass (OpenCV RowFilter): s = k[0] S[0]; s += k[j] S[j]
    for (int r = wave; r < IH; r += 4) {
        const double *row = sIn + r * IW + lane;
        double s = tp.k[0] * row[0];
#pragma unroll
        for (int j = 1; j <= 2 * R; ++j) s = s + tp.k[j] * row[j];
        sH[r * TW + lane] = s;
    }
    __syncthreads();
    // column pass (OpenCV SymmColumnFilter): centre, then pairs outwards
    double col[ROWS_PER_WAVE + 2 * R];
#pragma unroll
    for (int j = 0; j < ROWS_PER_WAVE + 2 * R; ++j)
        col[j] = sH[(wave * ROWS_PER_WAVE + j) * TW + lane];
#pragma unroll
    for (int o = 0; o < ROWS_PER_WAVE; ++o) {
        double s = tp.k[R] * col[o + R];
#pragma unroll
        for (int j = 1; j <= R; ++j) s = s + tp.k[R + j] * (col[o + R + j] + col[o + R - j]);
        out[o] = s;
    }
}

template <int R>
constexpr size_t blur_lds_bytes()
{
    return ((size_t)(TW + 2 * R) * (TH + 2 * R) + (size_t)TW * (TH + 2 * R)) * sizeof(double);
}

struct LrArgs {
    const void *wi;          // observed image(s), u16 or f64 [n][HW]
    const double *j1, *j2;   // [n][HW]
    double *ratio;           // [n][HW]
    double *jnew;            // = j2's buffer
    double *t1;              // [n][HW], updated in place
    const double *lam;       // [n] lambda of this iteration
    double *lam_next;        // [n]
    double *lam_trace;       // [n][trace_stride] or NULL: lambda used by iteration `iter`
    int64_t trace_stride;
    int iter;
    double *partials;        // [n][tiles][2]
    unsigned *counters;      // [n]
    int H, W, tiles_x, tiles;
    Taps tp;
};

template <int R, typename WT>
__global__ void __launch_bounds__(NT) lr_reblur_kernel(const LrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sIn = reinterpret_cast<double *>(smem);
    double *sH = sIn + (TW + 2 * R) * (TH + 2 * R);
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int64_t base = (int64_t)img * a.H * a.W;
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * TH;
    const double lam = a.lam[img];
    const double *j1 = a.j1 + base, *j2 = a.j2 + base;
    const int W = a.W;
    double out[ROWS_PER_WAVE];
    blur_tile<R>(a.tp, x0, y0, a.H, W, sIn, sH,
                 [&](int gy, int gx) {
                     const int64_t p = (int64_t)gy * W + gx;
                     return make_y(j1[p], j2[p], lam);
                 },
                 out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gx = x0 + lane;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + base;
    if (gx < W) {
#pragma unroll
        for (int o = 0; o < ROWS_PER_WAVE; ++o) {
            const int gy = y0 + wave * ROWS_PER_WAVE + o;
            if (gy < a.H) {
                const int64_t p = (int64_t)gy * W + gx;
                double rb = out[o];
                rb = rb < EPS ? EPS : rb;                       // np.clip(reblurred, eps, None)
                a.ratio[base + p] = load_wi(wi, p) / rb + EPS;  // cv2.divide; imR += eps
            }
        }
    }
}

__device__ __forceinline__ double atomic_load_f64(const double *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v;       // lane 0 holds the total
}

template <int R>
__global__ void __launch_bounds__(NT) lr_update_kernel(const LrArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *sIn = reinterpret_cast<double *>(smem);
    double *sH = sIn + (TW + 2 * R) * (TH + 2 * R);
    __shared__ double red[2][NT / 64];
    __shared__ int is_last;
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int64_t base = (int64_t)img * a.H * a.W;
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * TH;
    const double lam = a.lam[img];
    const double *ratio = a.ratio + base;
    const int W = a.W;
    double out[ROWS_PER_WAVE];
    blur_tile<R>(a.tp, x0, y0, a.H, W, sIn, sH,
                 [&](int gy, int gx) { return ratio[(int64_t)gy * W + gx]; }, out);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gx = x0 + lane;
    double s_ab = 0.0, s_bb = 0.0;
    if (gx < W) {
#pragma unroll
        for (int o = 0; o < ROWS_PER_WAVE; ++o) {
            const int gy = y0 + wave * ROWS_PER_WAVE + o;
            if (gy < a.H) {
                const int64_t p = base + (int64_t)gy * W + gx;
                const double y = make_y(a.j1[p], a.j2[p], lam);
                const double jn = y * out[o];            // np.multiply(Y, imR, out=J1)
                const double tn = jn - y;                // np.subtract(J1, Y, out=T1)
                const double to = a.t1[p];
                a.jnew[p] = jn;
                a.t1[p] = tn;
                s_ab = s_ab + tn * to;                   // cv2.multiply(T1, T2) summed
                s_bb = s_bb + to * to;
            }
        }
    }
    s_ab = wave_sum(s_ab);
    s_bb = wave_sum(s_bb);
    if (lane == 0) { red[0][wave] = s_ab; red[1][wave] = s_bb; }
    __syncthreads();
    if (tid == 0) {
        double *pp = a.partials + ((int64_t)img * a.tiles + t) * 2;
        pp[0] = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        pp[1] = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        __threadfence();                                  // partials before the ticket
        const unsigned ticket = atomicAdd(a.counters + img, 1u);
        is_last = ticket == (unsigned)a.tiles - 1u;
    }
    __syncthreads();
    if (!is_last) return;
    // last tile of this image: add the tile partials in tile order
    __threadfence();
    const double *pp = a.partials + (int64_t)img * a.tiles * 2;
    double ab = 0.0, bb = 0.0;
    for (int i = tid; i < a.tiles; i += NT) {
        ab = ab + atomic_load_f64(pp + 2 * i);
        bb = bb + atomic_load_f64(pp + 2 * i + 1);
    }
    ab = wave_sum(ab);
    bb = wave_sum(bb);
    __syncthreads();
    if (lane == 0) { red[0][wave] = ab; red[1][wave] = bb; }
    __syncthreads();
    if (tid == 0) {
        ab = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        bb = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        double l = ab / (bb + EPS);                       // deconvolve.py:110
        l = 1.0 < l ? 1.0 : l;                            // min(l, 1)
        l = 0.0 > l ? 0.0 : l;                            // max(., 0)
        // the reference only starts accelerating at its third iteration
        if (a.iter + 1 <= 1) l = 0.0;
        a.lam_next[img] = l;
        if (a.lam_trace) a.lam_trace[(int64_t)img * a.trace_stride + a.iter + 1] = l;
        a.counters[img] = 0u;
    }
}

template <typename WT>
__global__ void __launch_bounds__(NT) lr_init_kernel(const WT *wi, double *ja, double *jb, double *t1,
                                                     int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) {
        const double v = load_wi(wi, i);
        ja[i] = v;
        jb[i] = v;
        t1[i] = 0.0;
    }
}

__global__ void lr_reset_kernel(double *lam, double *lam_trace, int64_t trace_stride,
                                unsigned *counters, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        lam[i] = 0.0;
        counters[i] = 0u;
        if (lam_trace) lam_trace[(int64_t)i * trace_stride] = 0.0;
    }
}

// float64 -> uint16 the way numpy's astype does on x86: cvttsd2si to int32
// (0x80000000 when out of range or NaN), low 16 bits (preprocess.py:139)
__device__ __forceinline__ uint16_t to_u16(double v)
{
    if (!(fabs(v) < 2147483648.0)) return 0;
    return (uint16_t)((int32_t)v & 0xffff);
}

constexpr int HIST_LOW = 8192;       // bins kept in LDS; the rest go to global atomics
constexpr int HIST_BLOCKS = 64;      // blocks per image

// J1 -> output (+ histogram over np.arange(0, 65535): 65534 bins, the last one
// closed on the right, 65535 outside; preprocess.py:114-129)
__global__ void __launch_bounds__(NT) lr_finalize_kernel(const double *j, void *out, int out_f64,
                                                         int64_t hw, unsigned long long *hist,
                                                         int hist_group, int img0)
{
    __shared__ unsigned low[HIST_LOW];
    const int img = blockIdx.y;
    const double *src = j + (int64_t)img * hw;
    if (hist)
        for (int i = threadIdx.x; i < HIST_LOW; i += NT) low[i] = 0u;
    __syncthreads();
    unsigned long long *row = hist ? hist + (int64_t)((img0 + img) / hist_group) * HIST_BINS : nullptr;
    const int64_t per = (hw + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = min(hw, lo + per);
    for (int64_t i = lo + threadIdx.x; i < hi; i += NT) {
        const double v = src[i];
        if (out_f64) {
            reinterpret_cast<double *>(out)[(int64_t)img * hw + i] = v;
        } else {
            const uint16_t u = to_u16(v);
            reinterpret_cast<uint16_t *>(out)[(int64_t)img * hw + i] = u;
            if (hist) {
                if (u < HIST_LOW) atomicAdd(&low[u], 1u);
                else if (u < 65535) atomicAdd(row + min((int)u, HIST_BINS - 1), 1ull);
            }
        }
    }
    if (hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < HIST_LOW; i += NT)
            if (low[i]) atomicAdd(row + i, (unsigned long long)low[i]);
    }
}

// high_pass_filter on uint16 (imagefilters.py:22-28): exact fixed-point blur
// (u32 row sums, u64 column sums, (acc + 2^31) >> 32), BORDER_REPLICATE, then
// image - lowpass, 0 where lowpass > image.
struct HpArgs {
    const uint16_t *in;
    uint16_t *out;
    int H, W, R, tiles_x, tiles;
    Taps tp;
};

__global__ void __launch_bounds__(NT) high_pass_kernel(const HpArgs a)
{
    constexpr int MR = MF_PP_MAX_RADIUS;
    __shared__ uint16_t sIn[(TH + 2 * MR) * (TW + 2 * MR)];
    __shared__ uint32_t sH[(TH + 2 * MR) * TW];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int R = a.R, IW = TW + 2 * R, IH = TH + 2 * R;
    const int64_t base = (int64_t)blockIdx.y * a.H * a.W;
    const uint16_t *in = a.in + base;
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * TH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < IW * IH; e += NT) {
        const int r = e / IW, c = e - r * IW;
        sIn[e] = in[(int64_t)clampi(y0 - R + r, 0, a.H - 1) * a.W + clampi(x0 - R + c, 0, a.W - 1)];
    }
    __syncthreads();
    for (int r = wave; r < IH; r += 4) {
        uint32_t s = 0;
        for (int j = 0; j <= 2 * R; ++j) s += a.tp.kfix[j] * (uint32_t)sIn[r * IW + lane + j];
        sH[r * TW + lane] = s;
    }
    __syncthreads();
    const int gx = x0 + lane;
    if (gx >= a.W) return;
    for (int o = 0; o < ROWS_PER_WAVE; ++o) {
        const int ly = wave * ROWS_PER_WAVE + o, gy = y0 + ly;
        if (gy >= a.H) break;
        uint64_t s = 0;
        for (int j = 0; j <= 2 * R; ++j) s += (uint64_t)a.tp.kfix[j] * sH[(ly + j) * TW + lane];
        uint64_t lp = (s + (1ull << 31)) >> 32;
        if (lp > 65535) lp = 65535;
        const uint32_t v = sIn[(ly + R) * IW + lane + R];
        a.out[base + (int64_t)gy * a.W + gx] = lp > v ? (uint16_t)0 : (uint16_t)(v - (uint32_t)lp);
    }
}

int fill_taps(int ksize, double sigma, Taps &tp)
{
    if (ksize < 1 || !(ksize & 1) || ksize > 2 * MF_PP_MAX_RADIUS + 1)
        return mf::fail(MF_EUNSUPPORTED, "filter size %d must be odd and <= %d", ksize,
                        2 * MF_PP_MAX_RADIUS + 1);
    std::memset(&tp, 0, sizeof(tp));
    if (ksize == 1) { tp.k[0] = 1.0; tp.kfix[0] = 65536u; return MF_OK; }
    // OpenCV getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED
    if (sigma <= 0) sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double sum = 0.0;
    for (int i = 0; i < ksize; ++i) {
        const double xx = i - (ksize - 1) * 0.5;
        tp.k[i] = std::exp(scale2x * xx * xx);
        sum += tp.k[i];
    }
    const double inv = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) tp.k[i] *= inv;
    double err = 0.0;
    int64_t total = 0;
    const int half = ksize / 2;
    for (int i = 0; i < half; ++i) {
        const double adj = tp.k[i] * 65536.0 + err;
        const int64_t v0 = (int64_t)std::nearbyint(adj);
        err = adj - (double)v0;
        tp.kfix[i] = tp.kfix[ksize - 1 - i] = (uint32_t)v0;
        total += 2 * v0;
    }
    tp.kfix[half] = (uint32_t)(65536 - total);
    return MF_OK;
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

struct Workspace {
    double *ja, *jb, *t1, *ratio, *partials, *lam;
    unsigned *counters;
    uint16_t *wi16;
    size_t total;
};

void layout_workspace(int batch, int H, int W, unsigned char *base, Workspace &ws)
{
    const size_t hw = (size_t)H * W;
    const size_t tiles = (size_t)((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base + off; off += align256(bytes); return p; };
    ws.ja = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.jb = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.t1 = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.ratio = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.wi16 = reinterpret_cast<uint16_t *>(take(batch * hw * 2));
    ws.partials = reinterpret_cast<double *>(take(batch * tiles * 16));
    ws.lam = reinterpret_cast<double *>(take((size_t)batch * 2 * 8));
    ws.counters = reinterpret_cast<unsigned *>(take((size_t)batch * 4));
    ws.total = off;
}

template <int R, typename WT>
int launch_iteration(const LrArgs &a, int n, hipStream_t s)
{
    const size_t lds = blur_lds_bytes<R>();
    auto ka = lr_reblur_kernel<R, WT>;
    auto kb = lr_update_kernel<R>;
    static bool attr_done = false;       // per instantiation; idempotent
    if (!attr_done && lds > 48 * 1024) {
        MF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(ka),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kb),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)n);
    hipLaunchKernelGGL(ka, grid, dim3(NT), lds, s, a);
    hipLaunchKernelGGL(kb, grid, dim3(NT), lds, s, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

template <typename WT>
int dispatch_iteration(int R, const LrArgs &a, int n, hipStream_t s)
{
    switch (R) {
    case 0: return launch_iteration<0, WT>(a, n, s);
    case 1: return launch_iteration<1, WT>(a, n, s);
    case 2: return launch_iteration<2, WT>(a, n, s);
    case 3: return launch_iteration<3, WT>(a, n, s);
    case 4: return launch_iteration<4, WT>(a, n, s);
    case 5: return launch_iteration<5, WT>(a, n, s);
    case 6: return launch_iteration<6, WT>(a, n, s);
    }
    return mf::fail(MF_EUNSUPPORTED, "deconvolution filter radius %d > 6", R);
}

int check_shape(int n, int H, int W, int batch)
{
    if (n < 0 || H < 1 || W < 1 || batch < 1)
        return mf::fail(MF_EINVAL, "bad shape n=%d H=%d W=%d batch=%d", n, H, W, batch);
    if ((int64_t)H * W > INT32_MAX) return mf::fail(MF_EUNSUPPORTED, "image exceeds 2^31 pixels");
    if (batch > 65535) return mf::fail(MF_EUNSUPPORTED, "batch %d > 65535", batch);
    return MF_OK;
}

int run_high_pass(const uint16_t *in, int n, int H, int W, int ksize, double sigma, uint16_t *out,
                  hipStream_t s)
{
    HpArgs h;
    int rc = fill_taps(ksize, sigma, h.tp);
    if (rc) return rc;
    h.R = ksize / 2;
    h.H = H; h.W = W;
    h.tiles_x = (W + TW - 1) / TW;
    h.tiles = h.tiles_x * ((H + TH - 1) / TH);
    for (int i0 = 0; i0 < n; i0 += 65535) {
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        h.in = in + (int64_t)i0 * H * W;
        h.out = out + (int64_t)i0 * H * W;
        hipLaunchKernelGGL(high_pass_kernel, dim3((unsigned)(((h.tiles + 7) >> 3) << 3), (unsigned)nb),
                           dim3(NT), 0, s, h);
    }
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

// Lucy-Richardson on images [i0, i0 + nb) whose observed image is `wi`
template <typename WT>
int run_lr_batch(const WT *wi, int nb, int i0, int H, int W, int R, const Taps &tp, int iterations,
                 void *out, int out_f64, unsigned long long *hist, int hist_group, double *lam_trace,
                 Workspace &ws, hipStream_t s)
{
    const int64_t hw = (int64_t)H * W;
    const int64_t total = hw * nb;
    hipLaunchKernelGGL(lr_init_kernel<WT>, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s, wi,
                       ws.ja, ws.jb, ws.t1, total);
    double *trace = lam_trace ? lam_trace + (int64_t)i0 * iterations : nullptr;
    hipLaunchKernelGGL(lr_reset_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, ws.lam,
                       iterations > 0 ? trace : nullptr, (int64_t)iterations, ws.counters, nb);
    LrArgs a;
    std::memset(&a, 0, sizeof(a));
    a.wi = wi;
    a.ratio = ws.ratio;
    a.t1 = ws.t1;
    a.partials = ws.partials;
    a.counters = ws.counters;
    a.H = H; a.W = W;
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles = a.tiles_x * ((H + TH - 1) / TH);
    a.tp = tp;
    a.trace_stride = iterations;
    double *j1 = ws.ja, *j2 = ws.jb;
    for (int it = 0; it < iterations; ++it) {
        a.j1 = j1; a.j2 = j2; a.jnew = j2;
        a.lam = ws.lam + (size_t)(it & 1) * nb;
        a.lam_next = ws.lam + (size_t)((it + 1) & 1) * nb;
        a.lam_trace = it + 1 < iterations ? trace : nullptr;
        a.iter = it;
        int rc = dispatch_iteration<WT>(R, a, nb, s);
        if (rc) return rc;
        double *tmp = j1; j1 = j2; j2 = tmp;
    }
    void *dst = out_f64 ? (void *)(reinterpret_cast<double *>(out) + (int64_t)i0 * hw)
                        : (void *)(reinterpret_cast<uint16_t *>(out) + (int64_t)i0 * hw);
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, j1, dst,
                       out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

}  // namespace

extern "C" int64_t mf_preprocess_workspace_bytes(int batch, int H, int W)
{
    int rc = check_shape(0, H, W, batch);
    if (rc) return rc;
    Workspace ws;
    layout_workspace(batch, H, W, nullptr, ws);
    return (int64_t)ws.total;
}

extern "C" int mf_high_pass_filter(const uint16_t *image, int n_images, int H, int W,
                                   int window_size, double sigma, uint16_t *out, void *stream)
{
    if (!image || !out) return mf::fail(MF_EINVAL, "null pointer argument");
    int rc = check_shape(n_images, H, W, 1);
    if (rc) return rc;
    if (!(sigma >= 0)) return mf::fail(MF_EINVAL, "sigma must be >= 0");
    if (image == out) return mf::fail(MF_EINVAL, "high-pass filter cannot run in place");
    return run_high_pass(image, n_images, H, W, window_size, sigma, out,
                         reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mf_deconvolve_lucyrichardson(const void *image, int image_dtype, int n_images, int H,
                                            int W, int window_size, double sigma, int iterations,
                                            void *out, int out_dtype, int64_t *histogram,
                                            int hist_group, double *lambdas, int batch,
                                            void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!image || !out || !workspace) return mf::fail(MF_EINVAL, "null pointer argument");
    int rc = check_shape(n_images, H, W, batch);
    if (rc) return rc;
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f64", image_dtype);
    if (out_dtype != MF_DTYPE_U16 && out_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "output dtype %d not u16/f64", out_dtype);
    if (iterations < 0) return mf::fail(MF_EINVAL, "iteration count %d < 0", iterations);
    if (!(window_size & 1))
        return mf::fail(MF_EUNSUPPORTED, "even window size %d (the reference's ndimage.convolve "
                        "branch) is not built", window_size);
    if (histogram && (out_dtype != MF_DTYPE_U16 || hist_group < 1))
        return mf::fail(MF_EINVAL, "histogram needs uint16 output and hist_group >= 1");
    Taps tp;
    rc = fill_taps(window_size, sigma, tp);
    if (rc) return rc;
    const int R = window_size / 2;
    if (R > 6) return mf::fail(MF_EUNSUPPORTED, "deconvolution window %d > 13", window_size);
    Workspace ws;
    layout_workspace(batch, H, W, reinterpret_cast<unsigned char *>(workspace), ws);
    if ((int64_t)ws.total > workspace_bytes)
        return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes", (long long)workspace_bytes,
                        (long long)ws.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t hw = (int64_t)H * W;
    for (int i0 = 0; i0 < n_images; i0 += batch) {
        const int nb = n_images - i0 < batch ? n_images - i0 : batch;
        if (image_dtype == MF_DTYPE_U16)
            rc = run_lr_batch(reinterpret_cast<const uint16_t *>(image) + i0 * hw, nb, i0, H, W, R, tp,
                              iterations, out, out_dtype == MF_DTYPE_F64,
                              reinterpret_cast<unsigned long long *>(histogram), hist_group, lambdas,
                              ws, s);
        else
            rc = run_lr_batch(reinterpret_cast<const double *>(image) + i0 * hw, nb, i0, H, W, R, tp,
                              iterations, out, out_dtype == MF_DTYPE_F64,
                              reinterpret_cast<unsigned long long *>(histogram), hist_group, lambdas,
                              ws, s);
        if (rc) return rc;
    }
    return MF_OK;
}

extern "C" int mf_preprocess_images(const uint16_t *image, int n_images, int H, int W,
                                    double highpass_sigma, int decon_filter_size, double decon_sigma,
                                    int decon_iterations, uint16_t *out, int64_t *histogram,
                                    int hist_group, int batch, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    if (!image || !out || !workspace) return mf::fail(MF_EINVAL, "null pointer argument");
    int rc = check_shape(n_images, H, W, batch);
    if (rc) return rc;
    if (!(highpass_sigma >= 0)) return mf::fail(MF_EINVAL, "sigma must be >= 0");
    if (decon_iterations < 0) return mf::fail(MF_EINVAL, "iteration count %d < 0", decon_iterations);
    if (!(decon_filter_size & 1))
        return mf::fail(MF_EUNSUPPORTED, "even window size %d (the reference's ndimage.convolve "
                        "branch) is not built", decon_filter_size);
    if (histogram && hist_group < 1) return mf::fail(MF_EINVAL, "hist_group must be >= 1");
    Taps tp;
    rc = fill_taps(decon_filter_size, decon_sigma, tp);
    if (rc) return rc;
    const int R = decon_filter_size / 2;
    if (R > 6) return mf::fail(MF_EUNSUPPORTED, "deconvolution window %d > 13", decon_filter_size);
    Workspace ws;
    layout_workspace(batch, H, W, reinterpret_cast<unsigned char *>(workspace), ws);
    if ((int64_t)ws.total > workspace_bytes)
        return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes", (long long)workspace_bytes,
                        (long long)ws.total);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t hw = (int64_t)H * W;
    for (int i0 = 0; i0 < n_images; i0 += batch) {
        const int nb = n_images - i0 < batch ? n_images - i0 : batch;
        rc = run_high_pass(image + i0 * hw, nb, H, W,
                           (int)(2 * std::ceil(2 * highpass_sigma) + 1),   // preprocess.py:104
                           highpass_sigma, ws.wi16, s);
        if (rc) return rc;
        rc = run_lr_batch(ws.wi16, nb, i0, H, W, R, tp, decon_iterations, out, 0,
                          reinterpret_cast<unsigned long long *>(histogram), hist_group, nullptr, ws, s);
        if (rc) return rc;
    }
    return MF_OK;
}
