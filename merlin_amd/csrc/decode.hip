// decode.hip -- fused low-pass + scale/background + magnitude + L2-normalise +
// nearest-codeword kernel for gfx950 (MI355X).
//
// Replaces merlin/util/decoding.py:93-129 (decode_pixels) for Z planes at once.
//
// One workgroup (256 threads = 4 waves) owns a 64 x (4*PPT) pixel tile; lane =
// column, every thread carries PPT vertically adjacent pixels so that each
// wave-level load/store is one contiguous row segment.  Per bit the tile (+halo)
// is staged through LDS, blurred separably with exact integer arithmetic, and
// the scaled value lands in a per-thread register array x[PPT][MAXB]; the stack
// is read from HBM exactly once and the traces are written (optionally) once.
// The normalised codebook sits in LDS as float32 rows that every lane reads at
// the same address (broadcast), each lane keeps a running (best, second,
// argbest) over codewords for its own pixels, and only near-ties are re-scored
// in float64, so the winner equals the exact float64 1-NN of the reference.
//
// Built with -ffp-contract=off: every float op below is the single IEEE
// operation it is written as; fused multiply-adds are explicit fmaf() calls.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include "common.h"

namespace {

constexpr int TW = 64;              // tile width  = wave width
constexpr int NTHREADS = 256;       // 4 waves
#ifndef MF_TIE_EPS
#define MF_TIE_EPS 1.0e-5f
#endif
constexpr float TIE_EPS = MF_TIE_EPS;  // score gap below which float64 decides

struct DecodeArgs {
    const void *img;
    int64_t bit_stride, z_stride;
    int B, Z, H, W, N;
    const double *cb64;
    float sf[MF_MAX_BITS], bg[MF_MAX_BITS];
    uint32_t kfix[2 * MF_MAX_RADIUS + 1];
    float kflt[2 * MF_MAX_RADIUS + 1];
    double dthr;
    float mthr;
    int16_t *decoded;
    float *mag;
    void *dist;
    int dist_f64;
    float *traces;
    int64_t tr_z, tr_b;
    int tiles_x, tiles_y;
    int n_tiles;
    int chunk;        // codewords per LDS chunk
};

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// 0.5 * |c_n|^2 and the float32 row, computed identically wherever needed
__device__ __forceinline__ float half_norm(const double *row, int B)
{
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += row[b] * row[b];
    return (float)(0.5 * s);
}

template <typename PIX> struct HType { using type = uint32_t; };
template <> struct HType<float> { using type = float; };

template <typename PIX, int R, int MAXB, int PPT>
__global__ void __launch_bounds__(NTHREADS)
decode_kernel(const DecodeArgs a)
{
    using HT = typename HType<PIX>::type;
    constexpr int TH = 4 * PPT;
    constexpr int RH = TH + 2 * R;
    constexpr int RW = TW + 2 * R;
    constexpr int K = 2 * R + 1;
    constexpr size_t RAW_BYTES = ((size_t)RH * RW * sizeof(PIX) + 15) & ~size_t(15);
    constexpr size_t HS_BYTES = (size_t)RH * TW * sizeof(HT);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PIX *raw = reinterpret_cast<PIX *>(smem);
    HT *hs = reinterpret_cast<HT *>(smem + RAW_BYTES);
    float *cb = reinterpret_cast<float *>(smem + RAW_BYTES + HS_BYTES);  // [chunk][MAXB]
    float *hn = cb + (size_t)a.chunk * MAXB;                             // [chunk]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int B = a.B, H = a.H, W = a.W, N = a.N;
    const bool single_chunk = a.chunk >= N;
    const PIX *img = reinterpret_cast<const PIX *>(a.img);

    auto load_chunk = [&](int c0, int cn) {
        for (int i = tid; i < cn * MAXB; i += NTHREADS) {
            const int n = i / MAXB, b = i - n * MAXB;
            cb[i] = b < B ? (float)a.cb64[(size_t)(c0 + n) * B + b] : 0.0f;
        }
        for (int n = tid; n < cn; n += NTHREADS)
            hn[n] = half_norm(a.cb64 + (size_t)(c0 + n) * B, B);
    };
    if (single_chunk) load_chunk(0, N);   // made visible by the first barrier

    // XCD-aware tile walk: blocks with the same (blockIdx % 8) share an XCD
    // and therefore an L2; give each such group one contiguous run of tiles
    // so that the halo rows/columns of neighbouring tiles hit that L2.
    const int xcd = blockIdx.x & 7;
    const int per_xcd = (a.n_tiles + 7) >> 3;
    const int t_lo = xcd * per_xcd;
    const int t_hi = min(a.n_tiles, t_lo + per_xcd);
    const int t_step = gridDim.x >> 3;

    for (int t = t_lo + (blockIdx.x >> 3); t < t_hi; t += t_step) {
        const int tx = t % a.tiles_x;
        const int ty = (t / a.tiles_x) % a.tiles_y;
        const int z = t / (a.tiles_x * a.tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;

        float x[PPT][MAXB];

#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b < B) {
                const PIX *plane = img + (int64_t)b * a.bit_stride + (int64_t)z * a.z_stride;
                // ---- stage tile + halo (BORDER_REFLECT_101) ----
                for (int i = tid; i < RH * RW; i += NTHREADS) {
                    const int ry = i / RW, rx = i - ry * RW;
                    const int gy = reflect101(y0 - R + ry, H);
                    const int gx = reflect101(x0 - R + rx, W);
                    raw[i] = plane[(int64_t)gy * W + gx];
                }
                __syncthreads();
                // ---- horizontal pass: exact u32 (or f32) ----
                for (int i = tid; i < RH * TW; i += NTHREADS) {
                    const int ry = i >> 6, rx = i & 63;
                    const PIX *p = raw + ry * RW + rx + R;
                    if constexpr (std::is_same<PIX, uint16_t>::value) {
                        uint32_t acc = a.kfix[R] * (uint32_t)p[0];
#pragma unroll
                        for (int j = 1; j <= R; ++j)
                            acc += a.kfix[R + j] * ((uint32_t)p[-j] + (uint32_t)p[j]);
                        hs[i] = acc;
                    } else {
                        float acc = a.kflt[R] * p[0];
#pragma unroll
                        for (int j = 1; j <= R; ++j)
                            acc = acc + a.kflt[R + j] * (p[-j] + p[j]);
                        hs[i] = acc;
                    }
                }
                __syncthreads();
                // ---- vertical pass for this thread's PPT pixels ----
                HT col[PPT + 2 * R];
#pragma unroll
                for (int j = 0; j < PPT + 2 * R; ++j)
                    col[j] = hs[(wave * PPT + j) * TW + lane];
#pragma unroll
                for (int o = 0; o < PPT; ++o) {
                    float v;
                    if constexpr (std::is_same<PIX, uint16_t>::value) {
                        // V = sum k_j * H_j needs 48 bits: split H into 16-bit
                        // halves, two exact u32 sums, out = (V + 2^31) >> 32
                        uint32_t hi = a.kfix[R] * (col[o + R] >> 16);
                        uint32_t lo = a.kfix[R] * (col[o + R] & 0xffffu);
#pragma unroll
                        for (int j = 1; j <= R; ++j) {
                            hi += a.kfix[R + j] * ((col[o + R - j] >> 16) + (col[o + R + j] >> 16));
                            lo += a.kfix[R + j] * ((col[o + R - j] & 0xffffu) + (col[o + R + j] & 0xffffu));
                        }
                        const uint32_t out = (hi + (lo >> 16) + 0x8000u) >> 16;
                        v = (float)out;
                    } else {
                        float acc = a.kflt[R] * col[o + R];
#pragma unroll
                        for (int j = 1; j <= R; ++j)
                            acc = acc + a.kflt[R + j] * (col[o + R - j] + col[o + R + j]);
                        v = acc;
                    }
                    x[o][b] = (v - a.bg[b]) / a.sf[b];
                }
            } else {
#pragma unroll
                for (int o = 0; o < PPT; ++o) x[o][b] = 0.0f;
            }
        }
        (void)K;

        // ---- magnitude and L2 normalisation (decoding.py:106-110) ----
        float mag[PPT];
#pragma unroll
        for (int o = 0; o < PPT; ++o) {
            float acc = 0.0f;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) acc = acc + x[o][b] * x[o][b];
            float m = sqrtf(acc);
            if (m == 0.0f) m = 1.0f;
            mag[o] = m;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) x[o][b] = x[o][b] / m;
        }

        // ---- nearest codeword: float32 scores, running top-2 per pixel ----
        float best[PPT], second[PPT];
        int bi[PPT];
#pragma unroll
        for (int o = 0; o < PPT; ++o) { best[o] = -INFINITY; second[o] = -INFINITY; bi[o] = 0; }

        for (int c0 = 0; c0 < N; c0 += a.chunk) {
            const int cn = min(a.chunk, N - c0);
            if (!single_chunk) {
                __syncthreads();            // previous chunk fully consumed
                load_chunk(c0, cn);
                __syncthreads();
            }
            for (int n = 0; n < cn; ++n) {
                const float *c = cb + n * MAXB;
                float s[PPT];
                const float h = hn[n];
#pragma unroll
                for (int o = 0; o < PPT; ++o) s[o] = -h;
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    const float cv = c[b];
#pragma unroll
                    for (int o = 0; o < PPT; ++o) s[o] = fmaf(x[o][b], cv, s[o]);
                }
#pragma unroll
                for (int o = 0; o < PPT; ++o) {
                    second[o] = __builtin_amdgcn_fmed3f(best[o], second[o], s[o]);
                    bi[o] = s[o] > best[o] ? c0 + n : bi[o];
                    best[o] = fmaxf(best[o], s[o]);
                }
            }
        }

        // ---- exact float64 distance of the winner; near-ties re-scored ----
#pragma unroll
        for (int o = 0; o < PPT; ++o) {
            int win = bi[o];
            double d2 = 0.0;
            if (best[o] - second[o] < TIE_EPS) {
                // every codeword whose float32 score is within TIE_EPS of the
                // best is a candidate; float64 decides, lowest index on ties
                d2 = INFINITY;
                for (int n = 0; n < N; ++n) {
                    const double *row = a.cb64 + (size_t)n * B;
                    const float *crow = cb + n * MAXB;      // valid iff single_chunk
                    float s = single_chunk ? -hn[n] : -half_norm(row, B);
#pragma unroll
                    for (int b = 0; b < MAXB; ++b) {
                        const float cv = single_chunk ? crow[b] : (b < B ? (float)row[b] : 0.0f);
                        s = fmaf(x[o][b], cv, s);
                    }
                    if (s >= best[o] - TIE_EPS) {
                        double e = 0.0;
#pragma unroll
                        for (int b = 0; b < MAXB; ++b)
                            if (b < B) {
                                const double tdiff = (double)x[o][b] - row[b];
                                e = e + tdiff * tdiff;
                            }
                        if (e < d2) { d2 = e; win = n; }
                    }
                }
            } else {
                const double *row = a.cb64 + (size_t)win * B;
#pragma unroll
                for (int b = 0; b < MAXB; ++b)
                    if (b < B) {
                        const double tdiff = (double)x[o][b] - row[b];
                        d2 = d2 + tdiff * tdiff;
                    }
            }
            const double d = sqrt(d2);

            const int gy = y0 + wave * PPT + o, gx = x0 + lane;
            if (gy < H && gx < W) {
                const int64_t pix = ((int64_t)z * H + gy) * W + gx;
                int16_t code = d <= a.dthr ? (int16_t)win : (int16_t)-1;
                if (mag[o] < a.mthr) code = -1;
                a.decoded[pix] = code;
                a.mag[pix] = mag[o];
                if (a.dist_f64) reinterpret_cast<double *>(a.dist)[pix] = d;
                else reinterpret_cast<float *>(a.dist)[pix] = (float)d;
                if (a.traces) {
                    float *tp = a.traces + (int64_t)z * a.tr_z + (int64_t)gy * W + gx;
#pragma unroll
                    for (int b = 0; b < MAXB; ++b)
                        if (b < B) tp[(int64_t)b * a.tr_b] = x[o][b];
                }
            }
        }
        __syncthreads();   // tile buffers are reused by the next tile
    }
}

template <typename PIX, int R, int MAXB, int PPT>
size_t tile_lds_bytes()
{
    using HT = typename HType<PIX>::type;
    constexpr int TH = 4 * PPT, RH = TH + 2 * R, RW = TW + 2 * R;
    return (((size_t)RH * RW * sizeof(PIX) + 15) & ~size_t(15)) + (size_t)RH * TW * sizeof(HT);
}

template <typename PIX, int R, int MAXB, int PPT>
int launch(DecodeArgs &a, hipStream_t stream)
{
    constexpr int TH = 4 * PPT;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * a.Z;
    if (nt > INT32_MAX) return mf::fail(MF_EUNSUPPORTED, "too many tiles");
    a.n_tiles = (int)nt;

    const size_t tile_bytes = tile_lds_bytes<PIX, R, MAXB, PPT>();
    const size_t per_word = (size_t)MAXB * 4 + 4;
    const size_t budget = 64 * 1024;
    int chunk = (int)((budget - tile_bytes) / per_word);
    if (chunk > a.N) chunk = a.N;
    a.chunk = chunk;
    const size_t lds = tile_bytes + (size_t)chunk * per_word;

    const int cus = mf::device_cu_count();
    int64_t grid = (int64_t)cus * 4;
    if (grid > nt) grid = nt;
    grid = (grid + 7) & ~int64_t(7);

    auto kern = decode_kernel<PIX, R, MAXB, PPT>;
    if (lds > 48 * 1024)
        MF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, stream, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

template <typename PIX, int R>
int dispatch_bits(DecodeArgs &a, hipStream_t s)
{
    if (a.B <= 16) return launch<PIX, R, 16, 4>(a, s);
    if (a.B <= 24) return launch<PIX, R, 24, 2>(a, s);
    return launch<PIX, R, 32, 2>(a, s);
}

template <typename PIX>
int dispatch_radius(int R, DecodeArgs &a, hipStream_t s)
{
#ifdef MF_DEV_BUILD   // development: only the sigma=1 kernels, quick compile
    if (R == 2) return dispatch_bits<PIX, 2>(a, s);
    return mf::fail(MF_EUNSUPPORTED, "dev build: R=2 only");
#endif
    switch (R) {
    case 0: return dispatch_bits<PIX, 0>(a, s);
    case 1: return dispatch_bits<PIX, 1>(a, s);
    case 2: return dispatch_bits<PIX, 2>(a, s);
    case 3: return dispatch_bits<PIX, 3>(a, s);
    case 4: return dispatch_bits<PIX, 4>(a, s);
    }
    return mf::fail(MF_EUNSUPPORTED, "blur radius %d > %d", R, MF_MAX_RADIUS);
}

}  // namespace

extern "C" int mf_gaussian_taps(int ksize, double sigma, uint32_t *h_fixed, float *h_float)
{
    if (ksize < 1 || !(ksize & 1) || ksize > 2 * MF_MAX_RADIUS + 1)
        return mf::fail(MF_EINVAL, "ksize %d must be odd and <= %d", ksize, 2 * MF_MAX_RADIUS + 1);
    if (ksize == 1) {
        if (h_fixed) h_fixed[0] = 65536u;
        if (h_float) h_float[0] = 1.0f;
        return MF_OK;
    }
    // OpenCV getGaussianKernelBitExact: double-precision taps, normalised
    if (sigma <= 0) sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double k[2 * MF_MAX_RADIUS + 1], sum = 0.0;
    for (int i = 0; i < ksize; ++i) {
        const double xx = i - (ksize - 1) * 0.5;
        k[i] = std::exp(scale2x * xx * xx);
        sum += k[i];
    }
    const double inv = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) k[i] *= inv;
    if (h_float)
        for (int i = 0; i < ksize; ++i) h_float[i] = (float)k[i];
    if (h_fixed) {
        // getGaussianKernelFixedPoint_ED: 16 fraction bits, error diffusion,
        // centre tap takes the remainder so the taps sum to exactly 65536
        double err = 0.0;
        int64_t total = 0;
        const int half = ksize / 2;
        for (int i = 0; i < half; ++i) {
            const double adj = k[i] * 65536.0 + err;
            const int64_t v0 = (int64_t)std::nearbyint(adj);   // ties to even
            err = adj - (double)v0;
            h_fixed[i] = (uint32_t)v0;
            h_fixed[ksize - 1 - i] = (uint32_t)v0;
            total += 2 * v0;
        }
        h_fixed[half] = (uint32_t)(65536 - total);
    }
    return MF_OK;
}

extern "C" int mf_decode_pixels(const void *image, int image_dtype, int B, int Z, int H, int W,
                                int64_t img_bit_stride, int64_t img_z_stride,
                                const double *codebook, int N, const float *h_scale,
                                const float *h_background, double sigma,
                                double distance_threshold, float magnitude_threshold,
                                int16_t *decoded, float *magnitudes, void *distances,
                                int dist_dtype, float *traces, int64_t tr_z_stride,
                                int64_t tr_bit_stride, void *stream)
{
    if (!image || !codebook || !h_scale || !h_background || !decoded || !magnitudes || !distances)
        return mf::fail(MF_EINVAL, "null pointer argument");
    if (B < 1 || B > MF_MAX_BITS)
        return mf::fail(MF_EUNSUPPORTED, "bit count %d outside 1..%d", B, MF_MAX_BITS);
    if (Z < 1 || H < 1 || W < 1 || N < 1 || N > 32767)
        return mf::fail(MF_EINVAL, "bad shape Z=%d H=%d W=%d N=%d", Z, H, W, N);
    if ((int64_t)Z * H * W > INT32_MAX)
        return mf::fail(MF_EUNSUPPORTED, "volume exceeds 2^31 pixels");
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F32)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f32", image_dtype);
    if (dist_dtype != MF_DTYPE_F32 && dist_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "distance dtype %d not f32/f64", dist_dtype);
    if (!(sigma >= 0))
        return mf::fail(MF_EINVAL, "sigma must be >= 0");
    const int ksize = (int)(2 * std::ceil(2 * sigma) + 1);   // decoding.py:94
    const int R = ksize / 2;
    if (R > MF_MAX_RADIUS)
        return mf::fail(MF_EUNSUPPORTED, "lowPassSigma %g needs ksize %d > %d", sigma, ksize,
                        2 * MF_MAX_RADIUS + 1);
    if (R > H - 1 || R > W - 1)
        return mf::fail(MF_EINVAL, "image %dx%d smaller than blur radius %d", H, W, R);

    DecodeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.img = image;
    a.bit_stride = img_bit_stride;
    a.z_stride = img_z_stride;
    a.B = B; a.Z = Z; a.H = H; a.W = W; a.N = N;
    a.cb64 = codebook;
    for (int b = 0; b < B; ++b) { a.sf[b] = h_scale[b]; a.bg[b] = h_background[b]; }
    int rc = mf_gaussian_taps(ksize, sigma, a.kfix, a.kflt);
    if (rc) return rc;
    a.dthr = distance_threshold;
    a.mthr = magnitude_threshold;
    a.decoded = decoded;
    a.mag = magnitudes;
    a.dist = distances;
    a.dist_f64 = dist_dtype == MF_DTYPE_F64;
    a.traces = traces;
    a.tr_z = tr_z_stride;
    a.tr_b = tr_bit_stride;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (image_dtype == MF_DTYPE_U16) return dispatch_radius<uint16_t>(R, a, s);
    return dispatch_radius<float>(R, a, s);
}
