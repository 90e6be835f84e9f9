// Generic Gaussian low-pass for blur radii beyond the fused decode kernel's
// (lowPassSigma > 3, ksize > 13; merlin/util/decoding.py:93-97 accepts any
// sigma): the stack is blurred plane by plane into a second buffer, which
// mf_decode_pixels then takes with sigma = 0.  Same arithmetic as the fused
// path and the oracle: uint16 planes through OpenCV's 16.16 fixed-point kernel
// (exact u32 row sums, exact u64 column sums, (V + 2^31) >> 32), float32 planes
// through the float32 kernel with the centre tap first and the symmetric pairs
// summed before the product; BORDER_REFLECT_101.  Two plain kernels (row pass,
// column pass): this is the rare path, not a tuned one.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXK = 2 * MF_LOWPASS_MAX_RADIUS + 1;

struct Taps {
    uint32_t fixed[MAXK];
    float flt[MAXK];
};

__device__ __forceinline__ int reflect101(int i, int n)
{
    // radius <= n - 1 (checked by the host): one reflection suffices
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

__global__ void __launch_bounds__(NT)
row_pass_u16(const uint16_t *__restrict__ img, int64_t plane_stride, int H, int W, int R,
             const Taps t, uint32_t *__restrict__ sums)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint16_t *row = img + (int64_t)blockIdx.z * plane_stride + (int64_t)y * W;
    uint32_t acc = 0;                   // taps sum to 65536: < 2^32
    for (int j = -R; j <= R; ++j) acc += t.fixed[j + R] * (uint32_t)row[reflect101(x + j, W)];
    sums[((int64_t)blockIdx.z * H + y) * W + x] = acc;
}

__global__ void __launch_bounds__(NT)
col_pass_u16(const uint32_t *__restrict__ sums, int H, int W, int R, const Taps t,
             uint16_t *__restrict__ out)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint32_t *p = sums + (int64_t)blockIdx.z * H * W;
    unsigned long long acc = 0;         // < 2^48
    for (int j = -R; j <= R; ++j)
        acc += (unsigned long long)t.fixed[j + R] * p[(int64_t)reflect101(y + j, H) * W + x];
    const unsigned long long v = (acc + (1ull << 31)) >> 32;
    out[((int64_t)blockIdx.z * H + y) * W + x] = (uint16_t)(v > 65535ull ? 65535ull : v);
}

__global__ void __launch_bounds__(NT)
row_pass_f32(const float *__restrict__ img, int64_t plane_stride, int H, int W, int R,
             const Taps t, float *__restrict__ sums)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float *row = img + (int64_t)blockIdx.z * plane_stride + (int64_t)y * W;
    float acc = t.flt[R] * row[x];
    for (int j = 1; j <= R; ++j)
        acc = acc + t.flt[R + j] * (row[reflect101(x - j, W)] + row[reflect101(x + j, W)]);
    sums[((int64_t)blockIdx.z * H + y) * W + x] = acc;
}

__global__ void __launch_bounds__(NT)
col_pass_f32(const float *__restrict__ sums, int H, int W, int R, const Taps t,
             float *__restrict__ out)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float *p = sums + (int64_t)blockIdx.z * H * W;
    float acc = t.flt[R] * p[(int64_t)y * W + x];
    for (int j = 1; j <= R; ++j)
        acc = acc + t.flt[R + j] * (p[(int64_t)reflect101(y - j, H) * W + x] +
                                    p[(int64_t)reflect101(y + j, H) * W + x]);
    out[((int64_t)blockIdx.z * H + y) * W + x] = acc;
}

}  // namespace

extern "C" int64_t mf_lowpass_workspace_bytes(int image_dtype, int64_t n_planes, int H, int W)
{
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F32)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f32", image_dtype);
    if (n_planes < 0 || H < 1 || W < 1) return mf::fail(MF_EINVAL, "bad sizes");
    return n_planes * H * W * 4;        // u32 or f32 row sums
}

extern "C" int mf_lowpass_images(const void *image, int image_dtype, int64_t n_planes, int H, int W,
                                 int64_t plane_stride, double sigma, void *out, void *workspace,
                                 int64_t workspace_bytes, void *stream)
{
    if (n_planes == 0) return MF_OK;
    if (!image || !out || !workspace) return mf::fail(MF_EINVAL, "null pointer argument");
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F32)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f32", image_dtype);
    if (n_planes < 0 || n_planes > 65535 || H < 1 || H > 65535 || W < 1)
        return mf::fail(MF_EINVAL, "bad sizes");
    if (plane_stride < (int64_t)H * W) return mf::fail(MF_EINVAL, "plane stride below H * W");
    if (!(sigma >= 0)) return mf::fail(MF_EINVAL, "sigma must be >= 0");
    if (workspace_bytes < n_planes * H * W * 4)
        return mf::fail(MF_EWORKSPACE, "workspace %lld < %lld bytes", (long long)workspace_bytes,
                        (long long)(n_planes * H * W * 4));
    const int ksize = (int)(2 * std::ceil(2 * sigma) + 1);   // decoding.py:94
    const int R = ksize / 2;
    if (R > MF_LOWPASS_MAX_RADIUS)
        return mf::fail(MF_EUNSUPPORTED, "lowPassSigma %g needs ksize %d > %d", sigma, ksize, MAXK);
    if (R > H - 1 || R > W - 1)
        return mf::fail(MF_EINVAL, "image %dx%d smaller than blur radius %d", H, W, R);
    Taps t;
    const int rc = mf_gaussian_taps(ksize, sigma, t.fixed, t.flt);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((W + NT - 1) / NT), (unsigned)H, (unsigned)n_planes);
    if (image_dtype == MF_DTYPE_U16) {
        hipLaunchKernelGGL(row_pass_u16, grid, dim3(NT), 0, s, static_cast<const uint16_t *>(image),
                           plane_stride, H, W, R, t, static_cast<uint32_t *>(workspace));
        hipLaunchKernelGGL(col_pass_u16, grid, dim3(NT), 0, s,
                           static_cast<const uint32_t *>(workspace), H, W, R, t,
                           static_cast<uint16_t *>(out));
    } else {
        hipLaunchKernelGGL(row_pass_f32, grid, dim3(NT), 0, s, static_cast<const float *>(image),
                           plane_stride, H, W, R, t, static_cast<float *>(workspace));
        hipLaunchKernelGGL(col_pass_f32, grid, dim3(NT), 0, s,
                           static_cast<const float *>(workspace), H, W, R, t,
                           static_cast<float *>(out));
    }
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}
