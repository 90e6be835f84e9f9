// decode.hip -- fused low-pass + scale/background + magnitude + L2-normalise +
// nearest-codeword kernel for gfx950 (MI355X).
//
// Replaces merlin/util/decoding.py:93-129 (decode_pixels) for Z planes at once.
//
// One workgroup (256 threads = 4 waves) owns a 64 x (4*PPT) pixel tile; lane =
// column and every thread carries PPT vertically adjacent pixels, so each
// wave-level store is one contiguous row segment.  Workgroups are persistent
// (32 per CU, tiles in contiguous runs per XCD) and sized for 3 waves per SIMD
// at <= 16 bits (<= 168 VGPRs; 2 waves and 2 rows per thread above 16 bits).
//
//  * Low-pass: bits are staged GROUP (4) at a time.  Every thread pulls 16-byte
//    windows of the u16 rows straight from global memory (prefetched into
//    registers one group ahead; a window's address is a wave-uniform base plus
//    a 32-bit per-task offset computed once per tile -- 64-bit per-task
//    addresses hoisted out of the tile loop were what kept the kernel at two
//    waves per SIMD), does the horizontal pass in registers with
//    v_dot2_u32_u16 (two exact taps per instruction on the packed u16 pairs of
//    a window dword) and parks the u32 row sums in LDS; after one barrier each
//    thread reads its column of row sums back and finishes the vertical pass:
//    the 48-bit sum as three exact float64 FMAs, (V + 2^31) >> 32 as
//    trunc(fma(V, 2^-32, 0.5)).  The stack is read from HBM once.
//  * Scale / background / magnitude / normalise stay in registers
//    (x[PPT][MAXB]), in numpy's float32 operation order; the two divisions are
//    exact IEEE quotients from the rounded reciprocal and two residual steps.
//  * Nearest codeword: MERFISH codebooks are binary, so the score of codeword n
//    is v_n * sum_{on bits} x_b - |c_n|^2/2.  The sum over on-bits is a
//    [codewords x bits] x [bits x pixels] product with a 0/-t matrix: it runs on
//    the matrix cores (v_mfma_f32_32x32x16_f16) with 1024 x split into two
//    float16 terms (22 significant bits; r02 used three exact bfloat16 terms
//    and half as many again matrix instructions), float32 accumulation.  The
//    accumulator is the key (a float that
//    grows with the distance); each lane keeps the running minimum -- and the
//    runner-up unless the codebook's minimum distance rules a second candidate
//    out (EXCL) -- over its accumulator registers, 32-codeword tiles one after
//    the other; the two lanes that share a pixel merge through
//    v_permlane32_swap.  The winner's distance is recomputed in float64 exactly
//    as the reference does, and a pixel whose top two keys are closer than
//    TIE_EPS is re-decided by an exact float64 scan whenever it could pass the
//    distance threshold.
//
// Built with -ffp-contract=off: every float op below is the single IEEE
// operation it is written as; fused multiply-adds are explicit fmaf() calls.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include "common.h"

// One translation unit builds everything (development builds).  The shipped
// library compiles this file once per group of blur radii (-DMF_INST_GROUP=g:
// radii 2g and 2g + 1; group 0 also carries the host code), in parallel: the
// kernel templates then need a named namespace, and each group instantiates
// its own radii explicitly.
#ifdef MF_INST_GROUP
#define MF_DECODE_NS mf_decode
#define MF_OWN_RADIUS(r) ((r) / 2 == MF_INST_GROUP)
#define MF_HOST_CODE (MF_INST_GROUP == 0)
#else
#define MF_DECODE_NS
#define MF_HOST_CODE 1
#endif
namespace MF_DECODE_NS {

constexpr int TW = 64;              // tile width  = wave width
#ifndef MF_NWAVES
#define MF_NWAVES 4
#endif
constexpr int NWAVES = MF_NWAVES;   // waves per workgroup, stacked vertically
constexpr int NTHREADS = 64 * NWAVES;
#ifndef MF_GROUP
#define MF_GROUP 4
#endif
constexpr int GROUP = MF_GROUP;     // bits staged per barrier pair
#ifndef MF_TIE_EPS
#define MF_TIE_EPS 1.0e-5f
#endif
constexpr float TIE_EPS = MF_TIE_EPS;   // score gap below which float64 decides
constexpr double TIE_D2_MARGIN = 1.0e-4;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// Operands of the codeword search: 1 = x * 2^10 split into two float16 terms
// (round to nearest; 22 significant bits, the scaling keeps the low term out
// of the float16 subnormals), v_mfma_f32_32x32x16_f16, two matrix instructions
// per 32 x 32 tile; 0 = r02's exact three-term bfloat16 split, three
// instructions per tile.  The table built by mf_codebook_prepare matches.
#ifndef MF_SPLIT_F16
#define MF_SPLIT_F16 1
#endif
constexpr int NSPLIT = MF_SPLIT_F16 ? 2 : 3;
constexpr float X_SCALE = MF_SPLIT_F16 ? 1024.0f : 1.0f;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Device codebook table built by mf_codebook_prepare (offsets in bytes from
// the start of the blob, 256-byte aligned; a function of N and B only).
struct CbHeader {
    int32_t magic, N, B, Npad, KB, uniform;
    float key_c, key_scale;
    int64_t off_a, off_mask, off_v, off_hn, off_v64, off_cb64, total;
};
constexpr int32_t CB_MAGIC = 0x4d464342;   // "MFCB"

struct DecodeArgs {
    const void *img;
    int64_t bit_stride, z_stride;
    int B, Z, H, W, N, Npad, fast_ok, uniform, fast_div, exclusive;
    const unsigned char *cbt;          // codebook table blob
    int64_t off_a, off_mask, off_v, off_hn, off_v64;
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
    int64_t tr_z, tr_b, tr_p;
    int tr_vec;          // bit stride 1 and 16-byte aligned: float4 stores
    int tr_decoded_only; // skip the traces of pixels without a barcode
    int tiles_x, tiles_y;
    int n_tiles;
    int chunk;        // codewords per LDS chunk (multiple of 32)
};

// a uniform value the compiler must take as new at this point: keeps it from
// hoisting one lane mask per use out of the tile loop (and spilling them)
__device__ __forceinline__ int opaque_s(int v)
{
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    // a tile may overhang the image by more than the blur radius: those
    // positions feed no real pixel, they only have to stay inside the image
    return min(max(i, 0), n - 1);
}

template <typename PIX> struct HType { using type = uint32_t; };
template <> struct HType<float> { using type = float; };

// Running two smallest codeword keys of one pixel.  A key is a float that
// grows with the distance to the codeword (see the tile loop); its low 4
// mantissa bits carry the accumulator slot it came from (a relative
// perturbation of < 2^-19) and the 32-codeword tile is recorded whenever the
// smallest key changes.  Keys of either sign are compared as floats; four keys
// per step: 2 x (min3, med3) + min3.
struct Top2 {
    float best, second;
    int tile;
};
constexpr int KEY_SLOT_MASK = 15;
constexpr float KEY_INIT = 3.0e38f;

__device__ __forceinline__ float score_key(float s, int slot)
{
    return __int_as_float((__float_as_int(s) & ~KEY_SLOT_MASK) | slot);
}

// SEARCH_GROUPED (large codebooks): the keys carry no slot; the winner is found
// afterwards among the 16 codewords of the recorded tile that the lane holds
enum { SEARCH_TOP2 = 0, SEARCH_EXCLUSIVE = 1, SEARCH_GROUPED = 2 };
template <bool TAG>
__device__ __forceinline__ float tile_key(float s, int slot)
{
    if constexpr (TAG) return score_key(s, slot);
    else return s;
}

// v_min3_f32 without the canonicalisation the compiler puts in front of fminf
// (keys are never NaN)
__device__ __forceinline__ float fmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// IEEE square root of a squared distance.  The compiler's float64 sqrt is this
// Newton sequence wrapped in a rescaling for arguments below 2^-767; a sum of
// squares of differences of float32 values is either 0 or at least 2^-298, so
// the rescaling (two ldexp, a compare, two selects) never acts and is left out:
// same bits for every argument this kernel produces.
__device__ __forceinline__ double sqrt_d2(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;
    double h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double e = __builtin_fma(-g, g, a);
    g = __builtin_fma(e, h, g);
    e = __builtin_fma(-g, g, a);
    g = __builtin_fma(e, h, g);
    // 0 and +inf (no real codeword: d2 = INFINITY) map to themselves
    return (a == 0.0 || a == INFINITY) ? a : g;
}

// one 32 x 32 x 16 matrix instruction: codeword rows x 16 bits x pixel columns
__device__ __forceinline__ f32x16 mfma_tile(u32x4 a, u32x4 b, f32x16 c)
{
#if MF_SPLIT_F16
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}

// EXCL: the codebook and the threshold guarantee that no pixel has two
// codewords within reach (see codebook_exclusive): the runner-up is not needed.
template <bool EXCL>
__device__ __forceinline__ void top2_update4(Top2 &t, float k1, float k2, float k3, float k4)
{
    if constexpr (EXCL) {
        t.best = fmin3(fmin3(t.best, k1, k2), k3, k4);
        return;
    }
    const float m1 = __builtin_amdgcn_fmed3f(t.best, k1, k2);
    const float b1 = fmin3(t.best, k1, k2);
    const float m2 = __builtin_amdgcn_fmed3f(b1, k3, k4);
    t.best = fmin3(b1, k3, k4);
    t.second = fmin3(t.second, m1, m2);
}

// Correctly rounded n / d from the correctly rounded reciprocal r = RN(1/d):
// q0 = RN(n r) is within ~1 ulp, one residual step makes it faithful, and by
// Markstein's theorem a second residual step on a faithful quotient with a
// correctly rounded reciprocal returns RN(n / d) (no over/underflow here: the
// host only enables this path for 2^-60 < |d| < 2^60).  5 ops instead of the
// ~13-op IEEE division sequence.
__device__ __forceinline__ float div_by_recip(float n, float d, float r)
{
    float q = n * r;
    q = fmaf(fmaf(-q, d, n), r, q);
    q = fmaf(fmaf(-q, d, n), r, q);
    return q;
}

// Two quotients per instruction (v_pk_mul_f32 / v_pk_fma_f32): the same five
// roundings per element as div_by_recip.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 div_by_recip2(f32x2 n, f32x2 d, f32x2 r)
{
    f32x2 q = n * r;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, d, n), r, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, d, n), r, q);
    return q;
}

#ifndef MF_OPAQUE_B
#define MF_OPAQUE_B 1
#endif
#if MF_OPAQUE_B
#define MF_OPQ(v) opaque_s(v)
#else
#define MF_OPQ(v) (v)
#endif
// A/B switches of dev/mkvariant.sh (all on by default)
#ifndef MF_PK_F32
#define MF_PK_F32 1
#endif
#ifndef MF_VPASS_F64
#define MF_VPASS_F64 1
#endif
#ifndef MF_D2_FMA
#define MF_D2_FMA 1
#endif
#ifndef MF_HPASS_DOT2
#define MF_HPASS_DOT2 1
#endif
// two row-sum buffers in LDS, used alternately by the groups of bits: one
// barrier per group instead of two (the horizontal pass of group g+1 may start
// while other waves still read group g)
#ifndef MF_HS_DOUBLE
#define MF_HS_DOUBLE 0   // measured: no gain (1.191 vs 1.192 ms), twice the LDS
#endif
constexpr int HS_BUFFERS = MF_HS_DOUBLE ? 2 : 1;
// Per bit-count class (<= 16 bits / <= 32 bits): pixel rows per thread, waves
// per SIMD the register budget is set for, and whether the two 32-pixel
// accumulator tiles of a row are worked one after the other (16 registers
// instead of 32).  Measured per 7-z FOV (16 bits, 140 codewords): 4 rows x 2
// waves 1.40 ms, 4 rows x 3 waves sequential 1.22, 2 rows x 4 waves
// sequential 1.20; with the dot2 horizontal pass 1.16 and 1.19, and the
// three-wave kernel leaves room on the CUs for the region-table kernels of the
// previous FOV (1.93 vs 2.06 ms per FOV in the pipelined loop).
#ifndef MF_PPT16
#define MF_PPT16 4
#endif
#ifndef MF_WPS16
#define MF_WPS16 3
#endif
#ifndef MF_SEQ16
#define MF_SEQ16 1
#endif
#ifndef MF_PPT32
#define MF_PPT32 2
#endif
#ifndef MF_WPS32
#define MF_WPS32 2
#endif
#ifndef MF_SEQ32
#define MF_SEQ32 0
#endif
// (mixed-weight codebooks carry the per-codeword tables: one wave fewer)
constexpr int waves_per_simd(int maxb, bool uni)
{
    return maxb <= 16 ? (uni ? MF_WPS16 : (MF_WPS16 > 3 ? 3 : MF_WPS16)) : MF_WPS32;
}
constexpr bool tiles_in_sequence(int maxb) { return maxb <= 16 ? MF_SEQ16 : MF_SEQ32; }
// scheduling fences between the unrolled tasks / bits of the blur: keeps the
// compiler from interleaving them (register pressure)
#ifndef MF_SCHED_FENCE
#define MF_SCHED_FENCE 0
#endif
#if MF_SCHED_FENCE
#define MF_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MF_FENCE() ((void)0)
#endif

// instruction order of a step of the software-pipelined codeword scan: one
// matrix instruction, MF_PIPE_VALU vector instructions, four times over, then
// whatever is left; a full fence ends the step
#ifndef MF_PIPE_VALU
#define MF_PIPE_VALU 0
#endif
#if MF_PIPE_VALU > 0
#define MF_INTERLEAVE()                                                   \
    do {                                                                  \
        for (int q_ = 0; q_ < 4; ++q_) {                                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x002, MF_PIPE_VALU, 0); \
        }                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                \
    } while (0)
#else
#define MF_INTERLEAVE() __builtin_amdgcn_sched_barrier(0)
#endif

// timing-only ablations (wrong results): no workgroup barriers in the blur
#ifdef MF_ABL_NOBARRIER
#define MF_TILE_SYNC() ((void)0)
#else
#define MF_TILE_SYNC() __syncthreads()
#endif

// exact float64 squared distance between the normalised trace and codeword
// (mask, v): sum_b (x_b - c_b)^2 in bit order (sklearn's rdist loop)
template <int MAXB>
__device__ __forceinline__ double exact_d2(const float (&x)[MAXB], uint32_t mask, double v, int B)
{
    // bits >= B: x is 0 there (the vertical pass writes 0) and so is the mask
    // bit, so they add +0.0 to a non-negative sum -- no guard, which would cost
    // one lane mask per bit in (spilled) scalar registers
    (void)B;
    double d2 = 0.0;
#pragma unroll
    for (int b = 0; b < MAXB; ++b)
        {
#if MF_D2_FMA
            // x - c with c = v * bit (bit is 0 or 1, so the product is exact
            // and the fused result is the rounded difference)
            const double t = __builtin_fma(-(double)((mask >> b) & 1u), v, (double)x[b]);
#else
            const double c = (mask >> b) & 1u ? v : 0.0;
            const double t = (double)x[b] - c;
#endif
            d2 = d2 + t * t;
        }
    return d2;
}


// FULL: B == MAXB and the reciprocal division applies, both known at compile
// time (the common 16-bit / 32-bit codebooks): no per-bit guards.
template <typename PIX, int R, int MAXB, int PPT, bool UNI, bool FULL, int SM>
__global__ void __launch_bounds__(NTHREADS, waves_per_simd(MAXB, UNI))
decode_kernel(const DecodeArgs a)
{
    using HT = typename HType<PIX>::type;
    constexpr bool EXCL = SM == SEARCH_EXCLUSIVE;
    constexpr bool TAG = SM != SEARCH_GROUPED;
    constexpr bool IS_U16 = std::is_same<PIX, uint16_t>::value;
    constexpr int TH = NWAVES * PPT;
    constexpr int RH = TH + 2 * R;                 // rows staged per bit
    constexpr int RA = (R + 1) & ~1;               // radius rounded up to even
    constexpr int NW = 4 + 2 * RA;                 // window elements per task
    constexpr int WDW = NW * (int)sizeof(PIX) / 4; // window dwords
    constexpr int TASKS = GROUP * RH * 16;         // 4-output row pieces per group
    constexpr int NTASK = (TASKS + NTHREADS - 1) / NTHREADS;
    constexpr int NG = MAXB / GROUP;
    constexpr int KS = MAXB / 16;                  // MFMA k-steps
    constexpr size_t HS_BYTES = (size_t)HS_BUFFERS * GROUP * RH * TW * sizeof(HT);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    HT *hs0 = reinterpret_cast<HT *>(smem);                          // [buffers][GROUP][RH][TW]
    // codebook chunk: f16 [chunk][MAXB], v [chunk], hn [chunk]; a codebook that
    // does not fit has TWO such buffers (the next chunk is staged while the
    // current one is scanned), `bstride` bytes apart
    unsigned char *lds_a = smem + HS_BYTES;
    float *lds_v = reinterpret_cast<float *>(lds_a + (size_t)a.chunk * MAXB * 2);
    float *lds_hn = lds_v + a.chunk;
    const uint32_t bstride = (uint32_t)a.chunk * (MAXB * 2 + 8);
    // per-bit constants, each stored twice side by side: one 8-byte read puts
    // them where the packed two-row arithmetic wants them (no register moves)
    f32x2 *lds_sf = reinterpret_cast<f32x2 *>(lds_a + (a.chunk >= a.Npad ? 1u : 2u) * bstride);   // [MAXB] scale factors
    f32x2 *lds_bg = lds_sf + MAXB;                                   // [MAXB] backgrounds
    f32x2 *lds_rs = lds_bg + MAXB;                                   // [MAXB] RN(1 / scale)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int B = FULL ? MAXB : a.B, H = a.H, W = a.W;
    const bool fast_div = FULL || a.fast_div;
    const bool single_chunk = a.chunk >= a.Npad;
    const PIX *img = reinterpret_cast<const PIX *>(a.img);
    const uint32_t *g_mask = reinterpret_cast<const uint32_t *>(a.cbt + a.off_mask);
    const double *g_v64 = reinterpret_cast<const double *>(a.cbt + a.off_v64);
    const float key_scale = reinterpret_cast<const CbHeader *>(a.cbt)->key_scale;
    // 64-byte table rows (more than 16 bits): the four 16-byte pieces of row n
    // are stored at piece ^ ((n >> 2) & 3), which spreads the 32 rows one
    // ds_read_b128 touches over all LDS banks (mf_codebook_prepare)
    const int a_swz = MAXB == 32 ? ((lane & 31) >> 2) & 3 : 0;

    auto load_chunk = [&](int c0, int cn) {   // cn multiple of 32
        const u32x4 *src = reinterpret_cast<const u32x4 *>(a.cbt + a.off_a + (size_t)c0 * MAXB * 2);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds_a);
        for (int i = tid; i < cn * MAXB * 2 / 16; i += NTHREADS) dst[i] = src[i];
        const float *sv = reinterpret_cast<const float *>(a.cbt + a.off_v) + c0;
        const float *sh = reinterpret_cast<const float *>(a.cbt + a.off_hn) + c0;
        if constexpr (UNI) {
            // a uniform codebook needs neither table in the search; the slot
            // of lds_hn holds the on-bit masks for the exact distance instead
            // (used when the whole codebook is resident: single_chunk)
            uint32_t *lm = reinterpret_cast<uint32_t *>(lds_hn);
            for (int i = tid; i < cn; i += NTHREADS) lm[i] = g_mask[c0 + i];
        } else {
            for (int i = tid; i < cn; i += NTHREADS) { lds_v[i] = sv[i]; lds_hn[i] = sh[i]; }
        }
    };
    // The same chunk into the buffer at byte offset `boff`, asynchronously: loads
    // that write LDS directly (1 KB per wave instruction, lane i's 16 bytes at
    // base + 16 i; no registers), complete at the next s_waitcnt vmcnt(0).
    auto stage_chunk = [&](int c0, int cn, uint32_t boff) {
        typedef const __attribute__((address_space(1))) void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const unsigned char *ga = a.cbt + a.off_a + (size_t)c0 * MAXB * 2;
        const int nbytes = cn * MAXB * 2;                 // a multiple of 1024
        for (int off = wave * 1024; off < nbytes; off += NWAVES * 1024)
            __builtin_amdgcn_global_load_lds((gptr_t)(ga + off + lane * 16),
                                             (lptr_t)(lds_a + boff + off), 16, 0, 0);
        if constexpr (!UNI) {
            const unsigned char *gv = a.cbt + a.off_v + (size_t)c0 * 4;
            const unsigned char *gh = a.cbt + a.off_hn + (size_t)c0 * 4;
            unsigned char *lv = reinterpret_cast<unsigned char *>(lds_v) + boff;
            unsigned char *lh = reinterpret_cast<unsigned char *>(lds_hn) + boff;
            for (int off = wave * 256; off < cn * 4; off += NWAVES * 256)
                if (off + lane * 4 < cn * 4) {            // cn * 4 is a multiple of 128 only
                    __builtin_amdgcn_global_load_lds((gptr_t)(gv + off + lane * 4), (lptr_t)(lv + off), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(gh + off + lane * 4), (lptr_t)(lh + off), 4, 0, 0);
                }
        }
    };
    uint32_t cur = 0;      // byte offset of the buffer the next scan step reads
    // uniform codebook: every codeword carries the same value
    const double v64u = UNI ? g_v64[0] : 0.0;
    const uint32_t *lds_mask = reinterpret_cast<const uint32_t *>(lds_hn);
    if (single_chunk) load_chunk(0, a.Npad);   // made visible by the first barrier
    // per-bit constants live in LDS, not in 64 scalar registers
    if (tid < MAXB) {
        const float sfv = tid < B ? a.sf[tid] : 1.0f;
        const float bgv = tid < B ? a.bg[tid] : 0.0f;
        const float rsv = tid < B ? 1.0f / a.sf[tid] : 1.0f;     // IEEE division
        lds_sf[tid] = (f32x2){sfv, sfv};
        lds_bg[tid] = (f32x2){bgv, bgv};
        lds_rs[tid] = (f32x2){rsv, rsv};
    }

    // XCD-aware tile walk: blocks with the same (blockIdx % 8) share an XCD
    // and therefore an L2; give each such group one contiguous run of tiles
    // so that the halo rows/columns of neighbouring tiles hit that L2.
    const int xcd = blockIdx.x & 7;
    const int per_xcd = (a.n_tiles + 7) >> 3;
    const int t_lo = xcd * per_xcd;
    const int t_hi = min(a.n_tiles, t_lo + per_xcd);
    const int t_step = gridDim.x >> 3;

    // byte offset of task k's window inside a group's block of bit planes
    // (fast_ok guarantees it fits 32 bits)
    constexpr int NTASK_ = (GROUP * (NWAVES * PPT + 2 * R) * 16 + NTHREADS - 1) / NTHREADS;
    uint32_t voff[NTASK_];
#pragma unroll
    for (int k = 0; k < NTASK_; ++k) {
        const int q = tid + k * NTHREADS;
        const int rem = q % ((NWAVES * PPT + 2 * R) * 16);
        voff[k] = (uint32_t)(((int64_t)(q / ((NWAVES * PPT + 2 * R) * 16)) * a.bit_stride +
                              (int64_t)(rem >> 4) * W + 4 * (rem & 15)) *
                             (int64_t)sizeof(PIX));
    }

    for (int t = t_lo + (blockIdx.x >> 3); t < t_hi; t += t_step) {
        const int tx = t % a.tiles_x;
        const int ty = (t / a.tiles_x) % a.tiles_y;
        const int z = t / (a.tiles_x * a.tiles_y);
        const int x0 = tx * TW, y0 = ty * TH;
        const bool interior = a.fast_ok && y0 - R >= 0 && y0 + TH + R <= H && x0 - RA >= 0 &&
                              x0 + TW + RA <= W;

        float x[PPT][MAXB];
        uint32_t win[NTASK][WDW];
        double kd[2 * R + 1];
#pragma unroll
        // taps times 2^-32 (exact): the vertical sum then comes out as
        // V * 2^-32 + 0.5 without a scaling step of its own
        for (int j = 0; j <= 2 * R; ++j) kd[j] = (double)a.kfix[j] * 0x1p-32;

        // ---- global -> register windows for one group of bits ----
        // Interior tiles: the task's window sits at a tile-invariant byte
        // offset voff[k] (one register per task, worked out before the tile
        // loop) from a base that is uniform over the workgroup.
        auto prefetch = [&](int g) {
#ifdef MF_ABL_NOLOAD
#pragma unroll
            for (int k = 0; k < NTASK; ++k)
#pragma unroll
                for (int d = 0; d < WDW; ++d) win[k][d] = 0x00640064u + (uint32_t)(tid & 7) + g;
            return;
#endif
            if (interior) {
#ifdef MF_ABL_HOTLOAD   // timing only: every tile reads the windows of one fixed tile (cache hits)
                const char *gbase = reinterpret_cast<const char *>(img) +
                                    ((int64_t)(g * GROUP) * a.bit_stride +
                                     (int64_t)(64 - R) * W + (64 - RA)) * (int64_t)sizeof(PIX);
#else
                const char *gbase = reinterpret_cast<const char *>(img) +
                                    ((int64_t)(g * GROUP) * a.bit_stride + (int64_t)z * a.z_stride +
                                     (int64_t)(y0 - R) * W + (x0 - RA)) * (int64_t)sizeof(PIX);
#endif
                // every task of the group carries a real bit (the usual case:
                // B a multiple of the group): plain loads, no lane masks
                if (TASKS == NTASK * NTHREADS && (FULL || (g + 1) * GROUP <= MF_OPQ(B))) {
#pragma unroll
                    for (int k = 0; k < NTASK; ++k) {
#ifdef MF_ABL_ALIGNLOAD   // timing only: every window 16-byte aligned (wrong pixels)
                        const u32x4 w4 = *reinterpret_cast<const u32x4 *>(
                            reinterpret_cast<uintptr_t>(gbase + voff[k]) & ~uintptr_t(15));
                        win[k][0] = w4.x; win[k][1] = w4.y; win[k][2] = w4.z; win[k][3] = w4.w;
#else
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(gbase + voff[k]);
#pragma unroll
                        for (int d = 0; d < WDW; ++d) win[k][d] = src[d];
#endif
                    }
                    return;
                }
#pragma unroll
                for (int k = 0; k < NTASK; ++k) {
                    const bool live = (k + 1 < NTASK || tid + k * NTHREADS < TASKS) &&
                                      (FULL || g * GROUP + (tid + k * NTHREADS) / (RH * 16) < B);
                    if (live) {
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(gbase + voff[k]);
#pragma unroll
                        for (int d = 0; d < WDW; ++d) win[k][d] = src[d];
                    } else {
#pragma unroll
                        for (int d = 0; d < WDW; ++d) win[k][d] = 0u;
                    }
                }
                return;
            }
            // border tiles only.  The thread index is taken as new here so that
            // the index and reflection arithmetic stays inside this branch (the
            // compiler otherwise hoists it and runs it for every tile)
            int tid_l = tid;
            asm volatile("" : "+v"(tid_l));
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
                const int q = tid_l + k * NTHREADS;
                const int bit = g * GROUP + q / (RH * 16);
                const int rem = q % (RH * 16);
                const int row = rem >> 4, cg = rem & 15;
                if (q < TASKS && bit < B) {
                    const PIX *plane = img + (int64_t)bit * a.bit_stride + (int64_t)z * a.z_stride;
                    const int gy = y0 - R + row, gx = x0 + 4 * cg - RA;
                    {
                        int gys = gy, gxs = gx;
                        asm volatile("" : "+v"(gys), "+v"(gxs));
                        const PIX *rowp = plane + (int64_t)reflect101(gys, H) * W;
                        PIX tmp[NW];
#pragma unroll
                        for (int j = 0; j < NW; ++j) tmp[j] = rowp[reflect101(gxs + j, W)];
                        if constexpr (IS_U16) {
#pragma unroll
                            for (int d = 0; d < WDW; ++d)
                                win[k][d] = (uint32_t)tmp[2 * d] | ((uint32_t)tmp[2 * d + 1] << 16);
                        } else {
#pragma unroll
                            for (int d = 0; d < WDW; ++d) win[k][d] = __float_as_uint(tmp[d]);
                        }
                    }
                } else {
#pragma unroll
                    for (int d = 0; d < WDW; ++d) win[k][d] = 0u;
                }
            }
        };

        // ---- horizontal pass from the register windows into LDS ----
        auto hpass = [&](int g) {
            HT *hs = hs0 + (g % HS_BUFFERS) * (GROUP * RH * TW);
#ifdef MF_ABL_NOHPASS
            if (win[0][0] == 0x12345u) hs[tid] = (HT)win[1][1];
            return;
#endif
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
                const int q = tid + k * NTHREADS;
                if (q < TASKS) {
                    u32x4 v;
                    if constexpr (IS_U16) {
                        if constexpr (MF_HPASS_DOT2 && R > 0) {
                        // (R = 0: the single tap is 65536, no 16-bit operand)
                        // every window dword holds two pixels: one
                        // v_dot2_u32_u16 per dword with the pair of taps that
                        // falls on it (a zero tap outside the kernel); exact
                        // u32 sums as before (taps sum to 65536, pixels < 2^16)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = RA + j;
                            uint32_t acc = 0;
#pragma unroll
                            for (int d = 0; d < WDW; ++d) {
                                const int o0 = 2 * d - c, o1 = 2 * d + 1 - c;   // tap offsets
                                const bool in0 = o0 >= -R && o0 <= R, in1 = o1 >= -R && o1 <= R;
                                if (in0 || in1) {
                                    const uint32_t t0 = in0 ? a.kfix[R + (o0 < 0 ? -o0 : o0)] : 0u;
                                    const uint32_t t1 = in1 ? a.kfix[R + (o1 < 0 ? -o1 : o1)] : 0u;
                                    acc = __builtin_amdgcn_udot2(
                                        __builtin_bit_cast(u16x2, win[k][d]),
                                        __builtin_bit_cast(u16x2, t0 | (t1 << 16)), acc, false);
                                }
                            }
                            v[j] = acc;
                        }
                        } else {
                        uint32_t p[NW];
#pragma unroll
                        for (int d = 0; d < WDW; ++d) {
                            p[2 * d] = win[k][d] & 0xffffu;
                            p[2 * d + 1] = win[k][d] >> 16;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = RA + j;
                            uint32_t acc = __umul24(a.kfix[R], p[c]);
#pragma unroll
                            for (int i = 1; i <= R; ++i)
                                acc += __umul24(a.kfix[R + i], p[c - i] + p[c + i]);
                            v[j] = acc;
                        }
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int c = RA + j;
                            float acc = a.kflt[R] * __uint_as_float(win[k][c]);
#pragma unroll
                            for (int i = 1; i <= R; ++i)
                                acc = acc + a.kflt[R + i] * (__uint_as_float(win[k][c - i]) +
                                                             __uint_as_float(win[k][c + i]));
                            v[j] = __float_as_uint(acc);
                        }
                    }
                    // q = (bit_in_group * RH + row) * 16 + cg  ->  hs[4q .. 4q+3]
                    reinterpret_cast<u32x4 *>(hs)[q] = v;
                }
                MF_FENCE();
            }
        };

        prefetch(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * GROUP < MF_OPQ(B)) {
                hpass(g);
                MF_TILE_SYNC();
                const HT *hs = hs0 + (g % HS_BUFFERS) * (GROUP * RH * TW);
                if (g + 1 < NG && (g + 1) * GROUP < MF_OPQ(B)) prefetch(g + 1);
                // ---- vertical pass for this thread's PPT pixels ----
#pragma unroll
                for (int i = 0; i < GROUP; ++i) {
                    const int b = g * GROUP + i;
                    const f32x2 sf2 = lds_sf[b], bg2 = lds_bg[b], rs2 = lds_rs[b];
                    const float sfb = sf2.x, bgb = bg2.x, rsb = rs2.x;
                    HT col[PPT + 2 * R];
#pragma unroll
                    for (int j = 0; j < PPT + 2 * R; ++j)
                        col[j] = hs[(i * RH + wave * PPT + j) * TW + lane];
                    double cd[PPT + 2 * R];
#ifndef MF_ABL_NOVPASS
                    if constexpr (IS_U16 && MF_VPASS_F64) {
#pragma unroll
                        for (int j = 0; j < PPT + 2 * R; ++j) cd[j] = (double)col[j];
                    }
#endif
#pragma unroll
                    for (int o = 0; o < PPT; ++o) {
                        float v;
#ifdef MF_ABL_NOVPASS
                        if constexpr (IS_U16) {
                            v = (float)(col[o + R] >> 16);
                        } else
#endif
                        if constexpr (IS_U16 && MF_VPASS_F64) {
                            // V = sum k_j * H_j < 2^48 is exact in float64 (H < 2^32,
                            // pair sums < 2^33, taps < 2^16); V * 2^-32 + 0.5 is exact
                            // too (an integer multiple of 2^-32 below 2^17), so the
                            // truncating conversion is (V + 2^31) >> 32
                            // With the taps pre-scaled by 2^-32 and the sum started at
                            // 0.5, every partial sum is a multiple of 2^-32 below 2^17:
                            // still exact, one operation less per pixel.
                            double acc = __builtin_fma(kd[R], cd[o + R], 0.5);
#pragma unroll
                            for (int j = 1; j <= R; ++j)
                                acc = __builtin_fma(kd[R + j], cd[o + R - j] + cd[o + R + j], acc);
                            v = (float)(uint32_t)acc;
                        } else if constexpr (IS_U16) {
                            // V = sum k_j * H_j needs 48 bits: split H into 16-bit
                            // halves, two exact u32 sums, out = (V + 2^31) >> 32
                            // (operands < 2^24: full-rate 24-bit multiplies)
                            uint32_t hi = __umul24(a.kfix[R], col[o + R] >> 16);
                            uint32_t lo = __umul24(a.kfix[R], col[o + R] & 0xffffu);
#pragma unroll
                            for (int j = 1; j <= R; ++j) {
                                hi += __umul24(a.kfix[R + j],
                                               (col[o + R - j] >> 16) + (col[o + R + j] >> 16));
                                lo += __umul24(a.kfix[R + j], (col[o + R - j] & 0xffffu) +
                                                                  (col[o + R + j] & 0xffffu));
                            }
                            v = (float)((hi + (lo >> 16) + 0x8000u) >> 16);
                        } else {
                            float acc = a.kflt[R] * col[o + R];
#pragma unroll
                            for (int j = 1; j <= R; ++j)
                                acc = acc + a.kflt[R + j] * (col[o + R - j] + col[o + R + j]);
                            v = acc;
                        }
                        if constexpr (MF_PK_F32 && PPT % 2 == 0)
                            x[o][b] = v;
                        else
                            // bits >= B need no guard: their windows are loaded as
                            // zeros, their background is 0 and their scale 1 (set
                            // above), so the quotient is +0.0 by itself
                            x[o][b] = fast_div ? div_by_recip(v - bgb, sfb, rsb) : (v - bgb) / sfb;
                    }
                    if constexpr (MF_PK_F32 && PPT % 2 == 0) {
                        // pixel rows in pairs: one packed instruction per two quotients
#pragma unroll
                        for (int o = 0; o < PPT; o += 2) {
                            f32x2 n = {x[o][b], x[o + 1][b]};
                            n = n - bg2;
                            if (fast_div)
                                n = div_by_recip2(n, sf2, rs2);
                            else
                                n = n / sf2;
                            x[o][b] = n.x;              // bits >= B: +0.0 by itself, see above
                            x[o + 1][b] = n.y;
                        }
                    }
                    MF_FENCE();
                }
                if constexpr (HS_BUFFERS == 1) MF_TILE_SYNC();   // hs is rewritten by the next group
            } else {
#pragma unroll
                for (int i = 0; i < GROUP; ++i)
#pragma unroll
                    for (int o = 0; o < PPT; ++o) x[o][g * GROUP + i] = 0.0f;
            }
        }

        // ---- magnitude and L2 normalisation (decoding.py:106-110) ----
        float mag[PPT];
        if constexpr (MF_PK_F32 && PPT % 2 == 0) {
#pragma unroll
            for (int o = 0; o < PPT; o += 2) {
                f32x2 acc = {0.0f, 0.0f};
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    const f32x2 xx = {x[o][b], x[o + 1][b]};
                    acc = acc + xx * xx;               // unfused, in bit order
                }
                f32x2 m = {sqrtf(acc.x), sqrtf(acc.y)};
                if (m.x == 0.0f) m.x = 1.0f;
                if (m.y == 0.0f) m.y = 1.0f;
                mag[o] = m.x;
                mag[o + 1] = m.y;
#ifndef MF_ABL_NONORM
                if (fast_div && fminf(m.x, m.y) > 0x1p-60f && fmaxf(m.x, m.y) < 0x1p60f) {
                    const f32x2 rm = {1.0f / m.x, 1.0f / m.y};
#pragma unroll
                    for (int b = 0; b < MAXB; ++b) {
                        const f32x2 q = div_by_recip2((f32x2){x[o][b], x[o + 1][b]}, m, rm);
                        x[o][b] = q.x;
                        x[o + 1][b] = q.y;
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < MAXB; ++b) {
                        x[o][b] = x[o][b] / m.x;
                        x[o + 1][b] = x[o + 1][b] / m.y;
                    }
                }
#endif
            }
        } else
#pragma unroll
        for (int o = 0; o < PPT; ++o) {
            float acc = 0.0f;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) acc = acc + x[o][b] * x[o][b];
            float m = sqrtf(acc);
            if (m == 0.0f) m = 1.0f;
            mag[o] = m;
#ifndef MF_ABL_NONORM
            // m >= 2^-60 or exactly 1 whenever fast_div is set (see host)
            if (fast_div && m > 0x1p-60f && m < 0x1p60f) {
                const float rm = 1.0f / m;
#pragma unroll
                for (int b = 0; b < MAXB; ++b) x[o][b] = div_by_recip(x[o][b], m, rm);
            } else {
#pragma unroll
                for (int b = 0; b < MAXB; ++b) x[o][b] = x[o][b] / m;
            }
#endif
        }

        // ---- nearest codeword on the matrix cores, one pixel row at a time ----
        // bf16 operands: pk[s][ks][0..3] = bits 16ks+0..7, [4..7] = bits
        // 16ks+8..15 of split s.  After the half swap, regs [0..3] are the B
        // operand of pixel tile 0 (lanes 0-31's columns) and [4..7] of tile 1.
#pragma unroll
        for (int o = 0; o < PPT; ++o) {
            Top2 st[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                st[tt].best = KEY_INIT; st[tt].second = KEY_INIT; st[tt].tile = 0;
            }
            uint32_t pk[NSPLIT][KS][8];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float r[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) r[j] = x[o][16 * ks + j];
                if constexpr (MF_SPLIT_F16) {
                    // hi = RN16(1024 x), lo = RN16(1024 x - hi): the residual is
                    // exact in float32, so x ~ (hi + lo) / 1024 to 2^-22 relative;
                    // one v_cvt_pk_f16_f32 packs the two bits a B-operand dword holds
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        const f32x2 xs = (f32x2){r[2 * d], r[2 * d + 1]} * X_SCALE;
                        const f16x2 hi = __builtin_convertvector(xs, f16x2);
                        const f32x2 lo = xs - __builtin_convertvector(hi, f32x2);
                        pk[0][ks][d] = __builtin_bit_cast(uint32_t, hi);
                        pk[1][ks][d] = __builtin_bit_cast(uint32_t, __builtin_convertvector(lo, f16x2));
                    }
                } else {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int d = 0; d < 8; ++d)
                        pk[s % NSPLIT][ks][d] = __builtin_amdgcn_perm(__float_as_uint(r[2 * d + 1]),
                                                             __float_as_uint(r[2 * d]), 0x07060302u);
                    if (s < 2) {
                        // residual after the bf16 truncation (exact), two
                        // elements per v_pk_add_f32
#pragma unroll
                        for (int j = 0; j < 16; j += 2) {
                            const f32x2 hi = {__uint_as_float(__float_as_uint(r[j]) & 0xffff0000u),
                                              __uint_as_float(__float_as_uint(r[j + 1]) & 0xffff0000u)};
                            const f32x2 lo = (f32x2){r[j], r[j + 1]} - hi;
                            r[j] = lo.x;
                            r[j + 1] = lo.y;
                        }
                    }
                }
                }
#pragma unroll
                for (int s = 0; s < NSPLIT; ++s)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        auto sw = __builtin_amdgcn_permlane32_swap(pk[s][ks][d], pk[s][ks][4 + d],
                                                                   false, false);
                        pk[s][ks][d] = sw[0];
                        pk[s][ks][4 + d] = sw[1];
                    }
            }

#ifdef MF_ABL_NONN
            for (int c0 = 0; c0 < 0; c0 += a.chunk) {
#else
            // (every wave has passed the barriers of this tile's blur by now, so
            // nobody still reads the previous tile's last chunk)
            if (!single_chunk && o == 0) stage_chunk(0, min(a.chunk, a.Npad), cur);
            for (int c0 = 0; c0 < a.Npad; c0 += a.chunk) {
#endif
                const int cn = min(a.chunk, a.Npad - c0);
                if (!single_chunk) {
                    // this chunk was staged one step ago (the first one of a tile
                    // just above): wait for the own loads, then for everybody's --
                    // the barrier also says that every wave has finished the
                    // previous step, whose buffer now receives the next chunk
                    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0)
                    __syncthreads();
                    const int n0 = c0 + a.chunk;
                    if (n0 < a.Npad) stage_chunk(n0, min(a.chunk, a.Npad - n0), bstride - cur);
                    else if (o + 1 < PPT) stage_chunk(0, min(a.chunk, a.Npad), bstride - cur);
                }
                // A operand: 8 elements of codeword (ct + lane&31), bits 16ks + 8*half ..
                auto load_a = [&](int ct, u32x4 (&af)[KS]) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        af[ks] = *reinterpret_cast<const u32x4 *>(
                            lds_a + cur + ((size_t)(ct + (lane & 31)) * MAXB +
                                     8 * ((2 * ks + half) ^ a_swz)) * 2);
                };
                // the table entries are -t (t = 1, or 1/2 above 16 bits), so acc =
                // -t * (sum of the on bits).  For a uniform codebook that is the
                // key: d^2 / 2 = 1 + acc * v / t for a unit trace.  (The zero start
                // is an MFMA inline constant.)
                auto chain = [&](const u32x4 (&af)[KS], int tt, f32x16 &acc) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int s = 0; s < NSPLIT; ++s) {
                            u32x4 bv;
                            bv.x = pk[s][ks][4 * tt + 0]; bv.y = pk[s][ks][4 * tt + 1];
                            bv.z = pk[s][ks][4 * tt + 2]; bv.w = pk[s][ks][4 * tt + 3];
                            acc = mfma_tile(af[ks], bv, acc);
                        }
                };
                // lane holds codewords ct + (i&3) + 8*(i>>2) + 4*half of pixel
                // (lane&31) + 32*tt in accumulator slot i.  Zero padding rows of a
                // uniform table are caught after the merge.
                auto fold = [&](Top2 &t, const f32x16 &acc, int ct, auto trim) {
                    const int rows_left = a.N - (c0 + ct);     // real codewords from this tile on
                    const float before = t.best;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        // rows 16-31 of the codebook's last tile are padding when at
                        // most 16 codewords are left: accumulator slots 8-15 (gq 2,
                        // 3) cannot win (skipped where the codebook has few tiles;
                        // the pipelined scan stays free of branches)
                        if constexpr (decltype(trim)::value)
                            if (gq == 2 && rows_left <= 16) break;
                        float s4[4];
                        if constexpr (UNI) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) s4[i] = acc[4 * gq + i];
                        } else {
                            // (1 + |c|^2) / 2 - v * sum = d^2 / 2
                            const int nl = ct + 8 * gq + 4 * half;
                            const float4 vv = *reinterpret_cast<const float4 *>(
                                reinterpret_cast<const unsigned char *>(lds_v + nl) + cur);
                            const float4 hh = *reinterpret_cast<const float4 *>(
                                reinterpret_cast<const unsigned char *>(lds_hn + nl) + cur);
                            s4[0] = fmaf(vv.x, acc[4 * gq + 0], hh.x);
                            s4[1] = fmaf(vv.y, acc[4 * gq + 1], hh.y);
                            s4[2] = fmaf(vv.z, acc[4 * gq + 2], hh.z);
                            s4[3] = fmaf(vv.w, acc[4 * gq + 3], hh.w);
                        }
                        top2_update4<EXCL>(t, tile_key<TAG>(s4[0], 4 * gq + 0),
                                           tile_key<TAG>(s4[1], 4 * gq + 1),
                                           tile_key<TAG>(s4[2], 4 * gq + 2),
                                           tile_key<TAG>(s4[3], 4 * gq + 3));
                    }
                    t.tile = t.best != before ? c0 + ct : t.tile;
                };
                if constexpr (tiles_in_sequence(MAXB)) {
                    // one pixel tile at a time: 16 accumulator registers instead of
                    // 32; with three or four waves per SIMD the other waves cover
                    // the MFMA latency
                    for (int ct = 0; ct < cn; ct += 32) {
                        u32x4 af[KS];
                        load_a(ct, af);
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            f32x16 acc1;
                            chain(af, tt, acc1);
                            fold(st[tt], acc1, ct, std::true_type());
                        }
                    }
                } else {
                    // software pipeline over the (codeword tile, pixel tile) steps:
                    // the four matrix instructions of the next step (a dependent
                    // chain, 32 clocks each) are issued between the fold
                    // instructions of the current one, so that the matrix core and
                    // the vector ALU of a SIMD both work from ONE wave (left to
                    // itself the compiler issues the chain first: the wave then
                    // waits 96 clocks on it and the fold runs beside the last
                    // instruction only); the A operand of the next codeword tile
                    // is read from LDS a whole step ahead
                    f32x16 acc0, acc1;
                    u32x4 afa[KS], afb[KS];
                    load_a(0, afa);
                    chain(afa, 0, acc0);
                    // af: this codeword tile's A operand, afn: the next one's
                    auto step = [&](int ct, const u32x4 (&af)[KS], u32x4 (&afn)[KS], auto more) {
                        if constexpr (decltype(more)::value) load_a(ct + 32, afn);
                        chain(af, 1, acc1);
                        fold(st[0], acc0, ct, std::false_type());
                        MF_INTERLEAVE();
                        if constexpr (decltype(more)::value) chain(afn, 0, acc0);
                        fold(st[1], acc1, ct, std::false_type());
                        MF_INTERLEAVE();
                    };
                    // two codeword tiles per trip: the operand registers swap roles
                    int ct = 0;
                    for (; ct < cn - 64; ct += 64) {
                        step(ct, afa, afb, std::true_type());
                        step(ct + 32, afb, afa, std::true_type());
                    }
                    if (ct < cn - 32) {
                        step(ct, afa, afb, std::true_type());
                        step(ct + 32, afb, afa, std::false_type());
                    } else {
                        step(ct, afa, afb, std::false_type());
                    }
                }
                if (!single_chunk) cur = bstride - cur;     // the other buffer holds the next chunk
            }

            // ---- merge lane halves, exact float64 distance, thresholds ----
            // after the swaps both states in a lane belong to its own pixel
            auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(st[0].best),
                                                       __float_as_uint(st[1].best), false, false);
            auto ss = sb;
            if constexpr (!EXCL)
                ss = __builtin_amdgcn_permlane32_swap(__float_as_uint(st[0].second),
                                                      __float_as_uint(st[1].second), false, false);
            auto si = __builtin_amdgcn_permlane32_swap((unsigned)st[0].tile, (unsigned)st[1].tile,
                                                       false, false);
            // [0] came from the lane holding codeword rows 0-3 of every group of
            // eight (half 0), [1] from the lane holding rows 4-7
            const float b0 = __uint_as_float(sb[0]), b1 = __uint_as_float(sb[1]);
            const float s0 = __uint_as_float(ss[0]), s1 = __uint_as_float(ss[1]);
            const bool from1 = b1 < b0;
            const float kbest = from1 ? b1 : b0;
            // EXCL: no runner-up, so the tie test below never fires
            const float ksecond = EXCL ? KEY_INIT : fmin3(from1 ? b0 : b1, s0, s1);
            const int slot = __float_as_int(kbest) & KEY_SLOT_MASK;
            int win = (from1 ? (int)si[1] + 4 : (int)si[0]) + (TAG ? (slot & 3) + 8 * (slot >> 2) : 0);
            const float best = TAG ? __int_as_float(__float_as_int(kbest) & ~KEY_SLOT_MASK) : kbest;
            const float second = TAG ? __int_as_float(__float_as_int(ksecond) & ~KEY_SLOT_MASK)
                                     : ksecond;
            // TIE_EPS is in score (v * sum) units: a uniform key is t / v times
            // that; the slot bits move a key by < 2^-19 of itself
            const float eps = (UNI ? TIE_EPS * key_scale : TIE_EPS) + fabsf(second) * 0x1p-18f;

#ifdef MF_ABL_NOF64
            double d2 = (double)best;
#else
            double d2;
            if constexpr (TAG) {
                // the winner's mask and value: from LDS / a scalar for a resident
                // uniform codebook (no dependent global gather in front of the
                // stores), else from the table in global memory
                uint32_t wmask;
                double wv;
                if (UNI && single_chunk) {
                    wmask = lds_mask[win];
                    wv = v64u;
                } else {
                    wmask = g_mask[win];
                    wv = g_v64[win];
                }
                d2 = exact_d2<MAXB>(x[o], wmask, wv, B);
            } else {
                // the smallest key came from one of the 16 codewords this half
                // of the lane pair held of the recorded tile: exact distances of
                // those, in index order (the lowest index wins an exact tie)
                // (a zero padding row of a uniform table has key 0 and beats the
                // real codewords when all their sums are negative: no candidate
                // then, and the scan below decides)
                const int base = win;
                const bool real = !UNI || kbest < 0.0f;
                d2 = INFINITY;
                win = a.N;
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const int n = base + (i & 3) + 8 * (i >> 2);
                    if (real && n < a.N) {
                        const double e = exact_d2<MAXB>(x[o], g_mask[n], UNI ? v64u : g_v64[n], B);
                        if (e < d2) { d2 = e; win = n; }
                    }
                }
            }
#endif
            // a zero padding row can only "win" when every real sum is <= 0
            bool rescan = false;
#ifdef MF_ABL_NOSCAN
            if (win >= a.N) {
#else
            if (win >= a.N ||
                (second - best < eps && d2 <= a.dthr * a.dthr + TIE_D2_MARGIN)) {
#endif
                if constexpr (MAXB <= 16) {
                    // (at most 16 bits: a short codebook, and the cooperative
                    // scan below would cost this kernel registers it does not have)
                    d2 = INFINITY;
                    for (int n = 0; n < a.N; ++n) {
                        const double e = exact_d2<MAXB>(x[o], g_mask[n], g_v64[n], B);
                        if (e < d2) { d2 = e; win = n; }
                    }
                } else {
                    rescan = true;
                }
            }
            // ambiguous and possibly within the threshold: exact scan, lowest
            // index wins exact ties (the oracle's rule).  Above 16 bits the wave
            // takes its flagged pixels one at a time and spreads the N codewords of
            // each over its 64 lanes (a lane scanning alone keeps the other 63
            // waiting for N float64 distances: 10 % of the kernel at 10 000
            // codewords)
            if constexpr (MAXB > 16)
            for (unsigned long long todo = __ballot(rescan); todo; todo &= todo - 1) {
                const int src = __ffsll((long long)todo) - 1;
                float xs[MAXB];
#pragma unroll
                for (int b = 0; b < MAXB; ++b)
                    xs[b] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x[o][b]), src));
                double e_min = INFINITY;
                int n_min = a.N;
                for (int n = lane; n < a.N; n += 64) {        // ascending: a lane keeps its lowest index
                    const double e = exact_d2<MAXB>(xs, g_mask[n], g_v64[n], B);
                    if (e < e_min) { e_min = e; n_min = n; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double e2 = __shfl_xor(e_min, off);
                    const int n2 = __shfl_xor(n_min, off);
                    if (e2 < e_min || (e2 == e_min && n2 < n_min)) { e_min = e2; n_min = n2; }
                }
                if (lane == src) { d2 = e_min; win = n_min; }
            }
            const double d = sqrt_d2(d2);

            const int gy = y0 + wave * PPT + o, gx = x0 + lane;
#ifdef MF_ABL_NOSTORE
            if (gy < H && gx < W && d == 12345.0) {
#else
            if (gy < H && gx < W) {
#endif
                // the volume has fewer than 2^31 pixels (checked by the host): the
                // pixel index is 32-bit arithmetic, one shift-add per output array
                const uint32_t pin = (uint32_t)gy * (uint32_t)W + (uint32_t)gx;
                const uint32_t pix = (uint32_t)z * (uint32_t)(H * W) + pin;
                int16_t code = d <= a.dthr ? (int16_t)win : (int16_t)-1;
                if (mag[o] < a.mthr) code = -1;
                a.decoded[pix] = code;
                a.mag[pix] = mag[o];
                if (a.dist_f64) reinterpret_cast<double *>(a.dist)[pix] = d;
                else reinterpret_cast<float *>(a.dist)[pix] = (float)d;
                if (a.traces && (!a.tr_decoded_only || code >= 0)) {
                    float *tp = a.traces + (int64_t)z * a.tr_z + (int64_t)pin * a.tr_p;
                    // (B and the bit stride taken as new here: otherwise one lane
                    // mask and one offset per bit are hoisted out of the tile loop
                    // and live in spilled scalar registers)
                    const int Bs = MF_OPQ(B);
                    if (a.tr_vec) {
                        // pixel-major layout: the pixel's B floats are contiguous
#pragma unroll
                        for (int b = 0; b < MAXB; b += 4)
                            if (b < Bs)
                                *reinterpret_cast<float4 *>(tp + b) =
                                    make_float4(x[o][b], x[o][b + 1], x[o][b + 2], x[o][b + 3]);
                    } else {
                        float *q = tp;
#pragma unroll
                        for (int b = 0; b < MAXB; ++b) {
                            if (b < Bs) *q = x[o][b];
                            q += a.tr_b;
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the pixel rows from interleaving
        }
    }
}

template <typename PIX, int R, int MAXB, int PPT, bool UNI, bool FULL, int SM>
int launch(DecodeArgs &a, hipStream_t stream)
{
    using HT = typename HType<PIX>::type;
    constexpr int TH = NWAVES * PPT, RH = TH + 2 * R;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    const int64_t nt = (int64_t)a.tiles_x * a.tiles_y * a.Z;
    if (nt > INT32_MAX) return mf::fail(MF_EUNSUPPORTED, "too many tiles");
    a.n_tiles = (int)nt;

    const size_t hs_bytes = (size_t)HS_BUFFERS * GROUP * RH * TW * sizeof(HT);
    const size_t per_word = (size_t)MAXB * 2 + 8;
    // as many workgroups per CU as the register budget allows waves per SIMD
    // (160 KB of LDS shared between them)
    const size_t budget = (size_t)160 * 1024 / (waves_per_simd(MAXB, UNI) * 4 / NWAVES);
    int chunk = (int)((budget - hs_bytes - 768) / per_word) & ~31;
    int n_buffers = 1;
    if (chunk >= a.Npad) {
        chunk = a.Npad;
    } else {
        // two buffers (the next chunk is staged while this one is scanned) of
        // equal chunks, each a multiple of 32 codewords
        n_buffers = 2;
        chunk = (int)((budget - hs_bytes - 768) / 2 / per_word) & ~31;
        if (chunk < 32) return mf::fail(MF_EUNSUPPORTED, "no room for a codebook chunk in LDS");
        const int n_chunks = (a.Npad + chunk - 1) / chunk;
        chunk = ((a.Npad / 32 + n_chunks - 1) / n_chunks) * 32;
    }
    a.chunk = chunk;
    size_t lds = hs_bytes + (size_t)n_buffers * chunk * per_word + 3 * MAXB * 2 * sizeof(float);
#ifdef MF_LDS_MIN   // development: fewer workgroups per CU (room for other streams' kernels)
    if (lds < MF_LDS_MIN) lds = MF_LDS_MIN;
#endif

    const int cus = mf::device_cu_count();
    // far more workgroups than resident slots: the dispatcher balances the
    // CUs much better than an exactly-resident persistent grid (measured:
    // 2.86 ms at 2 WGs/CU vs 2.19 ms at 32 WGs/CU for one 7-z FOV)
    int64_t grid = (int64_t)cus * 32;
    if (grid > nt) grid = nt;
    grid = (grid + 7) & ~int64_t(7);

    auto kern = decode_kernel<PIX, R, MAXB, PPT, UNI, FULL, SM>;
    if (lds > 48 * 1024)
        MF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), lds, stream, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

// The FULL specialisation executes fewer instructions but the compiler spills
// registers in it: measured slower (2.04 vs 1.92 ms per 7-z FOV), so it is
// only built on request.
#ifdef MF_USE_FULL
constexpr bool USE_FULL = true;
#else
constexpr bool USE_FULL = false;
#endif

#ifndef MF_GROUPED_MIN_N
#define MF_GROUPED_MIN_N 3072
#endif
constexpr int GROUPED_MIN_N = MF_GROUPED_MIN_N;

template <typename PIX, int R, int MAXB, int PPT>
int dispatch_kind(DecodeArgs &a, hipStream_t s)
{
    if constexpr (USE_FULL) {
        if (a.B == MAXB && a.fast_div)
            return a.uniform ? launch<PIX, R, MAXB, PPT, true, true, SEARCH_TOP2>(a, s)
                             : launch<PIX, R, MAXB, PPT, false, true, SEARCH_TOP2>(a, s);
    }
    if (a.uniform && a.exclusive)
        return launch<PIX, R, MAXB, PPT, true, false, SEARCH_EXCLUSIVE>(a, s);
    if constexpr (MAXB == 32) {
        // from a few thousand codewords on, a slot tag per score costs more than
        // 16 exact distances per pixel at the end
        if (a.N >= GROUPED_MIN_N)
            return a.uniform ? launch<PIX, R, MAXB, PPT, true, false, SEARCH_GROUPED>(a, s)
                             : launch<PIX, R, MAXB, PPT, false, false, SEARCH_GROUPED>(a, s);
    }
    return a.uniform ? launch<PIX, R, MAXB, PPT, true, false, SEARCH_TOP2>(a, s)
                     : launch<PIX, R, MAXB, PPT, false, false, SEARCH_TOP2>(a, s);
}

template <typename PIX, int R>
int dispatch_bits(DecodeArgs &a, hipStream_t s)
{
    if (a.B <= 16) return dispatch_kind<PIX, R, 16, MF_PPT16>(a, s);
    return dispatch_kind<PIX, R, 32, MF_PPT32>(a, s);
}

#ifdef MF_INST_GROUP
#define MF_INST_DEF(R)                                                            \
    template int dispatch_bits<uint16_t, R>(DecodeArgs &, hipStream_t);           \
    template int dispatch_bits<float, R>(DecodeArgs &, hipStream_t);
#define MF_INST_DECL(R)                                                           \
    extern template int dispatch_bits<uint16_t, R>(DecodeArgs &, hipStream_t);    \
    extern template int dispatch_bits<float, R>(DecodeArgs &, hipStream_t);
#if MF_OWN_RADIUS(0)
MF_INST_DEF(0)
#else
MF_INST_DECL(0)
#endif
#if MF_OWN_RADIUS(1)
MF_INST_DEF(1)
#else
MF_INST_DECL(1)
#endif
#if MF_OWN_RADIUS(2)
MF_INST_DEF(2)
#else
MF_INST_DECL(2)
#endif
#if MF_OWN_RADIUS(3)
MF_INST_DEF(3)
#else
MF_INST_DECL(3)
#endif
#if MF_OWN_RADIUS(4)
MF_INST_DEF(4)
#else
MF_INST_DECL(4)
#endif
#if MF_OWN_RADIUS(5)
MF_INST_DEF(5)
#else
MF_INST_DECL(5)
#endif
#if MF_OWN_RADIUS(6)
MF_INST_DEF(6)
#else
MF_INST_DECL(6)
#endif
#endif

#if MF_HOST_CODE
template <typename PIX>
int dispatch_radius(int R, DecodeArgs &a, hipStream_t s)
{
#ifdef MF_DEV_BUILD   // development: only the sigma=1 u16 kernels, quick compile
    if constexpr (std::is_same<PIX, uint16_t>::value)
        if (R == 2) return dispatch_bits<PIX, 2>(a, s);
    return mf::fail(MF_EUNSUPPORTED, "dev build: u16, R=2 only");
#else
    switch (R) {
    case 0: return dispatch_bits<PIX, 0>(a, s);
    case 1: return dispatch_bits<PIX, 1>(a, s);
    case 2: return dispatch_bits<PIX, 2>(a, s);
    case 3: return dispatch_bits<PIX, 3>(a, s);
    case 4: return dispatch_bits<PIX, 4>(a, s);
    case 5: return dispatch_bits<PIX, 5>(a, s);
    case 6: return dispatch_bits<PIX, 6>(a, s);
    }
    return mf::fail(MF_EUNSUPPORTED, "blur radius %d > %d", R, MF_MAX_RADIUS);
#endif
}

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// Can two codewords of a uniform codebook (weight w, all values 1/sqrt(w),
// minimum Hamming distance hd) both lie within `thr` of one unit vector?  Two
// codewords share at most w - hd/2 on-bits: cos(theta) <= 1 - hd / (2 w).  A
// unit trace x at distance d1 <= thr of c1 has x.c1 = a1 >= 1 - thr^2 / 2 and
// then x.c2 <= cos(theta - acos(a1)).  If a1 exceeds that by more than the
// float32 keys can blur (slot bits 2^-19 of a key, accumulation rounding:
// < 5e-6 in these units; 2e-5 asked for), the float-key winner IS the exact
// winner for every pixel that can decode, and the kernel needs no runner-up.
// MHD4 (w = 4, hd = 4): margin 3.9e-5 at thr = 0.5176, 9.7e-4 at 0.5167; the two
// meet at thr = sqrt(2 - sqrt 3) = 0.51764, which is where 0.5176 comes from.
bool codebook_exclusive(int weight, int min_hd, double thr)
{
    if (weight < 1 || min_hd < 2 || !(thr >= 0)) return false;
    const double a1 = 1.0 - 0.5 * thr * thr;
    if (a1 <= 0.0) return false;
    const double cos_theta = 1.0 - (double)min_hd / (2.0 * weight);
    const double theta = std::acos(cos_theta < -1.0 ? -1.0 : cos_theta);
    const double phi = std::acos(a1 > 1.0 ? 1.0 : a1);
    if (phi >= 0.5 * theta) return false;
    return a1 - std::cos(theta - phi) >= 2.0e-5;
}

void table_layout(int N, int B, CbHeader &h)
{
    std::memset(&h, 0, sizeof(h));
    h.magic = CB_MAGIC;
    h.N = N;
    h.B = B;
    h.Npad = (N + 31) & ~31;
    h.KB = B <= 16 ? 16 : 32;
    size_t off = align256(sizeof(CbHeader));
    h.off_a = (int64_t)off;    off += align256((size_t)h.Npad * h.KB * 2);
    h.off_mask = (int64_t)off; off += align256((size_t)h.Npad * 4);
    h.off_v = (int64_t)off;    off += align256((size_t)h.Npad * 4);
    h.off_hn = (int64_t)off;   off += align256((size_t)h.Npad * 4);
    h.off_v64 = (int64_t)off;  off += align256((size_t)h.Npad * 8);
    h.off_cb64 = (int64_t)off; off += align256((size_t)N * B * 8);
    h.total = (int64_t)off;
}

#endif  // MF_HOST_CODE

}  // namespace

#if MF_HOST_CODE
#ifdef MF_INST_GROUP
using namespace mf_decode;
#endif

extern "C" int64_t mf_codebook_table_bytes(int N, int B)
{
    if (N < 1 || N > 32767 || B < 1 || B > MF_MAX_BITS)
        return mf::fail(MF_EUNSUPPORTED, "codebook shape N=%d B=%d unsupported", N, B);
    CbHeader h;
    table_layout(N, B, h);
    return h.total;
}

extern "C" int mf_codebook_prepare(const double *h_codebook, int N, int B, void *table,
                                   int64_t table_bytes)
{
    if (!h_codebook || !table) return mf::fail(MF_EINVAL, "null pointer argument");
    if (N < 1 || N > 32767 || B < 1 || B > MF_MAX_BITS)
        return mf::fail(MF_EUNSUPPORTED, "codebook shape N=%d B=%d unsupported", N, B);
    CbHeader h;
    table_layout(N, B, h);
    if (table_bytes < h.total)
        return mf::fail(MF_EWORKSPACE, "codebook table %lld < %lld bytes", (long long)table_bytes,
                        (long long)h.total);
    std::vector<unsigned char> blob((size_t)h.total, 0);
    uint16_t *am = reinterpret_cast<uint16_t *>(blob.data() + h.off_a);
    uint32_t *mask = reinterpret_cast<uint32_t *>(blob.data() + h.off_mask);
    float *vt = reinterpret_cast<float *>(blob.data() + h.off_v);
    float *hn = reinterpret_cast<float *>(blob.data() + h.off_hn);
    double *v64 = reinterpret_cast<double *>(blob.data() + h.off_v64);
    double *cb = reinterpret_cast<double *>(blob.data() + h.off_cb64);
    int uniform = 1;
    double v_first = 0.0;
    for (int n = 0; n < h.Npad; ++n) {
        hn[n] = 1.0e30f;            // padding rows (v = 0) can never win
        if (n >= N) continue;
        double v = 0.0, sq = 0.0;
        uint32_t m = 0;
        for (int b = 0; b < B; ++b) {
            const double c = h_codebook[(size_t)n * B + b];
            cb[(size_t)n * B + b] = c;
            if (c != 0.0) {
                // MERFISH codewords are binary: every non-zero entry of a
                // normalised row carries the same value
                if (v != 0.0 && c != v)
                    return mf::fail(MF_EUNSUPPORTED,
                                    "codeword %d is not binary-valued (entries %g and %g)", n, v, c);
                v = c;
                m |= 1u << b;
                // -1.0, or -0.5 above 16 bits, as float16 (bfloat16 for the
                // three-term split)
                if (h.KB == 16)
                    am[(size_t)n * 16 + b] = MF_SPLIT_F16 ? 0xbc00 : 0xbf80;
                else        // swizzled 16-byte pieces, see a_swz in the kernel
                    am[(size_t)n * 32 + 8 * ((b >> 3) ^ ((n >> 2) & 3)) + (b & 7)] =
                        MF_SPLIT_F16 ? 0xb800 : 0xbf00;
            }
            sq += c * c;
        }
        mask[n] = m;
        v64[n] = v;
        // score = vt * accumulator + hn; the accumulator is -t * X_SCALE * sum
        vt[n] = (float)((h.KB == 16 ? v : 2.0 * v) / (double)X_SCALE);
        hn[n] = (float)(0.5 + 0.5 * sq);
        if (n == 0) v_first = v;
        // uniform: every row has the same value and the same weight, so the
        // raw on-bit sum orders the codewords
        if (v != v_first || v <= 0.0 || __builtin_popcount(m) != __builtin_popcount(mask[0]))
            uniform = 0;
    }
    h.uniform = uniform;
    // minimum Hamming distance of a uniform codebook (for codebook_exclusive);
    // quadratic, so only up to 4096 codewords
    int min_hd = 0;
    if (uniform && N >= 2 && N <= 4096) {
        min_hd = 255;
        for (int i = 0; i < N && min_hd > 0; ++i)
            for (int j = i + 1; j < N; ++j) {
                const int d = __builtin_popcount(mask[i] ^ mask[j]);
                if (d < min_hd) min_hd = d;
            }
    }
    // a uniform key (-t * sum) is t / v = t * sqrt(weight) times the score
    h.key_c = 0.0f;
    h.key_scale = (float)((h.KB == 16 ? 1.0 : 0.5) * (double)X_SCALE *
                          std::sqrt((double)__builtin_popcount(mask[0])));
    std::memcpy(blob.data(), &h, sizeof(h));
    MF_HIP_CHECK(hipMemcpy(table, blob.data(), (size_t)h.total, hipMemcpyHostToDevice));
    return uniform ? (MF_CODEBOOK_UNIFORM | (min_hd << MF_CODEBOOK_MINHD_SHIFT) |
                      (__builtin_popcount(mask[0]) << MF_CODEBOOK_WEIGHT_SHIFT))
                   : 0;
}

extern "C" int mf_gaussian_taps(int ksize, double sigma, uint32_t *h_fixed, float *h_float)
{
    if (ksize < 1 || !(ksize & 1) || ksize > 2 * MF_LOWPASS_MAX_RADIUS + 1)
        return mf::fail(MF_EINVAL, "ksize %d must be odd and <= %d", ksize,
                        2 * MF_LOWPASS_MAX_RADIUS + 1);
    if (ksize == 1) {
        if (h_fixed) h_fixed[0] = 65536u;
        if (h_float) h_float[0] = 1.0f;
        return MF_OK;
    }
    // OpenCV getGaussianKernelBitExact: double-precision taps, normalised
    if (sigma <= 0) sigma = ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2x = -0.5 / (sigma * sigma);
    double k[2 * MF_LOWPASS_MAX_RADIUS + 1], sum = 0.0;
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
                                const void *codebook_table, int N, int codebook_flags,
                                const float *h_scale,
                                const float *h_background, double sigma,
                                double distance_threshold, float magnitude_threshold,
                                int16_t *decoded, float *magnitudes, void *distances,
                                int dist_dtype, float *traces, int64_t tr_z_stride,
                                int64_t tr_bit_stride, int64_t tr_pix_stride, int flags,
                                void *stream)
{
    if (!image || !codebook_table || !h_scale || !h_background || !decoded || !magnitudes ||
        !distances)
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
    if (!(sigma >= 0)) return mf::fail(MF_EINVAL, "sigma must be >= 0");
    if (flags & ~(MF_DECODE_TRACES_DECODED_ONLY | MF_DECODE_KEEP_RUNNER_UP))
        return mf::fail(MF_EINVAL, "unknown flags %d", flags);
    if (traces && tr_pix_stride < 1) return mf::fail(MF_EINVAL, "traces pixel stride must be >= 1");
    const int ksize = (int)(2 * std::ceil(2 * sigma) + 1);   // decoding.py:94
    const int R = ksize / 2;
    if (R > MF_MAX_RADIUS)
        return mf::fail(MF_EUNSUPPORTED,
                        "lowPassSigma %g needs ksize %d > %d: blur the stack with "
                        "mf_lowpass_images and pass sigma = 0",
                        sigma, ksize, 2 * MF_MAX_RADIUS + 1);
    if (R > H - 1 || R > W - 1)
        return mf::fail(MF_EINVAL, "image %dx%d smaller than blur radius %d", H, W, R);

    CbHeader h;
    table_layout(N, B, h);
    DecodeArgs a;
    std::memset(&a, 0, sizeof(a));
    a.img = image;
    a.bit_stride = img_bit_stride;
    a.z_stride = img_z_stride;
    a.B = B; a.Z = Z; a.H = H; a.W = W; a.N = N; a.Npad = h.Npad;
    a.uniform = (codebook_flags & MF_CODEBOOK_UNIFORM) != 0;
    a.exclusive = a.uniform && !(flags & MF_DECODE_KEEP_RUNNER_UP) &&
                  codebook_exclusive((codebook_flags >> MF_CODEBOOK_WEIGHT_SHIFT) & 0xff,
                                     (codebook_flags >> MF_CODEBOOK_MINHD_SHIFT) & 0xff,
                                     distance_threshold);
    // the 4-byte window loads need even element offsets for u16 rows
    a.fast_ok = image_dtype == MF_DTYPE_F32 ||
                (!(W & 1) && !((img_bit_stride | img_z_stride) & 1) &&
                 !(reinterpret_cast<uintptr_t>(image) & 3));
    // interior tiles address a task's window by a 32-bit byte offset inside a
    // group of GROUP bit planes
    if (img_bit_stride < 0 ||
        ((int64_t)GROUP * img_bit_stride + (int64_t)64 * W + 64) * 4 >= ((int64_t)1 << 32))
        a.fast_ok = 0;
    a.cbt = reinterpret_cast<const unsigned char *>(codebook_table);
    a.off_a = h.off_a; a.off_mask = h.off_mask; a.off_v = h.off_v; a.off_hn = h.off_hn;
    a.off_v64 = h.off_v64;
    a.fast_div = 1;
    for (int b = 0; b < B; ++b) {
        a.sf[b] = h_scale[b];
        a.bg[b] = h_background[b];
        const float s_abs = std::fabs(h_scale[b]);
        if (!(s_abs > 0x1p-60f && s_abs < 0x1p60f) || !(std::fabs(h_background[b]) < 0x1p60f))
            a.fast_div = 0;      // odd scale factors: plain IEEE division
    }
    int rc = mf_gaussian_taps(ksize, sigma, a.kfix, a.kflt);
    if (rc) return rc;
    int Reff = R;
    if (image_dtype == MF_DTYPE_U16 && R > 0 && a.kfix[R] >= 65536u) {
        // A very narrow Gaussian (sigma < ~0.21): every side tap rounds to 0 and
        // the centre tap is exactly 2^16, which does not fit the 16-bit halves
        // the v_dot2_u32_u16 row pass packs its taps into.  Such a kernel is
        // the identity on u16 pixels ((p * 2^32 + 2^31) >> 32 = p): run the
        // single-tap kernel, which carries 65536 as a 24-bit multiplier.
        Reff = 0;
        rc = mf_gaussian_taps(1, sigma, a.kfix, a.kflt);
        if (rc) return rc;
    }
    for (int j = 0; j <= 2 * Reff; ++j)
        if (image_dtype == MF_DTYPE_U16 && Reff > 0 && a.kfix[j] >= 65536u)
            return mf::fail(MF_EUNSUPPORTED, "fixed-point tap %u does not fit 16 bits", a.kfix[j]);
    a.dthr = distance_threshold;
    a.mthr = magnitude_threshold;
    a.decoded = decoded;
    a.mag = magnitudes;
    a.dist = distances;
    a.dist_f64 = dist_dtype == MF_DTYPE_F64;
    a.traces = traces;
    a.tr_z = tr_z_stride;
    a.tr_b = tr_bit_stride;
    a.tr_p = tr_pix_stride;
    a.tr_decoded_only = (flags & MF_DECODE_TRACES_DECODED_ONLY) != 0;
    a.tr_vec = traces && tr_bit_stride == 1 && !(B & 3) && !(tr_pix_stride & 3) &&
               !(tr_z_stride & 3) && !(reinterpret_cast<uintptr_t>(traces) & 15);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (image_dtype == MF_DTYPE_U16) return dispatch_radius<uint16_t>(Reff, a, s);
    return dispatch_radius<float>(Reff, a, s);
}

#endif  // MF_HOST_CODE
