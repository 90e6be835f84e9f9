// preprocess.hip -- high-pass filter + Lucy-Richardson deconvolution for gfx950
// (MI355X): the stage that produces the decoder's uint16 stack.
//
// Replaces merlin/util/imagefilters.py:9-28 (high_pass_filter),
// merlin/util/deconvolve.py:70-133 (deconvolve_lucyrichardson, odd windows) and
// merlin/analysis/preprocess.py:114-140 (_preprocess_image + pixel histogram).
//
// The deconvolution is float64 like the reference and bound by memory traffic.
// One launch per iteration: single-wave workgroups stream column strips (128
// columns, two per lane; 64 when the image width is odd) through both blurs of
// the iteration, so that per pixel only J1, J2, J3 and the observed image are
// read and the new J1 written (34 B).  The sums behind the acceleration
// parameter leave each strip as per-16-row partials and are added in a fixed
// order by a one-block kernel between the iterations (lr_lambda_kernel): the
// hand-off is a kernel boundary, and the result does not depend on how a frame
// was cut into strip segments.
//
// Frames are processed `batch` at a time (one launch covers the batch; the
// working set is 26 B/px per frame); the segment height is chosen per launch so
// that the strips fill whole rounds of the chip's wave slots (choose_segments).
//
// Built with -ffp-contract=off: every product and sum below rounds on its own,
// as the non-FMA scalar path of OpenCV and numpy do.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace {

constexpr int TW = 64;          // tile width = wave width
constexpr int BH = 16;          // rows per block of lambda partial sums; segments are whole blocks
constexpr int SEG_MIN_BLOCKS = 4;   // shortest strip segment: 64 rows
constexpr int TH = 32;          // tile height: 4 waves x 8 rows
constexpr int NT = 256;
constexpr int ROWS_PER_WAVE = TH / 4;
constexpr int HIST_BINS = MF_HIST_BINS;
constexpr double EPS = DBL_EPSILON;      // np.finfo(float).eps, deconvolve.py:89

struct alignas(16) D2 { double x, y; };      // two adjacent columns of one lane
struct alignas(4) U2 { uint16_t x, y; };

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

// float64 -> uint16 the way numpy's astype does on x86: cvttsd2si to int32
// (0x80000000 when out of range or NaN), low 16 bits (preprocess.py:139)
__device__ __forceinline__ uint16_t to_u16(double v)
{
    if (!(fabs(v) < 2147483648.0)) return 0;
    return (uint16_t)((int32_t)v & 0xffff);
}

// Y of the current iteration (deconvolve.py:112-114): cv2.subtract,
// cv2.addWeighted(J1, 1, Y, lambda, 0), clip at 0
__device__ __forceinline__ double make_y(double j1, double j2, double lam)
{
    const double d = j1 - j2;
    double y = j1 * 1.0 + d * lam;
    y = y + 0.0;
    return y < 0.0 ? 0.0 : y;      // np.clip keeps NaN
}

struct LrArgs {
    const void *wi;          // observed image(s), u16 or f64 [n][HW]
    const double *j1, *j2;   // [n][HW]
    const double *j3;        // the J before j2 (its buffer receives jnew)
    double *jnew;
    const double *lam;       // [n] lambda of this iteration
    const double *lam_prev;  // [n] lambda of the previous iteration
    double *partials;        // [n][nblk][tiles_x][2]
    // last iteration of a uint16 run (LAST): the hand-over is written instead of J1'
    uint16_t *out16;               // [n][HW]
    unsigned long long *hist;      // [rows][HIST_BINS] or NULL
    int hist_group, img0;
    int H, W;
    int jp;                  // row pitch of the J buffers, in doubles
    int64_t js;              // frame stride of the J buffers, in doubles
    int tiles_x;             // strips across the image
    int nseg;                // segments a strip is cut into (each a whole number of blocks)
    int nblk;                // blocks of BH rows per frame
    int tiles;               // tiles_x * nseg
    Taps tp;
};

// compile-time ordering of LDS accesses among the lanes of one wave
__device__ __forceinline__ void lds_wave_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
    return v;       // lane 0 holds the total
}

// A wave-uniform condition that must stay a branch: an empty volatile asm in
// the conditional block keeps the compiler from turning the block into
// per-lane selects (18 of them per row for the ring fill below).
#define MF_UNIFORM_BRANCH() asm volatile("" ::: "memory")

// The strip's sums over one block of BH output rows -> partials.  Blocks are
// aligned to image rows, so the set of partials (and the order lr_lambda_kernel
// adds them in) is the same however the frame is cut into segments.
__device__ __forceinline__ void store_partial(const LrArgs &a, int img, int blk, int sx, int lane,
                                              double s_ab, double s_bb)
{
    s_ab = wave_sum(s_ab);
    s_bb = wave_sum(s_bb);
    if (lane == 0) {
        double *pp = a.partials + (((int64_t)img * a.nblk + blk) * a.tiles_x + sx) * 2;
        pp[0] = s_ab;
        pp[1] = s_bb;
    }
}

// tile -> strip column sx, first image row y0 and row count of its segment
__device__ __forceinline__ void tile_rows(const LrArgs &a, int t, int &sx, int &y0, int &rows_out)
{
    sx = t % a.tiles_x;
    const int k = t / a.tiles_x;
    const int b0 = (int)((int64_t)k * a.nblk / a.nseg), b1 = (int)((int64_t)(k + 1) * a.nblk / a.nseg);
    y0 = b0 * BH;
    rows_out = min(b1 * BH, a.H) - y0;
}

// ---- fused iteration -------------------------------------------------------
// One launch per iteration.  Within an iteration nothing global separates the
// two blurs (lambda only links one iteration to the next), so a wave can carry
// both: it streams a 64-column strip of Y rows, of which the middle 64 - 4R
// columns are outputs.  Stage 1 (row pass in an LDS row, ring of 2R + 1 row
// sums, column pass) gives the reblurred row R rows behind the Y stream and from
// it the ratio row; stage 2 does the same to the ratio rows and finishes output
// row 2R rows behind: J1' = Y * blur.  The ratio never reaches memory, J1 / J2
// are fetched once per iteration (plus the strip's halo), and the previous
// difference T1 = J1 - Y_prev is rebuilt bit-for-bit from the J of two
// iterations ago (j3, whose buffer then receives J1') instead of being stored:
// 34 B per pixel and iteration (J1, J2, J3, observed u16 in; J1' out) instead
// of 74.
//
// BORDER_REPLICATE: for the first blur clamped loads replicate Y.  The second
// blur replicates the RATIO: beyond the left / right image edge its row pass
// reads the edge column's ratio (clamped LDS index, border strips only), above
// the top edge the ring is filled with row 0's row sum, below the bottom edge
// the last row sum is repeated.
//
// Y and the rebuilt T1 of the strip's own columns wait in an LDS ring for the 2R
// rows between the blur input and the epilogue (a second global load of J1 / J2
// there missed L2 too often: 384 waves per XCD stream more than its 4 MB
// between the two uses), so every loaded register is dead within the row it is
// consumed in.
template <int R, int P, typename WT>
__global__ void __launch_bounds__(64) lr_fused_kernel(const LrArgs a)
{
    constexpr int OUTW = TW - 4 * R;
    constexpr int NRING = R > 0 ? 2 * R : 1;   // rows parked between blur input and epilogue
    __shared__ double srow1[TW + 2 * R];
    __shared__ double srow2[TW + 2 * R];
    __shared__ double sj1[NRING][TW];
    __shared__ double sj2[NRING][TW];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int lane = threadIdx.x;
    const int H = a.H, W = a.W;
    const int64_t base = (int64_t)img * a.js;     // J buffers: padded rows and frames
    const int JP = a.jp;
    int sx, y0, rows_out;
    tile_rows(a, t, sx, y0, rows_out);
    const int x0 = sx * OUTW;
    const double lam = a.lam[img], lam_prev = a.lam_prev[img];
    const double *j1 = a.j1 + base, *j2 = a.j2 + base, *j3 = a.j3 + base;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + (int64_t)img * H * W;
    const int gx = x0 - 2 * R + lane;                 // this lane's column
    const int cm = clampi(gx, 0, W - 1);
    const bool out_lane = lane >= 2 * R && lane < TW - 2 * R && gx < W;
    // the strip reaches past the image's left or right edge (wave-uniform)
    const bool border_x = x0 - 2 * R < 0 || x0 - 2 * R + TW > W;
    if (lane < 2 * R) {                               // LDS pads: never written again
        const int padi = lane < R ? lane : TW + lane;
        srow1[padi] = 0.0;
        srow2[padi] = 0.0;
    }

    // P rows of every stream in flight, in fixed registers: slot u serves loop
    // rows r = -2R + u (mod P): Y row r, ratio row r - R, output row r - 2R
    double m0[P], m1[P], e2[P];
    WT w0[P];
    auto fetch = [&](int u, int r) {
        const int64_t yrow = (int64_t)clampi(y0 + min(r, rows_out + 2 * R - 1), 0, H - 1) * JP;
        m0[u] = j1[yrow + cm];
        m1[u] = j2[yrow + cm];
        const int64_t qrow = (int64_t)clampi(y0 + clampi(r - R, -R, rows_out + R - 1), 0, H - 1) * W;
        w0[u] = wi[qrow + cm];
        // J3 is only needed where row r is one of the strip's own output rows
        const int64_t p = (int64_t)(y0 + clampi(r, 0, rows_out - 1)) * JP + cm;
        e2[u] = j3[p];
    };

    double ring1[2 * R + 1], ring2[2 * R + 1];
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) ring1[j] = ring2[j] = 0.0;
    double s_ab = 0.0, s_bb = 0.0, last_s2 = 0.0;

    const int r_end = rows_out + 2 * R;
    int slot = 0;                      // ring slot of the Y row being consumed
#pragma unroll
    for (int u = 0; u < P; ++u) fetch(u, -2 * R + u);
    for (int rb = -2 * R; rb < r_end; rb += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int r = rb + u;
            const double a1 = m0[u], a2 = m1[u];
            const double y = make_y(a1, a2, lam);
            // the T1 of the last iteration at this pixel, rebuilt: J1 - Y_prev
            const double told = a1 - make_y(a2, e2[u], lam_prev);
            const WT cw = w0[u];
            // ---- stage 1: reblurred row r - R, ratio ----
            lds_wave_order();
            srow1[R + lane] = y;
            // the row parked 2R iterations ago is the output row's; its slot
            // takes this row's values
            double yo = y, to = told;
            if constexpr (R > 0) {
                yo = sj1[slot][lane];
                to = sj2[slot][lane];
                lds_wave_order();
                sj1[slot][lane] = y;
                sj2[slot][lane] = told;
                slot = slot + 1 == NRING ? 0 : slot + 1;
            }
            lds_wave_order();
            double s1 = a.tp.k[0] * srow1[lane];
#pragma unroll
            for (int j = 1; j <= 2 * R; ++j) s1 = s1 + a.tp.k[j] * srow1[lane + j];
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) ring1[j] = ring1[j + 1];
            ring1[2 * R] = s1;
            double rbl = a.tp.k[R] * ring1[R];
#pragma unroll
            for (int j = 1; j <= R; ++j) rbl = rbl + a.tp.k[R + j] * (ring1[R + j] + ring1[R - j]);
            rbl = rbl < EPS ? EPS : rbl;                       // np.clip(reblurred, eps, None)
            const double ratio = (double)cw / rbl + EPS;       // cv2.divide; imR += eps
            // ---- stage 2: blurred ratio row r - 2R ----
            srow2[R + lane] = ratio;
            lds_wave_order();
            double s2;
            if (!border_x) {
                s2 = a.tp.k[0] * srow2[lane];
#pragma unroll
                for (int j = 1; j <= 2 * R; ++j) s2 = s2 + a.tp.k[j] * srow2[lane + j];
            } else {
                // tap j sits at column gx - R + j: replicate the image's edge columns
                const int off = 2 * R - x0 + R;                // column -> srow2 index
                s2 = a.tp.k[0] * srow2[clampi(gx - R, 0, W - 1) + off];
#pragma unroll
                for (int j = 1; j <= 2 * R; ++j)
                    s2 = s2 + a.tp.k[j] * srow2[clampi(gx - R + j, 0, W - 1) + off];
            }
            const int gq = y0 + r - R;                          // image row of this ratio row
            if (gq >= H) {                                      // below the bottom edge
                MF_UNIFORM_BRANCH();
                s2 = last_s2;
            }
            last_s2 = s2;
            if (gq == 0) {                                      // rows above the top edge
                MF_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j <= 2 * R; ++j) ring2[j] = s2;
            } else {
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) ring2[j] = ring2[j + 1];
                ring2[2 * R] = s2;
            }
            double o = a.tp.k[R] * ring2[R];
#pragma unroll
            for (int j = 1; j <= R; ++j) o = o + a.tp.k[R + j] * (ring2[R + j] + ring2[R - j]);
            // ---- epilogue: output row r - 2R ----
            const int ro = r - 2 * R;
            if (ro >= 0 && ro < rows_out) {                     // not a warm-up or drain row
                MF_UNIFORM_BRANCH();
                const double jn = yo * o;                       // np.multiply(Y, imR, out=J1)
                const double tn = jn - yo;                      // np.subtract(J1, Y, out=T1)
                if (out_lane) a.jnew[base + (int64_t)(y0 + ro) * JP + gx] = jn;
                // cv2.multiply(T1, T2) summed; lanes that own no output are dropped at the flush
                s_ab = s_ab + tn * to;
                s_bb = s_bb + to * to;
                if (((ro + 1) & (BH - 1)) == 0 || ro == rows_out - 1) {
                    store_partial(a, img, (y0 + ro) / BH, sx, lane, out_lane ? s_ab : 0.0,
                                  out_lane ? s_bb : 0.0);
                    s_ab = s_bb = 0.0;
                }
            }
            // the slot's values are dead now: refill it.  The scheduling barrier
            // keeps the loads from being hoisted above the last use of the old
            // values, which would cost a register copy (and a wait for the
            // load) at the loop's back edge.
            __builtin_amdgcn_sched_barrier(0);
            fetch(u, r + P);
        }
    }
}

// ---- fused iteration, two columns per lane --------------------------------
// The same pipeline as lr_fused_kernel over 128-column strips: lane L carries
// columns 2L and 2L + 1 of the strip, so 128 - 4R of 128 columns are outputs
// (112 at R = 4, against 48 of 64), every global access is 16 bytes per lane,
// and one row pass reads its 2R + 2 LDS values as R + 1 (+1 for odd R) 16-byte
// reads for both columns.  The two rings of row sums rotate by renaming when
// the row loop is unrolled a multiple of 2R + 1 times (U; no register moves),
// otherwise they shift.  Needs an even image width (a lane's pair is then never
// split by the right edge) and 16-byte aligned rows; the host falls back to
// lr_fused_kernel otherwise.  Arithmetic per pixel is unchanged.
template <typename T> struct PairOf;
template <> struct PairOf<double> { using type = D2; };
template <> struct PairOf<uint16_t> { using type = U2; };

//
// MODE = min(iteration, 3).  The J buffers all start as the observed image, so
// the first iterations read it instead (the same values, converted on the fly;
// no initialisation pass over the three buffers): iteration 0 takes J1 = J2 =
// J3 from the observed image, iteration 1 J2 = J3, iteration 2 J3; from
// iteration 3 on every operand comes from its buffer.
// LAST: the run's final iteration when the caller wants uint16: J1' goes through
// numpy's float64 -> uint16 cast straight into the output (and into the pixel
// histogram: zeros are counted per lane, the few per cent of non-zero pixels by
// global atomics), the lambda sums are skipped, and no hand-over pass re-reads
// the image.
template <int R, int P, int U, int MODE, bool LAST, typename WT>
__global__ void __launch_bounds__(64, 2) lr_fused2_kernel(const LrArgs a)
{
    using WP = typename PairOf<WT>::type;
    constexpr int SW = 2 * TW;
    constexpr int OUTW = SW - 4 * R;
    constexpr int N = 2 * R + 1;
    constexpr bool ROT = U % N == 0;               // ring slots renamed instead of shifted
    static_assert(U % P == 0, "prefetch slots must divide the unroll");
    constexpr int NRING = R > 0 ? 2 * R : 1;
    constexpr int PADL = R + (R & 1);              // even: a lane's pair stays 16-byte aligned
    constexpr int O = PADL - R;                    // first tap of column 2L inside the lane's reads
    constexpr int NV = (O + 2 * R + 3) / 2;        // 16-byte reads per row pass
    constexpr int SROW = SW - 2 + 2 * NV;          // >= PADL + SW + R
    __shared__ __align__(16) double srow1[SROW];
    __shared__ __align__(16) double srow2[SROW];
    __shared__ D2 sj1[NRING][TW];
    __shared__ D2 sj2[NRING][TW];
    // LAST: counts of the small output values of this wave's pixels.  Most
    // non-zero deconvolved pixels are small, and one global atomic each on the
    // same few bins serialises (0.06 ms per frame at 15 % non-zero pixels).
    constexpr int HLOW = 256;
    __shared__ unsigned hlow[LAST ? HLOW : 1];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int lane = threadIdx.x;
    const int H = a.H, W = a.W;
    const int64_t base = (int64_t)img * a.js;     // J buffers: padded rows and frames
    const int JP = a.jp;
    int sx, y0, rows_out;
    tile_rows(a, t, sx, y0, rows_out);
    const int x0 = sx * OUTW;
    const double lam = a.lam[img], lam_prev = a.lam_prev[img];
    const double *j1 = a.j1 + base, *j2 = a.j2 + base, *j3 = a.j3 + base;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + (int64_t)img * H * W;
    if constexpr (LAST)
        for (int i = lane; i < HLOW; i += TW) hlow[i] = 0u;
    const int gx = x0 - 2 * R + 2 * lane;             // the first of this lane's two columns (even)
    const int cp = clampi(gx, 0, W - 2);              // the pair that is loaded for them
    const bool dup_lo = gx < 0, dup_hi = gx >= W;     // both columns left / right of the image
    const bool out_lane = lane >= R && lane < TW - R && gx < W;
    const bool border_x = x0 - 2 * R < 0 || x0 - 2 * R + SW > W;    // wave-uniform
    for (int i = lane; i < SROW; i += TW)
        if (i < PADL || i >= PADL + SW) srow1[i] = srow2[i] = 0.0;   // pads: never written again
    D2 *const row1 = reinterpret_cast<D2 *>(srow1);
    D2 *const row2 = reinterpret_cast<D2 *>(srow2);

    D2 m0[P], m1[P], e2[P];
    WP w0[P], wy[P];
    auto widen = [](const WP &v) { D2 d; d.x = (double)v.x; d.y = (double)v.y; return d; };
    auto fetch = [&](int u, int r) {
        const int yr = clampi(y0 + min(r, rows_out + 2 * R - 1), 0, H - 1);
        const int64_t yrow = (int64_t)yr * JP;
        if constexpr (MODE >= 1) m0[u] = *reinterpret_cast<const D2 *>(j1 + yrow + cp);
        if constexpr (MODE >= 2) m1[u] = *reinterpret_cast<const D2 *>(j2 + yrow + cp);
        const int64_t qrow = (int64_t)clampi(y0 + clampi(r - R, -R, rows_out + R - 1), 0, H - 1) * W;
        w0[u] = *reinterpret_cast<const WP *>(wi + qrow + cp);
        // J3 is only needed where row r is one of the strip's own output rows
        const int pr = y0 + clampi(r, 0, rows_out - 1);
        if constexpr (MODE >= 3) e2[u] = *reinterpret_cast<const D2 *>(j3 + (int64_t)pr * JP + cp);
        // the observed image standing in for buffers that still hold it (on the
        // strip's own rows pr is the Y row, which is where J3 matters)
        if constexpr (MODE < 2) wy[u] = *reinterpret_cast<const WP *>(wi + (int64_t)yr * W + cp);
        if constexpr (MODE == 2) wy[u] = *reinterpret_cast<const WP *>(wi + (int64_t)pr * W + cp);
    };

    double r1x[N], r1y[N], r2x[N], r2y[N];         // row sums of the last 2R + 1 rows, both stages
#pragma unroll
    for (int j = 0; j < N; ++j) r1x[j] = r1y[j] = r2x[j] = r2y[j] = 0.0;
    double ab0 = 0.0, ab1 = 0.0, bb0 = 0.0, bb1 = 0.0, last_x = 0.0, last_y = 0.0;
    unsigned zeros = 0;
    unsigned long long *hrow =
        LAST && a.hist ? a.hist + (int64_t)((a.img0 + img) / a.hist_group) * HIST_BINS : nullptr;

    // OpenCV RowFilter over the values the lane read: s = k[0] S[0]; s += k[j] S[j]
    auto row_pass = [&](const double *v, int first) {
        double s = a.tp.k[0] * v[first];
#pragma unroll
        for (int j = 1; j <= 2 * R; ++j) s = s + a.tp.k[j] * v[first + j];
        return s;
    };

    const int r_end = rows_out + 2 * R;
    int slot = 0;
#pragma unroll
    for (int u = 0; u < P; ++u) fetch(u, -2 * R + u);
    for (int rb = -2 * R; rb < r_end; rb += U) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int u = i % P;
            const int r = rb + i;
            // ring slot written by this row, and where row sum j (0 = oldest) then lives
            const int wr = ROT ? i % N : 2 * R;
            auto at = [&](int j) { return ROT ? (i % N + 1 + j) % N : j; };
            D2 a1, a2, a3;
            if constexpr (MODE >= 1) a1 = m0[u]; else a1 = widen(wy[u]);
            if constexpr (MODE >= 2) a2 = m1[u]; else a2 = widen(wy[u]);
            if constexpr (MODE >= 3) a3 = e2[u]; else a3 = widen(wy[u]);
            const WP cw = w0[u];
            if (border_x) {                     // BORDER_REPLICATE of Y beyond the image's sides
                if (dup_lo) { a1.y = a1.x; a2.y = a2.x; }
                if (dup_hi) { a1.x = a1.y; a2.x = a2.y; }
            }
            D2 y, told;
            y.x = make_y(a1.x, a2.x, lam);
            y.y = make_y(a1.y, a2.y, lam);
            told.x = a1.x - make_y(a2.x, a3.x, lam_prev);
            told.y = a1.y - make_y(a2.y, a3.y, lam_prev);
            // ---- stage 1: reblurred row r - R, ratio ----
            lds_wave_order();
            row1[PADL / 2 + lane] = y;
            D2 yo = y, to = told;
            if constexpr (R > 0) {
                yo = sj1[slot][lane];
                to = sj2[slot][lane];
                lds_wave_order();
                sj1[slot][lane] = y;
                sj2[slot][lane] = told;
                slot = slot + 1 == NRING ? 0 : slot + 1;
            }
            lds_wave_order();
            double v[2 * NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const D2 tq = row1[lane + q];
                v[2 * q] = tq.x;
                v[2 * q + 1] = tq.y;
            }
            const double s1x = row_pass(v, O), s1y = row_pass(v, O + 1);
            if constexpr (!ROT) {
#pragma unroll
                for (int j = 0; j < 2 * R; ++j) { r1x[j] = r1x[j + 1]; r1y[j] = r1y[j + 1]; }
            }
            r1x[wr] = s1x;
            r1y[wr] = s1y;
            double bx = a.tp.k[R] * r1x[at(R)], by = a.tp.k[R] * r1y[at(R)];
#pragma unroll
            for (int j = 1; j <= R; ++j) {
                bx = bx + a.tp.k[R + j] * (r1x[at(R + j)] + r1x[at(R - j)]);
                by = by + a.tp.k[R + j] * (r1y[at(R + j)] + r1y[at(R - j)]);
            }
            bx = bx < EPS ? EPS : bx;                           // np.clip(reblurred, eps, None)
            by = by < EPS ? EPS : by;
            D2 ratio;
            ratio.x = (double)cw.x / bx + EPS;                  // cv2.divide; imR += eps
            ratio.y = (double)cw.y / by + EPS;
            // ---- stage 2: blurred ratio row r - 2R ----
            row2[PADL / 2 + lane] = ratio;
            lds_wave_order();
            double s2x, s2y;
            if (!border_x) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const D2 tq = row2[lane + q];
                    v[2 * q] = tq.x;
                    v[2 * q + 1] = tq.y;
                }
                s2x = row_pass(v, O);
                s2y = row_pass(v, O + 1);
            } else {
                // tap j of column c sits at column c - R + j: replicate the image's edge columns
                const int off = PADL - (x0 - 2 * R);            // column -> srow2 index
                s2x = a.tp.k[0] * srow2[clampi(gx - R, 0, W - 1) + off];
                s2y = a.tp.k[0] * srow2[clampi(gx + 1 - R, 0, W - 1) + off];
#pragma unroll
                for (int j = 1; j <= 2 * R; ++j) {
                    s2x = s2x + a.tp.k[j] * srow2[clampi(gx - R + j, 0, W - 1) + off];
                    s2y = s2y + a.tp.k[j] * srow2[clampi(gx + 1 - R + j, 0, W - 1) + off];
                }
            }
            const int gq = y0 + r - R;                          // image row of this ratio row
            if (gq >= H) {                                      // below the bottom edge
                MF_UNIFORM_BRANCH();
                s2x = last_x;
                s2y = last_y;
            }
            last_x = s2x;
            last_y = s2y;
            if (gq == 0) {                                      // rows above the top edge
                MF_UNIFORM_BRANCH();
#pragma unroll
                for (int j = 0; j < N; ++j) { r2x[j] = s2x; r2y[j] = s2y; }
            } else {
                if constexpr (!ROT) {
#pragma unroll
                    for (int j = 0; j < 2 * R; ++j) { r2x[j] = r2x[j + 1]; r2y[j] = r2y[j + 1]; }
                }
                r2x[wr] = s2x;
                r2y[wr] = s2y;
            }
            double ox = a.tp.k[R] * r2x[at(R)], oy = a.tp.k[R] * r2y[at(R)];
#pragma unroll
            for (int j = 1; j <= R; ++j) {
                ox = ox + a.tp.k[R + j] * (r2x[at(R + j)] + r2x[at(R - j)]);
                oy = oy + a.tp.k[R + j] * (r2y[at(R + j)] + r2y[at(R - j)]);
            }
            // ---- epilogue: output row r - 2R ----
            const int ro = r - 2 * R;
            if (ro >= 0 && ro < rows_out) {                     // not a warm-up or drain row
                MF_UNIFORM_BRANCH();
                D2 jn;
                jn.x = yo.x * ox;                               // np.multiply(Y, imR, out=J1)
                jn.y = yo.y * oy;
                if constexpr (LAST) {
                    if (out_lane) {
                        U2 u;
                        u.x = to_u16(jn.x);
                        u.y = to_u16(jn.y);
                        *reinterpret_cast<U2 *>(a.out16 + ((int64_t)img * H + (y0 + ro)) * W + gx) = u;
                        if (hrow) {
                            // preprocess.py:114-129: bins of np.arange(0, 65535), 65535 outside
                            if (u.x == 0) ++zeros;
                            else if (u.x < HLOW) atomicAdd(&hlow[u.x], 1u);
                            else if (u.x < 65535) atomicAdd(hrow + min((int)u.x, HIST_BINS - 1), 1ull);
                            if (u.y == 0) ++zeros;
                            else if (u.y < HLOW) atomicAdd(&hlow[u.y], 1u);
                            else if (u.y < 65535) atomicAdd(hrow + min((int)u.y, HIST_BINS - 1), 1ull);
                        }
                    }
                } else {
                    const double tnx = jn.x - yo.x, tny = jn.y - yo.y;  // np.subtract(J1, Y, out=T1)
                    if (out_lane)
                        *reinterpret_cast<D2 *>(a.jnew + base + (int64_t)(y0 + ro) * JP + gx) = jn;
                    // cv2.multiply(T1, T2) summed; lanes that own no output are dropped at the flush
                    ab0 = ab0 + tnx * to.x;
                    ab1 = ab1 + tny * to.y;
                    bb0 = bb0 + to.x * to.x;
                    bb1 = bb1 + to.y * to.y;
                    if (((ro + 1) & (BH - 1)) == 0 || ro == rows_out - 1) {
                        store_partial(a, img, (y0 + ro) / BH, sx, lane, out_lane ? ab0 + ab1 : 0.0,
                                      out_lane ? bb0 + bb1 : 0.0);
                        ab0 = ab1 = bb0 = bb1 = 0.0;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // as in lr_fused_kernel
            fetch(u, r + P);
        }
    }
    if constexpr (LAST) {
        if (hrow) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zeros += __shfl_down(zeros, off, 64);
            if (lane == 0 && zeros) atomicAdd(hrow, (unsigned long long)zeros);
            lds_wave_order();
            for (int i = lane; i < HLOW; i += TW)
                if (hlow[i]) atomicAdd(hrow + i, (unsigned long long)hlow[i]);
        }
    }
}

// observed image [n][H][W] -> the three J buffers [n][js] with row pitch jp
template <typename WT>
__global__ void __launch_bounds__(NT) lr_init_kernel(const WT *wi, double *ja, double *jb, double *jc,
                                                     int W, int64_t hw, int jp, int64_t js)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;      // pixel of frame blockIdx.y
    if (i < hw) {
        const double v = load_wi(wi, (int64_t)blockIdx.y * hw + i);
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const int64_t o = (int64_t)blockIdx.y * js + (int64_t)y * jp + x;
        ja[o] = v;
        jb[o] = v;
        jc[o] = v;
    }
}

__global__ void lr_reset_kernel(double *lam, double *lam_trace, int64_t trace_stride, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        lam[i] = lam[n + i] = lam[2 * n + i] = 0.0;
        if (lam_trace) lam_trace[(int64_t)i * trace_stride] = 0.0;
    }
}

// lambda of the next iteration from the strips' partial sums (deconvolve.py:
// 107-110): one block per frame; thread i adds partials i, i + 256, ... in
// order, the four waves add their lanes in a fixed tree, then wave 0 .. 3 in
// order.  No float atomics: identical from run to run, for every batch size
// and every cut of the frame into segments.
__global__ void __launch_bounds__(NT) lr_lambda_kernel(const double *partials, int count, int iter,
                                                       double *lam_next, double *lam_trace,
                                                       int64_t trace_stride)
{
    __shared__ double sw[2][NT / 64];
    const int img = blockIdx.x, tid = threadIdx.x;
    const double *pp = partials + (int64_t)img * count * 2;
    double ab = 0.0, bb = 0.0;
    for (int i = tid; i < count; i += NT) {
        ab = ab + pp[2 * i];
        bb = bb + pp[2 * i + 1];
    }
    ab = wave_sum(ab);
    bb = wave_sum(bb);
    if ((tid & 63) == 0) { sw[0][tid >> 6] = ab; sw[1][tid >> 6] = bb; }
    __syncthreads();
    if (tid == 0) {
        ab = sw[0][0]; bb = sw[1][0];
        for (int w = 1; w < NT / 64; ++w) { ab = ab + sw[0][w]; bb = bb + sw[1][w]; }
        double l = ab / (bb + EPS);                   // deconvolve.py:110
        l = 1.0 < l ? 1.0 : l;                        // min(l, 1)
        l = 0.0 > l ? 0.0 : l;                        // max(., 0)
        // the reference only starts accelerating at its third iteration
        if (iter + 1 <= 1) l = 0.0;
        lam_next[img] = l;
        if (lam_trace) lam_trace[(int64_t)img * trace_stride + iter + 1] = l;
    }
}

constexpr int HIST_LOW = 8192;       // bins kept in LDS; the rest go to global atomics
constexpr int HIST_BLOCKS = 64;      // blocks per image

// J1 -> output (+ histogram over np.arange(0, 65535): 65534 bins, the last one
// closed on the right, 65535 outside; preprocess.py:114-129)
__global__ void __launch_bounds__(NT) lr_finalize_kernel(const double *j, int W, int jp, int64_t js,
                                                         void *out, int out_f64, int64_t hw,
                                                         unsigned long long *hist, int hist_group, int img0)
{
    __shared__ unsigned low[HIST_LOW];
    const int img = blockIdx.y;
    const double *src = j + (int64_t)img * js;
    if (hist)
        for (int i = threadIdx.x; i < HIST_LOW; i += NT) low[i] = 0u;
    __syncthreads();
    unsigned long long *row = hist ? hist + (int64_t)((img0 + img) / hist_group) * HIST_BINS : nullptr;
    unsigned zeros = 0;      // most deconvolved pixels are 0: one LDS add per thread, not per pixel
    auto count = [&](uint16_t u) {
        if (u == 0) ++zeros;
        else if (u < HIST_LOW) atomicAdd(&low[u], 1u);
        else if (u < 65535) atomicAdd(row + min((int)u, HIST_BINS - 1), 1ull);
    };
    // an even share of the pixels per block, so that pairs never straddle two blocks
    const int64_t per = (((hw + gridDim.x - 1) / gridDim.x) + 1) & ~int64_t(1);
    const int64_t lo = min(hw, (int64_t)blockIdx.x * per), hi = min(hw, lo + per);
    const bool pairs = jp == W && !(hw & 1) && !(js & 1) && !(reinterpret_cast<uintptr_t>(j) & 15)
        && !(reinterpret_cast<uintptr_t>(out) & 15);
    if (pairs) {            // dense rows: 16 bytes in, 4 (or 16) bytes out per lane
        const D2 *src2 = reinterpret_cast<const D2 *>(src);
        for (int64_t q = (lo >> 1) + threadIdx.x; q < (hi >> 1); q += NT) {
            const D2 v = src2[q];
            if (out_f64) {
                reinterpret_cast<D2 *>(out)[(((int64_t)img * hw) >> 1) + q] = v;
            } else {
                U2 u;
                u.x = to_u16(v.x);
                u.y = to_u16(v.y);
                reinterpret_cast<U2 *>(out)[(((int64_t)img * hw) >> 1) + q] = u;
                if (hist) { count(u.x); count(u.y); }
            }
        }
    } else {                // padded rows or odd sizes
        for (int64_t i = lo + threadIdx.x; i < hi; i += NT) {
            const unsigned y = (unsigned)i / (unsigned)W;         // hw < 2^31 (check_shape)
            const double v = src[(int64_t)y * jp + ((unsigned)i - y * (unsigned)W)];
            if (out_f64) {
                reinterpret_cast<double *>(out)[(int64_t)img * hw + i] = v;
            } else {
                const uint16_t u = to_u16(v);
                reinterpret_cast<uint16_t *>(out)[(int64_t)img * hw + i] = u;
                if (hist) count(u);
            }
        }
    }
    if (hist) {
        if (zeros) atomicAdd(&low[0], zeros);
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

template <int R>
__global__ void __launch_bounds__(NT) high_pass_kernel(const HpArgs a)
{
    constexpr int IW = TW + 2 * R, IH = TH + 2 * R;
    __shared__ uint16_t sIn[IH * IW];
    __shared__ uint32_t sH[IH * TW];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
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
        uint32_t s = 0;      // taps sum to 65536: fits 32 bits; operands < 2^24
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) s += __umul24(a.tp.kfix[j], (uint32_t)sIn[r * IW + lane + j]);
        sH[r * TW + lane] = s;
    }
    __syncthreads();
    const int gx = x0 + lane;
    if (gx >= a.W) return;
    for (int o = 0; o < ROWS_PER_WAVE; ++o) {
        const int ly = wave * ROWS_PER_WAVE + o, gy = y0 + ly;
        if (gy >= a.H) break;
        // V = sum k_j H_j needs 48 bits: split H into 16-bit halves, two exact
        // u32 sums; (V + 2^31) >> 32 = (hi + (lo >> 16) + 2^15) >> 16
        uint32_t shi = 0, slo = 0;
#pragma unroll
        for (int j = 0; j <= 2 * R; ++j) {
            const uint32_t h = sH[(ly + j) * TW + lane];
            shi += __umul24(a.tp.kfix[j], h >> 16);
            slo += __umul24(a.tp.kfix[j], h & 0xffffu);
        }
        uint32_t lp = (shi + (slo >> 16) + 0x8000u) >> 16;
        if (lp > 65535) lp = 65535;
        const uint32_t v = sIn[(ly + R) * IW + lane + R];
        a.out[base + (int64_t)gy * a.W + gx] = lp > v ? (uint16_t)0 : (uint16_t)(v - lp);
    }
}

// ---- high-pass, streaming form -----------------------------------------------
// The tiled kernel above re-reads a 12-pixel halo on every side of a 64 x 32
// tile and pays 39 integer multiply-adds plus 26 LDS reads per pixel: 0.7 ms per
// 28 frames of 2048 x 2048, i.e. 0.7 TB/s for a 4 B/px filter.  This one streams
// 256-column strips top to bottom, one wave per strip and segment, four columns
// per lane (8-byte loads and stores): the row pass reads the lane's window from
// one LDS row as packed u16 pairs and adds two taps per v_dot2_u32_u16 (the
// packed taps of the four outputs come from the host), the u32 row sums of the
// last 2R + 1 rows sit in registers already split into 16-bit halves (rings
// renamed by unrolling the row loop 2R + 1 times), and the column pass is two
// exact 32-bit sums, symmetric taps paired (24-bit multiplies; see
// high_pass_kernel for the rounding).  Same arithmetic, same results.
constexpr int HPQ = 4;                 // columns per lane
constexpr int HPW = TW * HPQ;          // strip width
constexpr int HPH = 8;                 // column halo on either side (>= R, a multiple of 4)
constexpr int HP_MAXD = MF_PP_MAX_RADIUS + (MF_PP_MAX_RADIUS & 1) + 2;

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct HpsArgs {
    const uint16_t *in;
    uint16_t *out;
    int H, W, tiles_x, nseg, nblk, tiles;
    uint32_t tap[HPQ][HP_MAXD];        // packed taps of output column i over the lane's window
    uint32_t kfix[2 * MF_PP_MAX_RADIUS + 1];
};

template <int R>
__global__ void __launch_bounds__(64, 2) high_pass_stream_kernel(const HpsArgs a)
{
    constexpr int RE = R + (R & 1);        // window starts RE columns left of the lane's first
    constexpr int ND = RE + 2;             // dwords (u16 pairs) in the window
    constexpr int N = 2 * R + 1;
    constexpr int OUTW = HPW - 2 * HPH;
    __shared__ uint32_t srow[HPW / 2 + RE];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int lane = threadIdx.x;
    const int H = a.H, W = a.W;
    const int sx = t % a.tiles_x, k = t / a.tiles_x;
    const int y0 = (int)((int64_t)k * a.nblk / a.nseg) * BH;
    const int rows_out = min((int)((int64_t)(k + 1) * a.nblk / a.nseg) * BH, H) - y0;
    const int64_t base = (int64_t)blockIdx.y * H * W;
    const uint16_t *in = a.in + base;
    const int x0 = sx * OUTW;
    const int cq = x0 - HPH + HPQ * lane;              // first of the lane's four columns
    const int cl = clampi(cq, 0, W - HPQ);             // the quad that is loaded for them
    const bool left = cq < 0, right = cq >= W;         // BORDER_REPLICATE of whole quads
    const bool out_lane = lane >= HPH / HPQ && lane < TW - HPH / HPQ && cq < W;
    for (int i = lane; i < HPW / 2 + RE; i += TW)
        if (i < RE / 2 || i >= RE / 2 + HPW / 2) srow[i] = 0u;     // pads: never written again

    uint2 qin[N], qout[N];
    auto fetch = [&](int u, int r) {
        const int yi = clampi(y0 + min(r, rows_out + R - 1), 0, H - 1);
        qin[u] = *reinterpret_cast<const uint2 *>(in + (int64_t)yi * W + cl);
        const int yo = y0 + clampi(r - R, 0, rows_out - 1);
        qout[u] = *reinterpret_cast<const uint2 *>(in + (int64_t)yo * W + cl);
    };
    uint32_t rh[HPQ][N], rl[HPQ][N];       // row sums of the last 2R + 1 rows, 16-bit halves
#pragma unroll
    for (int i = 0; i < HPQ; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) rh[i][j] = rl[i][j] = 0u;

    const int r_end = rows_out + R;
#pragma unroll
    for (int u = 0; u < N; ++u) fetch(u, -R + u);
    for (int rb = -R; rb < r_end; rb += N) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int r = rb + u;
            uint2 q = qin[u];
            const uint2 c = qout[u];
            if (left) q.x = q.y = (q.x & 0xffffu) * 0x10001u;
            if (right) q.x = q.y = (q.y >> 16) * 0x10001u;
            lds_wave_order();
            srow[RE / 2 + 2 * lane] = q.x;
            srow[RE / 2 + 2 * lane + 1] = q.y;
            lds_wave_order();
            uint32_t w[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) w[d] = srow[2 * lane + d];
#pragma unroll
            for (int i = 0; i < HPQ; ++i) {
                uint32_t h = 0u;       // taps sum to 65536: fits 32 bits
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    h = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, w[d]),
                                               __builtin_bit_cast(u16x2, a.tap[i][d]), h, false);
                rh[i][u] = h >> 16;     // slot u: the row loop is unrolled N times
                rl[i][u] = h & 0xffffu;
            }
            // row sum j (0 = oldest) of the ring now lives in slot (u + 1 + j) % N
            const int ro = r - R;
            if (ro >= 0 && ro < rows_out) {
                MF_UNIFORM_BRANCH();
                uint32_t o[HPQ];
#pragma unroll
                for (int i = 0; i < HPQ; ++i) {
                    uint32_t shi = __umul24(a.kfix[R], rh[i][(u + 1 + R) % N]);
                    uint32_t slo = __umul24(a.kfix[R], rl[i][(u + 1 + R) % N]);
#pragma unroll
                    for (int j = 1; j <= R; ++j) {
                        // symmetric taps: k (a + b), k <= 2^15, a + b < 2^17
                        shi += __umul24(a.kfix[R + j], rh[i][(u + 1 + R + j) % N] + rh[i][(u + 1 + R - j) % N]);
                        slo += __umul24(a.kfix[R + j], rl[i][(u + 1 + R + j) % N] + rl[i][(u + 1 + R - j) % N]);
                    }
                    uint32_t lp = (shi + (slo >> 16) + 0x8000u) >> 16;
                    if (lp > 65535u) lp = 65535u;
                    const uint32_t pair = i < 2 ? c.x : c.y;
                    const uint32_t v = (i & 1) ? pair >> 16 : pair & 0xffffu;
                    o[i] = lp > v ? 0u : v - lp;
                }
                if (out_lane) {
                    uint2 res;
                    res.x = o[0] | (o[1] << 16);
                    res.y = o[2] | (o[3] << 16);
                    *reinterpret_cast<uint2 *>(a.out + base + (int64_t)(y0 + ro) * W + cq) = res;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(u, r + N);
        }
    }
}

// ---- even windows: the reference's scipy.ndimage.convolve branch -------------
// deconvolve.py:102-104, 118-119, 126-128: with an even windowSize both blurs of
// an iteration are ndimage.convolve(x, matlab_gauss2D, mode='constant') -- a
// non-separable K x K kernel, zeros outside the image -- and the blurred ratio
// is zeroed where it exceeds 2^16.  ndimage flips the kernel, moves the origin
// of an even axis by one, drops taps with |w| <= DBL_EPSILON and adds the rest
// in row-major order into a float64 that starts at 0 (ni_filters.c
// NI_Correlate): the host hands over the flipped, thresholded kernel and its
// centre, the kernel below adds in that order.  A rare configuration (the
// reference's own default is odd): one LDS-tiled launch per iteration that
// carries both blurs (the ratio tile is rebuilt with its halo instead of being
// stored), the same three-buffer rotation, partial sums per tile.
constexpr int CONV_MAXK = 16;
constexpr int CONV_TH = 16;

struct ConvArgs {
    const void *wi;
    const double *j1, *j2, *j3;
    double *jnew;
    const double *lam, *lam_prev;
    double *partials;        // [n][tiles][2]
    int H, W, jp;
    int64_t js;
    int tiles_x, tiles;
    int K, c;                // kernel size; index of the centre tap in the flipped kernel
    double k[CONV_MAXK * CONV_MAXK];
};

template <typename WT>
__global__ void __launch_bounds__(NT) lr_conv2d_kernel(const ConvArgs a)
{
    // Y tile: the outputs' taps' taps; ratio tile: the outputs' taps
    __shared__ double ytile[(CONV_TH + 2 * CONV_MAXK - 2) * (TW + 2 * CONV_MAXK - 2)];
    __shared__ double rtile[(CONV_TH + CONV_MAXK - 1) * (TW + CONV_MAXK - 1)];
    __shared__ double sw[2][NT / 64];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int K = a.K, lo = a.c, H = a.H, W = a.W, JP = a.jp;
    const int YW = TW + 2 * K - 2, YH = CONV_TH + 2 * K - 2;
    const int RW = TW + K - 1, RH = CONV_TH + K - 1;
    const int64_t jbase = (int64_t)img * a.js;
    const double *j1 = a.j1 + jbase, *j2 = a.j2 + jbase, *j3 = a.j3 + jbase;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + (int64_t)img * H * W;
    const double lam = a.lam[img], lam_prev = a.lam_prev[img];
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * CONV_TH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < YW * YH; e += NT) {
        const int r = e / YW, c = e - r * YW;
        const int gy = y0 - 2 * lo + r, gx = x0 - 2 * lo + c;
        double v = 0.0;                                   // mode='constant', cval 0
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int64_t p = (int64_t)gy * JP + gx;
            v = make_y(j1[p], j2[p], lam);
        }
        ytile[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < RW * RH; e += NT) {
        const int r = e / RW, c = e - r * RW;
        const int gy = y0 - lo + r, gx = x0 - lo + c;
        double v = 0.0;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            double s = 0.0;
            for (int ky = 0; ky < K; ++ky)
                for (int kx = 0; kx < K; ++kx) {
                    const double kv = a.k[ky * K + kx];
                    if (kv != 0.0) s = s + ytile[(r + ky) * YW + c + kx] * kv;
                }
            s = s < EPS ? EPS : s;                         // np.clip(reblurred, eps, None)
            v = load_wi(wi, (int64_t)gy * W + gx) / s + EPS;      // cv2.divide; imR += eps
        }
        rtile[e] = v;
    }
    __syncthreads();
    const int gx = x0 + lane;
    double s_ab = 0.0, s_bb = 0.0;
    for (int o = 0; o < CONV_TH / 4; ++o) {
        const int ly = wave * (CONV_TH / 4) + o, gy = y0 + ly;
        if (gy >= H || gx >= W) continue;
        double s = 0.0;
        for (int ky = 0; ky < K; ++ky)
            for (int kx = 0; kx < K; ++kx) {
                const double kv = a.k[ky * K + kx];
                if (kv != 0.0) s = s + rtile[(ly + ky) * RW + lane + kx] * kv;
            }
        if (s > 65536.0) s = 0.0;                          // imR[imR > 2 ** 16] = 0
        const int64_t p = (int64_t)gy * JP + gx;
        const double a1 = j1[p], a2 = j2[p];
        const double y = make_y(a1, a2, lam);
        const double told = a1 - make_y(a2, j3[p], lam_prev);     // last iteration's T1
        const double jn = y * s;                           // np.multiply(Y, imR, out=J1)
        const double tn = jn - y;                          // np.subtract(J1, Y, out=T1)
        a.jnew[jbase + p] = jn;
        s_ab = s_ab + tn * told;
        s_bb = s_bb + told * told;
    }
    s_ab = wave_sum(s_ab);
    s_bb = wave_sum(s_bb);
    if (lane == 0) { sw[0][wave] = s_ab; sw[1][wave] = s_bb; }
    __syncthreads();
    if (tid == 0) {
        double ab = sw[0][0], bb = sw[1][0];
        for (int w = 1; w < NT / 64; ++w) { ab = ab + sw[0][w]; bb = bb + sw[1][w]; }
        double *pp = a.partials + ((int64_t)img * a.tiles + t) * 2;
        pp[0] = ab;
        pp[1] = bb;
    }
}

// ---- Guo et al. variant (deconvolve.py:136-174) ------------------------------
// ek <- clip(ek * filter2D(image / clip(filter2D(ek, pf), eps, 65535), pb), eps,
// 65535) with two non-separable K x K projectors (K <= 13) supplied by the host
// (they come out of an FFT, deconvolve.py:13-67).  cv2.filter2D is a correlation
// anchored at the kernel centre, BORDER_REPLICATE; OpenCV adds the non-zero taps
// in row-major order.  Two iterations by default: a plain LDS-tiled kernel.
constexpr int GUO_MAXK = 13;
constexpr int GUO_TH = 16;
constexpr double GUO_EPS = 1.0e-6, GUO_IMAX = 65535.0;

struct GuoArgs {
    const void *img;      // observed image(s) [n][HW], u16 or f64
    double *ek;           // estimate, updated in place by the backward step
    double *ratio;        // image / forward projection
    int H, W, K, tiles_x, tiles, backward;
    double k[GUO_MAXK * GUO_MAXK];
};

__device__ __forceinline__ double clip_guo(double v)
{
    v = v < GUO_EPS ? GUO_EPS : v;
    return v > GUO_IMAX ? GUO_IMAX : v;
}

template <typename WT>
__global__ void __launch_bounds__(NT) guo_step_kernel(const GuoArgs a)
{
    __shared__ double tile[(GUO_TH + GUO_MAXK - 1) * (TW + GUO_MAXK - 1)];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int K = a.K, R = K / 2, IW = TW + 2 * R, IH = GUO_TH + 2 * R;
    const int H = a.H, W = a.W;
    const int64_t base = (int64_t)blockIdx.y * H * W;
    const double *src = (a.backward ? a.ratio : a.ek) + base;
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * GUO_TH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < IW * IH; e += NT) {
        const int r = e / IW, c = e - r * IW;
        tile[e] = src[(int64_t)clampi(y0 - R + r, 0, H - 1) * W + clampi(x0 - R + c, 0, W - 1)];
    }
    __syncthreads();
    const int gx = x0 + lane;
    const WT *img = reinterpret_cast<const WT *>(a.img) + base;
    for (int o = 0; o < GUO_TH / 4; ++o) {
        const int ly = wave * (GUO_TH / 4) + o, gy = y0 + ly;
        if (gy >= H || gx >= W) continue;
        double s = 0.0;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                const double kv = a.k[i * K + j];
                if (kv != 0.0) s = s + kv * tile[(ly + i) * IW + lane + j];
            }
        const int64_t p = base + (int64_t)gy * W + gx;
        if (a.backward) a.ek[p] = clip_guo(a.ek[p] * s);
        else a.ratio[p] = load_wi(img, p - base) / clip_guo(s);
    }
}

template <typename WT>
__global__ void __launch_bounds__(NT) guo_init_kernel(const WT *img, double *ek, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) {
        const double v = load_wi(img, i);
        ek[i] = v < GUO_EPS ? GUO_EPS : v;          // np.clip(ek, eps, None)
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
    double *ja, *jb, *jc, *partials, *lam;
    uint16_t *wi16;
    int jp;          // row pitch of ja / jb / jc, in doubles
    int64_t js;      // their frame stride, in doubles
    size_t total;
};

#ifndef MF_LR_PADX
#define MF_LR_PADX 0
#endif
#ifndef MF_LR_PADF
#define MF_LR_PADF 0
#endif

// Row and frame padding of the J buffers, in doubles (both even, so that pairs
// of columns stay 16-byte aligned).
void j_padding(int &padx, int64_t &padf)
{
    padx = MF_LR_PADX;
    padf = MF_LR_PADF;
#ifdef MF_LR_TUNE
    if (std::getenv("MF_LR_PADX")) padx = std::atoi(std::getenv("MF_LR_PADX")) & ~1;
    if (std::getenv("MF_LR_PADF")) padf = std::atoll(std::getenv("MF_LR_PADF")) & ~1ll;
#endif
}

void layout_workspace(int batch, int H, int W, unsigned char *base, Workspace &ws)
{
    const size_t hw = (size_t)H * W;
    // strips of the one-column kernel at its narrowest (R = 6: 40 output columns)
    const size_t cells = (size_t)((W + 39) / 40) * ((H + BH - 1) / BH);
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base + off; off += align256(bytes); return p; };
    int padx;
    int64_t padf;
    j_padding(padx, padf);
    ws.jp = W + padx;
    ws.js = (int64_t)H * ws.jp + padf;
    ws.ja = reinterpret_cast<double *>(take(batch * ws.js * 8));
    ws.jb = reinterpret_cast<double *>(take(batch * ws.js * 8));
    ws.jc = reinterpret_cast<double *>(take(batch * ws.js * 8));
    ws.wi16 = reinterpret_cast<uint16_t *>(take(batch * hw * 2));
    ws.partials = reinterpret_cast<double *>(take(batch * cells * 16));
    ws.lam = reinterpret_cast<double *>(take((size_t)batch * 3 * 8));
    ws.total = off;
}

#ifndef MF_LR_PREFETCH
#define MF_LR_PREFETCH 4
#endif

// Wave slots of the chip for a single-wave kernel: resident workgroups per CU
// (registers, LDS) x CUs of the current device.
template <typename K>
int wave_slots(K kernel)
{
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu < 1)
        per_cu = 8;
    if (hipGetDevice(&dev) != hipSuccess
        || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    return per_cu * cus;
}

// Segments per strip for one launch of `strips` strip columns (tiles_x x
// frames).  Every segment streams its rows plus `halo` warm-up rows, and the
// launch runs in rounds of `slots` concurrent waves; a partly filled last round
// takes about as long as a full one (a wave alone is bound by its own latency:
// measured 1.39 us per row at half occupancy against 1.64 at full), so with one
// fixed height (r01: 128 rows) up to a fifth of a launch was spent there.
// Cost: rows of the longest segment x rounds, rounded up.  Measured at 28 frames
// of 2048 x 2048 (19 strips, 2048 slots), ms per frame: 7 segments 0.665, 11
// 0.651, 15 0.652, 16 0.659, 19 0.697; the rule picks 11 (and 8 at 112 frames).
int choose_segments(int64_t strips, int nblk, int halo, int slots)
{
#ifdef MF_LR_TUNE
    if (std::getenv("MF_LR_NSEG")) {
        const int v = std::atoi(std::getenv("MF_LR_NSEG"));
        return v < 1 ? 1 : v > nblk ? nblk : v;
    }
#endif
    const int most = nblk / SEG_MIN_BLOCKS > 1 ? nblk / SEG_MIN_BLOCKS : 1;
    int best = 1;
    double best_cost = 0.0;
    for (int n = 1; n <= most; ++n) {
        // the longest segment sets the length of a round: counts that cut the
        // frame into equal segments win (112 frames: 8 -> 0.578 ms per frame,
        // 16 -> 0.581, 11 -> 0.588, 10 / 12 -> 0.598, 14 -> 0.619)
        const double rows = (double)((nblk + n - 1) / n) * BH + halo;
        const int64_t waves = strips * n;
        const double cost = rows * (double)((waves + slots - 1) / slots);
        if (n == 1 || cost < best_cost) { best = n; best_cost = cost; }
    }
#ifdef MF_LR_TUNE
    if (std::getenv("MF_LR_VERBOSE"))
        fprintf(stderr, "choose_segments: strips %lld nblk %d halo %d slots %d -> %d\n", (long long)strips,
                nblk, halo, slots, best);
#endif
    return best;
}

template <int R, typename WT>
int launch_fused(LrArgs a, int n, hipStream_t s)
{
    auto kernel = lr_fused_kernel<R, MF_LR_PREFETCH, WT>;
    static const int slots = wave_slots(kernel);
    a.nseg = choose_segments((int64_t)a.tiles_x * n, a.nblk, 4 * R, slots);
    a.tiles = a.tiles_x * a.nseg;
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)n);
    hipLaunchKernelGGL(kernel, grid, dim3(64), 0, s, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

constexpr int MODE_LAST = 1 << 20;     // launch_fused2: general operands + fused hand-over

// prefetch depth P and unroll U of the two-column kernel per radius: U is a
// multiple of 2R + 1 where that keeps the code small (rings rotate by renaming)
template <int R> struct Fused2Shape {
    static constexpr int N = 2 * R + 1;
    static constexpr int U = N == 3 || N == 5 || N == 9 ? N : N < 9 ? 4 : 3;
    static constexpr int P = N == 9 ? 3 : U;
};

template <int R, typename WT>
int launch_fused2(LrArgs a, int n, int mode, hipStream_t s)
{
    constexpr int P = Fused2Shape<R>::P, U = Fused2Shape<R>::U;
    static const int slots = wave_slots(lr_fused2_kernel<R, P, U, 3, false, WT>);
    a.nseg = choose_segments((int64_t)a.tiles_x * n, a.nblk, 4 * R, slots);
    a.tiles = a.tiles_x * a.nseg;
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)n);
    switch (mode) {
    case 0: hipLaunchKernelGGL((lr_fused2_kernel<R, P, U, 0, false, WT>), grid, dim3(64), 0, s, a); break;
    case 1: hipLaunchKernelGGL((lr_fused2_kernel<R, P, U, 1, false, WT>), grid, dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL((lr_fused2_kernel<R, P, U, 2, false, WT>), grid, dim3(64), 0, s, a); break;
    case MODE_LAST:
        hipLaunchKernelGGL((lr_fused2_kernel<R, P, U, 3, true, WT>), grid, dim3(64), 0, s, a);
        break;
    default: hipLaunchKernelGGL((lr_fused2_kernel<R, P, U, 3, false, WT>), grid, dim3(64), 0, s, a); break;
    }
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

// mode: the iteration number (two-column kernel only; see lr_fused2_kernel)
template <typename WT>
int dispatch_fused(int R, bool pairs, const LrArgs &a, int n, int mode, hipStream_t s)
{
    switch (pairs ? R + 16 : R) {
    case 0: return launch_fused<0, WT>(a, n, s);
    case 1: return launch_fused<1, WT>(a, n, s);
    case 2: return launch_fused<2, WT>(a, n, s);
    case 3: return launch_fused<3, WT>(a, n, s);
    case 4: return launch_fused<4, WT>(a, n, s);
    case 5: return launch_fused<5, WT>(a, n, s);
    case 6: return launch_fused<6, WT>(a, n, s);
    case 16: return launch_fused2<0, WT>(a, n, mode, s);
    case 17: return launch_fused2<1, WT>(a, n, mode, s);
    case 18: return launch_fused2<2, WT>(a, n, mode, s);
    case 19: return launch_fused2<3, WT>(a, n, mode, s);
    case 20: return launch_fused2<4, WT>(a, n, mode, s);
    case 21: return launch_fused2<5, WT>(a, n, mode, s);
    case 22: return launch_fused2<6, WT>(a, n, mode, s);
    }
    return mf::fail(MF_EUNSUPPORTED, "deconvolution filter radius %d > 6", R);
}

// development builds can switch the two-column kernel off (MF_LR_PAIRS=0)
bool use_pairs()
{
#ifdef MF_LR_TUNE
    static const bool on = !(std::getenv("MF_LR_PAIRS") && !std::atoi(std::getenv("MF_LR_PAIRS")));
    return on;
#else
    return true;
#endif
}

int check_shape(int n, int H, int W, int batch)
{
    if (n < 0 || H < 1 || W < 1 || batch < 1)
        return mf::fail(MF_EINVAL, "bad shape n=%d H=%d W=%d batch=%d", n, H, W, batch);
    if ((int64_t)H * W > INT32_MAX) return mf::fail(MF_EUNSUPPORTED, "image exceeds 2^31 pixels");
    if (batch > 65535) return mf::fail(MF_EUNSUPPORTED, "batch %d > 65535", batch);
    return MF_OK;
}

template <int R>
int launch_high_pass_stream(HpsArgs a, int nb, hipStream_t s)
{
    static const int slots = wave_slots(high_pass_stream_kernel<R>);
    a.nseg = choose_segments((int64_t)a.tiles_x * nb, a.nblk, 2 * R, slots);
    a.tiles = a.tiles_x * a.nseg;
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)nb);
    hipLaunchKernelGGL(high_pass_stream_kernel<R>, grid, dim3(64), 0, s, a);
    return MF_OK;
}

// development builds can switch the streaming high-pass off (MF_HP_STREAM=0)
bool use_hp_stream()
{
#ifdef MF_LR_TUNE
    static const bool on = !(std::getenv("MF_HP_STREAM") && !std::atoi(std::getenv("MF_HP_STREAM")));
    return on;
#else
    return true;
#endif
}

int run_high_pass(const uint16_t *in, int n, int H, int W, int ksize, double sigma, uint16_t *out,
                  hipStream_t s)
{
    HpArgs h;
    int rc = fill_taps(ksize, sigma, h.tp);
    if (rc) return rc;
    h.R = ksize / 2;
    // streaming kernel: quads of columns must not straddle the image edge or an
    // 8-byte boundary, and every tap has to fit the 16 bits v_dot2_u32_u16 takes
    bool stream = use_hp_stream() && h.R >= 1 && h.R <= HPH && W >= HPQ && !(W & 3)
        && !(reinterpret_cast<uintptr_t>(in) & 7) && !(reinterpret_cast<uintptr_t>(out) & 7);
    for (int j = 0; j < ksize; ++j) stream = stream && h.tp.kfix[j] < 65536u;
    if (stream) {
        const int R = h.R, RE = R + (R & 1);
        HpsArgs a;
        std::memset(&a, 0, sizeof(a));
        a.H = H; a.W = W;
        a.tiles_x = (W + (HPW - 2 * HPH) - 1) / (HPW - 2 * HPH);
        a.nblk = (H + BH - 1) / BH;
        std::memcpy(a.kfix, h.tp.kfix, sizeof(a.kfix));
        // window element e of a lane is column 4 L - RE + e; output i sits at
        // column 4 L + i, its tap j at column 4 L + i - R + j
        for (int i = 0; i < HPQ; ++i)
            for (int d = 0; d < RE + 2; ++d) {
                uint32_t pair = 0;
                for (int half = 0; half < 2; ++half) {
                    const int j = 2 * d + half - i - (RE - R);
                    if (j >= 0 && j <= 2 * R) pair |= h.tp.kfix[j] << (16 * half);
                }
                a.tap[i][d] = pair;
            }
        for (int i0 = 0; i0 < n; i0 += 65535) {
            const int nb = n - i0 < 65535 ? n - i0 : 65535;
            a.in = in + (int64_t)i0 * H * W;
            a.out = out + (int64_t)i0 * H * W;
            switch (R) {
            case 1: launch_high_pass_stream<1>(a, nb, s); break;
            case 2: launch_high_pass_stream<2>(a, nb, s); break;
            case 3: launch_high_pass_stream<3>(a, nb, s); break;
            case 4: launch_high_pass_stream<4>(a, nb, s); break;
            case 5: launch_high_pass_stream<5>(a, nb, s); break;
            case 6: launch_high_pass_stream<6>(a, nb, s); break;
            case 7: launch_high_pass_stream<7>(a, nb, s); break;
            default: launch_high_pass_stream<8>(a, nb, s); break;
            }
        }
        MF_HIP_CHECK(hipGetLastError());
        return MF_OK;
    }
    h.H = H; h.W = W;
    h.tiles_x = (W + TW - 1) / TW;
    h.tiles = h.tiles_x * ((H + TH - 1) / TH);
    for (int i0 = 0; i0 < n; i0 += 65535) {
        const int nb = n - i0 < 65535 ? n - i0 : 65535;
        h.in = in + (int64_t)i0 * H * W;
        h.out = out + (int64_t)i0 * H * W;
        const dim3 grid((unsigned)(((h.tiles + 7) >> 3) << 3), (unsigned)nb);
        switch (h.R) {
        case 0: hipLaunchKernelGGL(high_pass_kernel<0>, grid, dim3(NT), 0, s, h); break;
        case 1: hipLaunchKernelGGL(high_pass_kernel<1>, grid, dim3(NT), 0, s, h); break;
        case 2: hipLaunchKernelGGL(high_pass_kernel<2>, grid, dim3(NT), 0, s, h); break;
        case 3: hipLaunchKernelGGL(high_pass_kernel<3>, grid, dim3(NT), 0, s, h); break;
        case 4: hipLaunchKernelGGL(high_pass_kernel<4>, grid, dim3(NT), 0, s, h); break;
        case 5: hipLaunchKernelGGL(high_pass_kernel<5>, grid, dim3(NT), 0, s, h); break;
        case 6: hipLaunchKernelGGL(high_pass_kernel<6>, grid, dim3(NT), 0, s, h); break;
        case 7: hipLaunchKernelGGL(high_pass_kernel<7>, grid, dim3(NT), 0, s, h); break;
        case 8: hipLaunchKernelGGL(high_pass_kernel<8>, grid, dim3(NT), 0, s, h); break;
        default: return mf::fail(MF_EUNSUPPORTED, "high-pass radius %d > %d", h.R, MF_PP_MAX_RADIUS);
        }
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
    double *trace = lam_trace ? lam_trace + (int64_t)i0 * iterations : nullptr;
    hipLaunchKernelGGL(lr_reset_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, ws.lam,
                       iterations > 0 ? trace : nullptr, (int64_t)iterations, nb);
    LrArgs a;
    std::memset(&a, 0, sizeof(a));
    a.wi = wi;
    a.partials = ws.partials;
    a.H = H; a.W = W;
    a.jp = ws.jp; a.js = ws.js;
    // two columns per lane when no pair of columns can straddle the right edge
    // and every row starts 16-byte aligned (4 bytes for the uint16 image)
    const bool pairs = use_pairs() && W >= 2 && !(W & 1)
        && !(reinterpret_cast<uintptr_t>(wi) & (2 * sizeof(WT) - 1))
        && !(reinterpret_cast<uintptr_t>(ws.ja) & 15) && !(ws.jp & 1) && !(ws.js & 1);
    // the two-column kernel reads the observed image in place of buffers that
    // would only hold a copy of it; the one-column kernel starts from copies
    if (!pairs || iterations == 0)
        hipLaunchKernelGGL(lr_init_kernel<WT>, dim3((unsigned)((hw + NT - 1) / NT), (unsigned)nb), dim3(NT),
                           0, s, wi, ws.ja, ws.jb, ws.jc, W, hw, ws.jp, ws.js);
    const int strip_w = (pairs ? 2 * TW : TW) - 4 * R;
    a.tiles_x = (W + strip_w - 1) / strip_w;
    a.nblk = (H + BH - 1) / BH;
    a.tp = tp;
    void *dst = out_f64 ? (void *)(reinterpret_cast<double *>(out) + (int64_t)i0 * hw)
                        : (void *)(reinterpret_cast<uint16_t *>(out) + (int64_t)i0 * hw);
    // uint16 output, an operand set without special cases (iteration >= 3) and
    // 4-byte aligned pairs: the last iteration writes the hand-over itself
    // (not when the output overlaps the observed image: other workgroups of the
    // last launch still read that image, halo rows included, while this one
    // writes; the separate hand-over kernel runs after every read instead)
    const uintptr_t wi_lo = reinterpret_cast<uintptr_t>(wi), wi_hi = wi_lo + (size_t)nb * hw * sizeof(WT);
    const uintptr_t d_lo = reinterpret_cast<uintptr_t>(dst), d_hi = d_lo + (size_t)nb * hw * sizeof(uint16_t);
    const bool aliased = d_lo < wi_hi && wi_lo < d_hi;
    const bool fuse_last = pairs && !out_f64 && iterations >= 4 && !aliased &&
                           !(reinterpret_cast<uintptr_t>(dst) & 3);
    a.out16 = reinterpret_cast<uint16_t *>(dst);
    a.hist = hist;
    a.hist_group = hist_group > 0 ? hist_group : 1;
    a.img0 = i0;
    double *j1 = ws.ja, *j2 = ws.jb, *j3 = ws.jc;
    for (int it = 0; it < iterations; ++it) {
        a.j1 = j1; a.j2 = j2; a.j3 = j3;
        a.jnew = j3;
        a.lam = ws.lam + (size_t)(it % 3) * nb;
        a.lam_prev = ws.lam + (size_t)((it + 2) % 3) * nb;
        // the last iteration of a uint16 run writes the hand-over itself
        const bool last = fuse_last && it + 1 == iterations;
        int rc = dispatch_fused<WT>(R, pairs, a, nb, last ? MODE_LAST : it, s);
        if (rc) return rc;
        if (last) {
            MF_HIP_CHECK(hipGetLastError());
            return MF_OK;
        }
        hipLaunchKernelGGL(lr_lambda_kernel, dim3((unsigned)nb), dim3(NT), 0, s, ws.partials,
                           a.nblk * a.tiles_x, it, ws.lam + (size_t)((it + 1) % 3) * nb,
                           it + 1 < iterations ? trace : nullptr, (int64_t)iterations);
        double *tmp = j3; j3 = j2; j2 = j1; j1 = tmp;     // (J1, J2, J3) <- (J1', J1, J2)
    }
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, j1, W, ws.jp,
                       ws.js, dst, out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

// Lucy-Richardson with a K x K ndimage.convolve kernel on images [i0, i0 + nb)
template <typename WT>
int run_lr_conv_batch(const WT *wi, int nb, int i0, int H, int W, int K, const double *h_weights,
                      int iterations, void *out, int out_f64, unsigned long long *hist, int hist_group,
                      double *lam_trace, Workspace &ws, hipStream_t s)
{
    const int64_t hw = (int64_t)H * W;
    hipLaunchKernelGGL(lr_init_kernel<WT>, dim3((unsigned)((hw + NT - 1) / NT), (unsigned)nb), dim3(NT), 0,
                       s, wi, ws.ja, ws.jb, ws.jc, W, hw, ws.jp, ws.js);
    double *trace = lam_trace ? lam_trace + (int64_t)i0 * iterations : nullptr;
    hipLaunchKernelGGL(lr_reset_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, s, ws.lam,
                       iterations > 0 ? trace : nullptr, (int64_t)iterations, nb);
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.wi = wi;
    a.partials = ws.partials;
    a.H = H; a.W = W;
    a.jp = ws.jp; a.js = ws.js;
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles = a.tiles_x * ((H + CONV_TH - 1) / CONV_TH);
    a.K = K;
    // scipy/ndimage/_filters.py _correlate_or_convolve: flip, origin -1 on even axes
    a.c = K / 2 - ((K & 1) ? 0 : 1);
    for (int i = 0; i < K * K; ++i) {
        const double w = h_weights[K * K - 1 - i];
        a.k[i] = std::fabs(w) > DBL_EPSILON ? w : 0.0;    // ni_filters.c: the footprint
    }
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)nb);
    double *j1 = ws.ja, *j2 = ws.jb, *j3 = ws.jc;
    for (int it = 0; it < iterations; ++it) {
        a.j1 = j1; a.j2 = j2; a.j3 = j3;
        a.jnew = j3;
        a.lam = ws.lam + (size_t)(it % 3) * nb;
        a.lam_prev = ws.lam + (size_t)((it + 2) % 3) * nb;
        hipLaunchKernelGGL(lr_conv2d_kernel<WT>, grid, dim3(NT), 0, s, a);
        hipLaunchKernelGGL(lr_lambda_kernel, dim3((unsigned)nb), dim3(NT), 0, s, ws.partials, a.tiles, it,
                           ws.lam + (size_t)((it + 1) % 3) * nb, it + 1 < iterations ? trace : nullptr,
                           (int64_t)iterations);
        double *tmp = j3; j3 = j2; j2 = j1; j1 = tmp;
    }
    void *dst = out_f64 ? (void *)(reinterpret_cast<double *>(out) + (int64_t)i0 * hw)
                        : (void *)(reinterpret_cast<uint16_t *>(out) + (int64_t)i0 * hw);
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, j1, W, ws.jp,
                       ws.js, dst, out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

template <typename WT>
int run_guo_batch(const WT *img, int nb, int i0, int H, int W, int K, const double *h_pf,
                  const double *h_pb, int iterations, void *out, int out_f64,
                  unsigned long long *hist, int hist_group, Workspace &ws, hipStream_t s)
{
    const int64_t hw = (int64_t)H * W, total = hw * nb;
    hipLaunchKernelGGL(guo_init_kernel<WT>, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s, img,
                       ws.ja, total);
    GuoArgs a;
    std::memset(&a, 0, sizeof(a));
    a.img = img;
    a.ek = ws.ja;
    a.ratio = ws.jb;
    a.H = H; a.W = W; a.K = K;
    a.tiles_x = (W + TW - 1) / TW;
    a.tiles = a.tiles_x * ((H + GUO_TH - 1) / GUO_TH);
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)nb);
    for (int it = 0; it < iterations; ++it) {
        a.backward = 0;
        std::memcpy(a.k, h_pf, sizeof(double) * K * K);
        hipLaunchKernelGGL(guo_step_kernel<WT>, grid, dim3(NT), 0, s, a);
        a.backward = 1;
        std::memcpy(a.k, h_pb, sizeof(double) * K * K);
        hipLaunchKernelGGL(guo_step_kernel<WT>, grid, dim3(NT), 0, s, a);
    }
    void *dst = out_f64 ? (void *)(reinterpret_cast<double *>(out) + (int64_t)i0 * hw)
                        : (void *)(reinterpret_cast<uint16_t *>(out) + (int64_t)i0 * hw);
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, ws.ja, W, W, hw,
                       dst, out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

}  // namespace

extern "C" int mf_deconvolve_lucyrichardson_guo(const void *image, int image_dtype, int n_images, int H,
                                                int W, int window_size, const double *h_pf,
                                                const double *h_pb, int iterations, void *out,
                                                int out_dtype, int64_t *histogram, int hist_group,
                                                int batch, void *workspace, int64_t workspace_bytes,
                                                void *stream)
{
    if (!image || !out || !workspace || !h_pf || !h_pb)
        return mf::fail(MF_EINVAL, "null pointer argument");
    int rc = check_shape(n_images, H, W, batch);
    if (rc) return rc;
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f64", image_dtype);
    if (out_dtype != MF_DTYPE_U16 && out_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "output dtype %d not u16/f64", out_dtype);
    if (iterations < 0) return mf::fail(MF_EINVAL, "iteration count %d < 0", iterations);
    if (window_size < 1 || !(window_size & 1) || window_size > GUO_MAXK)
        return mf::fail(MF_EUNSUPPORTED, "projector window %d must be odd and <= %d", window_size,
                        GUO_MAXK);
    if (histogram && (out_dtype != MF_DTYPE_U16 || hist_group < 1))
        return mf::fail(MF_EINVAL, "histogram needs uint16 output and hist_group >= 1");
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
            rc = run_guo_batch(reinterpret_cast<const uint16_t *>(image) + i0 * hw, nb, i0, H, W,
                               window_size, h_pf, h_pb, iterations, out, out_dtype == MF_DTYPE_F64,
                               reinterpret_cast<unsigned long long *>(histogram), hist_group, ws, s);
        else
            rc = run_guo_batch(reinterpret_cast<const double *>(image) + i0 * hw, nb, i0, H, W,
                               window_size, h_pf, h_pb, iterations, out, out_dtype == MF_DTYPE_F64,
                               reinterpret_cast<unsigned long long *>(histogram), hist_group, ws, s);
        if (rc) return rc;
    }
    return MF_OK;
}

extern "C" int mf_deconvolve_lucyrichardson_kernel2d(const void *image, int image_dtype, int n_images,
                                                     int H, int W, int window_size,
                                                     const double *h_weights, int iterations, void *out,
                                                     int out_dtype, int64_t *histogram, int hist_group,
                                                     double *lambdas, int batch, void *workspace,
                                                     int64_t workspace_bytes, void *stream)
{
    if (!image || !out || !workspace || !h_weights) return mf::fail(MF_EINVAL, "null pointer argument");
    int rc = check_shape(n_images, H, W, batch);
    if (rc) return rc;
    if (image_dtype != MF_DTYPE_U16 && image_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "image dtype %d not u16/f64", image_dtype);
    if (out_dtype != MF_DTYPE_U16 && out_dtype != MF_DTYPE_F64)
        return mf::fail(MF_EINVAL, "output dtype %d not u16/f64", out_dtype);
    if (iterations < 0) return mf::fail(MF_EINVAL, "iteration count %d < 0", iterations);
    if (window_size < 1 || window_size > CONV_MAXK)
        return mf::fail(MF_EUNSUPPORTED, "convolution window %d not in 1..%d", window_size, CONV_MAXK);
    if (histogram && (out_dtype != MF_DTYPE_U16 || hist_group < 1))
        return mf::fail(MF_EINVAL, "histogram needs uint16 output and hist_group >= 1");
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
            rc = run_lr_conv_batch(reinterpret_cast<const uint16_t *>(image) + i0 * hw, nb, i0, H, W,
                                   window_size, h_weights, iterations, out, out_dtype == MF_DTYPE_F64,
                                   reinterpret_cast<unsigned long long *>(histogram), hist_group, lambdas,
                                   ws, s);
        else
            rc = run_lr_conv_batch(reinterpret_cast<const double *>(image) + i0 * hw, nb, i0, H, W,
                                   window_size, h_weights, iterations, out, out_dtype == MF_DTYPE_F64,
                                   reinterpret_cast<unsigned long long *>(histogram), hist_group, lambdas,
                                   ws, s);
        if (rc) return rc;
    }
    return MF_OK;
}

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
        return mf::fail(MF_EUNSUPPORTED, "even window size %d takes the ndimage.convolve branch: "
                        "mf_deconvolve_lucyrichardson_kernel2d", window_size);
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
        return mf::fail(MF_EUNSUPPORTED, "even window size %d takes the ndimage.convolve branch: "
                        "mf_high_pass_filter + mf_deconvolve_lucyrichardson_kernel2d", decon_filter_size);
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
