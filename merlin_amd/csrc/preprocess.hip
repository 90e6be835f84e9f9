// preprocess.hip -- high-pass filter + Lucy-Richardson deconvolution for gfx950
// (MI355X): the stage that produces the decoder's uint16 stack.
//
// Replaces merlin/util/imagefilters.py:9-28 (high_pass_filter),
// merlin/util/deconvolve.py:70-133 (deconvolve_lucyrichardson, odd windows) and
// merlin/analysis/preprocess.py:114-140 (_preprocess_image + pixel histogram).
//
// The deconvolution is float64 like the reference and bound by memory traffic.
// One launch per iteration (lr_fused_kernel): single-wave workgroups stream
// 64-column strips through both blurs of the iteration, so that per pixel only
// J1, J2, J3 and the observed image are read and the new J1 written (34 B);
// the sums behind the acceleration parameter are reduced in a fixed order.
// lr_stream_kernel (two launches per iteration through a stored ratio image,
// 74 B) is the earlier form, kept behind MF_LR_SPLIT=1.
//
// Frames are processed `batch` at a time (one launch covers the batch; the
// working set is 34 B/px per frame); large batches keep the chip filled to the
// end of each launch.
//
// Built with -ffp-contract=off: every product and sum below rounds on its own,
// as the non-FMA scalar path of OpenCV and numpy do.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace {

constexpr int TW = 64;          // tile width = wave width
#ifndef MF_LR_SEG
#define MF_LR_SEG 128
#endif
constexpr int SEG = MF_LR_SEG;  // rows of one deconvolution strip segment
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

struct LrArgs {
    const void *wi;          // observed image(s), u16 or f64 [n][HW]
    const double *j1, *j2;   // [n][HW]
    const double *j3;        // fused kernel: the J before j2 (its buffer receives jnew)
    const double *lam_prev;  // fused kernel: [n] lambda of the previous iteration
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

__device__ __forceinline__ double atomic_load_f64(const double *p)
{
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

__device__ __forceinline__ void atomic_store_f64(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

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

// One wave streams a 64-column strip of SEG rows from top to bottom.  Per input
// row: the wave's 64 + 2R values go through one LDS row for the row pass
// (OpenCV RowFilter: s = k[0] S[0]; s += k[j] S[j], left to right), the row
// sums of the last 2R + 1 rows live in a register ring, and the column pass
// (SymmColumnFilter: centre, then k[R+j] (S[+j] + S[-j]) outwards) finishes
// output row r - R.  No workgroup barrier (a workgroup is one wave), the next
// row's loads are in flight while this one is filtered, and up to eight such
// waves per SIMD hide what is left of the latency.
//
// UPDATE = false (lr_reblur): source Y = max(J1 + lambda (J1 - J2), 0) rebuilt on
//   the fly; output ratio = wI / max(blur, eps) + eps       (deconvolve.py:112-122)
// UPDATE = true  (lr_update): source = ratio; J1' = Y * blur, T1' = J1' - Y, and
//   the tile's sums of T1' T1 and T1 T1; the LAST tile of the image to finish
//   adds the tile partials in tile order and publishes the next lambda
//   (no float atomics: identical from run to run)           (deconvolve.py:107-132)
template <int R, int SEGT, int P, typename WT, bool UPDATE>
__global__ void __launch_bounds__(64) lr_stream_kernel(const LrArgs a)
{
    __shared__ double srow[TW + 2 * R + 2];
    const int t = xcd_tile(a.tiles);
    if (t < 0) return;
    const int img = blockIdx.y;
    const int lane = threadIdx.x;
    const int H = a.H, W = a.W;
    const int64_t base = (int64_t)img * H * W;
    const int x0 = (t % a.tiles_x) * TW, y0 = (t / a.tiles_x) * SEGT;
    const int rows_out = min(SEGT, H - y0);
    const double lam = a.lam[img];
    const double *j1 = a.j1 + base, *j2 = a.j2 + base;
    const double *ratio = a.ratio + base;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + base;
    const double *t1 = a.t1 + base;
    // source columns: lane L holds x0 - R + L, lanes < 2R also x0 - R + 64 + L
    const int cm = clampi(x0 - R + lane, 0, W - 1);
    const int ch = clampi(x0 - R + TW + lane, 0, W - 1);
    const bool has_halo = lane < 2 * R;
    const int gx = x0 + lane;
    const int cg = min(gx, W - 1);

    // P source rows and P output rows' operands in flight, in fixed registers
    // (the row loop is unrolled P times so that no load target is ever moved)
    double m0[P], m1[P], h0[P], h1[P];
    double e0[P], e1[P], e2[P];
    WT w0[P];            // raw observed value: converted where it is used, not where it lands
#pragma unroll
    for (int u = 0; u < P; ++u) {
        m0[u] = m1[u] = h0[u] = h1[u] = e0[u] = e1[u] = e2[u] = 0.0;
        w0[u] = (WT)0;
    }
    // rows past either end of the strip (pipeline warm-up and drain) are
    // redirected to rows the wave reads anyway: an L1/L2 hit, no fabric traffic
    auto fetch = [&](int u, int r) {
        const int64_t row = (int64_t)clampi(y0 + min(r, rows_out + R - 1), 0, H - 1) * W;
        if constexpr (UPDATE) {
            m0[u] = ratio[row + cm];
            if (has_halo) h0[u] = ratio[row + ch];
        } else {
            m0[u] = j1[row + cm];
            m1[u] = j2[row + cm];
            if (has_halo) { h0[u] = j1[row + ch]; h1[u] = j2[row + ch]; }
        }
    };
    auto fetch_epilogue = [&](int u, int ro) {
        const int64_t p = (int64_t)(y0 + clampi(ro, 0, rows_out - 1)) * W + cg;
        if constexpr (UPDATE) { e0[u] = j1[p]; e1[u] = j2[p]; e2[u] = t1[p]; }
        else w0[u] = wi[p];
    };

    double ring[2 * R + 1];
#pragma unroll
    for (int j = 0; j <= 2 * R; ++j) ring[j] = 0.0;
    double s_ab = 0.0, s_bb = 0.0;

    // Straight-line software pipeline: every fetch is unconditional (row
    // indexes clamp, so rows past the strip only cost a wasted load) and the
    // trip count is padded to a multiple of P, so that no load target is copied
    // or waited for before its row is consumed.
    const int r_end = rows_out + R;
#pragma unroll
    for (int u = 0; u < P; ++u) { fetch(u, -R + u); fetch_epilogue(u, -2 * R + u); }
    for (int rb = -R; rb < r_end; rb += P) {
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int r = rb + u;
            double vm, vh;
            if constexpr (UPDATE) { vm = m0[u]; vh = h0[u]; }
            else { vm = make_y(m0[u], m1[u], lam); vh = make_y(h0[u], h1[u], lam); }
            const double c0 = e0[u], c1 = e1[u], c2 = e2[u];
            const WT cw = w0[u];
            fetch(u, r + P);
            fetch_epilogue(u, r + P - R);
            // one wave per workgroup: LDS operations of a wave execute in
            // order, so the row only needs the compiler to keep them in order
            // (a __syncthreads() would also wait for every global load in flight)
            lds_wave_order();
            srow[lane] = vm;
            if (has_halo) srow[TW + lane] = vh;
            lds_wave_order();
            double s = a.tp.k[0] * srow[lane];
#pragma unroll
            for (int j = 1; j <= 2 * R; ++j) s = s + a.tp.k[j] * srow[lane + j];
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) ring[j] = ring[j + 1];
            ring[2 * R] = s;
            const int ro = r - R;
            double o = a.tp.k[R] * ring[R];
#pragma unroll
            for (int j = 1; j <= R; ++j) o = o + a.tp.k[R + j] * (ring[R + j] + ring[R - j]);
            const bool ok = ro >= 0 && ro < rows_out && gx < W;
            const int64_t p = base + (int64_t)(y0 + ro) * W + gx;
            if constexpr (UPDATE) {
                const double y = make_y(c0, c1, lam);
                const double jn = y * o;             // np.multiply(Y, imR, out=J1)
                const double tn = jn - y;            // np.subtract(J1, Y, out=T1)
                if (ok) {
                    a.jnew[p] = jn;
                    a.t1[p] = tn;
                }
                s_ab = s_ab + (ok ? tn * c2 : 0.0);  // cv2.multiply(T1, T2) summed
                s_bb = s_bb + (ok ? c2 * c2 : 0.0);
            } else {
                const double rbl = o < EPS ? EPS : o;              // np.clip(reblurred, eps, None)
                if (ok) a.ratio[p] = (double)cw / rbl + EPS;       // cv2.divide; imR += eps
            }
        }
    }
    if constexpr (UPDATE) {
        // Tile partials cross to the reducing wave as 8-byte agent-scope
        // (sc1) stores and loads with the storing wave's vmcnt(0) before its
        // ticket: no L2 write-back / invalidate per tile.
        s_ab = wave_sum(s_ab);
        s_bb = wave_sum(s_bb);
        unsigned ticket = 0;
        if (lane == 0) {
            double *pp = a.partials + ((int64_t)img * a.tiles + t) * 2;
            atomic_store_f64(pp, s_ab);
            atomic_store_f64(pp + 1, s_bb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ticket = atomicAdd(a.counters + img, 1u);
        }
        ticket = __shfl(ticket, 0, 64);
        if (ticket != (unsigned)a.tiles - 1u) return;
        // last tile of this image: lane i adds tiles i, i + 64, ... in order
        const double *pp = a.partials + (int64_t)img * a.tiles * 2;
        double ab = 0.0, bb = 0.0;
        constexpr int U = 8;
        for (int i0 = lane; i0 < a.tiles; i0 += 64 * U) {
            double va[U], vb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 64 * u;
                va[u] = i < a.tiles ? atomic_load_f64(pp + 2 * i) : 0.0;
                vb[u] = i < a.tiles ? atomic_load_f64(pp + 2 * i + 1) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + 64 * u < a.tiles) { ab = ab + va[u]; bb = bb + vb[u]; }
            }
        }
        ab = wave_sum(ab);
        bb = wave_sum(bb);
        if (lane == 0) {
            double l = ab / (bb + EPS);                   // deconvolve.py:110
            l = 1.0 < l ? 1.0 : l;                        // min(l, 1)
            l = 0.0 > l ? 0.0 : l;                        // max(., 0)
            // the reference only starts accelerating at its third iteration
            if (a.iter + 1 <= 1) l = 0.0;
            a.lam_next[img] = l;
            if (a.lam_trace) a.lam_trace[(int64_t)img * a.trace_stride + a.iter + 1] = l;
            a.counters[img] = 0u;
        }
    }
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
template <int R, int SEGT, int P, typename WT>
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
    const int64_t base = (int64_t)img * H * W;
    const int x0 = (t % a.tiles_x) * OUTW, y0 = (t / a.tiles_x) * SEGT;
    const int rows_out = min(SEGT, H - y0);
    const double lam = a.lam[img], lam_prev = a.lam_prev[img];
    const double *j1 = a.j1 + base, *j2 = a.j2 + base, *j3 = a.j3 + base;
    const WT *wi = reinterpret_cast<const WT *>(a.wi) + base;
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
        const int64_t yrow = (int64_t)clampi(y0 + min(r, rows_out + 2 * R - 1), 0, H - 1) * W;
        m0[u] = j1[yrow + cm];
        m1[u] = j2[yrow + cm];
        const int64_t qrow = (int64_t)clampi(y0 + clampi(r - R, -R, rows_out + R - 1), 0, H - 1) * W;
        w0[u] = wi[qrow + cm];
        // J3 is only needed where row r is one of the strip's own output rows
        const int64_t p = (int64_t)(y0 + clampi(r, 0, rows_out - 1)) * W + cm;
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
            if (gq >= H) s2 = last_s2;                          // below the bottom edge
            last_s2 = s2;
            if (gq == 0) {                                      // rows above the top edge
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
            const bool ok = ro >= 0 && ro < rows_out && out_lane;
            const double jn = yo * o;                           // np.multiply(Y, imR, out=J1)
            const double tn = jn - yo;                          // np.subtract(J1, Y, out=T1)
            if (ok) a.jnew[base + (int64_t)(y0 + ro) * W + gx] = jn;
            s_ab = s_ab + (ok ? tn * to : 0.0);                 // cv2.multiply(T1, T2) summed
            s_bb = s_bb + (ok ? to * to : 0.0);
            // the slot's values are dead now: refill it.  The scheduling barrier
            // keeps the loads from being hoisted above the last use of the old
            // values, which would cost a register copy (and a wait for the
            // load) at the loop's back edge.
            __builtin_amdgcn_sched_barrier(0);
            fetch(u, r + P);
        }
    }
    // strip partials -> last strip of the frame -> next lambda (as lr_stream_kernel)
    s_ab = wave_sum(s_ab);
    s_bb = wave_sum(s_bb);
    unsigned ticket = 0;
    if (lane == 0) {
        double *pp = a.partials + ((int64_t)img * a.tiles + t) * 2;
        atomic_store_f64(pp, s_ab);
        atomic_store_f64(pp + 1, s_bb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = atomicAdd(a.counters + img, 1u);
    }
    ticket = __shfl(ticket, 0, 64);
    if (ticket != (unsigned)a.tiles - 1u) return;
    const double *pp = a.partials + (int64_t)img * a.tiles * 2;
    double ab = 0.0, bb = 0.0;
    constexpr int U = 8;
    for (int i0 = lane; i0 < a.tiles; i0 += 64 * U) {
        double va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 64 * u;
            va[u] = i < a.tiles ? atomic_load_f64(pp + 2 * i) : 0.0;
            vb[u] = i < a.tiles ? atomic_load_f64(pp + 2 * i + 1) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + 64 * u < a.tiles) { ab = ab + va[u]; bb = bb + vb[u]; }
        }
    }
    ab = wave_sum(ab);
    bb = wave_sum(bb);
    if (lane == 0) {
        double l = ab / (bb + EPS);                   // deconvolve.py:110
        l = 1.0 < l ? 1.0 : l;                        // min(l, 1)
        l = 0.0 > l ? 0.0 : l;                        // max(., 0)
        if (a.iter + 1 <= 1) l = 0.0;                 // acceleration starts at the third iteration
        a.lam_next[img] = l;
        if (a.lam_trace) a.lam_trace[(int64_t)img * a.trace_stride + a.iter + 1] = l;
        a.counters[img] = 0u;
    }
}

template <typename WT>
__global__ void __launch_bounds__(NT) lr_init_kernel(const WT *wi, double *ja, double *jb, double *t1,
                                                     int64_t n, int fused)
{
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) {
        const double v = load_wi(wi, i);
        ja[i] = v;
        jb[i] = v;
        t1[i] = fused ? v : 0.0;     // fused path: the third J buffer
    }
}

__global__ void lr_reset_kernel(double *lam, double *lam_trace, int64_t trace_stride,
                                unsigned *counters, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        lam[i] = lam[n + i] = lam[2 * n + i] = 0.0;
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
    unsigned zeros = 0;      // most deconvolved pixels are 0: one LDS add per thread, not per pixel
    for (int64_t i = lo + threadIdx.x; i < hi; i += NT) {
        const double v = src[i];
        if (out_f64) {
            reinterpret_cast<double *>(out)[(int64_t)img * hw + i] = v;
        } else {
            const uint16_t u = to_u16(v);
            reinterpret_cast<uint16_t *>(out)[(int64_t)img * hw + i] = u;
            if (hist) {
                if (u == 0) ++zeros;
                else if (u < HIST_LOW) atomicAdd(&low[u], 1u);
                else if (u < 65535) atomicAdd(row + min((int)u, HIST_BINS - 1), 1ull);
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
    double *ja, *jb, *t1, *ratio, *partials, *lam;
    unsigned *counters;
    uint16_t *wi16;
    size_t total;
};

void layout_workspace(int batch, int H, int W, unsigned char *base, Workspace &ws)
{
    const size_t hw = (size_t)H * W;
    // strips of the fused kernel at its narrowest (R = 6: 40 output columns)
    const size_t tiles = (size_t)((W + 39) / 40) * ((H + SEG - 1) / SEG);
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base + off; off += align256(bytes); return p; };
    ws.ja = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.jb = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.t1 = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.ratio = reinterpret_cast<double *>(take(batch * hw * 8));
    ws.wi16 = reinterpret_cast<uint16_t *>(take(batch * hw * 2));
    ws.partials = reinterpret_cast<double *>(take(batch * tiles * 16));
    ws.lam = reinterpret_cast<double *>(take((size_t)batch * 3 * 8));
    ws.counters = reinterpret_cast<unsigned *>(take((size_t)batch * 4));
    ws.total = off;
}

#ifndef MF_LR_PREFETCH
#define MF_LR_PREFETCH 4
#endif

template <int R, int P, typename WT>
int launch_iteration_p(const LrArgs &a, int n, hipStream_t s)
{
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)n);
    hipLaunchKernelGGL((lr_stream_kernel<R, SEG, P, WT, false>), grid, dim3(64), 0, s, a);
    hipLaunchKernelGGL((lr_stream_kernel<R, SEG, P, WT, true>), grid, dim3(64), 0, s, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

template <int R, typename WT>
int launch_iteration(const LrArgs &a, int n, hipStream_t s)
{
#ifdef MF_LR_TUNE      // development build: prefetch depth from the environment
    if (R == 4) {
        static const int p = std::getenv("MF_LR_P") ? std::atoi(std::getenv("MF_LR_P")) : MF_LR_PREFETCH;
        switch (p) {
        case 1: return launch_iteration_p<R, 1, WT>(a, n, s);
        case 2: return launch_iteration_p<R, 2, WT>(a, n, s);
        case 3: return launch_iteration_p<R, 3, WT>(a, n, s);
        case 6: return launch_iteration_p<R, 6, WT>(a, n, s);
        case 8: return launch_iteration_p<R, 8, WT>(a, n, s);
        }
    }
#endif
    return launch_iteration_p<R, MF_LR_PREFETCH, WT>(a, n, s);
}

template <int R, typename WT>
int launch_fused(const LrArgs &a, int n, hipStream_t s)
{
    const dim3 grid((unsigned)(((a.tiles + 7) >> 3) << 3), (unsigned)n);
    hipLaunchKernelGGL((lr_fused_kernel<R, SEG, MF_LR_PREFETCH, WT>), grid, dim3(64), 0, s, a);
    MF_HIP_CHECK(hipGetLastError());
    return MF_OK;
}

template <typename WT>
int dispatch_fused(int R, const LrArgs &a, int n, hipStream_t s)
{
    switch (R) {
    case 0: return launch_fused<0, WT>(a, n, s);
    case 1: return launch_fused<1, WT>(a, n, s);
    case 2: return launch_fused<2, WT>(a, n, s);
    case 3: return launch_fused<3, WT>(a, n, s);
    case 4: return launch_fused<4, WT>(a, n, s);
    case 5: return launch_fused<5, WT>(a, n, s);
    case 6: return launch_fused<6, WT>(a, n, s);
    }
    return mf::fail(MF_EUNSUPPORTED, "deconvolution filter radius %d > 6", R);
}

// The two-launch iteration (lr_stream_kernel reblur + update through a ratio
// buffer) instead of the fused one is kept for comparison: development builds
// (-DMF_LR_TUNE) read MF_LR_SPLIT=1 from the environment once; the release
// build has no environment knobs.
bool use_fused()
{
#ifdef MF_LR_TUNE
    static const bool fused = !(std::getenv("MF_LR_SPLIT") && std::atoi(std::getenv("MF_LR_SPLIT")));
    return fused;
#else
    return true;
#endif
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
    const int64_t total = hw * nb;
    const bool fused = use_fused();
    hipLaunchKernelGGL(lr_init_kernel<WT>, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, s, wi,
                       ws.ja, ws.jb, ws.t1, total, (int)fused);
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
    const int strip_w = fused ? TW - 4 * R : TW;
    a.tiles_x = (W + strip_w - 1) / strip_w;
    a.tiles = a.tiles_x * ((H + SEG - 1) / SEG);
    a.tp = tp;
    a.trace_stride = iterations;
    double *j1 = ws.ja, *j2 = ws.jb, *j3 = ws.t1;
    for (int it = 0; it < iterations; ++it) {
        a.j1 = j1; a.j2 = j2; a.j3 = j3;
        a.jnew = fused ? j3 : j2;
        a.lam = ws.lam + (size_t)(it % 3) * nb;
        a.lam_next = ws.lam + (size_t)((it + 1) % 3) * nb;
        a.lam_prev = ws.lam + (size_t)((it + 2) % 3) * nb;
        a.lam_trace = it + 1 < iterations ? trace : nullptr;
        a.iter = it;
        int rc = fused ? dispatch_fused<WT>(R, a, nb, s) : dispatch_iteration<WT>(R, a, nb, s);
        if (rc) return rc;
        if (fused) {         // (J1, J2, J3) <- (J1', J1, J2)
            double *tmp = j3; j3 = j2; j2 = j1; j1 = tmp;
        } else {
            double *tmp = j1; j1 = j2; j2 = tmp;
        }
    }
    void *dst = out_f64 ? (void *)(reinterpret_cast<double *>(out) + (int64_t)i0 * hw)
                        : (void *)(reinterpret_cast<uint16_t *>(out) + (int64_t)i0 * hw);
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, j1, dst,
                       out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
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
    hipLaunchKernelGGL(lr_finalize_kernel, dim3(HIST_BLOCKS, (unsigned)nb), dim3(NT), 0, s, ws.ja, dst,
                       out_f64, hw, out_f64 ? nullptr : hist, hist_group, i0);
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
