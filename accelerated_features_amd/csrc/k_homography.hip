// Robust homography from the match lists: the device side of
//     H, inliers = cv2.findHomography(points1, points2, cv2.USAC_MAGSAC, ransac_thr, maxIters=700, confidence=0.995)
// (realtime_demo.py:225 of the reference).  OpenCV is not available offline; what is implemented is the published algorithm behind
// USAC_MAGSAC -- RANSAC with the MAGSAC++ quality and sigma-consensus++ weights (Barath, Noskova, Ivashechkin, Matas, CVPR 2020) -- with
// the call's arguments (threshold, maxIters, confidence).  The specification (the test suite holds an independent numpy
// restatement of it):
//   * minimal sample: 4 distinct correspondences; draw d of hypothesis `it` of pair p is the upper half of
//     splitmix64-finaliser(seed + golden * (((p << 20) + it) * 16 + d + 1)) scaled to [0, n); duplicates are redrawn (16 draws at most);
//   * 4-point homography H = B adj(A), A / B the projective bases through the first three source / target points that send (1,1,1) to
//     the fourth (3x3 adjugates, no pivoting); a sample is rejected unless its four point triples all keep or all flip their orientation;
//   * residual = squared forward transfer error; quality = sum over residuals below (2 thr)^2 of 1 - rho(r) / rho(k sigma_max), rho the
//     MAGSAC++ loss for n = 4 degrees of freedom, k = 3.64, sigma_max = 2 thr / k, from a 4096-bin table over r^2 in 20-bit fixed point;
//   * termination: hypotheses in order, a strictly better quality updates the bound log(1 - confidence) / log(1 - w^4), w = inlier ratio at thr;
//   * refinement of the winner: up to 5 re-weighted least-squares steps (Hartley-normalised inhomogeneous DLT, weights w(r) / w(0) from the
//     same bins), each kept only if it raises the quality; mask = residual < thr^2 under the final model; found = at least 4 inliers.
//
// What makes it a device algorithm: hypothesis `it` is a function of (seed, pair, it) alone, so ALL maxIters hypotheses are built and
// scored at once -- thread = hypothesis, workgroups over (256 hypotheses) x (64-512 correspondences) x pairs -- and the sequential loop's
// stopping rule is applied afterwards to the score list, exactly as the loop would have applied it.  A score is a sum of integers
// (u64 atomics: no summation order), the geometry is fp64 with every product and sum rounded once (fp contraction off in this file), so
// the winning hypothesis, the iteration count and the inlier mask are reproducible bit for bit by any IEEE implementation of the
// specification; only the least-squares refinement carries a (1e-9) tolerance, from the order of its fp64 reductions.
//
// Up to five launches per call, fp64 VALU + latency bound (one pair with 1000 matches and 700 hypotheses is 0.7 M residuals of ~45 fp64 ops):
//   homog_tables_kernel : loss / weight tables of this threshold (closed forms of the incomplete gamma functions for n = 4), scores zeroed
//   homog_score_kernel  : hypotheses + MAGSAC++ quality + inlier counts; hypotheses 0..255 of every pair first, then homog_bound_kernel
//                         bounds the index the loop can still reach and the later hypothesis blocks run only below that bound (at 50 %
//                         inliers the loop needs 83 iterations: the 444 hypotheses beyond the first block are never built)
//   homog_select_kernel : one workgroup per pair: stopping rule, refinement (23 weighted sums -> block Cholesky of the 8x8 normal equations), mask
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace xfh {
namespace hg {
constexpr int NBINS = 4096, SCORE_ONE = 1 << 20, LO_ITERS = 5, MAX_DRAWS = 16;
constexpr int HYP_PER_WG = 256, PTS_PER_WG = 512, MAX_ITERS = 4096;
constexpr double K_QUANTILE = 3.64, MAX_THR_FACTOR = 2.0;
constexpr int NSUM = 23;
constexpr int SEL_CACHE = 2048;     // correspondences of a pair that homog_select_kernel keeps in LDS
}  // namespace hg

struct HgArgs {
    const float* p0;          // (P, kcap, 2): the correspondences themselves (idx0 == NULL, kcap == cap) or the key-point lists they index
    const float* p1;
    const int64_t* idx0;      // (P, cap) rows of p0 / p1 of correspondence i, or NULL
    const int64_t* idx1;
    const int32_t* counts;
    int n_const, P, cap, kcap, iters, iters_pad;
    int sel_cache_off;        // byte offset of homog_select_kernel's correspondence cache in its dynamic LDS
    int chunk;                // correspondences per workgroup of homog_score_kernel (64 .. PTS_PER_WG)
    double thr2, tmax2, bin_scale, log1mc;
    unsigned long long seed;
    unsigned* stab;
    double* wtab;
    unsigned long long* hscore;
    unsigned* hcnt;
    double* H;
    unsigned char* mask;
    int32_t* info;
};

// ---- tables -----------------------------------------------------------------------------------------------------------------------------
// Gamma(3/2, x) = sqrt(pi)/2 erfc(sqrt x) + sqrt x e^-x ;  gamma(5/2, x) = 3/4 sqrt(pi) erf(sqrt x) - e^-x sqrt x (3/2 + x)
__device__ inline double upper_gamma_3_2(double sx) { return 0.88622692545275801365 * erfc(sx) + sx * exp(-sx * sx); }
__device__ inline double lower_gamma_5_2(double sx) { return 1.32934038817913702047 * erf(sx) - exp(-sx * sx) * sx * (1.5 + sx * sx); }

__global__ __launch_bounds__(256) void homog_tables_kernel(double thr, unsigned* __restrict__ stab, double* __restrict__ wtab,
                                                           unsigned long long* __restrict__ hscore, unsigned* __restrict__ hcnt, int nhyp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < hg::NBINS && stab) {
        const double t_max = hg::MAX_THR_FACTOR * thr, sigma = t_max / hg::K_QUANTILE;
        const double bin_scale = hg::NBINS / (t_max * t_max);
        const double r = sqrt((i + 0.5) / bin_scale);
        const double sx = r / (sigma * 1.41421356237309504880), sk = hg::K_QUANTILE / 1.41421356237309504880;
        const double gk = upper_gamma_3_2(sk);
        const double diff = upper_gamma_3_2(sx) - gk;
        const double half_s2 = 0.5 * sigma * sigma;
        const double loss = half_s2 * lower_gamma_5_2(sx) + 0.25 * r * r * diff;          // common factor C 2^(5/2) / sigma dropped
        const double loss_out = half_s2 * lower_gamma_5_2(sk);
        double q = 1.0 - loss / loss_out;
        q = q < 0.0 ? 0.0 : (q > 1.0 ? 1.0 : q);
        stab[i] = (unsigned)floor(q * hg::SCORE_ONE + 0.5);
        wtab[i] = diff / (0.88622692545275801365 - gk);                                    // w(r) / w(0)
    }
    for (int j = i; j < nhyp; j += gridDim.x * 256) { hscore[j] = 0ull; hcnt[j] = 0u; }
}

// ---- correspondences of one pair ---------------------------------------------------------------------------------------------------------
struct PairPts {
    const float* p0;
    const float* p1;
    const int64_t* i0;
    const int64_t* i1;
    __device__ PairPts(const HgArgs& a, int pair)
        : p0(a.p0 + (size_t)pair * a.kcap * 2), p1(a.p1 + (size_t)pair * a.kcap * 2), i0(a.idx0 ? a.idx0 + (size_t)pair * a.cap : nullptr),
          i1(a.idx1 ? a.idx1 + (size_t)pair * a.cap : nullptr) {}
    __device__ inline void get(int i, float2& q0, float2& q1) const {
        const size_t r0 = i0 ? (size_t)i0[i] : (size_t)i, r1 = i1 ? (size_t)i1[i] : (size_t)i;
        q0 = *reinterpret_cast<const float2*>(p0 + 2 * r0);
        q1 = *reinterpret_cast<const float2*>(p1 + 2 * r1);
    }
};

// ---- hypotheses -------------------------------------------------------------------------------------------------------------------------
__device__ inline unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ inline int draw_index(unsigned long long seed, int pair, int it, int draw, int n) {
    const unsigned long long counter = ((unsigned long long)pair * (1ull << 20) + (unsigned long long)it) * hg::MAX_DRAWS + (unsigned long long)draw;
    const unsigned long long h = mix64(seed + 0x9e3779b97f4a7c15ull * (counter + 1ull));
    return (int)(((h >> 32) * (unsigned long long)n) >> 32);
}
__device__ inline double orient(double ax, double ay, double bx, double by, double cx, double cy) {
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
// columns lambda_j (x_j, y_j, 1) of the projective basis through points 0..2 that sends (1,1,1) to D * point 3; d[4] = triple orientations
__device__ inline void basis(const double (&x)[4], const double (&y)[4], double (&m)[3][3], double (&d)[4]) {
    const double l0 = orient(x[3], y[3], x[1], y[1], x[2], y[2]);
    const double l1 = orient(x[0], y[0], x[3], y[3], x[2], y[2]);
    const double l2 = orient(x[0], y[0], x[1], y[1], x[3], y[3]);
    d[0] = orient(x[0], y[0], x[1], y[1], x[2], y[2]); d[1] = l0; d[2] = l1; d[3] = l2;
    const double lam[3] = {l0, l1, l2};
#pragma unroll
    for (int j = 0; j < 3; ++j) { m[0][j] = lam[j] * x[j]; m[1][j] = lam[j] * y[j]; m[2][j] = lam[j]; }
}
// hypothesis `it` of pair `pair`: false when the draws ran out or the sample does not keep the orientation of its triples
__device__ inline bool make_hypothesis(const PairPts& pts, int n, unsigned long long seed, int pair, int it, double (&h)[9]) {
    int i0 = -1, i1 = -1, i2 = -1, i3 = -1, slot = 0;
#pragma unroll
    for (int d = 0; d < hg::MAX_DRAWS; ++d) {
        const int c = draw_index(seed, pair, it, d, n);
        const bool dup = (slot > 0 && c == i0) || (slot > 1 && c == i1) || (slot > 2 && c == i2);
        if (slot < 4 && !dup) {
            i0 = slot == 0 ? c : i0; i1 = slot == 1 ? c : i1; i2 = slot == 2 ? c : i2; i3 = slot == 3 ? c : i3;
            ++slot;
        }
    }
    if (slot < 4) return false;
    const int idx[4] = {i0, i1, i2, i3};
    double x0[4], y0[4], x1[4], y1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float2 a, b;
        pts.get(idx[k], a, b);
        x0[k] = a.x; y0[k] = a.y; x1[k] = b.x; y1[k] = b.y;
    }
    double A[3][3], Bm[3][3], da[4], db[4];
    basis(x0, y0, A, da);
    basis(x1, y1, Bm, db);
    double adj[3][3];
    adj[0][0] = A[1][1] * A[2][2] - A[1][2] * A[2][1];
    adj[0][1] = A[0][2] * A[2][1] - A[0][1] * A[2][2];
    adj[0][2] = A[0][1] * A[1][2] - A[0][2] * A[1][1];
    adj[1][0] = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    adj[1][1] = A[0][0] * A[2][2] - A[0][2] * A[2][0];
    adj[1][2] = A[0][2] * A[1][0] - A[0][0] * A[1][2];
    adj[2][0] = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    adj[2][1] = A[0][1] * A[2][0] - A[0][0] * A[2][1];
    adj[2][2] = A[0][0] * A[1][1] - A[0][1] * A[1][0];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) h[3 * i + j] = (Bm[i][0] * adj[0][j] + Bm[i][1] * adj[1][j]) + Bm[i][2] * adj[2][j];
    const double q0 = da[0] * db[0], q1 = da[1] * db[1], q2 = da[2] * db[2], q3 = da[3] * db[3];
    return (q0 > 0 && q1 > 0 && q2 > 0 && q3 > 0) || (q0 < 0 && q1 < 0 && q2 < 0 && q3 < 0);
}
// squared forward transfer error
__device__ inline double residual_sq(const double (&h)[9], double x, double y, double u1, double v1) {
    const double w = (h[6] * x + h[7] * y) + h[8];
    const double u = (h[0] * x + h[1] * y) + h[2];
    const double v = (h[3] * x + h[4] * y) + h[5];
    const double iw = 1.0 / w;
    const double dx = u1 - u * iw, dy = v1 - v * iw;
    return dx * dx + dy * dy;
}
__device__ inline int bin_of(double r2, double bin_scale) {
    const int b = (int)(r2 * bin_scale);
    return b < hg::NBINS - 1 ? b : hg::NBINS - 1;
}

// iterations the loop still needs once a model with `inliers` of n is the best one (the standard RANSAC bound)
__device__ inline int iterations_needed(unsigned inliers, int n, double log1mc, int max_iters) {
    const double w = (double)inliers / (double)n;
    const double p = 1.0 - w * w * w * w;
    if (p <= 0.0) return 1;
    if (p >= 1.0) return max_iters;
    const double k = ceil(log1mc / log(p));
    return k < (double)max_iters ? (int)k : max_iters;
}

// Hypotheses [256 (blockIdx.x + blk0), + 256) of pair blockIdx.z against correspondences [chunk blockIdx.y, + chunk).  The first 256 hypotheses of
// every pair are scored first (blk0 = 0, bound = NULL); the later blocks run only where homog_bound_kernel left a bound above their first index.
__global__ __launch_bounds__(256) void homog_score_kernel(HgArgs a, int blk0, const int* __restrict__ bound) {
    __shared__ unsigned stab[hg::NBINS];
    __shared__ float4 spt[hg::PTS_PER_WG];
    const int pair = blockIdx.z, tid = threadIdx.x;
    const int n = a.counts ? min(a.counts[pair], a.cap) : a.n_const;
    const int c0 = blockIdx.y * a.chunk;
    const int it0 = (blockIdx.x + blk0) * hg::HYP_PER_WG;
    if (n < 4 || c0 >= n) return;
    if (bound && bound[pair] <= it0) return;
    const PairPts pts(a, pair);
    const int c1 = min(c0 + a.chunk, n);
#pragma unroll
    for (int k = 0; k < hg::NBINS / 256; ++k) stab[tid + 256 * k] = a.stab[tid + 256 * k];
#pragma unroll
    for (int k = 0; k < hg::PTS_PER_WG / 256; ++k) {      // the chunk's correspondences: fetched (and de-referenced) once per workgroup
        const int i = c0 + tid + 256 * k;
        float2 q0 = make_float2(0.f, 0.f), q1 = q0;
        if (i < c1) pts.get(i, q0, q1);
        spt[tid + 256 * k] = make_float4(q0.x, q0.y, q1.x, q1.y);
    }
    __syncthreads();
    const int it = it0 + tid;
    if (it >= a.iters) return;
    double h[9];
    if (!make_hypothesis(pts, n, a.seed, pair, it, h)) return;
    unsigned long long s = 0;
    unsigned c = 0;
    const int m = c1 - c0;
#pragma unroll 4
    for (int i = 0; i < m; ++i) {                         // the same correspondence in every lane: LDS broadcast; branch-free, so that the
        const float4 q = spt[i];                          // scheduler interleaves the fp64 chains of consecutive correspondences
        const double r2 = residual_sq(h, q.x, q.y, q.z, q.w);
        const bool near = r2 < a.tmax2;
        const unsigned e = stab[bin_of(near ? r2 : 0.0, a.bin_scale)];
        s += near ? e : 0u;
        c += r2 < a.thr2 ? 1u : 0u;
    }
    atomicAdd(a.hscore + (size_t)pair * a.iters_pad + it, s);
    atomicAdd(a.hcnt + (size_t)pair * a.iters_pad + it, c);
}

// After the first 256 hypotheses: an upper bound of the index the sequential loop stops at = min over the records (strict prefix maxima of
// the quality) among them of iterations_needed.  If the loop stops inside the first 256 the value does not matter (no later hypothesis is
// visited); if it does not, every record among the first 256 is one the loop sees, so its own bound is <= this one.
__global__ __launch_bounds__(256) void homog_bound_kernel(HgArgs a, int* __restrict__ bound) {
    __shared__ unsigned long long sc[256];
    __shared__ int bmin;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n = a.counts ? min(a.counts[pair], a.cap) : a.n_const;
    const unsigned long long mine = tid < a.iters ? a.hscore[(size_t)pair * a.iters_pad + tid] : 0ull;
    sc[tid] = mine;
    if (tid == 0) bmin = a.iters;
    __syncthreads();
    unsigned long long before = 0;
    for (int j = 0; j < tid; ++j) before = sc[j] > before ? sc[j] : before;
    if (n >= 4 && mine > before) atomicMin(&bmin, iterations_needed(a.hcnt[(size_t)pair * a.iters_pad + tid], n, a.log1mc, a.iters));
    __syncthreads();
    if (tid == 0) bound[pair] = bmin;
}

// ---- selection, refinement, mask --------------------------------------------------------------------------------------------------------
// Totals of N per-thread values over the workgroup, every thread gets them; the order of the additions is fixed.  Through LDS as a transpose:
// thread (k, j) adds 32 of the 256 entries of row k, thread k the 8 partial sums -- ~40 dependent additions and four barriers.  (A butterfly of
// wave shuffles costs 6 steps x 2 ds_bpermute per value, each waited for: with it the select kernel took 76-84 us instead of 52.)
constexpr int RED_PITCH = 257;
template <int N>
__device__ inline void block_sums(double (&v)[N], double* buf /* N * RED_PITCH + 9 * N doubles */) {
    static_assert(N * 8 <= 256, "one thread per (row, segment)");
    const int tid = threadIdx.x;
    double* part = buf + N * RED_PITCH;
    double* tot = part + N * 8;
    __syncthreads();                                       // buf free (previous use)
#pragma unroll
    for (int k = 0; k < N; ++k) buf[k * RED_PITCH + tid] = v[k];
    __syncthreads();
    if (tid < N * 8) {
        const int k = tid >> 3, j = tid & 7;
        const double* row = buf + k * RED_PITCH + j * 32;
        double t = 0.0;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) t += row[i];
        part[tid] = t;
    }
    __syncthreads();
    if (tid < N) {
        const double* q = part + tid * 8;
        tot[tid] = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])));
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = tot[k];
}

// Solve the 8x8 normal equations of the weighted inhomogeneous DLT (unknowns a = h00 h01 h02, b = h10 h11 h12, c = h20 h21; h22 = 1) from the 23 sums.
// The matrix is [[P 0 Cu] [0 P Cv] [Cu' Cv' R]] with one 3x3 block P = sum w p p' for both rows: Cholesky of P once, the 2x2 Schur complement
// S = R - Xu'Xu - Xv'Xv (X = L^-1 C) for c, back-substitution for a and b -- the block form of the dense 8x8 Cholesky solve (5 square roots
// instead of 8, ~1/5 of the dependent fp64 chain that every step of the refinement waits for).
__device__ inline bool solve_dlt(const double (&s)[hg::NSUM], double (&hn)[9]) {
    // sums: 0 xx 1 xy 2 x 3 yy 4 y 5 1 | 6 uxx 7 uxy 8 uyy 9 ux 10 uy 11 u | 12 vxx 13 vxy 14 vyy 15 vx 16 vy 17 v | 18 qxx 19 qxy 20 qyy 21 qx 22 qy (q = u^2+v^2)
    // P = [[xx xy x] [xy yy y] [x y 1]]
    bool ok = s[0] > 0.0;
    const double l00 = sqrt(s[0]), i00 = 1.0 / l00;
    const double l10 = s[1] * i00, l20 = s[2] * i00;
    const double d1 = s[3] - l10 * l10;
    ok = ok && d1 > 0.0;
    const double l11 = sqrt(d1), i11 = 1.0 / l11;
    const double l21 = (s[4] - l20 * l10) * i11;
    const double d2 = s[5] - l20 * l20 - l21 * l21;
    ok = ok && d2 > 0.0;
    const double l22 = sqrt(d2), i22 = 1.0 / l22;
    auto fwd = [&](double v0, double v1, double v2, double (&y)[3]) {
        y[0] = v0 * i00;
        y[1] = (v1 - l10 * y[0]) * i11;
        y[2] = (v2 - l20 * y[0] - l21 * y[1]) * i22;
    };
    auto bwd = [&](const double (&y)[3], double* x) {
        x[2] = y[2] * i22;
        x[1] = (y[1] - l21 * x[2]) * i11;
        x[0] = (y[0] - l10 * x[1] - l20 * x[2]) * i00;
    };
    // columns of Cu = -[[uxx uxy] [uxy uyy] [ux uy]], Cv likewise; right-hand sides gu = [ux uy u], gv = [vx vy v], gr = -[qx qy]
    double xu0[3], xu1[3], xv0[3], xv1[3], yu[3], yv[3];
    fwd(-s[6], -s[7], -s[9], xu0);
    fwd(-s[7], -s[8], -s[10], xu1);
    fwd(-s[12], -s[13], -s[15], xv0);
    fwd(-s[13], -s[14], -s[16], xv1);
    fwd(s[9], s[10], s[11], yu);
    fwd(s[15], s[16], s[17], yv);
    auto dot = [](const double (&p)[3], const double (&q)[3]) { return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]; };
    const double S00 = s[18] - dot(xu0, xu0) - dot(xv0, xv0);
    const double S01 = s[19] - dot(xu0, xu1) - dot(xv0, xv1);
    const double S11 = s[20] - dot(xu1, xu1) - dot(xv1, xv1);
    const double t0 = -s[21] - dot(xu0, yu) - dot(xv0, yv);
    const double t1 = -s[22] - dot(xu1, yu) - dot(xv1, yv);
    ok = ok && S00 > 0.0;
    const double m00 = sqrt(S00), j00 = 1.0 / m00;
    const double m10 = S01 * j00;
    const double e1 = S11 - m10 * m10;
    ok = ok && e1 > 0.0;
    const double m11 = sqrt(e1), j11 = 1.0 / m11;
    const double z0 = t0 * j00, z1 = (t1 - m10 * z0) * j11;
    const double c1 = z1 * j11, c0 = (z0 - m10 * c1) * j00;
    double ra[3], rb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { ra[k] = yu[k] - (xu0[k] * c0 + xu1[k] * c1); rb[k] = yv[k] - (xv0[k] * c0 + xv1[k] * c1); }
    bwd(ra, &hn[0]);
    bwd(rb, &hn[3]);
    hn[6] = c0; hn[7] = c1; hn[8] = 1.0;
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) fin = fin && (hn[i] - hn[i] == 0.0);
    return ok && fin;
}

__global__ __launch_bounds__(256) void homog_select_kernel(HgArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ double hsh[9];
    __shared__ int sel[4];
    __shared__ unsigned long long sc_sh;
    __shared__ unsigned cnt_sh;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int n = a.counts ? min(a.counts[pair], a.cap) : a.n_const;
    const PairPts pts(a, pair);
    unsigned char* mask = a.mask + (size_t)pair * a.cap;
    int32_t* info = a.info + pair * 8;
    double* Hout = a.H + pair * 9;

    // ---- the stopping rule of the sequential loop, on the score list
    unsigned long long* hs = reinterpret_cast<unsigned long long*>(lds_raw);
    unsigned* hc = reinterpret_cast<unsigned*>(lds_raw + (size_t)a.iters_pad * 8);
    for (int i = tid; i < a.iters; i += 256) { hs[i] = a.hscore[(size_t)pair * a.iters_pad + i]; hc[i] = a.hcnt[(size_t)pair * a.iters_pad + i]; }
    __syncthreads();
    if (tid == 0) {
        int best = -1, k_stop = a.iters, it = 0;
        unsigned long long best_s = 0;
        if (n >= 4) {
            for (; it < a.iters && it < k_stop; ++it) {
                if (hs[it] > best_s) {
                    best = it; best_s = hs[it];
                    const int need = iterations_needed(hc[it], n, a.log1mc, a.iters);
                    k_stop = need < k_stop ? need : k_stop;
                }
            }
        }
        sel[0] = best; sel[1] = it;
        double h[9];
        if (best >= 0) {
            make_hypothesis(pts, n, a.seed, pair, best, h);
#pragma unroll
            for (int k = 0; k < 9; ++k) hsh[k] = h[k];
        }
    }
    __syncthreads();
    const int best = sel[0], iters_run = sel[1];
    double* red = reinterpret_cast<double*>(lds_raw);       // the score list is dead: its LDS is the reduction buffer from here on
    if (best < 0) {
        for (int i = tid; i < a.cap; i += 256) mask[i] = 0;
        if (tid < 9) Hout[tid] = 0.0;
        if (tid < 8) info[tid] = tid == 2 ? iters_run : (tid == 1 ? -1 : (tid == 5 ? n : 0));
        return;
    }
    // ---- the first SEL_CACHE correspondences stay in LDS for all passes (every pass otherwise pays the index -> key-point round trip
    // again: ~1.5 us each, nine passes); longer lists re-read their tail
    float4* spt = reinterpret_cast<float4*>(lds_raw + a.sel_cache_off);
    for (int i = tid; i < min(n, hg::SEL_CACHE); i += 256) {
        float2 q0, q1;
        pts.get(i, q0, q1);
        spt[i] = make_float4(q0.x, q0.y, q1.x, q1.y);
    }
    __syncthreads();
    auto for_each = [&](auto&& f) {                        // f(i, (x0, y0, x1, y1)) for this thread's correspondences tid, tid + 256, ...
        for (int i = tid; i < n; i += 256) {
            float4 q;
            if (i < hg::SEL_CACHE) q = spt[i];
            else {
                float2 q0, q1;
                pts.get(i, q0, q1);
                q = make_float4(q0.x, q0.y, q1.x, q1.y);
            }
            f(i, q);
        }
    };
    // ---- Hartley normalisation of both point sets (conditioning of the normal equations only)
    double c[4] = {0, 0, 0, 0};
    for_each([&](int, const float4& q) { c[0] += q.x; c[1] += q.y; c[2] += q.z; c[3] += q.w; });
    block_sums(c, red);
    const double cx0 = c[0] / n, cy0 = c[1] / n, cx1 = c[2] / n, cy1 = c[3] / n;
    double dd[2] = {0, 0};
    for_each([&](int, const float4& q) {
        const double ax = q.x - cx0, ay = q.y - cy0, bx = q.z - cx1, by = q.w - cy1;
        dd[0] += sqrt(ax * ax + ay * ay); dd[1] += sqrt(bx * bx + by * by);
    });
    block_sums(dd, red);
    const double s0 = dd[0] > 0 ? 1.41421356237309504880 / (dd[0] / n) : 1.0, s1 = dd[1] > 0 ? 1.41421356237309504880 / (dd[1] / n) : 1.0;

    // ---- sigma-consensus++: re-weighted least squares while the quality rises
    double hcur[9], hbest[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { hcur[k] = hsh[k]; hbest[k] = hsh[k]; }
    unsigned long long s_best = 0;
    int lo_accepted = 0;
    for (int step = 0; step <= hg::LO_ITERS; ++step) {
        if (tid == 0) { sc_sh = 0ull; cnt_sh = 0u; }
        __syncthreads();
        double sm[hg::NSUM];
#pragma unroll
        for (int k = 0; k < hg::NSUM; ++k) sm[k] = 0.0;
        unsigned long long sc = 0;
        for_each([&](int, const float4& p) {
            const double r2 = residual_sq(hcur, p.x, p.y, p.z, p.w);
            if (r2 < a.tmax2) {
                const int b = bin_of(r2, a.bin_scale);
                sc += a.stab[b];
                const double w = a.wtab[b];
                const double x = (p.x - cx0) * s0, y = (p.y - cy0) * s0, u = (p.z - cx1) * s1, v = (p.w - cy1) * s1;
                const double wx = w * x, wy = w * y, wxx = wx * x, wxy = wx * y, wyy = wy * y, q = u * u + v * v;
                sm[0] += wxx; sm[1] += wxy; sm[2] += wx; sm[3] += wyy; sm[4] += wy; sm[5] += w;
                sm[6] += u * wxx; sm[7] += u * wxy; sm[8] += u * wyy; sm[9] += u * wx; sm[10] += u * wy; sm[11] += u * w;
                sm[12] += v * wxx; sm[13] += v * wxy; sm[14] += v * wyy; sm[15] += v * wx; sm[16] += v * wy; sm[17] += v * w;
                sm[18] += q * wxx; sm[19] += q * wxy; sm[20] += q * wyy; sm[21] += q * wx; sm[22] += q * wy;
            }
        });
        atomicAdd(&sc_sh, sc);
        block_sums(sm, red);                               // (its barriers also publish sc_sh)
        const unsigned long long s_now = sc_sh;
        if (s_now <= s_best) break;
#pragma unroll
        for (int k = 0; k < 9; ++k) hbest[k] = hcur[k];
        s_best = s_now;
        lo_accepted = step;
        if (step == hg::LO_ITERS) break;
        double hn[9];
        if (!solve_dlt(sm, hn)) break;
        // H = T1^-1 Hn T0
        double t[9];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            t[3 * r + 0] = hn[3 * r + 0] * s0;
            t[3 * r + 1] = hn[3 * r + 1] * s0;
            t[3 * r + 2] = hn[3 * r + 2] - (hn[3 * r + 0] * s0 * cx0 + hn[3 * r + 1] * s0 * cy0);
        }
        const double is1 = 1.0 / s1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            hcur[k] = t[k] * is1 + cx1 * t[6 + k];
            hcur[3 + k] = t[3 + k] * is1 + cy1 * t[6 + k];
            hcur[6 + k] = t[6 + k];
        }
        bool fin = true;
#pragma unroll
        for (int k = 0; k < 9; ++k) fin = fin && (hcur[k] - hcur[k] == 0.0);
        if (!fin) break;
        __syncthreads();                                   // sc_sh read by everybody before it is cleared again
    }
    // ---- inlier mask under the final model
    __syncthreads();
    if (tid == 0) cnt_sh = 0u;
    __syncthreads();
    unsigned cn = 0;
    for_each([&](int i, const float4& p) {
        const unsigned char mk = residual_sq(hbest, p.x, p.y, p.z, p.w) < a.thr2 ? 1 : 0;
        mask[i] = mk;
        cn += mk;
    });
    for (int i = (n > 0 ? n : 0) + tid; i < a.cap; i += 256) mask[i] = 0;      // rows beyond the pair's count
    atomicAdd(&cnt_sh, cn);
    __syncthreads();
    const int n_in = (int)cnt_sh;
    if (tid == 0) {
        const bool found = n_in >= 4;
        double nrm = hbest[8];
        if (!(fabs(nrm) > 1e-300)) {
            nrm = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) nrm += hbest[k] * hbest[k];
            nrm = sqrt(nrm);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) Hout[k] = found ? hbest[k] / nrm : 0.0;
        info[0] = found ? 1 : 0; info[1] = best; info[2] = iters_run; info[3] = n_in; info[4] = lo_accepted; info[5] = n;
        info[6] = (int)(s_best & 0xffffffffull); info[7] = (int)(s_best >> 32);
    }
}

size_t homography_workspace_bytes(int P, int max_iters) {
    const size_t pad = (size_t)ceil_div(max_iters, 256) * 256;
    return (size_t)hg::NBINS * 4 + (size_t)hg::NBINS * 8 + (size_t)P * pad * 12 + (size_t)P * 4 + 256;
}

void launch_homography_tables(double thr, unsigned* stab, double* wtab, hipStream_t st) {
    homog_tables_kernel<<<hg::NBINS / 256, 256, 0, st>>>(thr, stab, wtab, nullptr, nullptr, 0);
}

int launch_find_homography(const float* p0, const float* p1, const int64_t* idx0, const int64_t* idx1, int kcap, const int32_t* counts, int n_const, int P, int cap, double thr, int max_iters,
                           double confidence, unsigned long long seed, double* H, unsigned char* mask, int32_t* info, void* ws, hipStream_t st) {
    if (max_iters < 1 || max_iters > hg::MAX_ITERS || P > 65535) return -1;
    HgArgs a;
    a.p0 = p0; a.p1 = p1; a.idx0 = idx0; a.idx1 = idx1; a.kcap = idx0 ? kcap : cap; a.counts = counts; a.n_const = n_const; a.P = P; a.cap = cap; a.iters = max_iters;
    a.iters_pad = ceil_div(max_iters, 256) * 256;
    const double t_max = hg::MAX_THR_FACTOR * thr;
    a.thr2 = thr * thr; a.tmax2 = t_max * t_max; a.bin_scale = hg::NBINS / (t_max * t_max); a.log1mc = log(1.0 - confidence);
    a.seed = seed; a.sel_cache_off = 0; a.chunk = hg::PTS_PER_WG;
    unsigned char* w = static_cast<unsigned char*>(ws);
    a.wtab = reinterpret_cast<double*>(w); w += (size_t)hg::NBINS * 8;
    a.hscore = reinterpret_cast<unsigned long long*>(w); w += (size_t)P * a.iters_pad * 8;
    a.stab = reinterpret_cast<unsigned*>(w); w += (size_t)hg::NBINS * 4;
    a.hcnt = reinterpret_cast<unsigned*>(w); w += (size_t)P * a.iters_pad * 4;
    int* bound = reinterpret_cast<int*>(w);
    a.H = H; a.mask = mask; a.info = info;
    const int nhyp = P * a.iters_pad;
    int tg = ceil_div(nhyp, 256);
    tg = tg < hg::NBINS / 256 ? hg::NBINS / 256 : (tg > 1024 ? 1024 : tg);
    homog_tables_kernel<<<tg, 256, 0, st>>>(thr, a.stab, a.wtab, a.hscore, a.hcnt, nhyp);
    // few pairs: smaller chunks of correspondences, so that one pair still spreads over the chip (integer scores: any split gives the same sums)
    a.chunk = hg::PTS_PER_WG;
    while (a.chunk > 64 && (long)P * ceil_div(cap, a.chunk) < 256) a.chunk >>= 1;
    const int nblk = ceil_div(max_iters, hg::HYP_PER_WG), nch = ceil_div(cap, a.chunk);
    homog_score_kernel<<<dim3(1, nch, P), 256, 0, st>>>(a, 0, nullptr);
    if (nblk > 1) {
        homog_bound_kernel<<<P, 256, 0, st>>>(a, bound);
        homog_score_kernel<<<dim3(nblk - 1, nch, P), 256, 0, st>>>(a, 1, bound);
    }
    const size_t red_bytes = (size_t)(hg::NSUM * RED_PITCH + 9 * hg::NSUM) * sizeof(double);
    const size_t front = ((size_t)a.iters_pad * 12 > red_bytes ? (size_t)a.iters_pad * 12 : red_bytes) + 15 & ~(size_t)15;
    a.sel_cache_off = (int)front;
    const size_t lds = front + (size_t)hg::SEL_CACHE * sizeof(float4);
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(homog_select_kernel), 96 * 1024, attr);
    homog_select_kernel<<<P, 256, lds, st>>>(a);
    return 0;
}

}  // namespace xfh
