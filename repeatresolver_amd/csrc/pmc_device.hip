// pmc_device.hip -- MI355X (gfx950) implementation of MaxCorrelation's pair loop behind include/pmc.h.
//
// Reference: PhilippBongartz/RepeatResolver, MaxCorrelation.c ("MC:"), HilfsMaxCorrsRechner (MC:745-837): for every
// relevant variation i = (column ii, symbol k) and every relevant variation j in the columns jj >= ii + 20, as long as the
// two columns share at least `mincov` rows (the loop ends at the FIRST column that does not, MC:801-804), four sizes of
// intersections of row bit sets (MC:421-426) go into one upper tail of a hypergeometric distribution (MC:413-419, GSL);
// MaxCorrs[i] and MaxCorrs[j] keep the maximum (MC:816-817).
//   k_mc_bits   the bit sets, word-major: G[w][column * 5 + symbol], LC[w][column]  (MC:340-382), group sizes, coverage;
//               rows take their bit positions in the order of their first covered column, so that the rows of a column
//               sit in a narrow range of words (k_mc_colrange) and the products skip the words where a side is all zero
//   k_mc_end    per column the end of its jj loop
//   k_mc_pairs  a tiled bit-set product: a tile of PMC_TI relevant variations i stays in LDS (their group and their
//               column's coverage, a chunk of words at a time), every thread owns one relevant variation j and streams
//               its two bit sets once; the four counts per (i, j) live in registers, the epilogue evaluates the tail
//               in double precision and folds the maxima.
// The tail follows GSL's scheme (sum of pdf terms by ratio recurrences away from k; pdf = exp of three lnchoose): equal to
// the reference's up to rounding of lgamma / exp / log10, not bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "pmc.h"

#define PMC_TI 16                   // variations i per tile
#define PMC_NT 256                  // threads per block = variations j per tile
#define PMC_WC 32                   // words per LDS chunk
#define PMC_EI 8                    // columns per block of k_mc_end

static double g_ms[5] = {0, 0, 0, 0, 0};

__device__ __forceinline__ int code_of(unsigned char ch)
{
    switch (ch) {                                                     // MC:303-330
    case 'a': case 'A': return 0;
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    case '-': case '_': return 4;
    default: return 5;
    }
}

// first covered column of every row (W if it has none): block per row
__global__ __launch_bounds__(256) void k_mc_rowstart(int W, const unsigned char *__restrict__ text, int *__restrict__ start)
{
    __shared__ int s_min;
    if (threadIdx.x == 0) s_min = W;
    __syncthreads();
    const unsigned char *row = text + (size_t)blockIdx.x * W;
    for (int c0 = 0; c0 < W; c0 += 256) {                            // the start is usually near the row's left end
        const int c = c0 + threadIdx.x;
        if (c < W && code_of(row[c]) < 5) atomicMin(&s_min, c);
        __syncthreads();
        if (s_min < W) break;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[blockIdx.x] = s_min;
}

// thread = (column c, word w): 64 bit positions of one column.  The rows take their bit positions in the order of their first
// covered column (inv[position] = row): no count depends on the order, but the rows that cover a column then sit in a narrow
// range of words, and the products below only visit the words where both sides can be non-zero.
__global__ __launch_bounds__(256) void k_mc_bits(int T, int W, int sc, const unsigned char *__restrict__ text, const int *__restrict__ inv,
                                                 unsigned long long *__restrict__ G, unsigned long long *__restrict__ LC,
                                                 int *__restrict__ gsize, int *__restrict__ cover)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (c >= W) return;
    unsigned long long g[5] = {0, 0, 0, 0, 0};
    for (int r = 0; r < 64; ++r) {
        const int pos = w * 64 + r;
        if (pos >= T) break;
        const int k = code_of(text[(size_t)inv[pos] * W + c]);
        if (k < 5) g[k] |= 1ull << r;
    }
    unsigned long long lc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        G[(size_t)w * W * 5 + (size_t)c * 5 + k] = g[k];
        lc |= g[k];
        if (g[k]) atomicAdd(&gsize[c * 5 + k], __popcll(g[k]));
    }
    LC[(size_t)w * W + c] = lc;
    if (lc) atomicAdd(&cover[c], __popcll(lc));
}

// the words of a column's coverage set that are not zero: [wlo, whi)
__global__ __launch_bounds__(256) void k_mc_colrange(int W, int sc, const unsigned long long *__restrict__ LC, int2 *__restrict__ wr)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W) return;
    int lo = sc, hi = 0;
    for (int w = 0; w < sc; ++w)
        if (LC[(size_t)w * W + c]) { lo = min(lo, w); hi = w + 1; }
    wr[c] = make_int2(min(lo, hi), hi);
}

// MC:798-804: jend[ii] = the first jj >= ii + 20 whose column shares fewer than mincov rows with column ii (or W)
__global__ __launch_bounds__(256) void k_mc_end(int W, int sc, int mincov, const unsigned long long *__restrict__ LC, const int2 *__restrict__ wr,
                                                int *__restrict__ jend)
{
    extern __shared__ unsigned long long sL[];                         // [PMC_EI][sc]
    __shared__ int s_end[PMC_EI], s_open;
    const int ii0 = blockIdx.x * PMC_EI, tid = threadIdx.x;
    int wb = sc, we = 0;                                               // the words where any of the block's columns has a row
    for (int e = 0; e < PMC_EI; ++e)
        if (ii0 + e < W) { const int2 r = wr[ii0 + e]; if (r.y > r.x) { wb = min(wb, r.x); we = max(we, r.y); } }
    for (int i = tid; i < PMC_EI * sc; i += 256) {
        const int e = i / sc, w = i - e * sc;
        sL[i] = ii0 + e < W ? LC[(size_t)w * W + ii0 + e] : 0ull;
    }
    if (tid < PMC_EI) s_end[tid] = ii0 + tid < W ? W : -1;             // -1: no such column
    if (tid == 0) s_open = 0;
    __syncthreads();
    for (int base = ii0 + 20; base < W; base += 256) {
        const int jj = base + tid;
        int cnt[PMC_EI];
#pragma unroll
        for (int e = 0; e < PMC_EI; ++e) cnt[e] = 0;
        if (jj < W)
            for (int w = wb; w < we; ++w) {
                const unsigned long long l = LC[(size_t)w * W + jj];
#pragma unroll
                for (int e = 0; e < PMC_EI; ++e) cnt[e] += __popcll(l & sL[e * sc + w]);
            }
#pragma unroll
        for (int e = 0; e < PMC_EI; ++e)
            if (jj < W && jj >= ii0 + e + 20 && cnt[e] < mincov) atomicMin(&s_end[e], jj);
        __syncthreads();
        if (tid == 0) {
            int open = 0;
            for (int e = 0; e < PMC_EI; ++e) if (s_end[e] >= base + 256) open = 1;      // still unbroken beyond this chunk
            s_open = open;
        }
        __syncthreads();
        if (!s_open) break;
    }
    if (tid < PMC_EI && ii0 + tid < W) jend[ii0 + tid] = s_end[tid];
}

// lnf[n] = lgamma(n + 1), n <= rows (made on the host: no count exceeds the number of rows)
__device__ __forceinline__ double d_lnchoose(const double *__restrict__ lnf, unsigned n, unsigned m)
{
    if (m == n || m == 0) return 0.0;
    return lnf[n] - lnf[m] - lnf[n - m];
}

__device__ __forceinline__ double d_ln_hyper_pdf(const double *__restrict__ lnf, unsigned k, unsigned n1, unsigned n2, unsigned t)
{
    return d_lnchoose(lnf, n1, k) + d_lnchoose(lnf, n2, t - k) - d_lnchoose(lnf, n1 + n2, t);
}

__device__ __forceinline__ double d_hyper_pdf(const double *__restrict__ lnf, unsigned k, unsigned n1, unsigned n2, unsigned t)
{
    if (t > n1 + n2) t = n1 + n2;
    if (k > n1 || k > t) return 0.0;
    if (t > n2 && k + n2 < t) return 0.0;
    return exp(d_ln_hyper_pdf(lnf, k, n1, n2, t));
}

// gsl_cdf_hypergeometric_Q(k, n1, n2, t) = P(X > k)
__device__ double d_hyper_Q(const double *__restrict__ lnf, unsigned k, unsigned n1, unsigned n2, unsigned t)
{
    if (k >= n1 || k >= t) return 0.0;
    const double midpoint = ((double)t * n1) / ((double)n1 + n2);
    if (k < midpoint) {
        unsigned i = k;
        double s = d_hyper_pdf(lnf, i, n1, n2, t), P = s;
        while (i > 0) {
            s *= (i / (n1 - i + 1.0)) * ((n2 + i - t) / (t - i + 1.0));
            P += s;
            if (s / P < 2.2204460492503131e-16) break;
            i--;
        }
        return 1.0 - P;
    }
    unsigned i = k + 1;
    double s = d_hyper_pdf(lnf, i, n1, n2, t), Q = s;
    while (i < t) {
        s *= ((n1 - i) / (i + 1.0)) * ((t - i) / (n2 + i + 1.0 - t));
        Q += s;
        if (s / Q < 2.2204460492503131e-16) break;
        i++;
    }
    return Q;
}

// MC:413-434 PositiveSignificance (not inlined: the epilogue of k_mc_pairs calls it from an unrolled loop over register arrays)
// `floor`: what the two maxima this pair could raise already hold.  The tail is at least its first term, so
// -log10 pdf(schnitt) bounds the result from above: a pair that cannot raise either maximum is dropped before the sum
// (after a few thousand partners every variation's maximum is past what chance co-occurrence reaches, and almost every
// pair ends here -- the sums took 93 % of the kernel before).
__device__ __noinline__ double d_significance(const double *__restrict__ lnf, int schnitt, int cov, int gr1, int gr2, int size1, int size2, double floor)
{
    if (gr1 == 0 || gr2 == 0 || schnitt < 1) return 0.0;
    if ((unsigned)schnitt <= (unsigned)gr2 && schnitt <= gr1 && gr1 - schnitt <= cov - gr2) {
        const double zb = d_ln_hyper_pdf(lnf, (unsigned)schnitt, (unsigned)gr2, (unsigned)(cov - gr2), (unsigned)gr1) * -0.43429448190325182;
        if (zb < floor - 1e-6 && zb < 97.9) return 0.0;               // (beyond 98 the value is 98 + F, which the first term does not bound)
    }
    double Z = -1.0 * log10(d_hyper_Q(lnf, (unsigned)(schnitt - 1), (unsigned)gr2, (unsigned)(cov - gr2), (unsigned)gr1));
    if (isinf(Z) || Z > 99) Z = 99.0;
    if (isinf(Z) || Z > 98.0) Z = 98.0 + 2.0 * schnitt / (2.0 * schnitt + (size1 - schnitt) + (size2 - schnitt));     // F_beta(., ., 1), MC:396-410
    return Z;
}

struct McTile { int jlo, jhi; };    // positions in the list of relevant j this tile of i has to visit: [jlo, jhi)

// grid (tiles of i, chunks of PMC_NT relevant j); Ivar / Jvar = the relevant variations, ascending
__global__ __launch_bounds__(PMC_NT) void k_mc_pairs(int W, int sc, int nI, const int *__restrict__ Ivar, const int *__restrict__ Jvar,
                                                     const McTile *__restrict__ tiles, const unsigned long long *__restrict__ G,
                                                     const unsigned long long *__restrict__ LC, const int *__restrict__ gsize,
                                                     const int *__restrict__ jend, const double *__restrict__ lnf, const int2 *__restrict__ wr,
                                                     unsigned long long *maxc, unsigned long long *npairs)
{
    __shared__ unsigned long long sG[PMC_TI][PMC_WC], sL[PMC_TI][PMC_WC];
    __shared__ int s_i[PMC_TI], s_wb, s_we, s_jb, s_je;
    const McTile tl = tiles[blockIdx.x];
    const int b0 = tl.jlo + blockIdx.y * PMC_NT;
    if (b0 >= tl.jhi) return;
    const int tid = threadIdx.x, a0 = blockIdx.x * PMC_TI;
    if (tid < PMC_TI) s_i[tid] = a0 + tid < nI ? Ivar[a0 + tid] : -1;
    if (tid == 0) { s_wb = sc; s_we = 0; s_jb = sc; s_je = 0; }
    const int b = b0 + tid;
    const bool have = b < tl.jhi;
    const int j = have ? Jvar[b] : 0, jj = j / 5;
    __syncthreads();
    // the words where a row can be in a set of the tile's i AND in a set of the block's j (bit positions follow the rows' first
    // columns, so both are narrow ranges): everything else contributes zero to all four counts
    if (tid < PMC_TI && s_i[tid] >= 0) { const int2 r = wr[s_i[tid] / 5]; if (r.y > r.x) { atomicMin(&s_wb, r.x); atomicMax(&s_we, r.y); } }
    if (have) { const int2 r = wr[jj]; if (r.y > r.x) { atomicMin(&s_jb, r.x); atomicMax(&s_je, r.y); } }
    __syncthreads();
    const int wbeg = max(s_wb, s_jb), wend = min(s_we, s_je);
    int s[PMC_TI], g1[PMC_TI], g2[PMC_TI], cv[PMC_TI];
#pragma unroll
    for (int a = 0; a < PMC_TI; ++a) s[a] = g1[a] = g2[a] = cv[a] = 0;
    for (int w0 = wbeg; w0 < wend; w0 += PMC_WC) {
        __syncthreads();
        for (int t = tid; t < PMC_TI * PMC_WC; t += PMC_NT) {
            const int a = t / PMC_WC, w = w0 + t % PMC_WC, i = s_i[a];
            const bool ok = i >= 0 && w < wend;
            sG[a][t % PMC_WC] = ok ? G[(size_t)w * W * 5 + i] : 0ull;
            sL[a][t % PMC_WC] = ok ? LC[(size_t)w * W + i / 5] : 0ull;
        }
        __syncthreads();
        if (have) {
            const int wn = min(PMC_WC, wend - w0);
            for (int w = 0; w < wn; ++w) {
                const unsigned long long gj = G[(size_t)(w0 + w) * W * 5 + j], lj = LC[(size_t)(w0 + w) * W + jj];
#pragma unroll
                for (int a = 0; a < PMC_TI; ++a) {
                    const unsigned long long gi = sG[a][w], li = sL[a][w];
                    s[a] += __popcll(gi & gj); g1[a] += __popcll(gi & lj); g2[a] += __popcll(gj & li); cv[a] += __popcll(li & lj);
                }
            }
        }
    }
    if (!have) return;
    const int sj = gsize[j];
    const double curj = __longlong_as_double((long long)maxc[j]);
    double zj = 0.0;
    unsigned cnt = 0;
#pragma unroll
    for (int a = 0; a < PMC_TI; ++a) {
        const int i = s_i[a], ii = i / 5;
        if (i >= 0 && jj >= ii + 20 && jj < jend[ii]) {                // MC:798, MC:801-804
            ++cnt;
            const double floor = fmin(fmax(curj, zj), __longlong_as_double((long long)maxc[i]));
            double Z = d_significance(lnf, s[a], cv[a], g1[a], g2[a], gsize[i], sj, floor);
            if (!(Z > 0.0)) Z = 0.0;                                   // (-log10(1) is -0.0, whose bit pattern is the largest of all)
            zj = fmax(zj, Z);
            const unsigned long long zb = (unsigned long long)__double_as_longlong(Z);   // Z >= 0: the bit patterns order like the values
            if (zb > maxc[i]) atomicMax(&maxc[i], zb);                 // MC:816
        }
    }
    const unsigned long long zb = (unsigned long long)__double_as_longlong(zj);
    if (zb > maxc[j]) atomicMax(&maxc[j], zb);                         // MC:817
    if (cnt) atomicAdd(npairs, (unsigned long long)cnt);
}

#define HIPC(call)                                                                     \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "pmc: %s failed: %s\n", #call, hipGetErrorString(e_));    \
            return PWR_ERR_DEVICE;                                                     \
        }                                                                              \
    } while (0)

struct McBufs {
    unsigned char *text = nullptr; unsigned long long *G = nullptr, *LC = nullptr, *maxc = nullptr, *npairs = nullptr;
    int *gsize = nullptr, *cover = nullptr, *jend = nullptr, *Ivar = nullptr, *Jvar = nullptr; McTile *tiles = nullptr; double *lnf = nullptr;
    int *start = nullptr, *inv = nullptr; int2 *wr = nullptr;
    ~McBufs()
    {
        (void)hipFree(text); (void)hipFree(G); (void)hipFree(LC); (void)hipFree(maxc); (void)hipFree(npairs); (void)hipFree(gsize);
        (void)hipFree(cover); (void)hipFree(jend); (void)hipFree(Ivar); (void)hipFree(Jvar); (void)hipFree(tiles); (void)hipFree(lnf); (void)hipFree(start); (void)hipFree(inv); (void)hipFree(wr);
    }
};

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" int pmc_last_timing(double *ms5)
{
    if (!ms5) return PWR_ERR_ARG;
    for (int i = 0; i < 5; ++i) ms5[i] = g_ms[i];
    return PWR_OK;
}

extern "C" int pmc_maxcorrs(int T, int W, const unsigned char *text, int mincov, int device, double *maxcorrs)
{
    if (T <= 0 || W <= 0 || !text || !maxcorrs || mincov < 0) return PWR_ERR_ARG;
    if (T > PMC_MAX_ROWS || W > PMC_MAX_COLUMNS) return PWR_ERR_RANGE;
    if (hipSetDevice(device) != hipSuccess) return PWR_ERR_DEVICE;
    const double t0 = now_ms();
    const int sc = T / 64 + 1;                                                       // MC:338
    const size_t nv = (size_t)W * 5;
    McBufs d;
    if (hipMalloc(&d.text, (size_t)T * W) != hipSuccess || hipMalloc(&d.G, nv * sc * 8) != hipSuccess || hipMalloc(&d.LC, (size_t)W * sc * 8) != hipSuccess ||
        hipMalloc(&d.maxc, nv * 8) != hipSuccess || hipMalloc(&d.gsize, nv * 4) != hipSuccess || hipMalloc(&d.cover, (size_t)W * 4) != hipSuccess ||
        hipMalloc(&d.jend, (size_t)W * 4) != hipSuccess || hipMalloc(&d.npairs, 8) != hipSuccess || hipMalloc(&d.start, (size_t)T * 4) != hipSuccess ||
        hipMalloc(&d.inv, (size_t)T * 4) != hipSuccess || hipMalloc(&d.wr, (size_t)W * sizeof(int2)) != hipSuccess) return PWR_ERR_NOMEM;
    HIPC(hipMemcpy(d.text, text, (size_t)T * W, hipMemcpyHostToDevice));
    HIPC(hipMemset(d.gsize, 0, nv * 4)); HIPC(hipMemset(d.cover, 0, (size_t)W * 4)); HIPC(hipMemset(d.maxc, 0, nv * 8)); HIPC(hipMemset(d.npairs, 0, 8));
    // bit positions in the order of the rows' first covered columns
    hipLaunchKernelGGL(k_mc_rowstart, dim3(T), dim3(256), 0, 0, W, d.text, d.start);
    HIPC(hipGetLastError());
    std::vector<int> start(T), inv(T);
    HIPC(hipMemcpy(start.data(), d.start, (size_t)T * 4, hipMemcpyDeviceToHost));
    for (int r = 0; r < T; ++r) inv[r] = r;
    std::stable_sort(inv.begin(), inv.end(), [&](int a, int b) { return start[a] < start[b]; });
    HIPC(hipMemcpy(d.inv, inv.data(), (size_t)T * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mc_bits, dim3((W + 255) / 256, sc), dim3(256), 0, 0, T, W, sc, d.text, d.inv, d.G, d.LC, d.gsize, d.cover);
    HIPC(hipGetLastError());
    hipLaunchKernelGGL(k_mc_colrange, dim3((W + 255) / 256), dim3(256), 0, 0, W, sc, d.LC, d.wr);
    HIPC(hipGetLastError());
    std::vector<int> gsize(nv), cover(W), jend(W);
    HIPC(hipMemcpy(gsize.data(), d.gsize, nv * 4, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(cover.data(), d.cover, (size_t)W * 4, hipMemcpyDeviceToHost));
    const double t1 = now_ms();
    const size_t lds_end = (size_t)PMC_EI * sc * 8;
    hipLaunchKernelGGL(k_mc_end, dim3((W + PMC_EI - 1) / PMC_EI), dim3(256), lds_end, 0, W, sc, mincov, d.LC, d.wr, d.jend);
    HIPC(hipGetLastError());
    HIPC(hipMemcpy(jend.data(), d.jend, (size_t)W * 4, hipMemcpyDeviceToHost));
    const double t2 = now_ms();
    // the relevant variations (MC:796, MC:811): i needs more than mincov / 4 and fewer than all rows, and bases in more than
    // half of its column's covered rows; j only the first two
    const int maxgroup = T;                                                           // MC:1007
    std::vector<int> Ivar, Jvar;
    for (int c = 0; c < W; ++c) {
        const int baseno = gsize[c * 5] + gsize[c * 5 + 1] + gsize[c * 5 + 2] + gsize[c * 5 + 3];
        for (int k = 0; k < 5; ++k) {
            const int v = c * 5 + k;
            if (gsize[v] > mincov / 4 && gsize[v] < maxgroup) {
                Jvar.push_back(v);
                if (baseno > cover[c] / 2) Ivar.push_back(v);
            }
        }
    }
    const int nI = (int)Ivar.size(), nJ = (int)Jvar.size();
    unsigned long long npairs = 0;
    if (nI && nJ) {
        const int ntiles = (nI + PMC_TI - 1) / PMC_TI;
        std::vector<McTile> tiles(ntiles);
        int maxspan = 0;
        for (int t = 0; t < ntiles; ++t) {
            int clo = W, chi = 0;                                                     // columns any i of the tile visits
            for (int a = t * PMC_TI; a < std::min(nI, (t + 1) * PMC_TI); ++a) {
                const int ii = Ivar[a] / 5;
                clo = std::min(clo, ii + 20); chi = std::max(chi, jend[ii]);
            }
            const int jlo = (int)(std::lower_bound(Jvar.begin(), Jvar.end(), clo * 5) - Jvar.begin());
            const int jhi = (int)(std::lower_bound(Jvar.begin(), Jvar.end(), chi * 5) - Jvar.begin());
            tiles[t].jlo = jlo; tiles[t].jhi = std::max(jlo, jhi);
            maxspan = std::max(maxspan, tiles[t].jhi - tiles[t].jlo);
        }
        if (maxspan > 0) {
            if (hipMalloc(&d.Ivar, (size_t)nI * 4) != hipSuccess || hipMalloc(&d.Jvar, (size_t)nJ * 4) != hipSuccess ||
                hipMalloc(&d.tiles, sizeof(McTile) * ntiles) != hipSuccess) return PWR_ERR_NOMEM;
            HIPC(hipMemcpy(d.Ivar, Ivar.data(), (size_t)nI * 4, hipMemcpyHostToDevice));
            HIPC(hipMemcpy(d.Jvar, Jvar.data(), (size_t)nJ * 4, hipMemcpyHostToDevice));
            HIPC(hipMemcpy(d.tiles, tiles.data(), sizeof(McTile) * ntiles, hipMemcpyHostToDevice));
            std::vector<double> lnf((size_t)T + 2);
            for (int n = 0; n < T + 2; ++n) lnf[n] = std::lgamma(n + 1.0);
            if (hipMalloc(&d.lnf, lnf.size() * 8) != hipSuccess) return PWR_ERR_NOMEM;
            HIPC(hipMemcpy(d.lnf, lnf.data(), lnf.size() * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_mc_pairs, dim3(ntiles, (maxspan + PMC_NT - 1) / PMC_NT), dim3(PMC_NT), 0, 0, W, sc, nI, d.Ivar, d.Jvar, d.tiles,
                               d.G, d.LC, d.gsize, d.jend, d.lnf, d.wr, d.maxc, d.npairs);
            HIPC(hipGetLastError());
        }
    }
    HIPC(hipMemcpy(maxcorrs, d.maxc, nv * 8, hipMemcpyDeviceToHost));                 // (bit patterns of non-negative doubles)
    HIPC(hipMemcpy(&npairs, d.npairs, 8, hipMemcpyDeviceToHost));
    const double t3 = now_ms();
    g_ms[0] = t3 - t0; g_ms[1] = t1 - t0; g_ms[2] = t2 - t1; g_ms[3] = t3 - t2; g_ms[4] = (double)npairs;
    return PWR_OK;
}
