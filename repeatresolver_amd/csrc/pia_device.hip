// pia_device.hip -- MI355X (gfx950) implementation of the InitialAligner's hot loop behind include/pia.h.
//
// Reference: PhilippBongartz/RepeatResolver, InitialAligner.c ("IA:"), IntoAligner (IA:282-453): for every read a full
// (read length) x (template length) edit-distance matrix with free start and end along the template, one byte of direction
// per cell, then a traceback.  The reads are independent (the reference strides them over pthreads, IA:457-549).
//   recurrence (IA:300-328): E(x,y) = min(E(x-1,y-1) + [read x != template y], E(x,y-1) + 1, E(x-1,y) + 1),
//   E(-1,y) = 0, E(x,-1) = x + 1; ties: diagonal first, then left ONLY IF strictly smaller, then up only if strictly smaller.
//
// Unit costs make neighbouring cells differ by -1, 0 or +1, so a row is kept as two bit vectors over the template (Pv: E(x,y) -
// E(x,y-1) = +1, Mv: = -1) and advanced 32 columns per handful of integer instructions (Myers 1999, in the block form of
// Hyyro 2001 with a carried-in step delta per word).  What the traceback needs of the reference's direction byte follows from
// the same vectors:  the diagonal wins (codes 0 / 3, IA:308-312) unless the cell equals its diagonal neighbour although the
// bases differ (D0 & ~Eq), and then `left` (code 1) wins over `up` (code 2) exactly when E(x,y) = E(x,y-1) + 1 (the new Pv
// bit), because one of the two must equal the minimum and `up` is only taken when strictly smaller (IA:314-322).
//
// One wave owns one read: lane l holds the words [l * WPL, (l + 1) * WPL) of the row vectors in registers and works on read
// base x = step - l, so the step delta leaving its last word reaches lane l + 1 one step later through a lane shift -- no LDS,
// no barrier, 64 rows in flight per wave.  Pass 1 stores nothing and yields the distance and the entry column (IA:333-345);
// the traceback from there cannot leave the diagonal band of half-width `dist` around the entry's diagonal (each step off the
// diagonal costs one), so pass 2 repeats the rows and stores two bits per cell for that band only -- a tenth of the full
// matrices at sequencing error rates -- and the same wave then walks the traceback (IA:347-446) through what it has stored.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <new>
#include <numeric>
#include <vector>

#include "pia.h"

#define IA_MAXWPL 35                // 70 000 template bases / (64 lanes x 32 bits)
#define IA_NBUF 4                   // batches of pass 2 in flight, each with its buffer and stream

struct IaRead {
    long long boff;                 // offset of its bases (and of its alignment) in the call's arrays
    long long coff;                 // offset (uint2) of its band of direction bits, pass 2
    int L1;
    int dist, entry;                // pass 1: Row[entry] and the entry column (IA:333-345); dist = -1: traceback left the band
    int nl;                         // lanes stored per row (pass 2)
};

// The stored band of row x are the columns [c - dist, c + dist] around the entry's diagonal c = entry - (L1 - 1 - x), widened
// to whole lanes (32 * WPL columns each): first stored lane of row x
__device__ __forceinline__ int band_lane0(const IaRead &rd, int x, int colshift_mul)
{
    const int lo = rd.entry - (rd.L1 - 1 - x) - rd.dist;
    return (int)(((unsigned)(lo > 0 ? lo : 0) >> 5) / (unsigned)colshift_mul);
}

// IA:347-446, the traceback, walked by the wave that has just stored the read's band.  The walk is one chain of dependent
// steps, so it is kept in scalar registers: each round, lane i fetches the three words of row x - i around the diagonal from
// the current cell (64 rows per memory latency), the wave then steps through them with v_readlane, and lane i keeps the
// placement of its row for one coalesced store.  A step that needs a word outside the fetched three starts the next round.
template <int WPL>
__device__ __forceinline__ void wave_trace(const IaRead &rd, IaRead *rdp, const uint2 *__restrict__ cw, int *__restrict__ al, int lane)
{
    constexpr int WPS = (WPL + 1) & ~1;
    int x = rd.L1 - 1, y = rd.entry;                                  // the columns right of the entry are skipped (IA:359-364)
    bool lost = false;
    while (x >= 0 && y >= 0) {                                        // IA:366-383
        const int xtop = x, ytop = y;
        const int xr = xtop - lane, gb = (ytop - lane) >> 5;          // lane i: row xtop - i, the word of column ytop - i
        uint2 w0 = make_uint2(0u, 0u), w1 = w0, w2 = w0;
        if (xr >= 0 && gb >= -1) {
            const int l0 = band_lane0(rd, xr, WPL);
            const uint2 *rowp = cw + (size_t)xr * rd.nl * WPS;
            auto ld = [&](int g, uint2 &v) {
                const int ln = g / WPL, rel = ln - l0;
                if (g >= 0 && g < 64 * WPL && (unsigned)rel < (unsigned)rd.nl) v = rowp[rel * WPS + (g - ln * WPL)];
            };
            ld(gb - 1, w0); ld(gb, w1); ld(gb + 1, w2);
        }
        int myal = -1;
        for (;;) {
            const int k = xtop - x;
            if (k >= 64 || x < 0 || y < 0) break;
            const int g = y >> 5, b = y & 31;
            const int j = g - ((ytop - k) >> 5) + 1;
            if ((unsigned)j > 2u) break;                              // (never at k = 0: every round moves on)
            if ((unsigned)(g / WPL - band_lane0(rd, x, WPL)) >= (unsigned)rd.nl) { lost = true; break; }
            uint32_t nd, lf;
            if (j == 0) { nd = __builtin_amdgcn_readlane(w0.x, k); lf = __builtin_amdgcn_readlane(w0.y, k); }
            else if (j == 1) { nd = __builtin_amdgcn_readlane(w1.x, k); lf = __builtin_amdgcn_readlane(w1.y, k); }
            else { nd = __builtin_amdgcn_readlane(w2.x, k); lf = __builtin_amdgcn_readlane(w2.y, k); }
            if (!((nd >> b) & 1u)) {                                  // substitution / match: base x sits on template position y
                if (lane == k) myal = y;
                x = __builtin_amdgcn_readfirstlane(x - 1); y = __builtin_amdgcn_readfirstlane(y - 1);
            } else if ((lf >> b) & 1u) y = __builtin_amdgcn_readfirstlane(y - 1);      // template base skipped
            else x = __builtin_amdgcn_readfirstlane(x - 1);           // read base between two template bases: stays -1
        }
        if (lost) break;
        if (lane < xtop - x) al[xtop - lane] = myal;                  // the rows this round has finished
    }
    if (lost) { if (lane == 0) rdp->dist = -1; return; }              // left the stored band: cannot happen
    for (int i = x - lane; i >= 0; i -= 64) al[i] = -1;               // IA:389-394
}

template <int WPL, bool STORE>
__global__ __launch_bounds__(64) void k_ia_bits(const uint32_t *__restrict__ tplanes, int L2, const char *__restrict__ bases,
                                                IaRead *reads, const int *__restrict__ order, uint2 *__restrict__ codes, int *__restrict__ align)
{
    IaRead *rdp = &reads[order[blockIdx.x]];
    const IaRead rd = *rdp;
    const int L1 = rd.L1, lane = threadIdx.x;
    if (L1 <= 0) { if (!STORE && lane == 0) { rdp->dist = 0; rdp->entry = L2 - 1; } return; }
    constexpr int NW = 64 * WPL;
    constexpr int WPS = (WPL + 1) & ~1;           // stored words per lane and row (even: 16-byte stores)
    // lanes right of the last column that matters never influence the ones left of them
    const int lastcol = STORE ? rd.entry : L2 - 1;
    const int lastlane = (lastcol >> 5) / WPL;
    uint32_t T0[WPL], T1[WPL], Pv[WPL], Mv[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) { T0[w] = tplanes[lane * WPL + w]; T1[w] = tplanes[NW + lane * WPL + w]; Pv[w] = 0; Mv[w] = 0; }   // IA:296: row -1 is flat
    const char *read = bases + rd.boff;
    int msg = 0;                                  // base code (2 bits) | step delta leaving the lane's last word: +1 (bit 2), -1 (bit 3)
    int chunk = 0;
    const int steps = L1 + lastlane;
    for (int tau = 0; tau < steps; ++tau) {
        if ((tau & 63) == 0) { const int i = tau + lane; chunk = i < L1 ? (read[i] >> 1) & 3 : 0; }     // a c g t -> 0 1 3 2
        int m = __shfl_up(msg, 1);
        const int c0 = __builtin_amdgcn_readlane(chunk, tau & 63);
        if (lane == 0) m = c0 | 4;                // E(x,-1) - E(x-1,-1) = +1 (IA:301)
        const int x = tau - lane;
        if (x >= 0 && x < L1 && lane <= lastlane) {
            const uint32_t r0 = (m & 1) ? ~0u : 0u, r1 = (m & 2) ? ~0u : 0u;
            uint32_t hp = (m >> 2) & 1, hn = (m >> 3) & 1;
            uint32_t ND[STORE ? WPL : 1];
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                const uint32_t Eq = ~(T0[w] ^ r0) & ~(T1[w] ^ r1);
                const uint32_t Xv = Eq | Mv[w], Eqh = Eq | hn;
                const uint32_t Xh = (((Eqh & Pv[w]) + Pv[w]) ^ Pv[w]) | Eqh;
                uint32_t Ph = Mv[w] | ~(Xh | Pv[w]), Mh = Pv[w] & Xh;
                const uint32_t nd = (Xh | Mv[w]) & ~Eq;                 // equals its diagonal neighbour although the bases differ
                const uint32_t hp2 = Ph >> 31, hn2 = Mh >> 31;
                Ph = (Ph << 1) | hp; Mh = (Mh << 1) | hn;
                Pv[w] = Mh | ~(Xv | Ph);
                Mv[w] = Ph & Xv;
                hp = hp2; hn = hn2;
                if (STORE) ND[w] = nd;
            }
            msg = (m & 3) | (hp << 2) | (hn << 3);
            if (STORE) {
                // few lanes of a wave are inside the band at any step: they store all their words, no test per word.
                // (Measured on the benchmark reads, one batch: pass 2 takes 54 ms without these stores, 84 ms with them; the
                // traceback behind it adds 4 ms.  Each lane's 128-byte line leaves as eight 16-byte requests, but staging
                // the lines of a step in LDS so that eight lanes write one line per instruction was slower, 124-140 ms
                // against 110 ms in nine batches: more registers or eight more LDS writes per lane and step; non-temporal
                // stores 309 ms: the pieces then reach HBM unmerged.)
                const int rel = lane - band_lane0(rd, x, WPL);
                if ((unsigned)rel < (unsigned)rd.nl) {
                    uint2 *dst = codes + rd.coff + ((size_t)x * rd.nl + rel) * WPS;
#pragma unroll
                    for (int w = 0; w + 1 < WPL; w += 2) *reinterpret_cast<uint4 *>(dst + w) = make_uint4(ND[w], Pv[w], ND[w + 1], Pv[w + 1]);
                    if (WPL & 1) dst[WPL - 1] = make_uint2(ND[WPL - 1], Pv[WPL - 1]);
                }
            }
        }
    }
    if (STORE) {
        __threadfence();                          // the band is read back by other lanes of this wave
        wave_trace<WPL>(rd, rdp, codes + rd.coff, align + rd.boff, lane);
        return;
    }
    // IA:333-345: minimum of the last row over the columns L2-1 .. 1 (column 0 only if it is the only one), ties -> largest y;
    // E(L1-1, y) = L1 + sum over j <= y of (Pv - Mv)
    __shared__ uint32_t sP[64 * WPL], sM[64 * WPL];        // (a walk over register arrays would be unrolled 32 * WPL times)
    int tot = 0;
#pragma unroll
    for (int w = 0; w < WPL; ++w) { tot += __popc(Pv[w]) - __popc(Mv[w]); sP[lane * WPL + w] = Pv[w]; sM[lane * WPL + w] = Mv[w]; }
    int incl = tot;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    int cur = L1 + incl - tot;
    unsigned long long key = ~0ull;
#pragma unroll 1
    for (int w = 0; w < WPL; ++w) {
        const uint32_t p = sP[lane * WPL + w], q = sM[lane * WPL + w];
        const int ybase = (lane * WPL + w) * 32;
#pragma unroll 1
        for (int b = 0; b < 32; ++b) {
            cur += (int)((p >> b) & 1u) - (int)((q >> b) & 1u);
            const int y = ybase + b;
            if (y < L2 && (y >= 1 || L2 == 1)) {
                const unsigned long long k2 = ((unsigned long long)(unsigned)cur << 32) | (unsigned)(~(unsigned)y);
                key = k2 < key ? k2 : key;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(key, o); key = v < key ? v : key; }
    if (lane == 0) { rdp->dist = (int)(key >> 32); rdp->entry = (int)(~(unsigned)key); }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct pia_ctx {
    int device = 0, L2 = 0, WPL = 1;
    uint32_t *d_planes = nullptr;                 // [2][64 * WPL]: bit y of plane p = bit p of the code of template base y
    hipStream_t stream = nullptr;                 // = streams[0]
    hipStream_t streams[IA_NBUF] = {};
    uint2 *codes[IA_NBUF] = {};                   // direction bits of the batches of pass 2 in flight, kept between calls
    size_t codes_cap[IA_NBUF] = {};               // in uint2
    unsigned long long cells = 0;
    double fill_ms = 0.0;
    double t_ms[6] = {0, 0, 0, 0, 0, 0};          // last pia_align: total, set-up + upload, pass 1, pass 2 with the tracebacks, its batches, download
    size_t mem_budget = 0;                        // bytes of direction bits per batch of pass 2; 0: 22 GB (measured: the first ~96 GB a process
                                                  // allocates come at once, every further GB takes 30 ms)
};

#define HIPC(call)                                                                     \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "pia: %s failed: %s\n", #call, hipGetErrorString(e_));    \
            return PWR_ERR_DEVICE;                                                     \
        }                                                                              \
    } while (0)

// the kernels take a base's code from two bits of its character ((c >> 1) & 3: a c g t, either case): anything else would
// silently alias one of the four, so it is refused at the boundary
static inline bool ia_is_base(char ch)
{
    switch (ch) { case 'a': case 'c': case 'g': case 't': case 'A': case 'C': case 'G': case 'T': return true; default: return false; }
}

extern "C" int pia_create(pia_ctx **out, const char *templ, int templ_len, int device)
{
    if (!out || !templ || templ_len < 0) return PWR_ERR_ARG;
    if (templ_len > PIA_MAX_LINE) return PWR_ERR_RANGE;                              // IA:214 Template[70000]
    for (int y = 0; y < templ_len; ++y) if (!ia_is_base(templ[y])) return PWR_ERR_INPUT;     // (the reference's reader leaves nothing else, IA:190-209)
    pia_ctx *c = new (std::nothrow) pia_ctx();
    if (!c) return PWR_ERR_NOMEM;
    c->device = device; c->L2 = templ_len;
    c->WPL = std::max(1, (templ_len + 2047) / 2048);
    const int NW = 64 * c->WPL;
    std::vector<uint32_t> planes(2 * (size_t)NW, 0u);
    for (int y = 0; y < templ_len; ++y) {
        const int code = (templ[y] >> 1) & 3;
        if (code & 1) planes[y >> 5] |= 1u << (y & 31);
        if (code & 2) planes[NW + (y >> 5)] |= 1u << (y & 31);
    }
    if (hipSetDevice(device) != hipSuccess) { delete c; return PWR_ERR_DEVICE; }
    if (hipMalloc(&c->d_planes, planes.size() * 4) != hipSuccess) { delete c; return PWR_ERR_NOMEM; }
    if (hipMemcpy(c->d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreate(&c->streams[0]) != hipSuccess || hipStreamCreate(&c->streams[1]) != hipSuccess ||
        hipStreamCreate(&c->streams[2]) != hipSuccess || hipStreamCreate(&c->streams[3]) != hipSuccess) {
        for (hipStream_t st : c->streams) if (st) (void)hipStreamDestroy(st);
        (void)hipFree(c->d_planes); delete c; return PWR_ERR_DEVICE;
    }
    static_assert(IA_NBUF == 4, "one hipStreamCreate per buffer above");
    c->stream = c->streams[0];
    *out = c;
    return PWR_OK;
}

extern "C" void pia_destroy(pia_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (hipStream_t st : c->streams) if (st) (void)hipStreamDestroy(st);
    if (c->d_planes) (void)hipFree(c->d_planes);
    for (uint2 *p : c->codes) (void)hipFree(p);
    delete c;
}

extern "C" int pia_get_stats(pia_ctx *c, unsigned long long *cells, double *fill_ms)
{
    if (!c) return PWR_ERR_ARG;
    if (cells) *cells = c->cells;
    if (fill_ms) *fill_ms = c->fill_ms;
    return PWR_OK;
}

extern "C" int pia_set_option(pia_ctx *c, const char *key, long long value)
{
    if (!c || !key) return PWR_ERR_ARG;
    if (!strcmp(key, "mem_budget")) { if (value < 0) return PWR_ERR_ARG; c->mem_budget = (size_t)value; return PWR_OK; }
    return PWR_ERR_ARG;
}

extern "C" int pia_get_timing(pia_ctx *c, double *ms6)
{
    if (!c || !ms6) return PWR_ERR_ARG;
    for (int i = 0; i < 6; ++i) ms6[i] = c->t_ms[i];
    return PWR_OK;
}

static double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

template <int WPL>
static void launch_bits(pia_ctx *c, bool store, int n, const char *d_bases, IaRead *d_reads, const int *d_order, uint2 *d_codes, int *d_align, hipStream_t st)
{
    if (c->WPL == WPL) {
        if (store) hipLaunchKernelGGL((k_ia_bits<WPL, true>), dim3(n), dim3(64), 0, st, c->d_planes, c->L2, d_bases, d_reads, d_order, d_codes, d_align);
        else hipLaunchKernelGGL((k_ia_bits<WPL, false>), dim3(n), dim3(64), 0, st, c->d_planes, c->L2, d_bases, d_reads, d_order, d_codes, d_align);
        return;
    }
    if constexpr (WPL < IA_MAXWPL) launch_bits<WPL + 1>(c, store, n, d_bases, d_reads, d_order, d_codes, d_align, st);
}

struct DevBufs {                    // freed on every way out of pia_align
    char *bases = nullptr; IaRead *reads = nullptr; int *order = nullptr; int *align = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    std::vector<hipEvent_t> ev;
    ~DevBufs()
    {
        (void)hipFree(bases); (void)hipFree(reads); (void)hipFree(order); (void)hipFree(align);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

extern "C" int pia_align(pia_ctx *c, int nreads, const char *bases, const long long *off, int *align, int *dist)
{
    if (!c || nreads < 0 || !off || (nreads && (!bases || !align || !dist))) return PWR_ERR_ARG;
    if (nreads == 0) return PWR_OK;
    if (hipSetDevice(c->device) != hipSuccess) return PWR_ERR_DEVICE;
    for (int j = 0; j < nreads; ++j) {
        const long long l = off[j + 1] - off[j];
        if (l < 0) return PWR_ERR_ARG;
        if (l > PIA_MAX_READ) return PWR_ERR_RANGE;                                   // IA:742
    }
    const double t_start = now_ms();
    for (double &t : c->t_ms) t = 0;
    const long long b0 = off[0], nb = off[nreads] - off[0];
    for (long long i = 0; i < nb; ++i) if (!ia_is_base(bases[b0 + i])) return PWR_ERR_INPUT;
    std::vector<IaRead> hr(nreads);
    std::vector<int> order(nreads);
    for (int j = 0; j < nreads; ++j) {
        hr[j].boff = off[j] - b0; hr[j].coff = 0; hr[j].L1 = (int)(off[j + 1] - off[j]); hr[j].dist = 0; hr[j].entry = c->L2 - 1; hr[j].nl = 0;
        c->cells += (unsigned long long)hr[j].L1 * (unsigned long long)c->L2;
    }
    // the longest reads first: the waves that run longest start first
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hr[a].L1 > hr[b].L1; });
    if (c->L2 == 0) {                                                                 // nothing to align to: every base unaligned
        for (long long i = 0; i < nb; ++i) align[b0 + i] = -1;
        for (int j = 0; j < nreads; ++j) dist[j] = hr[j].L1;                          // Row[-1 + 0] is never read by the reference; E(x,-1) = x + 1
        return PWR_OK;
    }
    DevBufs d;
    HIPC(hipEventCreate(&d.e0)); HIPC(hipEventCreate(&d.e1));
    if (hipMalloc(&d.bases, std::max<long long>(nb, 1)) != hipSuccess || hipMalloc(&d.reads, sizeof(IaRead) * nreads) != hipSuccess ||
        hipMalloc(&d.order, sizeof(int) * nreads) != hipSuccess || hipMalloc(&d.align, sizeof(int) * std::max<long long>(nb, 1)) != hipSuccess) return PWR_ERR_NOMEM;
    if (nb) HIPC(hipMemcpyAsync(d.bases, bases + b0, nb, hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemcpyAsync(d.reads, hr.data(), sizeof(IaRead) * nreads, hipMemcpyHostToDevice, c->stream));
    HIPC(hipMemcpyAsync(d.order, order.data(), sizeof(int) * nreads, hipMemcpyHostToDevice, c->stream));
    float ms = 0;
    HIPC(hipStreamSynchronize(c->stream));
    c->t_ms[1] = now_ms() - t_start;
    // pass 1: distance and entry column of every read
    HIPC(hipEventRecord(d.e0, c->stream));
    launch_bits<1>(c, false, nreads, d.bases, d.reads, d.order, nullptr, nullptr, c->stream);
    HIPC(hipGetLastError());
    HIPC(hipEventRecord(d.e1, c->stream));
    HIPC(hipMemcpyAsync(hr.data(), d.reads, sizeof(IaRead) * nreads, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (hipEventElapsedTime(&ms, d.e0, d.e1) == hipSuccess) { c->fill_ms += ms; c->t_ms[2] = ms; }
    // pass 2 in batches whose bands fit the budget, IA_NBUF of them in flight, each with its buffer and stream: the tail of
    // one batch (its longest reads) runs beside the bulk of the next ones
    // The budget per buffer: 22 GB by default, but never more than the card has to spare right now (other contexts or ranks
    // may share it); if an allocation fails all the same, the batches are planned again with half of it.
    size_t budget = c->mem_budget ? c->mem_budget : (size_t)22 << 30;
    if (!c->mem_budget) {
        size_t free_b = 0, total_b = 0, have = 0;
        for (size_t cap : c->codes_cap) have += cap * sizeof(uint2);                 // (what this context holds already counts as available)
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, std::max<size_t>((free_b + have) / 4 * 3 / IA_NBUF, (size_t)64 << 20));
    }
    const size_t WPS = (size_t)((c->WPL + 1) & ~1);
    size_t maxwords = 0;
    std::vector<int> bend;                                                            // batch ends in `order`
    for (;;) {
        maxwords = 0;
        bend.clear();
        size_t words = 0;
        for (int k = 0; k < nreads; ++k) {
            IaRead &r = hr[order[k]];
            r.nl = (2 * r.dist) / (32 * c->WPL) + 2;
            const size_t w = (size_t)r.L1 * (size_t)r.nl * WPS;
            if (words && (words + w) * sizeof(uint2) > budget) { bend.push_back(k); maxwords = std::max(maxwords, words); words = 0; }
            r.coff = (long long)words;
            words += (w + 15) & ~(size_t)15;                                          // every read's band starts on its own 128-byte line
        }
        bend.push_back(nreads);
        maxwords = std::max(maxwords, words);
        bool ok = true;
        for (size_t b = 0; b < std::min<size_t>(bend.size(), IA_NBUF) && ok; ++b)
            if (c->codes_cap[b] < maxwords) {                                         // kept for the next call: large allocations are slow
                (void)hipFree(c->codes[b]); c->codes[b] = nullptr; c->codes_cap[b] = 0;
                if (hipMalloc(&c->codes[b], std::max<size_t>(maxwords, 1) * sizeof(uint2)) != hipSuccess) { (void)hipGetLastError(); ok = false; }
                else c->codes_cap[b] = maxwords;
            }
        if (ok) break;
        if (bend.size() == (size_t)nreads || budget <= ((size_t)16 << 20)) return PWR_ERR_NOMEM;   // one read per batch already, or nothing left to halve
        budget /= 2;
    }
    HIPC(hipMemcpyAsync(d.reads, hr.data(), sizeof(IaRead) * nreads, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    const double t_p2 = now_ms();
    d.ev.resize(2 * bend.size(), nullptr);
    for (hipEvent_t &e : d.ev) HIPC(hipEventCreate(&e));
    int k0 = 0;
    for (size_t bi = 0; bi < bend.size(); ++bi) {
        const int k1 = bend[bi], n = k1 - k0;
        hipStream_t st = c->streams[bi % IA_NBUF];
        HIPC(hipEventRecord(d.ev[2 * bi], st));
        launch_bits<1>(c, true, n, d.bases, d.reads, d.order + k0, c->codes[bi % IA_NBUF], d.align, st);
        HIPC(hipGetLastError());
        HIPC(hipEventRecord(d.ev[2 * bi + 1], st));
        k0 = k1;
    }
    for (hipStream_t st : c->streams) HIPC(hipStreamSynchronize(st));
    c->t_ms[3] = now_ms() - t_p2;
    c->t_ms[4] = (double)bend.size();
    for (size_t bi = 0; bi < bend.size(); ++bi) {
        if (hipEventElapsedTime(&ms, d.ev[2 * bi], d.ev[2 * bi + 1]) == hipSuccess) c->fill_ms += ms;
    }
    const double t_dl = now_ms();
    if (nb) HIPC(hipMemcpyAsync(align + b0, d.align, sizeof(int) * nb, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(hr.data(), d.reads, sizeof(IaRead) * nreads, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int j = 0; j < nreads; ++j) {
        if (hr[j].dist < 0) return PWR_ERR_INTERNAL;                                  // a traceback left its band: cannot happen
        dist[j] = hr[j].dist;
    }
    c->t_ms[5] = now_ms() - t_dl;
    c->t_ms[0] = now_ms() - t_start;
    return PWR_OK;
}
