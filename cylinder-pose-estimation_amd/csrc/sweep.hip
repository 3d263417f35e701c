// The 17 binarisations of SimpleBlobDetector (util_cylinder.py:1857-1864; thresholds 50..210 step 10) as two growing
// union-find forests, built tile by tile in LDS.
//
// What the blob detector needs from a binarisation at threshold t is its component list: the enclosed dark components
// (4-connected; cv2.findContours follows their hole borders) and the bright components (8-connected; outer borders), each
// with the raster-first pixel where Suzuki-Abe start the border, plus pixel totals for the exact area prunes.  The dark
// set {v <= t} only grows with t and the bright set {v > t} only grows as t falls, so every pixel joins each forest once
// over the whole sweep.  The first version did those joins with device-scope atomicMin on a label plane in HBM; those
// execute at the memory side on this part and were the cost of the stage.  Here:
//
//   phase A (k_sw_tile, one workgroup per 64x64 tile): the tile's pixels join a union-find that lives in LDS, threshold by
//     threshold (ds atomics).  After every threshold the tile's local components are classified:
//       closed -- no pixel on a tile side that continues inside the working rectangle: it IS a component of the frame and
//                 goes straight to the result list of that threshold;
//       open   -- may continue in a neighbouring tile: it is handed to phase B as (local root, value).
//     The tile also emits, per threshold, the unions that glue open local components together: a tile-border pixel with
//     its local root, an open local root with the root that absorbed it, and a tile-border pixel with its member
//     neighbour across the tile border.
//   phase B (k_sw_global, one workgroup per frame): a second, small union-find over exactly those pixels (HBM atomics, but
//     only ~6 % of the pixels ever appear): per threshold apply that threshold's unions, add up the values of the open local
//     components per global root and emit the global components.
//
// Roots are always the smallest pixel index of their set (larger root linked under the smaller), i.e. the raster-first pixel,
// locally (tile raster order is frame raster order restricted to the tile) and globally.
// Bright components carry the pixel total of the holes they enclose (a hole of threshold t belongs to the bright component
// of the pixel west of its first pixel): the holes found by the dark sweep are binned by the tile of that west pixel and
// added inside the bright tile pass at the matching threshold.
#include "sweep.h"

namespace cpe {
namespace {

constexpr int TS = SW_TILE, TP = TS * TS, HWD = TS + 2;
constexpr int EV_CAP = 1280;                 // unions one tile can emit at one threshold (252 border pixels x (1 + 3) + absorbed open roots)
constexpr unsigned F_OPEN = 1u << 30, F_TOUCH = 1u << 31, F_MASK = F_OPEN | F_TOUCH, V_MASK = ~F_MASK;
enum { T_E = 0, T_A = 1, T_OUT = 2, T_TL = 3 };

__device__ __forceinline__ unsigned lds_find(unsigned *par, unsigned x)
{
    unsigned p = par[x];
    while (p != x) {
        const unsigned g = par[p];
        if (g != p) par[x] = g;      // path halving; a stale write re-points x at another ancestor (parents only decrease)
        x = p;
        p = g;
    }
    return x;
}
__device__ __forceinline__ void lds_unite(unsigned *par, unsigned a, unsigned b)
{
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const unsigned t = a; a = b; b = t; }
        const unsigned old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

struct TileRect { int x0, y0, x1, y1; };

template <bool DARK>
__global__ __launch_bounds__(256) void k_sw_tile(SweepBuffers B, int h, int w, FrameState *__restrict__ st)
{
    __shared__ uint8_t s_lev[HWD * HWD];
    __shared__ unsigned s_par[TP];
    __shared__ unsigned s_val[TP];            // dark: pixel count | flags; bright: enclosed-hole pixels of this threshold | F_OPEN
    __shared__ unsigned short s_roots[2][TP];
    __shared__ int2 s_ev[EV_CAP];
    __shared__ int s_nroots[2], s_cnt[4], s_n[4], s_base[4], s_cnt2[4];
    const int f = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int tiles_x = sw_tiles_x(w);
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const int X0 = tx * TS, Y0 = ty * TS;
    const TileRect r{st[f].crect[0], st[f].crect[1], st[f].crect[2], st[f].crect[3]};
    if (r.x1 < r.x0 || X0 > r.x1 || X0 + TS - 1 < r.x0 || Y0 > r.y1 || Y0 + TS - 1 < r.y0) return;
    const size_t N = (size_t)h * w;
    const uint8_t *im = B.cl + f * N;
    int *Gf = B.G + f * N, *accf = B.acc + f * N;
    int *S = B.sw + (size_t)f * SW_STRIDE;
    const int cap = sweep_cap(h, w);
    constexpr uint8_t OUTSIDE = DARK ? 255 : 0;      // never a member
    for (int i = t; i < HWD * HWD; i += 256) {
        const int ly = i / HWD - 1, lx = i - (ly + 1) * HWD - 1;
        const int gx = X0 + lx, gy = Y0 + ly;
        uint8_t l = OUTSIDE;
        if (gx >= r.x0 && gx <= r.x1 && gy >= r.y0 && gy <= r.y1) l = (uint8_t)sw_level(im[(size_t)gy * w + gx]);
        s_lev[i] = l;
    }
    for (int i = t; i < TP; i += 256) { s_par[i] = i; s_val[i] = 0; }
    if (t < 4) { s_cnt[t] = 0; s_cnt2[t] = 0; }
    if (t < 2) s_nroots[t] = 0;
    // which tile sides continue inside the working rectangle
    const bool openL = X0 - 1 >= r.x0, openR = X0 + TS <= r.x1, openT = Y0 - 1 >= r.y0, openB = Y0 + TS <= r.y1;
    // holes whose west pixel lies in this tile (bright pass)
    int hb0 = 0, hb1 = 0;
    if (!DARK) {
        const int ntiles = tiles_x * sw_tiles_y(h);
        hb0 = B.hb_off[(size_t)f * (2 * ntiles + 2) + tile];
        hb1 = B.hb_off[(size_t)f * (2 * ntiles + 2) + tile + 1];
    }
    __syncthreads();
    auto LEV = [&](int lx, int ly) -> int { return s_lev[(ly + 1) * HWD + lx + 1]; };
    auto gidx = [&](int lx, int ly) -> int { return (Y0 + ly) * w + X0 + lx; };
    auto gidx_of = [&](unsigned i) -> int { return (Y0 + (int)(i >> 6)) * w + X0 + (int)(i & 63); };
    auto node_init = [&](int g) { Gf[g] = g; accf[g] = 0; };
    auto stage = [&](int a, int b) {
        const int q = atomicAdd(&s_cnt[T_E], 1);
        if (q < EV_CAP) s_ev[q] = make_int2(a, b);
    };

    for (int step = 0; step < NTHR; step++) {
        const int k = DARK ? step : NTHR - 1 - step;          // threshold slot
        const int bucket = DARK ? k : k + 1;                  // the pixels that join now
        auto member = [&](int l) { return DARK ? l <= k : l >= k + 1; };
        auto older = [&](int l) { return DARK ? l < bucket : l > bucket; };
        const int cur = step & 1, nxt = cur ^ 1;
        // ---- (a) the new pixels join: unions with the member neighbours inside the tile
        for (int j = 0; j < TP / 256; j++) {
            const unsigned i = t + 256 * j;
            const int lx = i & 63, ly = i >> 6;
            if (LEV(lx, ly) != bucket) continue;
            auto link = [&](int dx, int dy) {
                const int nx = lx + dx, ny = ly + dy;
                if ((unsigned)nx >= (unsigned)TS || (unsigned)ny >= (unsigned)TS) return;
                const int ln = LEV(nx, ny);
                if (!member(ln)) return;
                const unsigned q = ny * TS + nx;
                if (ln == bucket && q > i) return;            // a pair of new pixels is united by the later one
                lds_unite(s_par, i, q);
            };
            link(-1, 0); link(0, -1); link(1, 0); link(0, 1);
            if (!DARK) { link(-1, -1); link(1, -1); link(-1, 1); link(1, 1); }
        }
        __syncthreads();
        // ---- (b) the new pixels: flatten, count, flags, unions that leave the tile
        for (int j = 0; j < TP / 256; j++) {
            const unsigned i = t + 256 * j;
            const int lx = i & 63, ly = i >> 6;
            const bool isnew = LEV(lx, ly) == bucket;
            unsigned root = 0xFFFFFFFFu;
            if (isnew) {
                root = lds_find(s_par, i);
                if (root != i) s_par[i] = root;
            }
            if (DARK) {   // pixel counts, one LDS atomic per distinct root of the wavefront
                unsigned long long active = __ballot(isnew);
                while (active) {
                    const int leader = __ffsll((long long)active) - 1;
                    const unsigned lk = __shfl(root, leader, 64);
                    const unsigned long long same = __ballot(root == lk) & active;
                    if (lane == leader) atomicAdd(&s_val[lk], (unsigned)__popcll(same));
                    active &= ~same;
                }
            }
            if (!isnew) continue;
            const int gx = X0 + lx, gy = Y0 + ly;
            unsigned fl = 0;
            if (DARK && (gx == r.x0 || gx == r.x1 || gy == r.y0 || gy == r.y1)) fl |= F_TOUCH;
            const bool onL = lx == 0 && openL, onR = lx == TS - 1 && openR, onT = ly == 0 && openT, onB = ly == TS - 1 && openB;
            const bool ring = onL || onR || onT || onB;
            if (ring) fl |= F_OPEN;
            if (fl) atomicOr(&s_val[root], fl);
            if (root == i) { const int q = atomicAdd(&s_nroots[nxt], 1); s_roots[nxt][q] = (unsigned short)i; }
            if (!ring) continue;
            const int pg = gidx(lx, ly);
            node_init(pg);
            if (root != i) { const int rg = gidx_of(root); node_init(rg); stage(pg, rg); }
            // member neighbours across the tile border: the pair is emitted by the pixel that joins later (the larger index
            // when both join now).  Diagonal pairs (bright) only where no straight pair already makes the connection.
            auto cross = [&](int dx, int dy) {
                const int l2 = LEV(lx + dx, ly + dy);
                if (!member(l2)) return;
                const int qg = gidx(lx + dx, ly + dy);
                if (older(l2) || qg < pg) stage(pg, qg);
            };
            if (onL) cross(-1, 0);
            if (onR) cross(1, 0);
            if (onT) cross(0, -1);
            if (onB) cross(0, 1);
            if (!DARK) {
                // diagonal neighbour (dx, dy) outside the tile; skipped when one of the two pixels completing the 2x2 square is
                // a member (then the pair is connected through that pixel: one straight tile-internal / cross link each)
                auto diag = [&](int dx, int dy) {
                    const int nx = lx + dx, ny = ly + dy;
                    if ((unsigned)nx < (unsigned)TS && (unsigned)ny < (unsigned)TS) return;   // inside the tile: done in (a)
                    const int gx2 = X0 + nx, gy2 = Y0 + ny;
                    if (gx2 < r.x0 || gx2 > r.x1 || gy2 < r.y0 || gy2 > r.y1) return;
                    if (member(LEV(lx + dx, ly)) || member(LEV(lx, ly + dy))) return;
                    cross(dx, dy);
                };
                diag(-1, -1); diag(1, -1); diag(-1, 1); diag(1, 1);
            }
        }
        // ---- (c) the local roots of the previous threshold: still a root, or absorbed (hand over value and flags)
        {
            const int nold = s_nroots[cur];
            for (int idx = t; idx < nold; idx += 256) {
                const unsigned r0 = s_roots[cur][idx];
                if (s_par[r0] == r0) {
                    const int q = atomicAdd(&s_nroots[nxt], 1);
                    s_roots[nxt][q] = (unsigned short)r0;
                    if (!DARK) atomicAnd(&s_val[r0], F_MASK);              // the enclosed total is per threshold
                } else {
                    const unsigned R = lds_find(s_par, r0);
                    const unsigned v = s_val[r0];
                    if (DARK && (v & V_MASK)) atomicAdd(&s_val[R], v & V_MASK);
                    if (v & F_MASK) atomicOr(&s_val[R], v & F_MASK);
                    if (v & F_OPEN) { const int rg = gidx_of(R); node_init(rg); stage(gidx_of(r0), rg); }
                }
            }
        }
        __syncthreads();
        if (!DARK) {
            // pixels of the holes of this threshold, each added to the bright component west of its first pixel
            for (int e = hb0 + t; e < hb1; e += 256) {
                const int2 he = B.hb_ent[(size_t)f * B.hb_cap + e];
                if ((he.x >> 12) != k) continue;
                atomicAdd(&s_val[lds_find(s_par, (unsigned)(he.x & 4095))], (unsigned)he.y);
            }
            __syncthreads();
        }
        // ---- (d1) classify the local components of this threshold, count the records
        const int nr = s_nroots[nxt];
        auto classify = [&](unsigned v, bool &isA, bool &isOut, bool &isTl) {
            isA = (v & F_OPEN) != 0;
            isOut = !isA && !(DARK && (v & F_TOUCH));
            const unsigned c = v & V_MASK;
            isTl = DARK && isOut && c > 3 && c < 5000;
        };
        for (int idx0 = 0; idx0 < nr; idx0 += 256) {
            const int idx = idx0 + t;
            bool isA = false, isOut = false, isTl = false;
            if (idx < nr) classify(s_val[s_roots[nxt][idx]], isA, isOut, isTl);
            const unsigned long long bA = __ballot(isA), bO = __ballot(isOut), bT = __ballot(isTl);
            if (lane == 0) {
                if (bA) atomicAdd(&s_cnt[T_A], __popcll(bA));
                if (bO) atomicAdd(&s_cnt[T_OUT], __popcll(bO));
                if (bT) atomicAdd(&s_cnt[T_TL], __popcll(bT));
            }
        }
        __syncthreads();
        // ---- reserve: one global atomic per record type for the whole tile
        if (t < 4) {
            int cnt = s_cnt[t];
            if (t == T_E && cnt > EV_CAP) { set_overflow(st[f], OVF_SWEEP); cnt = EV_CAP; }
            int base = 0;
            if (cnt > 0) {
                if (t == T_E || t == T_A) {
                    base = atomicAdd(&S[SW_PN], cnt);
                    const int b = atomicAdd(&S[SW_BN + t * NTHR + k], 1);
                    if (base + cnt > B.pool_cap || b >= B.bcap) { set_overflow(st[f], OVF_SWEEP); cnt = 0; }
                    else B.bh[(((size_t)f * 2 + t) * NTHR + k) * B.bcap + b] = make_int2(base, cnt);
                } else {
                    base = atomicAdd(&S[(t == T_TL ? SW_NT : (DARK ? SW_NH : SW_NL)) + k], cnt);
                    if (base + cnt > cap) { set_overflow(st[f], OVF_SWEEP); cnt = 0; }
                }
            }
            s_n[t] = cnt; s_base[t] = base; s_cnt[t] = 0; s_cnt2[t] = 0;
            if (t == 0) s_nroots[cur] = 0;                      // the next threshold's new list
        }
        __syncthreads();
        // ---- (d2) write
        {
            int2 *pool = B.pool + (size_t)f * B.pool_cap;
            const int nE = s_n[T_E];
            for (int idx = t; idx < nE; idx += 256) pool[s_base[T_E] + idx] = s_ev[idx];
            int2 *out = (DARK ? B.hl : B.bl) + ((size_t)f * NTHR + k) * cap;
            int2 *tlo = B.tl + ((size_t)f * NTHR + k) * cap;
            const bool wA = s_n[T_A] > 0, wO = s_n[T_OUT] > 0, wT = s_n[T_TL] > 0;
            for (int idx = t; idx < nr; idx += 256) {
                const unsigned r0 = s_roots[nxt][idx];
                const unsigned v = s_val[r0];
                bool isA, isOut, isTl;
                classify(v, isA, isOut, isTl);
                const int rg = gidx_of(r0);
                const int val = (int)(v & V_MASK);
                if (isA && wA) pool[s_base[T_A] + atomicAdd(&s_cnt2[T_A], 1)] = make_int2(rg, DARK ? (int)(val | ((v & F_TOUCH) ? (1 << 30) : 0)) : min(val, 5000));
                if (isOut && wO) out[s_base[T_OUT] + atomicAdd(&s_cnt2[T_OUT], 1)] = make_int2(rg, DARK ? val : min(val, 5000));
                if (isTl && wT) tlo[s_base[T_TL] + atomicAdd(&s_cnt2[T_TL], 1)] = make_int2(rg, val);
            }
        }
        // (the next threshold's (a) only touches s_par; its (b) / (c) come after a barrier)
    }
}

__device__ __forceinline__ void list_append(bool want, int2 value, int *counter, int2 *list, int cap, FrameState &S)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long b = __ballot(want);
    if (!b) return;
    const int leader = __ffsll((long long)b) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(b));
    base = __shfl(base, leader, 64);
    if (want) {
        const int q = base + __popcll(b & ((1ull << lane) - 1ull));
        if (q < cap) list[q] = value;
        else set_overflow(S, OVF_SWEEP);
    }
}

// phase B: the open local components of a frame, glued across the tile borders threshold by threshold
constexpr int GL_NT = 1024;
template <bool DARK>
__global__ __launch_bounds__(GL_NT) void k_sw_global(SweepBuffers B, int h, int w, FrameState *__restrict__ st)
{
    const size_t f = blockIdx.x, N = (size_t)h * w;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int *Gf = B.G + f * N, *accf = B.acc + f * N;
    uint8_t *tchf = B.tch + f * N;
    int *S = B.sw + f * SW_STRIDE;
    const int2 *pool = B.pool + f * B.pool_cap;
    const int cap = sweep_cap(h, w);
    for (int step = 0; step < NTHR; step++) {
        const int k = DARK ? step : NTHR - 1 - step;
        const uint8_t epoch = (uint8_t)(k + 1);
        const int2 *bhE = B.bh + ((f * 2 + 0) * NTHR + k) * B.bcap, *bhA = B.bh + ((f * 2 + 1) * NTHR + k) * B.bcap;
        const int nbE = min(S[SW_BN + k], B.bcap), nbA = min(S[SW_BN + NTHR + k], B.bcap);
        for (int b = wave; b < nbE; b += GL_NT / 64) {
            const int2 hd = bhE[b];
            for (int i = lane; i < hd.y; i += 64) { const int2 e = pool[hd.x + i]; uf_unite(Gf, e.x, e.y); }
        }
        __syncthreads();
        for (int b = wave; b < nbA; b += GL_NT / 64) {
            const int2 hd = bhA[b];
            for (int i = lane; i < hd.y; i += 64) {
                const int2 e = pool[hd.x + i];
                const int g = uf_find(Gf, e.x);
                const int val = e.y & 0x3FFFFFFF;
                if (val) atomicAdd(&accf[g], val);
                if (DARK && (e.y >> 30)) __hip_atomic_store(tchf + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        int2 *out = (DARK ? B.hl : B.bl) + (f * NTHR + k) * cap;
        int2 *tlo = B.tl + (f * NTHR + k) * cap;
        for (int b = wave; b < nbA; b += GL_NT / 64) {
            const int2 hd = bhA[b];
            for (int i0 = 0; i0 < hd.y; i0 += 64) {          // wave-uniform: list_append is a wavefront collective
                const int i = i0 + lane;
                bool isroot = false;
                int2 e = make_int2(0, 0);
                if (i < hd.y) { e = pool[hd.x + i]; isroot = uf_load(Gf, e.x) == e.x; }
                int total = 0;
                bool keep = false;
                if (isroot) {
                    total = atomicExch(&accf[e.x], 0);
                    keep = !DARK || __hip_atomic_load(tchf + e.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch;
                }
                list_append(keep, make_int2(e.x, total), &S[(DARK ? SW_NH : SW_NL) + k], out, cap, st[f]);
                if (DARK) list_append(keep && total > 3 && total < 5000, make_int2(e.x, total), &S[SW_NT + k], tlo, cap, st[f]);
            }
        }
        __syncthreads();
    }
}

// ---- holes -> tiles of their west pixels (input of the bright pass) ------------------------------------------------------
// hb_off[f]: [0, ntiles] offsets, [ntiles + 1, 2 ntiles + 1] fill cursors
__global__ __launch_bounds__(256) void k_hole_count(SweepBuffers B, int h, int w, int pass)
{
    const size_t f = blockIdx.y;
    const int k = blockIdx.z;
    const int *S = B.sw + f * SW_STRIDE;
    const int cap = sweep_cap(h, w);
    const int n = min(S[SW_NH + k], cap);
    const int tiles_x = sw_tiles_x(w), ntiles = tiles_x * sw_tiles_y(h);
    int *off = B.hb_off + f * (2 * ntiles + 2);
    const int2 *hl = B.hl + (f * NTHR + k) * cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int2 e = hl[i];
        const int p = e.x - 1, y = p / w, x = p - y * w;
        const int tile = (y / TS) * tiles_x + x / TS;
        if (pass == 0) atomicAdd(&off[tile], 1);
        else {
            const int q = atomicAdd(&off[ntiles + 1 + tile], 1);
            if (q < B.hb_cap) B.hb_ent[f * B.hb_cap + q] = make_int2(((y % TS) * TS + x % TS) | (k << 12), min(e.y, 5000));
        }
    }
}

__global__ __launch_bounds__(256) void k_hole_scan(SweepBuffers B, int h, int w, FrameState *__restrict__ st)
{
    // exclusive scan of the per-tile counts of one frame (<= 4225 tiles): one workgroup, chunks of 256 with a running carry
    __shared__ int s_part[256];
    __shared__ int s_carry;
    const size_t f = blockIdx.x;
    const int t = threadIdx.x;
    const int ntiles = sw_tiles_x(w) * sw_tiles_y(h);
    int *off = B.hb_off + f * (2 * ntiles + 2);
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < ntiles; i0 += 256) {
        const int i = i0 + t;
        const int v = i < ntiles ? off[i] : 0;
        s_part[t] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int add = t >= d ? s_part[t - d] : 0;
            __syncthreads();
            s_part[t] += add;
            __syncthreads();
        }
        const int excl = s_carry + s_part[t] - v;
        if (i < ntiles) { off[i] = excl; off[ntiles + 1 + i] = excl; }
        __syncthreads();
        if (t == 255) s_carry += s_part[255];
        __syncthreads();
    }
    if (t == 0) {
        off[ntiles] = s_carry;
        if (s_carry > B.hb_cap) { set_overflow(st[f], OVF_SWEEP); off[ntiles] = B.hb_cap; }
    }
}

__global__ void k_sw_reset(SweepBuffers B, int n, int ntiles)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    int *S = B.sw + (size_t)f * SW_STRIDE;
    S[SW_PN] = 0;
    for (int i = 0; i < 2 * NTHR; i++) S[SW_BN + i] = 0;
}

}  // namespace

int sweep_dark(const SweepBuffers &B, int n, int h, int w, FrameState *st, hipStream_t s)
{
    const int ntiles = sw_tiles_x(w) * sw_tiles_y(h);
    CPE_CHECK_ARG(B.bcap >= ntiles && w * h < (1 << 30), "sweep_dark: batch-header capacity below the tile count");
    CPE_LAUNCH_BEGIN();
    CPE_CHECK_HIP(hipMemsetAsync(B.tch, 0, (size_t)n * h * w, s));
    CPE_KLAUNCH(k_sw_reset, dim3((n + 63) / 64), dim3(64), 0, s, B, n, ntiles);
    CPE_KLAUNCH(k_sw_tile<true>, dim3(ntiles, n), dim3(256), 0, s, B, h, w, st);
    CPE_KLAUNCH(k_sw_global<true>, dim3(n), dim3(GL_NT), 0, s, B, h, w, st);
    CPE_CHECK_LAUNCH("sweep_dark");
    return CPE_OK;
}

int sweep_bright(const SweepBuffers &B, int n, int h, int w, FrameState *st, hipStream_t s)
{
    const int ntiles = sw_tiles_x(w) * sw_tiles_y(h);
    CPE_CHECK_ARG(B.bcap >= ntiles, "sweep_bright: batch-header capacity below the tile count");
    CPE_LAUNCH_BEGIN();
    // holes of all thresholds, binned by the tile of the pixel west of their first pixel
    CPE_CHECK_HIP(hipMemsetAsync(B.hb_off, 0, (size_t)n * (2 * ntiles + 2) * sizeof(int), s));
    const dim3 gh(frame_waves(n * NTHR, 2, 64), n, NTHR);
    CPE_KLAUNCH(k_hole_count, gh, dim3(256), 0, s, B, h, w, 0);
    CPE_KLAUNCH(k_hole_scan, dim3(n), dim3(256), 0, s, B, h, w, st);
    CPE_KLAUNCH(k_hole_count, gh, dim3(256), 0, s, B, h, w, 1);
    CPE_KLAUNCH(k_sw_reset, dim3((n + 63) / 64), dim3(64), 0, s, B, n, ntiles);
    CPE_KLAUNCH(k_sw_tile<false>, dim3(ntiles, n), dim3(256), 0, s, B, h, w, st);
    CPE_KLAUNCH(k_sw_global<false>, dim3(n), dim3(GL_NT), 0, s, B, h, w, st);
    CPE_CHECK_LAUNCH("sweep_bright");
    return CPE_OK;
}

}  // namespace cpe
