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
constexpr int TILE_NT = 256, TILE_NW = TILE_NT / 64;   // threads / wavefronts of a tile's workgroup
constexpr int REC_CAP = 2560;                // output records a tile keeps in LDS between two flushes
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

// LDS of k_sw_tile (dynamic: more than the 64 KB a kernel may declare statically)
constexpr int MAXLR = TP / 2;   // local components of a tile: at most one per two pixels (4-connectivity; fewer with 8)
struct TileLds {
    unsigned par[TP];
    unsigned val[TP];              // dark: pixel count | flags; bright: enclosed-hole pixels of this threshold | F_OPEN
    unsigned long long rnew[NBK][TS];   // per grey-level bucket and tile row: the pixels of that bucket (bit = column)
    unsigned long long rold[TS];        // per row: the members of the previous thresholds
    int2 rec[REC_CAP];             // records waiting for the next flush: unions (T_E), open components (T_A), results
    unsigned short roots[2][MAXLR];
    uint8_t rtag[REC_CAP];         // record type * 32 + threshold slot
    uint8_t lev[HWD * HWD];
    int anynew[NBK];
    int nroots[2], nrec, lost;
    int fcnt[4 * 32], fbase[4 * 32], fcnt2[4 * 32];   // flush: records / first output slot / write cursor per (type, threshold)
};

template <bool DARK>
__global__ __launch_bounds__(TILE_NT) void k_sw_tile(SweepBuffers B, int h, int w, FrameState *__restrict__ st)
{
    extern __shared__ unsigned long long sw_dyn[];
    TileLds &L = *reinterpret_cast<TileLds *>(sw_dyn);
    const int f = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
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
    for (int i = t; i < HWD * HWD; i += TILE_NT) {
        const int ly = i / HWD - 1, lx = i - (ly + 1) * HWD - 1;
        const int gx = X0 + lx, gy = Y0 + ly;
        uint8_t l = OUTSIDE;
        if (gx >= r.x0 && gx <= r.x1 && gy >= r.y0 && gy <= r.y1) l = (uint8_t)sw_level(im[(size_t)gy * w + gx]);
        L.lev[i] = l;
    }
    for (int i = t; i < TP; i += TILE_NT) { L.par[i] = i; L.val[i] = 0; }
    if (t < 2) L.nroots[t] = 0;
    if (t == 0) { L.nrec = 0; L.lost = 0; }
    if (t < NBK) L.anynew[t] = 0;
    if (t < TS) L.rold[t] = 0;
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
    if (B.dbg == 1) return;
    auto LEV = [&](int lx, int ly) -> int { return L.lev[(ly + 1) * HWD + lx + 1]; };
    auto gidx = [&](int lx, int ly) -> int { return (Y0 + ly) * w + X0 + lx; };
    auto gidx_of = [&](unsigned i) -> int { return (Y0 + (int)(i >> 6)) * w + X0 + (int)(i & 63); };
    auto node_init = [&](int g) { if (B.dbg == 5) return; Gf[g] = g; accf[g] = 0; };
    // Output records wait in LDS and leave the tile together: the list slots of all record types and thresholds are
    // reserved with one round of global atomics per flush (a returning device-scope atomic costs microseconds; once per
    // threshold it was most of the tile's time).
    auto push = [&](int type, int k, int a, int b) {
        const int q = atomicAdd(&L.nrec, 1);
        if (q < REC_CAP) { L.rec[q] = make_int2(a, b); L.rtag[q] = (uint8_t)(type * 32 + k); }
        else L.lost = 1;
    };
    auto flush = [&]() {        // called by all threads
        __syncthreads();
        const int nrec = min(L.nrec, REC_CAP);
        if (t < 4 * 32) { L.fcnt[t] = 0; L.fcnt2[t] = 0; }
        __syncthreads();
        for (int q = t; q < nrec; q += TILE_NT) atomicAdd(&L.fcnt[L.rtag[q]], 1);
        __syncthreads();
        if (t < 4 * 32 && L.fcnt[t] > 0) {
            const int type = t >> 5, k = t & 31, cnt = L.fcnt[t];
            int base;
            if (type == T_E || type == T_A) {
                base = atomicAdd(&S[SW_PN], cnt);
                const int b = atomicAdd(&S[SW_BN + type * NTHR + k], 1);
                if (base + cnt > B.pool_cap || b >= B.bcap) { set_overflow(st[f], OVF_SWEEP); base = -1; }
                else B.bh[(((size_t)f * 2 + type) * NTHR + k) * B.bcap + b] = make_int2(base, cnt);
            } else {
                base = atomicAdd(&S[(type == T_TL ? SW_NT : (DARK ? SW_NH : SW_NL)) + k], cnt);
                if (base + cnt > cap) { set_overflow(st[f], OVF_SWEEP); base = -1; }
            }
            L.fbase[t] = base;
        }
        __syncthreads();
        for (int q = t; q < nrec; q += TILE_NT) {
            const int tag = L.rtag[q], type = tag >> 5, k = tag & 31;
            const int base = L.fbase[tag];
            if (base < 0) continue;
            const int pos = base + atomicAdd(&L.fcnt2[tag], 1);
            int2 *dst = (type == T_E || type == T_A) ? B.pool + (size_t)f * B.pool_cap
                                                      : (type == T_TL ? B.tl : (DARK ? B.hl : B.bl)) + ((size_t)f * NTHR + k) * cap;
            dst[pos] = L.rec[q];
        }
        __syncthreads();
        if (t == 0) L.nrec = 0;
        __syncthreads();
    };
    auto stage = [&](int k, int a, int b) { push(T_E, k, a, b); };
    // A wavefront owns 16 rows of the tile (row = wave + 4 j), its lanes the 64 pixels of a row: rows are handled as bit masks.
    for (int j = 0; j < TS / TILE_NW; j++) {
        const int ly = wave + TILE_NW * j;
        const int l = LEV(lane, ly);
        for (int b = 0; b < NBK; b++) {
            const unsigned long long m = __ballot(l == b);
            if (lane == 0) {
                L.rnew[b][ly] = m;
                if (m) L.anynew[b] = 1;
            }
        }
    }
    __syncthreads();

    if (B.dbg == 2) return;
    int cur = 0;                                              // which root list is current
    for (int step = 0; step < NTHR; step++) {
        const int k = DARK ? step : NTHR - 1 - step;          // threshold slot
        const int bucket = DARK ? k : k + 1;                  // the pixels that join now
        auto member = [&](int l) { return DARK ? l <= k : l >= k + 1; };
        auto older = [&](int l) { return DARK ? l < bucket : l > bucket; };
        const bool grow = L.anynew[bucket] != 0;              // uniform: does any pixel of the tile join at this threshold?
        if (B.dbg == 3) continue;
        if (grow) {
            const int nxt = cur ^ 1;
            if (L.nrec > REC_CAP - EV_CAP) flush();             // room for the unions of this threshold (uniform: LDS word)
            // Work is handed out so that no thread waits on a chain of dependent LDS operations of another row:
            //   (a1) lane = pixel of a row, 16 rows per wavefront, masks loaded up front;
            //   (a2), (b1) thread = (row, slot): the thread walks the runs of new pixels of its row with pure bit arithmetic
            //        and takes every fourth one; (b2) thread = one of the 252 tile-border pixels.
            // ---- (a1) every horizontal run of new pixels points at its first pixel (plain stores: nothing else touches a
            //           new pixel's entry yet)
            {
                unsigned long long nb[TS / TILE_NW];
#pragma unroll
                for (int j = 0; j < TS / TILE_NW; j++) nb[j] = L.rnew[bucket][wave + TILE_NW * j];
#pragma unroll
                for (int j = 0; j < TS / TILE_NW; j++) {
                    const unsigned long long NB = nb[j];
                    if (!((NB >> lane) & 1ull)) continue;
                    const unsigned long long starts = NB & ~(NB << 1);
                    const int a = 63 - __clzll(starts & (~0ull >> (63 - lane)));
                    if (a != lane) L.par[(wave + TILE_NW * j) * TS + lane] = (wave + TILE_NW * j) * TS + a;
                }
            }
            __syncthreads();
            const int ry = t & 63, rq = t >> 6;                     // this thread's row and run slot
            const unsigned long long NBr = L.rnew[bucket][ry];
            const unsigned long long OLr = L.rold[ry];
            const unsigned long long UMr = ry > 0 ? (L.rold[ry - 1] | L.rnew[bucket][ry - 1]) : 0ull;
            const unsigned long long DOr = ry < TS - 1 ? L.rold[ry + 1] : 0ull;
            const unsigned long long DMr = ry < TS - 1 ? (DOr | L.rnew[bucket][ry + 1]) : 0ull;
            const unsigned long long MBr = OLr | NBr;
            // ---- (a2) unions: a run of new pixels with the old members left / right of it; vertical pairs where a stretch
            //           of vertically adjacent member pairs begins (the pair one column to the left is not a member pair: the
            //           rest of the stretch is connected through the horizontal links of the two rows); diagonal pairs (bright)
            //           only where neither pixel completing the 2x2 square is a member.  Pairs that leave the tile: see (b2).
            {
                unsigned long long m = NBr;
                for (int idx = 0; m; idx++) {
                    const int a = __builtin_ctzll(m);
                    const unsigned long long above = m >> a;                       // the run starts at bit 0 of `above`
                    const int len = above == ~0ull ? 64 : __builtin_ctzll(~above);
                    const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << a;
                    m &= ~run;
                    if ((idx % TILE_NW) != rq) continue;
                    const unsigned base = ry * TS;
                    if (a > 0 && ((OLr >> (a - 1)) & 1ull)) lds_unite(L.par, base + a, base + a - 1);
                    if (a + len < TS && ((OLr >> (a + len)) & 1ull)) lds_unite(L.par, base + a, base + a + len);
                    unsigned long long v = run & UMr & ~((MBr & UMr) << 1);
                    while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x - TS); }
                    v = run & DOr & ~((MBr & DMr) << 1);
                    while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x + TS); }
                    if (!DARK) {
                        v = run & ~UMr & (UMr >> 1) & ~(MBr >> 1);
                        while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x - TS + 1); }
                        v = run & ~UMr & (UMr << 1) & ~(MBr << 1);
                        while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x - TS - 1); }
                        v = run & ~DMr & (DOr >> 1) & ~(MBr >> 1);
                        while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x + TS + 1); }
                        v = run & ~DMr & (DOr << 1) & ~(MBr << 1);
                        while (v) { const int x = __builtin_ctzll(v); v &= v - 1; lds_unite(L.par, base + x, base + x + TS - 1); }
                    }
                }
            }
            __syncthreads();
            // ---- (b1) the runs of new pixels: flatten their first pixel, pixel counts, flags, new local roots
            {
                const int gy = Y0 + ry;
                // pixels of this row on the working rectangle's border / on a tile side that continues inside the rectangle
                unsigned long long tmask = 0, omask = 0;
                if (DARK) {
                    if (gy == r.y0 || gy == r.y1) tmask = ~0ull;
                    if (r.x0 >= X0 && r.x0 < X0 + TS) tmask |= 1ull << (r.x0 - X0);
                    if (r.x1 >= X0 && r.x1 < X0 + TS) tmask |= 1ull << (r.x1 - X0);
                }
                if ((ry == 0 && openT) || (ry == TS - 1 && openB)) omask = ~0ull;
                if (openL) omask |= 1ull;
                if (openR) omask |= 1ull << 63;
                unsigned long long m = NBr;
                for (int idx = 0; m; idx++) {
                    const int a = __builtin_ctzll(m);
                    const unsigned long long above = m >> a;
                    const int len = above == ~0ull ? 64 : __builtin_ctzll(~above);
                    const unsigned long long run = (len == 64 ? ~0ull : ((1ull << len) - 1ull)) << a;
                    m &= ~run;
                    if ((idx % TILE_NW) != rq) continue;
                    const unsigned i = ry * TS + a;
                    const unsigned root = lds_find(L.par, i);
                    if (root != i) L.par[i] = root;
                    else { const int q = atomicAdd(&L.nroots[nxt], 1); L.roots[nxt][q] = (unsigned short)i; }
                    unsigned add = DARK ? (unsigned)len : 0u;
                    if (run & tmask) add |= F_TOUCH;
                    if (run & omask) add |= F_OPEN;
                    // count in the low bits, flags in the two top bits: one atomic when the sum cannot carry into the flags
                    if (add & F_MASK) { if (add & V_MASK) atomicAdd(&L.val[root], add & V_MASK); atomicOr(&L.val[root], add & F_MASK); }
                    else if (add) atomicAdd(&L.val[root], add);
                }
            }
            // ---- (b2) the tile-border pixels that join now: nodes of phase B, with the unions that leave the tile
            if (t < 4 * TS - 4) {
                int lx, ly;
                if (t < TS) { lx = t; ly = 0; }
                else if (t < 2 * TS) { lx = t - TS; ly = TS - 1; }
                else if (t < 3 * TS - 2) { lx = 0; ly = t - 2 * TS + 1; }
                else { lx = TS - 1; ly = t - (3 * TS - 2) + 1; }
                const bool onL = lx == 0 && openL, onR = lx == TS - 1 && openR, onT = ly == 0 && openT, onB = ly == TS - 1 && openB;
                if ((onL || onR || onT || onB) && LEV(lx, ly) == bucket) {
                    const unsigned i = ly * TS + lx;
                    const unsigned root = lds_find(L.par, i);
                    const int pg = gidx(lx, ly);
                    node_init(pg);
                    if (root != i) { const int rg = gidx_of(root); node_init(rg); stage(k, pg, rg); }
                    // member neighbours across the tile border: the pair is emitted by the pixel that joins later (the larger
                    // index when both join now).  Diagonal pairs (bright) only where no straight pair makes the connection.
                    auto cross = [&](int dx, int dy) {
                        const int l2 = LEV(lx + dx, ly + dy);
                        if (!member(l2)) return;
                        const int qg = gidx(lx + dx, ly + dy);
                        if (older(l2) || qg < pg) stage(k, pg, qg);
                    };
                    if (onL) cross(-1, 0);
                    if (onR) cross(1, 0);
                    if (onT) cross(0, -1);
                    if (onB) cross(0, 1);
                    if (!DARK) {
                        // diagonal neighbour (dx, dy) outside the tile; skipped when one of the two pixels completing the 2x2
                        // square is a member (then the pair is connected through that pixel: one straight link each)
                        auto diag = [&](int dx, int dy) {
                            const int nx = lx + dx, ny = ly + dy;
                            if ((unsigned)nx < (unsigned)TS && (unsigned)ny < (unsigned)TS) return;   // inside the tile: (a2)
                            const int gx2 = X0 + nx, gy2 = Y0 + ny;
                            if (gx2 < r.x0 || gx2 > r.x1 || gy2 < r.y0 || gy2 > r.y1) return;
                            if (member(LEV(lx + dx, ly)) || member(LEV(lx, ly + dy))) return;
                            cross(dx, dy);
                        };
                        diag(-1, -1); diag(1, -1); diag(-1, 1); diag(1, 1);
                    }
                }
            }
            // ---- (c) the local roots of the previous threshold: still a root, or absorbed (hand over value and flags)
            {
                const int nold = L.nroots[cur];
                for (int idx = t; idx < nold; idx += TILE_NT) {
                    const unsigned r0 = L.roots[cur][idx];
                    if (L.par[r0] == r0) {
                        const int q = atomicAdd(&L.nroots[nxt], 1);
                        L.roots[nxt][q] = (unsigned short)r0;
                    } else {
                        const unsigned R = lds_find(L.par, r0);
                        const unsigned v = L.val[r0];
                        if (DARK && (v & V_MASK)) atomicAdd(&L.val[R], v & V_MASK);
                        if (v & F_MASK) atomicOr(&L.val[R], v & F_MASK);
                        if (v & F_OPEN) { const int rg = gidx_of(R); node_init(rg); stage(k, gidx_of(r0), rg); }
                    }
                }
            }
            // the rows' member masks for the next threshold (read again only after the barriers below)
            for (int j = 0; j < TS / TILE_NW; j++) {
                const int ly = wave + TILE_NW * j;
                if (lane == 0) L.rold[ly] |= L.rnew[bucket][ly];
            }
            __syncthreads();
            if (t == 0) L.nroots[cur] = 0;                        // becomes the new list of the next growing threshold
            cur = nxt;
        }
        if (!DARK) {
            // pixels of the holes of this threshold, each added to the bright component west of its first pixel
            for (int e = hb0 + t; e < hb1; e += TILE_NT) {
                const int2 he = B.hb_ent[(size_t)f * B.hb_cap + e];
                if ((he.x >> 12) != k) continue;
                atomicAdd(&L.val[lds_find(L.par, (unsigned)(he.x & 4095))], (unsigned)he.y);
            }
            __syncthreads();
        }
        if (B.dbg == 4) continue;
        // ---- (d) the local components of this threshold become records: closed ones are results, open ones go to phase B
        const int nr = L.nroots[cur];
        constexpr int CH = REC_CAP / 4;                         // roots per round: at most two records each
        for (int c0 = 0; c0 < nr; c0 += CH) {
            const int c1 = min(nr, c0 + CH);
            if (L.nrec + 2 * (c1 - c0) > REC_CAP) flush();
            for (int idx = c0 + t; idx < c1; idx += TILE_NT) {
                const unsigned r0 = L.roots[cur][idx];
                const unsigned v = L.val[r0];
                const int rg = gidx_of(r0);
                const int val = (int)(v & V_MASK);
                if (v & F_OPEN) push(T_A, k, rg, DARK ? (int)(val | ((v & F_TOUCH) ? (1 << 30) : 0)) : min(val, 5000));
                else if (!(DARK && (v & F_TOUCH))) {
                    push(T_OUT, k, rg, DARK ? val : min(val, 5000));
                    if (DARK && val > 3 && val < 5000) push(T_TL, k, rg, val);
                }
                if (!DARK && val) L.val[r0] = v & F_MASK;     // the enclosed total is per threshold
            }
            __syncthreads();
        }
    }
    flush();
    if (t == 0 && L.lost) set_overflow(st[f], OVF_SWEEP);
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
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sw_tile<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TileLds)));
    CPE_KLAUNCH(k_sw_tile<true>, dim3(ntiles, n), dim3(TILE_NT), sizeof(TileLds), s, B, h, w, st);
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
    CPE_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sw_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TileLds)));
    CPE_KLAUNCH(k_sw_tile<false>, dim3(ntiles, n), dim3(TILE_NT), sizeof(TileLds), s, B, h, w, st);
    CPE_KLAUNCH(k_sw_global<false>, dim3(n), dim3(GL_NT), 0, s, B, h, w, st);
    CPE_CHECK_LAUNCH("sweep_bright");
    return CPE_OK;
}

}  // namespace cpe
