// Timing probe (not part of the product): one lane follows the outer border of a mask with the product's BitWin tracer;
// counts steps, window refills and cycles.   usage: trace_probe <mask.bin>   (int32 h, int32 w, h*w bytes)
#include "../../cylinder-pose-estimation_amd/csrc/cpe_dev.h"
#include <vector>
#include <cstdio>
using namespace cpe;

struct ProbeWin : BitWin {
    int refills = 0;
    long long refill_cycles = 0;
};
__device__ __forceinline__ unsigned nbr_mask(ProbeWin &bw, int x, int y)
{
    int p = x + 32 - bw.wx0, r = y - bw.wy0;
    if (p < 1 || p > 62 || r < 1 || r > BW_ROWS - 2) {
        long long t0 = clock64();
        unsigned v = bw.nbrs(x, y);
        bw.refills++;
        bw.refill_cycles += clock64() - t0;
        return v;
    }
    return bw.nbrs(x, y);
}

__global__ void k_probe(const uint32_t *bits, int h, int w, int root, long long *out_all, int lanes)
{
    long long *out = out_all + 8 * blockIdx.x;
    __shared__ unsigned long long s_win[BW_ROWS * 64];
    if ((int)threadIdx.x >= lanes) return;
    ProbeWin nz;
    nz.plane = bits; nz.ws = bit_row_words(w); nz.h = h; nz.win = s_win + threadIdx.x;
    StatVisitor sv;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    bool ok = trace_border(nz, root % w, root / w, false, sv, 1 << 24);
    long long t1 = clock64();
    long long w1 = wall_clock64();
    sv.finish();
    if (threadIdx.x != 0) return;
    out[0] = sv.npts; out[1] = nz.refills; out[2] = t1 - t0; out[3] = nz.refill_cycles; out[4] = ok; out[5] = sv.a00; out[6] = w1 - w0;
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    int h, w;
    if (!f || fread(&h, 4, 1, f) != 1 || fread(&w, 4, 1, f) != 1) return 1;
    std::vector<uint8_t> m((size_t)h * w);
    if (fread(m.data(), 1, m.size(), f) != m.size()) return 1;
    const int ws = bit_row_words(w);
    std::vector<uint32_t> bits((size_t)h * ws, 0);
    int root = -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (m[(size_t)y * w + x]) { bits[(size_t)y * ws + ((x + 32) >> 5)] |= 1u << ((x + 32) & 31); if (root < 0) root = y * w + x; }
    uint32_t *d; long long *o;
    hipMalloc(&d, bits.size() * 4); hipMalloc(&o, 64 * 4096);
    hipMemcpy(d, bits.data(), bits.size() * 4, hipMemcpyHostToDevice);
    const int blocks = argc > 2 ? atoi(argv[2]) : 1, lanes = argc > 3 ? atoi(argv[3]) : 1;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), 0, 0, d, h, w, root, o, lanes);
        long long r[7];
        hipMemcpy(r, o, 56, hipMemcpyDeviceToHost);
        printf("steps %lld refills %lld cycles %lld (%.0f per step) refill cycles %lld (%.0f each) ok %lld wall %.3f ms (100 MHz ticks %lld)\n", r[0], r[1], r[2],
               (double)r[2] / r[0], r[3], r[1] ? (double)r[3] / r[1] : 0.0, r[4], r[6] / 1e5, r[6]);
    }
    return 0;
}
