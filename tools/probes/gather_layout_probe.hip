// Probe for VERDICT r05 item 6 ("one 16-bit copy instead of two -- only if a layout serves both"): what do stage 2's row gathers cost when
// a 16-bit row (E = 768: 1 536 bytes) is read from
//   mode 0  the row-major GATHER slab (12 consecutive 128-byte lines: what mfar_score_rows_kernel<SRC_F16G> reads today)
//   mode 1  the scan's TILE layout as it is ([64 rows][16 dims] 2 KB tiles: 48 segments of 32 bytes, 2 KB apart)
//   mode 2  a tile layout with 64-byte segments ([64 rows][32 dims] 4 KB tiles: 24 segments, 4 KB apart) -- needs a re-tiled scan kernel
//   mode 3  128-byte segments ([64 rows][64 dims] 8 KB tiles: 12 segments, 8 KB apart)
// for random rows of a 1 M-row field (1.5 GB: far larger than L2 / MALL).  Every lane group fetches one row with 16-byte loads and folds it
// into a checksum; reported: rows/s, useful GB/s, and rocprofv3 FETCH_SIZE per row when run under --pmc FETCH_SIZE.
// build: hipcc -O3 --offload-arch=gfx950 gather_layout_probe.hip -o /tmp/gather_layout_probe ; run: /tmp/gather_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one wave = 8 rows per iteration: 8 lanes per row
template <int MODE>
__global__ void __launch_bounds__(256) gather(const char* __restrict__ slab, const unsigned* __restrict__ rows, int n_rows_req, unsigned* out) {
    const int lane = threadIdx.x & 63, sub = lane & 7;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    unsigned acc = 0;
    for (int i = wave * 8 + (lane >> 3); i < n_rows_req; i += n_waves * 8) {
        const size_t r = rows[i];
        // 96 granules of 16 bytes per row; lane `sub` takes granules sub, sub + 8, ...
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int gran = sub + 8 * g;                       // 0 .. 95: dims 8 gran .. 8 gran + 7
            size_t off;
            if (MODE == 0) off = r * 1536 + (size_t)gran * 16;
            else {
                constexpr int SEG = MODE == 1 ? 32 : (MODE == 2 ? 64 : 128);     // bytes of one row inside a tile
                constexpr int GPS = SEG / 16;                    // granules per segment
                const int seg = gran / GPS, gi = gran % GPS;
                const size_t blk = r >> 6, rr = r & 63;
                const size_t n_seg = 1536 / SEG;
                off = (blk * n_seg + seg) * (size_t)(64 * SEG) + rr * SEG + gi * 16;
            }
            const u32x4 v = *(const u32x4*)(slab + off);
            acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
    }
    if (acc == 0x12345678u) out[0] = acc;                       // keep the loads alive
}

int main() {
    const size_t n_rows = 1u << 20, bytes = n_rows * 1536;
    char* slab;
    CHK(hipMalloc(&slab, bytes));
    CHK(hipMemset(slab, 1, bytes));
    const int N = 1 << 20;                                      // gathered rows per launch (~ 128 queries x 8 fields x 1 000 candidates)
    std::vector<unsigned> h(N);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (unsigned)(s % n_rows); }
    unsigned *rows, *out;
    CHK(hipMalloc(&rows, N * 4));
    CHK(hipMalloc(&out, 4));
    CHK(hipMemcpy(rows, h.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const char* names[4] = {"row-major gather slab (12 x 128 B contiguous)", "scan tiles, 32-byte segments (48 x 32 B, 2 KB apart)",
                            "tiles with 64-byte segments (24 x 64 B, 4 KB apart)", "tiles with 128-byte segments (12 x 128 B, 8 KB apart)"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipEventRecord(e0));
            if (mode == 0) gather<0><<<2048, 256>>>(slab, rows, N, out);
            if (mode == 1) gather<1><<<2048, 256>>>(slab, rows, N, out);
            if (mode == 2) gather<2><<<2048, 256>>>(slab, rows, N, out);
            if (mode == 3) gather<3><<<2048, 256>>>(slab, rows, N, out);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("mode %d  %-58s %8.3f ms  %7.1f M rows/s  %7.1f GB/s useful\n", mode, names[mode], best, N / best / 1e3, (double)N * 1536 / best / 1e6);
    }
    return 0;
}
