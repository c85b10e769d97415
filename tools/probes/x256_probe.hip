// Micro-probe: would a 256-query-column fp16 scan keep up with HBM?  (the wide pass of mfar_stage1.h has 128 columns)
//   MODE 0: the k-loop of mfar_stage1_f16w_kernel: per wave and k-step 64 rows x 128 queries (2 x 16 B of docs per lane,
//           4 query fragments from LDS, 8 MFMAs)
//   MODE 1: 32 rows x 256 queries per wave and k-step (16 B of docs per lane, 8 query fragments from LDS, 8 MFMAs): twice the
//           MFMAs and LDS reads per doc byte, the same 128 accumulator registers
//   MODE 2: MODE 1 + the query stage of every k-step DMA-loaded into a 6-slot LDS ring (8 KB per step from L2)
// MODES 0/1 read query fragments from a fixed LDS tile; none has a selection epilogue.  Run on constant and on random data
// (MFMA power depends on the operands).
// 4 waves x 2 workgroups per CU, 6-slot register ring.
// build: hipcc -O3 --offload-arch=gfx950 x256_probe.hip -o x256_probe ; run: ./x256_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const char* __restrict__ g, float* out, int n_stages, size_t bytes_per_wave, const char* __restrict__ qsrc) {
    constexpr int R = 6;
    __shared__ __attribute__((aligned(16))) char qtile[MODE == 2 ? R * 8192 : 8192];
    for (int i = threadIdx.x; i < (MODE == 2 ? R * 2048 : 2048); i += 256) ((u32*)qtile)[i] = 0x3c003c00u + i;     // fp16 ~1.0
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int off = j * 32 + ((h ^ ((j >> 3) & 1)) << 4);
    const char* dnext = g + ((size_t)blockIdx.x * 4 + w) * bytes_per_wave + (MODE ? lane * 16 : off);
    u32x4 dr0[R], dr1[R];
    int s_next = 0;
#define ISSUE(SLOT)                                                                                           \
    do {                                                                                                      \
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dr0[SLOT]) : "v"(dnext) : "memory");        \
        if (!MODE) asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(dr1[SLOT]) : "v"(dnext) : "memory"); \
        dnext += MODE ? 1024 : 2048;                                                                          \
        if (MODE == 2) {                                                                                      \
            const char* qs_ = qsrc + (size_t)s_next * 8192 + w * 2048 + lane * 16;                            \
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(qs_), "s"(__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(qtile + (SLOT) * 8192 + w * 2048))) : "memory"); \
            asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(qs_ + 1024), "s"(__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(qtile + (SLOT) * 8192 + w * 2048 + 1024))) : "memory"); \
            if (++s_next == 48) s_next = 0;                                                                   \
        }                                                                                                     \
    } while (0)
#pragma unroll
    for (int i = 0; i < R - 1; ++i) ISSUE(i);
    f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, b00 = {0}, b01 = {0}, b10 = {0}, b11 = {0};
    const char* curq0 = qtile + off;
    for (int s0 = 0; s0 < n_stages; s0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (MODE == 2) asm volatile("s_waitcnt vmcnt(%1)\n\ts_barrier" : "+v"(dr0[u]) : "n"((R - 2) * 3) : "memory");
            else if (MODE) asm volatile("s_waitcnt vmcnt(%1)\n\ts_barrier" : "+v"(dr0[u]) : "n"((R - 2) * 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(dr0[u]), "+v"(dr1[u]) : "n"((R - 2) * 2) : "memory");
            const char* curq = curq0 + (MODE == 2 ? u * 8192 : 0);
            const f16x8 qa0 = *(const f16x8*)(curq), qa1 = *(const f16x8*)(curq + 1024);
            const f16x8 qb0 = *(const f16x8*)(curq + 2048), qb1 = *(const f16x8*)(curq + 3072);
            const u32x4 x0 = dr0[u], x1 = dr1[u];
            if (MODE == 0) {
                ISSUE((u + R - 1) % R);
                const f16x8 e0 = __builtin_bit_cast(f16x8, x0), e1 = __builtin_bit_cast(f16x8, x1);
                a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa0, a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa1, a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa0, a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qa1, a11, 0, 0, 0);
                b00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb0, b00, 0, 0, 0);
                b01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb1, b01, 0, 0, 0);
                b10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb0, b10, 0, 0, 0);
                b11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, qb1, b11, 0, 0, 0);
            } else {
                const f16x8 qc0 = *(const f16x8*)(curq + 4096), qc1 = *(const f16x8*)(curq + 5120);
                const f16x8 qd0 = *(const f16x8*)(curq + 6144), qd1 = *(const f16x8*)(curq + 7168);
                ISSUE((u + R - 1) % R);
                const f16x8 e0 = __builtin_bit_cast(f16x8, x0);
                a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa0, a00, 0, 0, 0);
                a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qa1, a01, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb0, a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qb1, a11, 0, 0, 0);
                b00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qc0, b00, 0, 0, 0);
                b01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qc1, b01, 0, 0, 0);
                b10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qd0, b10, 0, 0, 0);
                b11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, qd1, b11, 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a00[i] + a01[i] + a10[i] + a11[i] + b00[i] + b01[i] + b10[i] + b11[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void fill_random(u32* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32 x = (u32)i * 2654435761u ^ (u32)(i >> 32);
        x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
        p[i] = (x & 0x83ff83ffu) | 0x3c003c00u;      // sign + mantissa random, exponent of 1.0
    }
}

int main() {
    const size_t rows = 7364780, E = 768;                        // the bench corpus's unique rows
    const size_t bytes = rows * E * 2;
    const int grid = 512;
    const size_t per_wave = bytes / (grid * 4) / (2048 * 6) * (2048 * 6);
    char* g;
    float* out;
    CHK(hipMalloc(&g, per_wave * grid * 4 + (1 << 20)));
    CHK(hipMemset(g, 0x3c, per_wave * grid * 4 + (1 << 20)));
    char* qsrc;
    CHK(hipMalloc(&qsrc, 48 * 8192));
    CHK(hipMemset(qsrc, 0x3c, 48 * 8192));
    CHK(hipMalloc(&out, grid * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int round = 0; round < 2; ++round) {
        if (round == 1) {   // random docs and queries: fp16 values in (-2, 2) with random mantissas
            fill_random<<<4096, 256>>>((u32*)g, (per_wave * grid * 4) / 4);
            fill_random<<<64, 256>>>((u32*)qsrc, 48 * 8192 / 4);
            CHK(hipDeviceSynchronize());
            printf("random data:\n");
        }
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(e0));
                if (mode == 2) probe<2><<<grid, 256>>>(g, out, (int)(per_wave / 1024), per_wave, qsrc);
                else if (mode) probe<1><<<grid, 256>>>(g, out, (int)(per_wave / 1024), per_wave, qsrc);
                else probe<0><<<grid, 256>>>(g, out, (int)(per_wave / 2048), per_wave, qsrc);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("mode %d (%s): %.3f ms for %.2f GB = %.2f TB/s\n", mode, mode == 2 ? "32 rows x 256 queries + query DMA ring" : mode ? "32 rows x 256 queries" : "64 rows x 128 queries", ms,
                                per_wave * grid * 4 / 1e9, per_wave * grid * 4 / 1e9 / ms);
            }
        }
    }
    return 0;
}
