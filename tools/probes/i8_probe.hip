// Micro-probe for DESIGN.md section 9.1 (int8 first level under the fp16 screen): does the wide scan's k-loop keep up with
// HBM when the docs arrive as biased uint8 (1 byte per element) and are turned into exact fp16 operands in registers?
//   MODE 0: the fp16 loop of mfar_stage1_f16w_kernel (per k-step: 2 x 16 B of fp16 docs per lane, 8 MFMAs)
//   MODE 1: int8 docs (per PAIR of k-steps: 2 x 16 B per lane; v_perm_b32 + v_pk_add_f16 conversion: the fp16 pattern of
//           1024 + u is 0x6400 | u, minus 1152 gives the signed value exactly; 16 MFMAs)
// Both read their query fragments from a fixed LDS tile (the query ring's DMA is left out: it is L2 traffic) and have no
// selection epilogue: this isolates doc stream + conversion + MFMA issue.  4 waves x 2 workgroups per CU, 6-slot register ring.
// build: hipcc -O3 --offload-arch=gfx950 i8_probe.hip -o i8_probe ; run: ./i8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ u32 cvt2(u32 d, u32 sel) {          // two bytes of d -> two exact fp16 (u - 128)
    const u32 p = __builtin_amdgcn_perm(0x64646464u, d, sel);
    f16x2 h = __builtin_bit_cast(f16x2, p) - f16x2{(_Float16)1152.0f, (_Float16)1152.0f};
    return __builtin_bit_cast(u32, h);
}
__device__ __forceinline__ f16x8 cvt8(u32 lo, u32 hi) {        // 8 bytes -> 8 fp16
    u32x4 o = {cvt2(lo, 0x04010400u), cvt2(lo, 0x04030402u), cvt2(hi, 0x04010400u), cvt2(hi, 0x04030402u)};
    return __builtin_bit_cast(f16x8, o);
}

template <int MODE>
__global__ void __launch_bounds__(256, 2) probe(const char* __restrict__ g, float* out, int n_stages, size_t bytes_per_wave) {
    constexpr int R = 6;
    __shared__ __attribute__((aligned(16))) char qtile[8192];
    for (int i = threadIdx.x; i < 2048; i += 256) ((u32*)qtile)[i] = 0x3c003c00u + i;     // fp16 ~1.0
    __syncthreads();
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int off = MODE ? j * 32 + 16 * h : j * 32 + ((h ^ ((j >> 3) & 1)) << 4);
    const char* dnext = g + ((size_t)blockIdx.x * 4 + w) * bytes_per_wave + off;
    u32x4 dr0[R], dr1[R];
#define ISSUE(SLOT)                                                                                           \
    do {                                                                                                      \
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dr0[SLOT]) : "v"(dnext) : "memory");        \
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024 nt" : "=&v"(dr1[SLOT]) : "v"(dnext) : "memory"); \
        dnext += 2048;                                                                                        \
    } while (0)
#pragma unroll
    for (int i = 0; i < R - 1; ++i) ISSUE(i);
    f32x16 a00 = {0}, a01 = {0}, a10 = {0}, a11 = {0}, b00 = {0}, b01 = {0}, b10 = {0}, b11 = {0};
    const char* curq = qtile + j * 32 + (h << 4);
    for (int s0 = 0; s0 < n_stages; s0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(dr0[u]), "+v"(dr1[u]) : "n"((R - 2) * 2) : "memory");
            const f16x8 qa0 = *(const f16x8*)(curq), qa1 = *(const f16x8*)(curq + 1024);
            const f16x8 qb0 = *(const f16x8*)(curq + 2048), qb1 = *(const f16x8*)(curq + 3072);
            const u32x4 x0 = dr0[u], x1 = dr1[u];
            ISSUE((u + R - 1) % R);
#define MF(E0, E1, QA0, QA1, QB0, QB1)                                       \
    a00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E0, QA0, a00, 0, 0, 0);     \
    a01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E0, QA1, a01, 0, 0, 0);     \
    a10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E1, QA0, a10, 0, 0, 0);     \
    a11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E1, QA1, a11, 0, 0, 0);     \
    b00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E0, QB0, b00, 0, 0, 0);     \
    b01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E0, QB1, b01, 0, 0, 0);     \
    b10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E1, QB0, b10, 0, 0, 0);     \
    b11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(E1, QB1, b11, 0, 0, 0);
            if (MODE == 0) {
                const f16x8 e0 = __builtin_bit_cast(f16x8, x0), e1 = __builtin_bit_cast(f16x8, x1);
                MF(e0, e1, qa0, qa1, qb0, qb1)
            } else {
                const f16x8 qc0 = *(const f16x8*)(curq + 4096), qc1 = *(const f16x8*)(curq + 5120);
                const f16x8 qd0 = *(const f16x8*)(curq + 6144), qd1 = *(const f16x8*)(curq + 7168);
                const f16x8 e0 = cvt8(x0[0], x0[1]), e1 = cvt8(x1[0], x1[1]);        // k-step A of the pair
                MF(e0, e1, qa0, qa1, qb0, qb1)
                const f16x8 g0 = cvt8(x0[2], x0[3]), g1 = cvt8(x1[2], x1[3]);        // k-step B
                MF(g0, g1, qc0, qc1, qd0, qd1)
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a00[i] + a01[i] + a10[i] + a11[i] + b00[i] + b01[i] + b10[i] + b11[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const size_t rows = 7364780, E = 768;                        // the bench corpus's unique rows
    for (int mode = 0; mode < 2; ++mode) {
        const size_t bytes = rows * E * (mode ? 1 : 2);
        const int grid = 512;
        size_t per_wave = bytes / (grid * 4) / (2048 * 6) * (2048 * 6);
        const int n_stages = (int)(per_wave / 2048);
        char* g;
        float* out;
        CHK(hipMalloc(&g, per_wave * grid * 4 + (1 << 20)));
        CHK(hipMemset(g, mode ? 0x81 : 0x3c, per_wave * grid * 4 + (1 << 20)));
        CHK(hipMalloc(&out, grid * 256 * 4));
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 4; ++rep) {
            CHK(hipEventRecord(e0));
            if (mode) probe<1><<<grid, 256>>>(g, out, n_stages, per_wave);
            else probe<0><<<grid, 256>>>(g, out, n_stages, per_wave);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("mode %d (%s): %.3f ms for %.2f GB = %.2f TB/s; MFMAs per wave-stage %d\n", mode, mode ? "int8 docs" : "fp16 docs", ms,
                            per_wave * grid * 4 / 1e9, per_wave * grid * 4 / 1e9 / ms, mode ? 16 : 8);
        }
        CHK(hipFree(g));
        CHK(hipFree(out));
    }
    return 0;
}
