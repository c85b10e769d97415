// Micro-probe: what limits v_mfma_f32_32x32x2_f32 issue in a stage-1-like loop on MI355X?
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: pure MFMA, fixed operands.  1: + 8 ds_read_b128 per 32 MFMA.  2: + 5 LDS-DMA loads per step (streaming).
// 3: like 2 with a raw barrier per step
template <int MODE, int AUX>
__global__ void __launch_bounds__(256, 2) probe(const float* __restrict__ g, float* out, int iters, size_t stride_per_wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* buf = smem + w * 12288;
    for (int i = threadIdx.x; i < 12288; i += 256) ((float*)smem)[i] = g[i + blockIdx.x * 12288];
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f32x4 b0[16];
    for (int b = 0; b < 16; ++b) b0[b] = f32x4{0, 0, 0, 0};
    f32x4 d00 = {1, 2, 3, 4}, d01 = d00, d10 = d00, d11 = d00, q00 = d00, q01 = d00, q10 = d00, q11 = d00;
    const char* src = (const char*)g + ((size_t)blockIdx.x * 4 + w) * stride_per_wave + lane * 16;
    const int off = (lane & 31) * 64 + ((lane >> 5) << 4);
    int st = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
            if (MODE == 3 || MODE == 4 || MODE == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
        if (MODE >= 1) {
            const char* c = buf + st * 4096;
            d00 = *(const f32x4*)(c + off); d01 = *(const f32x4*)(c + off + 32);
            d10 = *(const f32x4*)(c + 2048 + off); d11 = *(const f32x4*)(c + 2048 + off + 32);
            q00 = *(const f32x4*)(c + off + 16); q01 = *(const f32x4*)(c + off + 48);
            q10 = *(const f32x4*)(c + 2048 + off + 16); q11 = *(const f32x4*)(c + 2048 + off + 48);
        }
        if (MODE >= 2) {
            char* dst = buf + ((st + 2) % 3) * 4096;
#pragma unroll
            for (int p = 0; p < 5; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (p & 3) * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + (p & 3) * 1024), 16, 0, AUX);
            src += 4096;
        }
        st = st == 2 ? 0 : st + 1;
        for (int rep = 0; rep < (MODE == 5 ? 2 : 1); ++rep)
        if (MODE != 4) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d00[x], q00[x], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d00[x], q10[x], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d10[x], q00[x], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(d10[x], q10[x], a3, 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d01[x], q01[x], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d01[x], q11[x], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d11[x], q01[x], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(d11[x], q11[x], a3, 0, 0, 0);
        }
        } else {
            // same flops with v_mfma_f32_16x16x4_f32: 16 blocks of 16x16 (4 acc regs each), K = 4 per instruction
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                f32x4 c = {b0[b][0], b0[b][1], b0[b][2], b0[b][3]};
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(((b & 1) ? d01 : d00)[x] + ((b & 2) ? d10[x] : d11[x]), ((b & 4) ? q01 : q00)[x] + ((b & 8) ? q10[x] : q11[x]), c, 0, 0, 0);
                b0[b] = c;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i] + b0[i][0] + b0[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int AUX = 0>
int run(const char* name, const float* g, float* out, int wgs, int iters, size_t stride) {
    CHK(hipFuncSetAttribute((const void*)probe<MODE, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0));
        probe<MODE, AUX><<<wgs, 256, 49152>>>(g, out, iters, stride);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double mf = (double)wgs * 4 * iters * 32 * (MODE == 5 ? 2 : 1);
        if (rep == 2) printf("%-38s wgs=%4d iters=%6d  %8.3f ms  %7.1f TFLOP/s  %.1f ns/MFMA/SIMD-slot  stream %.2f TB/s\n", name, wgs, iters, ms,
               mf * 4096 / ms / 1e9, ms * 1e6 / (mf / 1024), MODE >= 2 ? (double)wgs * 4 * iters * 4096 / ms / 1e9 : 0.0);
    }
    return 0;
}

__global__ void fill_random(float* g, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27;
        g[i] = ((int)(z & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000);
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)24 << 30;
    float *g, *out;
    CHK(hipMalloc(&g, bytes)); if (argc > 1) { fill_random<<<4096, 256>>>(g, bytes / 4); CHK(hipDeviceSynchronize()); printf("random data\n"); } else { CHK(hipMemset(g, 0x3c, bytes)); printf("constant data\n"); } CHK(hipMalloc(&out, 4 << 20));
    const int iters = 3000;
    for (int wgs : {256, 512}) {
        const size_t stride = bytes / ((size_t)wgs * 4);
        run<0>("pure MFMA", g, out, wgs, iters, stride);
        run<1>("MFMA + 8 ds_read_b128/step", g, out, wgs, iters, stride);
        run<2>("MFMA + ds_read + 5 LDS-DMA/step", g, out, wgs, iters, stride);
        run<3>("MFMA + ds_read + LDS-DMA + barrier", g, out, wgs, iters, stride);
        run<3, 2>("  ... with nt (aux=2) loads", g, out, wgs, iters, stride);
        run<4>("16x16x4 MFMA + ds_read + DMA + barrier", g, out, wgs, iters, stride);
        run<5, 2>("64 MFMA per 5 DMA (128-query pass), nt", g, out, wgs, iters / 2, stride);
    }
    return 0;
}
