// Probe: do scalar stores (s_store_dwordx2) work on gfx950, and how do they interact with vector stores to the same
// 64-byte lines?  build: hipcc -O3 --offload-arch=gfx950 sstore_probe.hip -o sstore_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

// every wave owns a region of 64 slots (8 B each = 8 lines).  Even slots are written by vector stores (lane = slot),
// odd slots by scalar stores (one at a time, value read with v_readlane).  MODE 1: vector first then scalar; 2: scalar
// first then vector; 3: interleaved rounds with a wb in between.
__global__ void probe(u64* out, int mode) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    u64* base = out + (size_t)wave * 64;
    const u64 val = ((u64)wave << 32) | (u64)(lane * 2654435761u >> 8);
    const unsigned lo = (unsigned)val, hi = (unsigned)(val >> 32);
    auto vec = [&]() { if ((lane & 1) == 0) base[lane] = val; };
    auto sca = [&](int first = 1, int stride = 2) {
        for (int l = first; l < 64; l += stride) {
            const unsigned slo = __builtin_amdgcn_readlane(lo, l), shi = __builtin_amdgcn_readlane(hi, l);
            const u64 sv = ((u64)shi << 32) | slo;
            const u64 a = (u64)(base + l);
            const u64 addr = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
            asm volatile("s_store_dwordx2 %0, %1, 0x0" :: "s"(sv), "s"(addr) : "memory");
        }
    };
    if (mode == 1) { vec(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sca(); }
    else if (mode == 2) { sca(); vec(); }
    else if (mode == 3) { sca(); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); vec(); }
    else {  // scalar (slots 1 mod 4) -> write back -> vector (even slots) -> scalar (slots 3 mod 4) on the SAME lines
        sca(1, 4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        vec();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sca(3, 4);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

int main() {
    const int wgs = 2048, thr = 256, waves = wgs * thr / 64;
    u64* d;
    CHK(hipMalloc(&d, (size_t)waves * 64 * 8));
    std::vector<u64> h((size_t)waves * 64);
    for (int mode = 1; mode <= 4; ++mode) {
        CHK(hipMemset(d, 0xAB, (size_t)waves * 64 * 8));
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        probe<<<wgs, thr>>>(d, mode);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        long bad_v = 0, bad_s = 0;
        for (int w = 0; w < waves; ++w)
            for (int l = 0; l < 64; ++l) {
                const u64 want = ((u64)w << 32) | (u64)((unsigned)(l * 2654435761u) >> 8);
                if (h[(size_t)w * 64 + l] != want) { if (l & 1) ++bad_s; else ++bad_v; }
            }
        printf("mode %d: %.3f ms, wrong vector-written slots %ld, wrong scalar-written slots %ld (of %d each)\n", mode, ms, bad_v, bad_s, waves * 32);
    }
    return 0;
}
