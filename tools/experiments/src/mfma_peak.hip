// What does the fp32 matrix pipe of an MI355X really sustain, and does VALU work of the SIMD's
// other wave overlap with it?   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
//   mode 0: every wave issues MFMAs (4 independent 32x32x2 f32 accumulators)
//   mode 1: waves 0-3 of a 512-thread workgroup issue MFMAs, waves 4-7 idle at the barrier
//   mode 2: waves 0-3 MFMAs, waves 4-7 a dependent v_fma chain (VALU only)
//   mode 3: waves 0-3 idle, waves 4-7 the same v_fma chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float* out, int iters, int valu_iters, int mode, int threads) {
    const int wv = threadIdx.x >> 6;
    const bool mfma_wave = (mode == 0) || (wv < 4 && mode != 3);
    const bool valu_wave = (wv >= 4) && (mode == 2 || mode == 3);
    float r = 0.f;
    if (mfma_wave) {
        f32x16 a0, a1, a2, a3;
        for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; a2[i] = 0.f; a3[i] = 0.f; }
        const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-6f;
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + a2[i] + a3[i];
    }
    if (valu_wave) {
        float a = (float)threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
        for (int it = 0; it < valu_iters; ++it) {
            a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, a);
            c = __builtin_fmaf(c, b, d); b = __builtin_fmaf(b, 0.999f, 1e-7f);
        }
        r += a + b + c + d;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;   // keep the work
}

int main(int argc, char** argv) {
    float* d; hipMalloc(&d, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    struct { int mode, threads, valu; const char* what; } runs[] = {
        {0, 256, 0, "4 waves/CU, all MFMA (1 per SIMD)"},
        {0, 512, 0, "8 waves/CU, all MFMA (2 per SIMD)"},
        {1, 512, 0, "8 waves/CU, waves 0-3 MFMA, 4-7 idle"},
        {3, 512, 40000, "8 waves/CU, waves 4-7 VALU chain only (160k v_fma)"},
        {2, 512, 40000, "8 waves/CU, waves 0-3 MFMA + waves 4-7 VALU chain"},
        {2, 512, 10000, "8 waves/CU, waves 0-3 MFMA + waves 4-7 VALU chain (40k v_fma)"},
    };
    for (auto& r : runs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(r.threads), 0, 0, d, iters, r.valu, r.mode, r.threads);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
                const int mfma_waves = r.mode == 3 ? 0 : (r.mode == 0 ? r.threads / 64 : 4);
                const double flops = 256.0 * 4 * mfma_waves * (double)iters * 4 * (32.0 * 32 * 2 * 2);
                printf("%-70s %8.3f ms  %7.1f TFLOP/s (fp32 MFMA)\n", r.what, ms, flops / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
