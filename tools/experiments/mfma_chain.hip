// Micro-benchmark (gfx950): issue rate of v_mfma_f32_32x32x2_f32 as a function of the number of INDEPENDENT accumulator chains a wave
// rotates through.  Question behind it: the 128x64 convolution tile gives a wave only TWO accumulator tiles (2x1), the 128x128 tile four
// (2x2); is a two-chain rotation issue-limited by the dependency on the accumulator written two MFMAs earlier?
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_chain.hip -o gpurun_out/mfma_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ void __launch_bounds__(256) chain_kernel(float* out, int iters, float a0, float b0) {
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / NCH; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NCH>
void run(float* d, int nblk, int nthr, const char* what) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain_kernel<NCH>, dim3(nblk), dim3(nthr), 0, 0, d, 64, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<NCH>, dim3(nblk), dim3(nthr), 0, 0, d, iters, 1.0f, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)iters * 16.0 * (nblk * (nthr / 64));      // MFMAs in total
    const double tf = nm * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-28s chains %d: %8.3f ms  %7.1f TFLOP/s  (%.1f ns per MFMA per wave)\n", what, NCH, ms, tf, ms * 1e6 / ((double)iters * 16.0));
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * sizeof(float));
    // one wave per SIMD: 256 CUs x 4 SIMDs = 1024 waves = 256 workgroups of 4 waves; two waves per SIMD: 512 workgroups
    for (int rep = 0; rep < 2; ++rep) {
        run<1>(d, 256, 256, "1 wave/SIMD");
        run<2>(d, 256, 256, "1 wave/SIMD");
        run<4>(d, 256, 256, "1 wave/SIMD");
        run<8>(d, 256, 256, "1 wave/SIMD");
        run<1>(d, 512, 256, "2 waves/SIMD");
        run<2>(d, 512, 256, "2 waves/SIMD");
        run<4>(d, 512, 256, "2 waves/SIMD");
    }
    return 0;
}
