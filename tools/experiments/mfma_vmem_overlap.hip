// Micro-benchmark (gfx950, round 6): do global loads into VGPRs overlap with bf16 MFMAs issued by the SAME wave (one wave per SIMD)?
// Question behind it: wino_gemm_x3_kernel with register staging ran at T(data only) + T(MFMA only) instead of max(...) — see
// tools/experiments/README.md, round 6.  Each wave streams NL x 1 KiB per iteration (buffer_load_dwordx4, two register sets: the loads of
// iteration i + 2 are issued in iteration i and consumed — summed — at the start of iteration i + 2) and issues NM independent
// v_mfma_f32_32x32x16_bf16 (4 accumulator chains) per iteration.  MODE 1 additionally pushes the loaded data through LDS (ds_write_b128,
// barrier, ds_read_b128) like the GEMM does.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_vmem_overlap.hip -o gpurun_out/mfma_vmem_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NL, int NM, int MODE>
__global__ void __launch_bounds__(256, 1) k(const float* __restrict__ src, size_t span_f4, float* out, int iters) {
    __shared__ f32x4 lds[2][NL > 0 ? NL * 256 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * 4 + wave, nw = (size_t)gridDim.x * 4;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + lane * 1e-3f); b[e] = (__bf16)1.0f; }
    f32x4 st[2][NL > 0 ? NL : 1];
    float sum = 0.f;
    auto issue = [&](int set, int it) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const size_t idx = (((size_t)it * nw + gw) * NL + i) * 64 + lane;
            st[set][i] = s4[idx % span_f4];
        }
    };
    auto body = [&](int set, int it) {
        // consume the set loaded two iterations ago
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i) sum += st[set][i][0];
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) lds[set][(i * 4 + wave) * 64 + lane] = st[set][i];
        }
        issue(set, it + 2);
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        if (MODE == 1) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NL; ++i) sum += lds[set][(i * 4 + ((wave + 1) & 3)) * 64 + lane][0];
        }
    };
    issue(0, 0);
    issue(1, 1);
    for (int it = 0; it < iters; it += 2) {
        body(0, it);
        body(1, it + 1);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// MODE 2: eight waves — waves 0..3 only contract (NM MFMAs per iteration + the LDS reads of the data the loaders staged), waves 4..7 only
// stream (NL x 1 KiB each per iteration -> LDS); one barrier per iteration
// NSET = register staging sets of a loader wave = iterations of lookahead (bytes in flight per CU = NSET x 4 waves x NL KiB)
template <int NL, int NM, int NSET = 2>
__global__ void __launch_bounds__(512, 2) k2(const float* __restrict__ src, size_t span_f4, float* out, int iters) {
    __shared__ f32x4 lds[2][NL * 256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= 4;
    const int w4 = wave & 3;
    const size_t gw = (size_t)blockIdx.x * 4 + w4, nw = (size_t)gridDim.x * 4;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    float sum = 0.f;
    if (loader) {
        f32x4 st[NSET][NL];
        auto issue = [&](int set, int it) {
#pragma unroll
            for (int i = 0; i < NL; ++i) st[set][i] = s4[((((size_t)it * nw + gw) * NL + i) * 64 + lane) % span_f4];
        };
        auto body = [&](int set, int it) {
#pragma unroll
            for (int i = 0; i < NL; ++i) lds[it & 1][(i * 4 + w4) * 64 + lane] = st[set][i];
            issue(set, it + NSET);
            __syncthreads();
        };
#pragma unroll
        for (int s_ = 0; s_ < NSET; ++s_) issue(s_, s_);
        for (int it = 0; it < iters; it += NSET) {
#pragma unroll
            for (int s_ = 0; s_ < NSET; ++s_) body(s_, it + s_);
        }
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + lane * 1e-3f); b[e] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
            const int set = (it + 1) & 1;       // the buffer staged in the previous iteration
#pragma unroll
            for (int i = 0; i < NL; ++i) sum += lds[set][(i * 4 + ((w4 + 1) & 3)) * 64 + lane][0];
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc[c][e];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int NL, int NM, int NSET = 2>
void run2(const float* src, size_t span_bytes, float* d, const char* what) {
    const int iters = 768, nblk = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<NL, NM, NSET>), dim3(nblk), dim3(512), 0, 0, src, span_bytes / 16, d, 48);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k2<NL, NM, NSET>), dim3(nblk), dim3(512), 0, 0, src, span_bytes / 16, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_it = ms * 1e3 / iters;
    const double tbs = (double)nblk * 4 * NL * 1024.0 * iters / (ms * 1e-3) / 1e12;
    const double tf = (double)nblk * 4 * NM * 32768.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-10s span %6.0f MB  loads/it %2d  mfma/it %2d  mode 2, %d sets (%3d KiB in flight per CU): %7.3f us per iteration  %6.2f TB/s  %7.1f TF/s\n", what, span_bytes / 1048576.0, NL, NM, NSET, NSET * 4 * NL, us_it, tbs, tf);
}

template <int NL, int NM, int MODE>
void run(const float* src, size_t span_bytes, float* d, const char* what) {
    const int iters = 512, nblk = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NL, NM, MODE>), dim3(nblk), dim3(256), 0, 0, src, span_bytes / 16, d, 32);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, NM, MODE>), dim3(nblk), dim3(256), 0, 0, src, span_bytes / 16, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us_it = ms * 1e3 / iters;
    const double tbs = (double)nblk * 4 * NL * 1024.0 * iters / (ms * 1e-3) / 1e12;
    const double tf = (double)nblk * 4 * NM * 32768.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-10s span %6.0f MB  loads/it %2d  mfma/it %2d  mode %d: %7.3f us per iteration  %6.2f TB/s  %7.1f TF/s\n", what, span_bytes / 1048576.0, NL, NM, MODE, us_it, tbs,
           tf);
}

int main() {
    const size_t big = (size_t)1 << 30;
    float *src, *d;
    hipMalloc(&src, big);
    hipMemset(src, 0, big);
    hipMalloc(&d, 4096 * 256 * sizeof(float));
    for (size_t span : {(size_t)160 << 20, (size_t)12 << 20}) {
        run<12, 0, 0>(src, span, d, "data");
        run<0, 48, 0>(src, span, d, "mfma");
        run<12, 48, 0>(src, span, d, "both");
        run<12, 96, 0>(src, span, d, "both");
        run<12, 0, 1>(src, span, d, "data+lds");
        run<12, 48, 1>(src, span, d, "both+lds");
        run<6, 48, 1>(src, span, d, "both+lds");
        run<24, 48, 0>(src, span, d, "both");
        run2<12, 0>(src, span, d, "4+4 data");
        run2<12, 48>(src, span, d, "4+4 both");
        run2<12, 96>(src, span, d, "4+4 both");
        run2<6, 48>(src, span, d, "4+4 both");
        run2<12, 48, 3>(src, span, d, "4+4 both");
        run2<12, 48, 4>(src, span, d, "4+4 both");
        run2<12, 0, 3>(src, span, d, "4+4 data");
        run2<12, 0, 4>(src, span, d, "4+4 data");
        run2<6, 48, 4>(src, span, d, "4+4 both");
        run2<6, 48, 8>(src, span, d, "4+4 both");
    }
    return 0;
}
