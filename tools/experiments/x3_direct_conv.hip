// Experiment (gfx950, round 6): a DIRECT 3x3 stride-1 SAME convolution 64 -> 64 channels on the bf16 matrix pipe with split operands —
// would it beat the Winograd F(4x4) fp32-pipe route on the narrow, large-map layers (cls1 64->64 @ 256^2, B = 16: 0.59 ms forward,
// traffic-bound on the 2.25x transformed tensors)?  Standalone: synthetic data, checked against a naive fp32 kernel, timed with HIP events.
//   * input x [N][H][W][64] fp32: a 16 x 16 output tile's 18 x 18 x 32-channel halo patch is loaded ONCE per channel half by three
//     loader waves (global -> VGPR -> split into three bf16 planes whose sum is the fp32 value -> LDS), double-buffered across halves;
//   * filters pre-split on the host: [half 2][tap 9][plane 3][K 64][32 ch] bf16, 12 KB per (half, tap) stage, streamed by one loader
//     wave with LDS-DMA, double-buffered;
//   * four consumer waves (64 pixels x 64 filters each): per stage 24 ds_read_b128 fragments, 48 v_mfma_f32_32x32x16_bf16 (six plane
//     products per fragment pair); taps are LDS address offsets of the patch; one raw s_barrier per stage for all eight waves;
//   * persistent workgroups (one per CU), the loaders run ahead across tiles.
// build: hipcc --offload-arch=gfx950 -O3 tools/experiments/x3_direct_conv.hip -o gpurun_out/x3_direct_conv ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int C = 64, K = 64, TH = 16, TW = 16, PH = TH + 2, PW = TW + 2, NPX = PH * PW;      // 324 patch pixels
constexpr int ROWB = 64;                           // bytes per LDS row: 32 channels of one plane
constexpr int PLANE_P = NPX * ROWB;                // 20 736 B per patch plane
constexpr int PATCH = 3 * PLANE_P;                 // 62 208 B per patch (one channel half)
constexpr int PLANE_F = K * ROWB;                  // 4 096 B per filter plane
constexpr int FSTG = 3 * PLANE_F;                  // 12 288 B per (half, tap) stage
constexpr int LDS_BYTES = 2 * PATCH + 2 * FSTG;    // 148 992 B
constexpr int NSTG = 18;                           // stages per tile: 2 halves x 9 taps
constexpr int NPL = 192;                           // patch-loader lanes (3 waves)
constexpr int NPI = (NPX * 8 + NPL - 1) / NPL;     // float4 loads per patch-loader lane: 14

__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

struct Args {
    const float* x;               // [N][H][W][64]
    const unsigned short* w3;     // [2][9][3][64][32] bf16
    float* y;                     // [N][H][W][64]
    int N, H, W;
};

__global__ void __launch_bounds__(512, 1) x3_direct_kernel(Args a) {
    __shared__ __attribute__((aligned(256))) unsigned char lds[LDS_BYTES];
    unsigned char* const patch0 = lds;
    unsigned char* const filt0 = lds + 2 * PATCH;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = a.W / TW, tiles_y = a.H / TH, ntiles = a.N * tiles_x * tiles_y;
    int mytiles = 0;
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) ++mytiles;
    const int gstages = mytiles * NSTG;
    auto tile_origin = [&](int it, int& n, int& oh0, int& ow0) {
        const int tl = blockIdx.x + it * gridDim.x;
        n = tl / (tiles_x * tiles_y);
        const int r = tl - n * tiles_x * tiles_y;
        oh0 = (r / tiles_x) * TH;
        ow0 = (r % tiles_x) * TW;
    };

    if (wave == 4) {
        // ============================ filter loader: one 12 KB stage per barrier, LDS-DMA ============================
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w3, 2u * 9u * FSTG);
        // piece i (1 KiB = 16 rows of 64 B): lane (lrow = lane / 4, lchk = lane % 4) fetches chunk lchk ^ swz(row) of its row
        const int lrow = lane >> 2, lchk = lane & 3;
        unsigned vo[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int row = (i & 3) * 16 + lrow;                      // row within the plane (K index); plane = i / 4
            vo[i] = (unsigned)((i >> 2) * PLANE_F + row * ROWB + ((lchk ^ swz(row)) << 4));
        }
        auto issue = [&](int g) {
            const int s = g % NSTG;                                   // (half, tap) = the stage's slot of the filter image
            unsigned char* bp = filt0 + (g & 1) * FSTG;
#pragma unroll
            for (int i = 0; i < 12; ++i) dma16(rw, (lds_void*)(bp + i * 1024), vo[i], s * FSTG);
        };
        issue(0);
        for (int g = 0; g < gstages; ++g) {
            wait_vm0();
            __builtin_amdgcn_s_barrier();
            if (g + 1 < gstages) issue(g + 1);
        }
        wait_vm0();
        return;
    }
    if (wave > 4) {
        // ============================ patch loaders: global fp32 -> three bf16 planes -> LDS ============================
        const int pl = (wave - 5) * 64 + lane;                        // 0..191
        f32x4 st[NPI];
        auto load = [&](int slot) {                                   // slot = tile_iter * 2 + half
            int n, oh0, ow0;
            tile_origin(slot >> 1, n, oh0, ow0);
            const int hsel = slot & 1;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int e = pl + i * NPL;
                const int q = e >> 3, j = e & 7;
                const int pr = q / PW, pc = q - pr * PW;
                const int ih = oh0 - 1 + pr, iw = ow0 - 1 + pc;
                const bool ok = (e < NPX * 8) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)n * a.H + ih) * a.W + iw) * C + hsel * 32 + j * 4);
                st[i] = v;
            }
        };
        auto store = [&](int slot) {
            unsigned char* pb = patch0 + (slot & 1) * PATCH;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int e = pl + i * NPL;
                if (e >= NPX * 8) continue;
                const int q = e >> 3, j = e & 7;
                bf16x4 hi, mi, lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = st[i][c];
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    const float r2 = r1 - (float)m;
                    hi[c] = h; mi[c] = m; lo[c] = (__bf16)r2;
                }
                const int off = q * ROWB + (((j >> 1) ^ swz(q)) << 4) + (j & 1) * 8;
                *reinterpret_cast<bf16x4*>(pb + off) = hi;
                *reinterpret_cast<bf16x4*>(pb + PLANE_P + off) = mi;
                *reinterpret_cast<bf16x4*>(pb + 2 * PLANE_P + off) = lo;
            }
        };
        const int nslots = mytiles * 2;
        load(0);
        wait_vm0();
        store(0);
        for (int g = 0; g < gstages; ++g) {
            const int slot = g / 9, j = g - slot * 9;
            if (j == 0) wait_lgkm0();                                  // this slot's patch is in LDS
            __builtin_amdgcn_s_barrier();
            if (j == 0 && slot + 1 < nslots) load(slot + 1);
            if (j == 4 && slot + 1 < nslots) {
                wait_vm0();
                store(slot + 1);
            }
        }
        return;
    }
    // ============================ consumers: 64 pixels (4 output rows x 16) x 64 filters per wave ============================
    const int l31 = lane & 31, h = lane >> 5;
    int aoff[2];                                    // patch byte offset of this lane's pixel for tm = 0, 1 at tap (0, 0), chunk bits left out
    int aq[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        aq[tm] = (4 * wave + 2 * tm + (l31 >> 4)) * PW + (l31 & 15);
        aoff[tm] = aq[tm] * ROWB;
    }
    int boff[2][2];                                 // [tn][ks]
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) boff[tn][ks] = (tn * 32 + l31) * ROWB + (((2 * ks + h) ^ swz(tn * 32 + l31)) << 4);
    constexpr int kTermA[6] = {2, 1, 0, 1, 0, 0}, kTermB[6] = {0, 1, 2, 0, 1, 0};
    f32x16 acc[2][2];
    int g = 0;
    for (int it = 0; it < mytiles; ++it) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
        for (int s = 0; s < NSTG; ++s, ++g) {
            const int slot = g / 9, tap = g - slot * 9;
            const int tr = tap / 3, ts = tap - tr * 3;
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned char* A = patch0 + (slot & 1) * PATCH;
            const unsigned char* B = filt0 + (g & 1) * FSTG;
            const int dq = tr * PW + ts;
            bf16x8 af[2][3][2], bfr[2][3][2];       // [ks][plane][tm / tn]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const int q = aq[tm] + dq;
                        af[ks][p][tm] = *reinterpret_cast<const bf16x8*>(A + p * PLANE_P + aoff[tm] + dq * ROWB + (((2 * ks + h) ^ swz(q)) << 4));
                    }
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn) bfr[ks][p][tn] = *reinterpret_cast<const bf16x8*>(B + p * PLANE_F + boff[tn][ks]);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int trm = 0; trm < 6; ++trm)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][kTermA[trm]][tm], bfr[ks][kTermB[trm]][tn], acc[tm][tn], 0, 0, 0);
        }
        // epilogue: D[m][n], lane: n = l31 (filter), rows m = 8 (i / 4) + 4 h + (i % 4) (pixel of the 32-pixel block)
        int n, oh0, ow0;
        tile_origin(it, n, oh0, ow0);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = 8 * (i >> 2) + 4 * h + (i & 3);
                    const int oh = oh0 + 4 * wave + 2 * tm + (m >> 4), ow = ow0 + (m & 15);
                    a.y[(((size_t)n * a.H + oh) * a.W + ow) * K + tn * 32 + l31] = acc[tm][tn][i];
                }
    }
}

__global__ void ref_kernel(const float* x, const float* w, float* y, int N, int H, int W) {      // w [3][3][64][64] (HWIO)
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * H * W * K) return;
    const int k = idx % K;
    size_t p = idx / K;
    const int ow = p % W; p /= W;
    const int oh = p % H;
    const int n = p / H;
    double s = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            const int ih = oh - 1 + r, iw = ow - 1 + c;
            if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
            const float* xp = x + (((size_t)n * H + ih) * W + iw) * C;
            const float* wp = w + ((size_t)(r * 3 + c) * C) * K + k;
            for (int ch = 0; ch < C; ++ch) s += (double)xp[ch] * (double)wp[(size_t)ch * K];
        }
    y[idx] = (float)s;
}

static unsigned short bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    const int H = 256, W = 256;
    for (int N : {2, 16}) {
        const size_t nx = (size_t)N * H * W * C;
        std::vector<float> hx(nx), hw(9 * C * K);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd() * 0.06f;
        std::vector<unsigned short> hw3((size_t)2 * 9 * 3 * K * 32);
        for (int hh = 0; hh < 2; ++hh)
            for (int tap = 0; tap < 9; ++tap)
                for (int k = 0; k < K; ++k)
                    for (int c = 0; c < 32; ++c) {
                        const float v = hw[((size_t)tap * C + hh * 32 + c) * K + k];
                        const unsigned short b0 = bf16_rne(v);
                        const float r1 = v - bf16_f(b0);
                        const unsigned short b1 = bf16_rne(r1);
                        const float r2 = r1 - bf16_f(b1);
                        const unsigned short b2 = bf16_rne(r2);
                        const unsigned short pl[3] = {b0, b1, b2};
                        for (int p = 0; p < 3; ++p) hw3[((((size_t)hh * 9 + tap) * 3 + p) * K + k) * 32 + c] = pl[p];
                    }
        float *dx, *dw, *dy, *dr;
        unsigned short* dw3;
        hipMalloc(&dx, nx * 4); hipMalloc(&dy, nx * 4); hipMalloc(&dr, nx * 4);
        hipMalloc(&dw, hw.size() * 4); hipMalloc(&dw3, hw3.size() * 2);
        hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice);
        hipMemset(dy, 0, nx * 4);
        Args a{dx, dw3, dy, N, H, W};
        hipLaunchKernelGGL(x3_direct_kernel, dim3(256), dim3(512), 0, 0, a);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
        if (N == 2) {
            hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, dx, dw, dr, N, H, W);
            hipDeviceSynchronize();
            std::vector<float> y(nx), r(nx);
            hipMemcpy(y.data(), dy, nx * 4, hipMemcpyDeviceToHost);
            hipMemcpy(r.data(), dr, nx * 4, hipMemcpyDeviceToHost);
            double emax = 0, rmax = 0;
            for (size_t i = 0; i < nx; ++i) { emax = fmax(emax, fabs((double)y[i] - r[i])); rmax = fmax(rmax, fabs((double)r[i])); }
            printf("N=%d parity vs float64-accumulated reference: max|err| / max|ref| = %.3e\n", N, emax / rmax);
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 10;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(x3_direct_kernel, dim3(256), dim3(512), 0, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        const double gf = 2.0 * N * H * W * 9.0 * C * K / 1e9;
        printf("N=%d 64->64 3x3 @%dx%d: %.3f ms  %.1f TF/s fp32-equivalent (%.0f TF/s executed bf16)  in+out %.2f TB/s\n", N, H, W, ms, gf / ms, 6 * gf / ms,
               2.0 * nx * 4 / ms / 1e9);
        hipFree(dx); hipFree(dy); hipFree(dr); hipFree(dw); hipFree(dw3);
    }
    return 0;
}
