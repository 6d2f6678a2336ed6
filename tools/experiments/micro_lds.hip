// micro_lds.hip — two hardware facts the bf16-resident kernels (csrc/conv_bf16r.hip) stand on, printed by the GPU itself:
//   1. LDS-DMA (buffer_load_dwordx4 ... lds) of a lane whose voffset is out of range: does it write ZEROS to LDS (TF zero padding for
//      free) or leave the previous LDS bytes?
//   2. ds_read_b64_tr_b16: which (lane, element) of a 16-lane group ends up where.
// build: hipcc --offload-arch=gfx950 -O2 tools/experiments/micro_lds.hip -o tools/experiments/bin/micro_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__global__ void dma_oob(const float* g, unsigned bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -777.f;      // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, bytes, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = 0x80000000u;                          // odd lanes: out of range
    if ((threadIdx.x & 7) == 2) voff = bytes;                         // exactly one past the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

__global__ void tr16(short* out, int variant) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = (short)i;
    __syncthreads();
    // lane l reads the 4 contiguous shorts at element offset 4*l (variant 0) — ids 4l..4l+3
    const int l = threadIdx.x;
    int off = 4 * l;
    if (variant == 1) off = 4 * (((l & 15) >> 2) * 16 + (l >> 4) * 4 + (l & 3));   // lane (key = (l&15)>>2, q = l&3) of group l>>4: row key of a [4][64] matrix
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

int main() {
    const int n = 256;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
    float *g, *o;
    hipMalloc(&g, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dma_oob, dim3(1), dim3(64), 0, 0, g, (unsigned)(n * 4), o);
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA: lane -> 4 floats landed in LDS (sentinel -777; source 1000+i; odd lanes out of range, lanes 2,10,.. at offset == size)\n");
    int zeros = 0, sent = 0, good = 0;
    for (int l = 0; l < 64; ++l) {
        if (l < 12) printf("  lane %2d: %8.1f %8.1f %8.1f %8.1f\n", l, r[4 * l], r[4 * l + 1], r[4 * l + 2], r[4 * l + 3]);
        const bool oob = (l & 1) || ((l & 7) == 2);
        for (int e = 0; e < 4; ++e) {
            const float v = r[4 * l + e];
            if (oob) { zeros += v == 0.f; sent += v == -777.f; } else good += v == 1000.f + 4 * l + e;
        }
    }
    printf("LDS-DMA RESULT: in-range ok %d/%d ; out-of-range lanes: zeros %d sentinel-left %d of %d  => %s\n", good, 4 * 28, zeros, sent, 4 * 36,
           zeros == 4 * 36 ? "OOB_WRITES_ZERO" : (sent == 4 * 36 ? "OOB_LEAVES_LDS" : "MIXED"));
    short* so;
    hipMalloc(&so, 64 * 4 * 2);
    for (int variant = 0; variant < 2; ++variant) {
        hipLaunchKernelGGL(tr16, dim3(1), dim3(64), 0, 0, so, variant);
        std::vector<short> s(256);
        hipMemcpy(s.data(), so, 512, hipMemcpyDeviceToHost);
        printf("ds_read_b64_tr_b16 variant %d: lane -> 4 source element ids (lane l supplied ids at its own address, see source)\n", variant);
        for (int l = 0; l < 64; ++l) {
            if (l < 20 || l == 32 || l == 33 || l == 48) printf("  lane %2d: %4d %4d %4d %4d\n", l, s[4 * l], s[4 * l + 1], s[4 * l + 2], s[4 * l + 3]);
        }
        if (variant == 0) {
            // hypothesis H1: within a 16-lane group, out lane i element e = in lane (4e + i/4) element (i%4)  [rows = lane quads]
            int h1 = 0, h2 = 0;
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int g16 = l & ~15, i = l & 15;
                    h1 += s[4 * l + e] == 4 * (g16 + 4 * e + i / 4) + (i % 4);
                    h2 += s[4 * l + e] == 4 * (g16 + e + 4 * (i / 4)) + (i % 4);
                }
            printf("tr16 RESULT: H1 (out[i][e] = in[lane 4e + i/4][i%%4]) matches %d/256 ; H2 (in[lane e + 4(i/4)][i%%4]) matches %d/256\n", h1, h2);
        }
    }
    return 0;
}
