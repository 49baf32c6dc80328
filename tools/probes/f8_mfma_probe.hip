// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 on 0/1 data (fp8 e4m3: 1.0 = 0x38), each lane supplying 32 bytes of its row as
// the two 16-byte chunks {g, 4 + g} (g = lane >> 4) of a 128-byte k row.  k is a dummy index, so any assignment of k to
// (lane group, byte) is right as long as A and B use the same one.  Prints the number of mismatches against a CPU count.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/f8probe tools/probes/f8_mfma_probe.hip && /tmp/f8probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* C, int scale) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const u32x4 a0 = *reinterpret_cast<const u32x4*>(A + r * 128 + 16 * g);
    const u32x4 a1 = *reinterpret_cast<const u32x4*>(A + r * 128 + 16 * (4 + g));
    const u32x4 b0 = *reinterpret_cast<const u32x4*>(B + r * 128 + 16 * g);
    const u32x4 b1 = *reinterpret_cast<const u32x4*>(B + r * 128 + 16 * (4 + g));
    i32x8 a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    i32x8 b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale, 0, scale);
    for (int q = 0; q < 4; ++q) C[(4 * g + q) * 16 + r] = c[q];   // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    srand(7);
    for (int i = 0; i < 16 * 128; ++i) { hA[i] = (rand() % 5 == 0) ? 0x38 : 0; hB[i] = (rand() % 3 == 0) ? 0x38 : 0; }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    const int scales[3] = {0x7F7F7F7F, 0, 0x7F};
    for (int s = 0; s < 3; ++s) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, scales[s]);
        float hC[256];
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        int bad = 0; float c00 = hC[0];
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                int cnt = 0;
                for (int kk = 0; kk < 128; ++kk) cnt += (hA[i * 128 + kk] && hB[j * 128 + kk]) ? 1 : 0;
                if (hC[i * 16 + j] != (float)cnt) ++bad;
            }
        int cnt00 = 0;
        for (int kk = 0; kk < 128; ++kk) cnt00 += (hA[kk] && hB[kk]) ? 1 : 0;
        printf("scale 0x%08x: mismatches %d of 256 (C[0][0] = %g, expected %d)\n", scales[s], bad, c00, cnt00);
    }
    return 0;
}
