// Probe: ds_read_b64_tr_b8 (gfx950).  By analogy with ds_read_b64_tr_b16 (cdna_hip_programming.md T10: per 16 lanes a block of
// 4 rows x 16 columns of 16-bit elements, lane 4q + p supplies row q / columns 4p .. 4p+3, lane i receives column i), the 8-bit
// form should take a block of 8 rows x 16 columns of bytes: lane 2q + p supplies the address of row q, bytes 8p .. 8p+7, and
// lane i receives column i of the 8 rows, row q in byte q.  LDS holds value(row, col) = 16 * row + col for an 8 x 16 block per
// 16-lane group (rows 256 bytes apart); the kernel prints what every lane received.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/trb8 tools/probes/tr_b8_probe.hip && /tmp/trb8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void k(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 8 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4 * 8 * 256; i += 64) {
        const int g = i / 2048, row = (i % 2048) / 256, col = i % 256;
        lds[i] = (unsigned char)(col < 16 ? 16 * row + col + 128 * (g & 1) : 0xEE);
    }
    __syncthreads();
    const int g = lane >> 4, j = lane & 15, q = j >> 1, p = j & 1;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds) + g * 2048 + q * 256 + 8 * p;   // (LDS addresses are 32-bit offsets)
    u32x2 r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[2 * lane] = r.x;
    out[2 * lane + 1] = r.y;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[128];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        unsigned char got[8];
        for (int b = 0; b < 8; ++b) got[b] = (unsigned char)(h[2 * lane + b / 4] >> (8 * (b % 4)));
        int ok = 1;
        for (int b = 0; b < 8; ++b) ok &= (got[b] == (unsigned char)(16 * b + i + 128 * (g & 1)));
        if (!ok) ++bad;
        if (lane < 4 || lane == 17 || !ok)
            printf("lane %2d: %02x %02x %02x %02x %02x %02x %02x %02x %s\n", lane, got[0], got[1], got[2], got[3], got[4], got[5], got[6],
                   got[7], ok ? "= column i, rows 0..7" : "UNEXPECTED");
    }
    printf("lanes that did not receive (column i of rows 0..7): %d of 64\n", bad);
    return 0;
}
