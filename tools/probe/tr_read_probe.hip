// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit elements does each lane receive?
// LDS holds a [64 rows][64 cols] int16 image with value = row * 64 + col.  Every 16-lane group g reads the 4 x 16 block
// at rows 0..3, columns 16 g .. 16 g + 15: lane 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3.
// hipcc --offload-arch=gfx950 tools/probe/tr_read_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(v4s* y) {
    __shared__ short sm[64 * 64];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    typedef __attribute__((address_space(3))) v4s lv4s;
    lv4s* ptr = (lv4s*)((__attribute__((address_space(3))) char*)sm + q * 128 + g * 32 + p * 8);
    y[lane] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
}
int main() {
    v4s* d; hipMalloc(&d, 64 * sizeof(v4s));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    v4s h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        for (int e = 0; e < 4; ++e) if (h[l][e] != e * 64 + 16 * g + i) ++bad;
        if (l < 20 || l % 16 == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l][0], h[l][1], h[l][2], h[l][3]);
    }
    printf("expected lane i of group g -> (row e, col 16 g + i): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
