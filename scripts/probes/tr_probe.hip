#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[16 * 64];       // [16 rows][64 cols], value = row*64+col
    for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x, t = lane & 15, g = lane >> 4;
    // group g reads the [4 rows][16 cols] block: rows 4*(g>>1).. , cols 16*(g&1)..  ; lane t -> row t>>2, cols 4*(t&3)
    int row = 4 * (g >> 1) + (t >> 2), col = 16 * (g & 1) + 4 * (t & 3);
    if (mode == 1) { row = 4 * (g >> 1) + (t & 3); col = 16 * (g & 1) + 4 * (t >> 2); }   // alternative lane order
    const uint16_t* p = &lds[row * 64 + col];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode); hipDeviceSynchronize();
        uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%2d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
    }
    return 0;
}
