#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint8_t* in, uint8_t* out, uint32_t* res, uint32_t sh)
{
    __shared__ uint8_t ring[4096];
    uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 4096; i += 64) ring[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t a; __builtin_memcpy(&a, in + lane * 5 + sh, 4);     // unaligned global load
    uint32_t b; __builtin_memcpy(&b, ring + lane * 5 + sh, 4);   // unaligned LDS load
    __builtin_memcpy(out + lane * 7 + sh, &a, 4);                // unaligned global store
    __syncthreads();
    __builtin_memcpy(ring + 2048 + lane * 5 + sh, &a, 4);        // unaligned LDS store
    __syncthreads();
    uint32_t c = 0;
    for (int q = 0; q < 4; ++q) c |= (uint32_t)ring[2048 + lane * 5 + sh + q] << (8 * q);
    res[lane * 3] = a; res[lane * 3 + 1] = b; res[lane * 3 + 2] = c;
}
int main()
{
    uint8_t *in, *out; uint32_t* res;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&res, 64 * 12);
    uint8_t h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (uint8_t)(i * 13 + 1);
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    int bad = 0;
    for (uint32_t sh = 0; sh < 4; ++sh) {
        hipMemset(out, 0, 4096);
        k<<<1, 64>>>(in, out, res, sh);
        uint32_t r[192]; uint8_t o[4096];
        hipMemcpy(r, res, sizeof r, hipMemcpyDeviceToHost); hipMemcpy(o, out, 4096, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) {
            uint32_t ea = 0, eb = 0;
            for (int q = 0; q < 4; ++q) { ea |= (uint32_t)h[l * 5 + sh + q] << (8 * q); eb |= (uint32_t)(uint8_t)((l * 5 + sh + q) * 7 + 3) << (8 * q); }
            if (r[l * 3] != ea || r[l * 3 + 1] != eb || r[l * 3 + 2] != ea) ++bad;
        }
        // global store check for last lane only (others overlap by design: 7 stride >4 so no overlap)
        for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) if (o[l * 7 + sh + q] != h[l * 5 + sh + q]) ++bad;
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
