#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(const int* in, uint32_t* out) {
    int a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
    unsigned short r = __builtin_amdgcn_ashr_pk_u8_i32(a, b, 17);
    uint32_t raw;
    asm volatile("v_ashr_pk_u8_i32 %0, %1, %2, 17" : "=v"(raw) : "v"(a), "v"(b));
    out[threadIdx.x * 2] = r;
    out[threadIdx.x * 2 + 1] = raw;
}
int main() {
    int h[8] = {118 << 17, 92 << 17, -5 << 17, 300 << 17, 0x7fffffff, (int)0x80000000, 255 << 17, 256 << 17};
    int* d; uint32_t* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 32);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 4>>>(d, o);
    uint32_t r[8]; hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("pair %d: builtin=0x%08x raw=0x%08x\n", i, r[2*i], r[2*i+1]);
    return 0;
}
