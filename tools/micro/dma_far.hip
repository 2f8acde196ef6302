// Diagnostic: does an LDS-DMA load (global_load_lds_dwordx4) reach LDS addresses beyond 64 KiB of a block's allocation?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const float* src, float* out, int off_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40960; i += 64) ((float*)smem)[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((gptr_t)(src + 4 * lane), (lptr_t)(smem + off_bytes), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[4 * lane + i] = ((float*)(smem + off_bytes))[4 * lane + i];
    // and where did it land if not there?
    if (lane == 0) { int hit = -1; for (int i = 0; i < 40960; ++i) if (((float*)smem)[i] == 7.0f) { hit = i * 4; break; } out[256] = (float)hit; }
}
int main() {
    float *src, *out; hipMalloc(&src, 1024); hipMalloc(&out, 2048);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = 7.0f + i; hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int offs[] = {0, 32768, 65536 - 1024, 65536, 81920, 131072, 162816};
    for (int o : offs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, src, out, o);
        float r[257]; hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        int ok = 1; for (int i = 0; i < 256; ++i) ok &= (r[i] == h[i]);
        printf("offset %6d: %s (first value found at byte %d)\n", o, ok ? "ok" : "WRONG", (int)r[256]);
    }
    return 0;
}
