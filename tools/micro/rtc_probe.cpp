// probe: does hipRTC (dlopen'ed) compile and run a gfx950 kernel on this box?
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string>
#include <vector>
int main() {
    void* h = dlopen("libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("no hiprtc: %s\n", dlerror()); return 1; }
#define SYM(n) auto p_##n = (decltype(&n))dlsym(h, #n); if (!p_##n) { printf("missing %s\n", #n); return 1; }
    SYM(hiprtcCreateProgram) SYM(hiprtcCompileProgram) SYM(hiprtcGetCodeSize) SYM(hiprtcGetCode)
    SYM(hiprtcGetProgramLogSize) SYM(hiprtcGetProgramLog) SYM(hiprtcDestroyProgram)
    const char* src = R"(
#include "mymath.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ void k(float* out, float a) {
    f32x4 acc = {0,0,0,0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 2.0f, acc, 0, 0, 0);
    out[threadIdx.x] = twice(acc[0]) + acosf(0.5f) + (float)fma(1.0, 2.0, 3.0);
})";
    const char* hdr = "__device__ inline float twice(float x){return 2*x;}\n";
    const char* hn = "mymath.h";
    hiprtcProgram prog;
    if (p_hiprtcCreateProgram(&prog, src, "k.hip", 1, &hdr, &hn) != HIPRTC_SUCCESS) { printf("create failed\n"); return 1; }
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult r = p_hiprtcCompileProgram(prog, 3, opts);
    size_t ls = 0; p_hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { std::string log(ls, 0); p_hiprtcGetProgramLog(prog, &log[0]); printf("log: %s\n", log.c_str()); }
    if (r != HIPRTC_SUCCESS) { printf("compile failed %d\n", (int)r); return 1; }
    size_t cs = 0; p_hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs); p_hiprtcGetCode(prog, code.data());
    hipModule_t mod; hipFunction_t fn;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { printf("module load failed\n"); return 1; }
    if (hipModuleGetFunction(&fn, mod, "k") != hipSuccess) { printf("get function failed\n"); return 1; }
    float* d; hipMalloc(&d, 64 * 4);
    struct { float* out; float a; } args = {d, 3.0f};
    size_t asz = sizeof(args);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
    if (hipModuleLaunchKernel(fn, 1, 1, 1, 64, 1, 1, 0, 0, nullptr, cfg) != hipSuccess) { printf("launch failed\n"); return 1; }
    float hout[64]; hipMemcpy(hout, d, sizeof(hout), hipMemcpyDeviceToHost);
    printf("rtc ok: code %zu bytes, out[0]=%f (expect %f)\n", cs, hout[0], 2 * 24.0f + 1.0471976f + 5.0f);
    return 0;
}
