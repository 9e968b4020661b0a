#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* out, int nelem, int shift) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nelem * 4, 0x00020000);
    unsigned vo = (unsigned)((threadIdx.x * 4 + shift) * 4);
    i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0);
    f4 f = __builtin_bit_cast(f4, v);
    for (int q = 0; q < 4; ++q) out[threadIdx.x * 4 + q] = f[q];
}
int main() {
    const int n = 1024; float h[n], o[256]; for (int i = 0; i < n; ++i) h[i] = i;
    float *dx, *dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, 256 * 4); hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int shift : {0, 1, 3, 5, 11, 25, -1, -11, 1000}) {
        hipMemset(dout, 0xff, 256 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n, shift);
        hipMemcpy(o, dout, 256 * 4, hipMemcpyDeviceToHost);
        int e = 0;
        for (int i = 0; i < 256; ++i) { int s = i + shift; float want = (s >= 0 && s < n) ? (float)s : 0.f; if (o[i] != want) { if (e < 3) printf("shift %d i %d got %f want %f\n", shift, i, o[i], want); ++e; } }
        printf("shift %d: %d mismatches\n", shift, e); bad += e;
    }
    printf(bad ? "FAIL\n" : "ALL OK\n");
    return 0;
}
