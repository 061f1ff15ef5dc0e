// What ANY HIP program pays before its first result on this box: the floor under tools/first_call.py's numbers.
//   hipcc --offload-arch=gfx950 -O2 -o tools/hip_floor tools/hip_floor.hip && tools/hip_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_touch(int *p) { p[threadIdx.x] = threadIdx.x; }
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double t0 = now_ms();
    int n = 0;
    (void)hipGetDeviceCount(&n);
    (void)hipSetDevice(0);
    double t1 = now_ms();
    hipStream_t s[3];
    for (auto &x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    double t2 = now_ms();
    int *d = nullptr;
    (void)hipMalloc(&d, 1 << 20);
    double t3 = now_ms();
    void *h = nullptr;
    (void)hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
    double t4 = now_ms();
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s[0], d);
    (void)hipStreamSynchronize(s[0]);
    double t5 = now_ms();
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s[1], d);
    (void)hipStreamSynchronize(s[1]);
    double t6 = now_ms();
    hipLaunchKernelGGL(k_touch, dim3(1), dim3(64), 0, s[0], d);
    (void)hipStreamSynchronize(s[0]);
    double t7 = now_ms();
    (void)hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, s[0]);
    (void)hipStreamSynchronize(s[0]);
    double t8 = now_ms();
    printf("{\"devices\": %d, \"runtime_init_ms\": %.3f, \"three_streams_ms\": %.3f, \"first_hipMalloc_ms\": %.3f, \"first_hipHostMalloc_ms\": %.3f, "
           "\"first_launch_ms\": %.3f, \"first_launch_second_stream_ms\": %.3f, \"warm_launch_ms\": %.3f, \"first_h2d_copy_ms\": %.3f}\n",
           n, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7);
    return 0;
}
