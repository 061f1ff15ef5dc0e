// Host->device strategies for the drop-in entry (96 MiB of pageable caller memory at 2^20): what does each cost?
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/h2d_probe tools/h2d_probe.hip -lpthread && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const size_t bytes = (size_t)96 << 20;
    char *src = (char *)aligned_alloc(4096, bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;  // touch
    void *dst;
    CK(hipMalloc(&dst, bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        printf("pageable hipMemcpy            %.3f ms  (%.1f GB/s)\n", now() - t0, bytes / (now() - t0) / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        CK(hipHostRegister(src, bytes, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        double t2 = now();
        CK(hipHostUnregister(src));
        double t3 = now();
        printf("register %.3f + copy %.3f + unregister %.3f = %.3f ms\n", t1 - t0, t2 - t1, t3 - t2, t3 - t0);
    }
    // staged: T threads memcpy chunks into pinned staging buffers, DMA follows chunk by chunk
    for (int T : {1, 2, 4, 8, 16}) {
        const size_t chunk = (size_t)4 << 20;
        const int nbuf = 8;
        char *stage;
        CK(hipHostMalloc((void **)&stage, chunk * nbuf, hipHostMallocDefault));
        hipEvent_t done[nbuf];
        for (auto &e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            const size_t nchunks = (bytes + chunk - 1) / chunk;
            for (size_t k = 0; k < nchunks; ++k) {
                const int b = (int)(k % nbuf);
                if (k >= (size_t)nbuf) CK(hipEventSynchronize(done[b]));
                const size_t off = k * chunk, len = std::min(chunk, bytes - off);
                if (T == 1) memcpy(stage + b * chunk, src + off, len);
                else {
                    std::vector<std::thread> th;
                    const size_t per = (len + T - 1) / T;
                    for (int t = 0; t < T; ++t) {
                        const size_t lo = t * per, hi = std::min(len, lo + per);
                        if (lo < hi) th.emplace_back([=] { memcpy(stage + b * chunk + lo, src + off + lo, hi - lo); });
                    }
                    for (auto &x : th) x.join();
                }
                CK(hipMemcpyAsync((char *)dst + off, stage + b * chunk, len, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(done[b], s));
            }
            CK(hipStreamSynchronize(s));
            if (rep == 2) printf("staged, %2d copy threads        %.3f ms  (%.1f GB/s)\n", T, now() - t0, bytes / (now() - t0) / 1e6);
        }
        CK(hipHostFree(stage));
    }
    // persistent worker threads variant: each thread owns a slice of the source and its own staging ring
    for (int T : {4, 8, 16}) {
        const size_t chunk = (size_t)1 << 20;
        const int nbuf = 4;
        char *stage;
        CK(hipHostMalloc((void **)&stage, chunk * nbuf * T, hipHostMallocDefault));
        std::vector<hipStream_t> ss(T);
        for (auto &x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            std::vector<std::thread> th;
            const size_t per = ((bytes / T) + 4095) & ~(size_t)4095;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    hipEvent_t ev[nbuf];
                    for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
                    const size_t lo = t * per, hi = std::min(bytes, lo + per);
                    size_t k = 0;
                    for (size_t off = lo; off < hi; off += chunk, ++k) {
                        const int b = (int)(k % nbuf);
                        if (k >= (size_t)nbuf) (void)hipEventSynchronize(ev[b]);
                        const size_t len = std::min(chunk, hi - off);
                        char *st = stage + ((size_t)t * nbuf + b) * chunk;
                        memcpy(st, src + off, len);
                        (void)hipMemcpyAsync((char *)dst + off, st, len, hipMemcpyHostToDevice, ss[t]);
                        (void)hipEventRecord(ev[b], ss[t]);
                    }
                    (void)hipStreamSynchronize(ss[t]);
                    for (auto &e : ev) (void)hipEventDestroy(e);
                });
            for (auto &x : th) x.join();
            if (rep == 2) printf("per-thread rings, %2d threads    %.3f ms  (%.1f GB/s)\n", T, now() - t0, bytes / (now() - t0) / 1e6);
        }
        CK(hipHostFree(stage));
    }
    return 0;
}
