// Microbenchmark: what a host-in / host-out control step pays around its kernel.
//   (a) launch: host writes a request word, launches a one-thread kernel that copies it to a pinned "ack" word, spins on ack
//   (b) resident: a one-workgroup kernel stays on the GPU polling a pinned mailbox over PCIe; the host writes the
//       request word and spins on the ack the kernel writes back (no launch at all).  The kernel leaves after `max_iter`
//       requests or ~2 s without one.
// Build: hipcc --offload-arch=gfx950 -O3 launch_latency.hip -o launch_latency.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_echo(const volatile unsigned* req, volatile unsigned* ack) { *ack = *req; }

__global__ void k_resident(const unsigned* req, unsigned* ack, unsigned* exited, unsigned max_iter) {
    unsigned last = 0;
    const long long t_start = wall_clock64();
    for (unsigned it = 0; it < max_iter;) {
        const unsigned r = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r != last) {
            last = r;
            __hip_atomic_store(ack, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            ++it;
        } else if (wall_clock64() - t_start > 200000000LL) {     // 100 MHz: 2 s
            break;
        }
    }
    __hip_atomic_store(exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    unsigned *h, *d;
    hipHostMalloc((void**)&h, 256, hipHostMallocCoherent | hipHostMallocMapped);
    hipHostGetDevicePointer((void**)&d, h, 0);
    volatile unsigned* req = h;            // separate cache lines
    volatile unsigned* ack = h + 16;
    volatile unsigned* exited = h + 32;
    *req = 0; *ack = 0; *exited = 0;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 3000;
    std::vector<double> ta, tb;
    for (int i = 1; i <= N + 200; ++i) {                          // (a) launch per request
        const auto t0 = std::chrono::steady_clock::now();
        *req = (unsigned)i;
        hipLaunchKernelGGL(k_echo, dim3(1), dim3(1), 0, s, d, d + 16);
        while (*ack != (unsigned)i) {}
        const auto t1 = std::chrono::steady_clock::now();
        if (i > 200) ta.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    hipStreamSynchronize(s);
    *req = 0; *ack = 0;
    hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, s, d, d + 16, d + 32, (unsigned)(N + 200));
    for (int i = 1; i <= N + 200; ++i) {                          // (b) resident kernel, mailbox
        const auto t0 = std::chrono::steady_clock::now();
        *req = (unsigned)i;
        const auto lim = t0 + std::chrono::seconds(3);
        while (*ack != (unsigned)i) { if (*exited || std::chrono::steady_clock::now() > lim) { printf("resident kernel left early at %d\n", i); return 1; } }
        const auto t1 = std::chrono::steady_clock::now();
        if (i > 200) tb.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        // a host-side gap like a real caller's (env.step): the poll phase is random with respect to the request
        const auto g = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g).count() < 3.0) {}
    }
    hipStreamSynchronize(s);
    std::vector<double> a2 = ta, b2 = tb;
    printf("launch + echo + host spin : median %.2f us  p10 %.2f  p90 %.2f\n", med(a2), a2[a2.size() / 10], a2[a2.size() * 9 / 10]);
    printf("resident kernel mailbox   : median %.2f us  p10 %.2f  p90 %.2f\n", med(b2), b2[b2.size() / 10], b2[b2.size() * 9 / 10]);
    return 0;
}
