// Is hipMallocSignalMemory host-readable, and does hipStreamWriteValue32 land where the host can poll it?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    uint32_t* f = nullptr;
    CK(hipExtMallocWithFlags((void**)&f, 8, hipMallocSignalMemory));
    hipPointerAttribute_t at;
    CK(hipPointerGetAttributes(&at, f));
    printf("type %d device %d host %p dev %p managed %d\n", (int)at.type, at.device, at.hostPointer, at.devicePointer, at.isManaged);
    CK(hipMemset(f, 0, 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamWriteValue32(s, f + 0, 1234u, 0));
    CK(hipStreamSynchronize(s));
    uint32_t v = 0;
    CK(hipMemcpy(&v, f + 0, 4, hipMemcpyDeviceToHost));
    printf("memcpy read %u\n", v);
    fflush(stdout);
    volatile uint32_t* hv = (volatile uint32_t*)f;
    printf("host deref %u\n", hv[0]);
    return 0;
}
