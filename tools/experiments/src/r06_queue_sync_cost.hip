// Round 6 micro-benchmark: what a cross-queue synchronisation point costs the queue it sits on (MI355X, ROCm 7.2).
// A chain of 2 x N short kernels on stream A with one of these between every pair:
//   none        nothing (back-to-back dispatch)
//   wait_event  hipStreamWaitEvent on an event recorded long ago on stream B (already complete: a barrier-AND packet)
//   stop_event  the first kernel of the pair launched with a completion ("stop") event through hipExtLaunchKernelGGL
//   wait_value  hipStreamWaitValue32 (>=) on signal memory that already holds the value (a barrier-value packet)
//   write_value hipStreamWriteValue32 between the kernels
//   flag_store  the SECOND kernel's first thread stores a flag to signal memory (no packet at all)
// Build: hipcc --offload-arch=gfx950 -O2 r06_queue_sync_cost.hip -o /tmp/qsc ; run: /tmp/qsc
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long cycles, unsigned* flag, unsigned value) {
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
}

int main() {
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    hipEvent_t old_ev, stop_ev, t0, t1;
    CK(hipEventCreateWithFlags(&old_ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&stop_ev, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, b, 100LL, (unsigned*)nullptr, 0u);
    CK(hipEventRecord(old_ev, b));
    CK(hipStreamWriteValue32(b, sig, 7u, 0));
    CK(hipDeviceSynchronize());
    const long long cyc = 500;      // wall_clock64 runs at 100 MHz: 5 us per kernel
    const int N = 200;
    const char* names[] = {"none", "wait_event", "stop_event", "wait_value", "write_value", "flag_store"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
        if ((mode == 3 || mode == 4) && !can) continue;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(t0, a));
        for (int i = 0; i < N; ++i) {
            if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, nullptr, stop_ev, 0, cyc, (unsigned*)nullptr, 0u);
            else hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, cyc, (unsigned*)nullptr, 0u);
            if (mode == 1) CK(hipStreamWaitEvent(a, old_ev, 0));
            if (mode == 3) CK(hipStreamWaitValue32(a, sig, 7u, hipStreamWaitValueGte, 0xffffffffu));
            if (mode == 4) CK(hipStreamWriteValue32(a, sig, 7u, 0));
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, cyc, mode == 5 ? sig : (unsigned*)nullptr, 7u);
        }
        CK(hipEventRecord(t1, a));
        CK(hipEventSynchronize(t1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, t0, t1));
        printf("%-12s %7.2f us per pair of 5 us kernels\n", names[mode], 1000.0 * ms / N);
    }
    // The same with waits that are NOT yet satisfied when the host enqueues them (as in the training step, whose launches run ahead of
    // the GPU) but are when the queue reaches them: both streams start with a long kernel, B's N events / value follow a short kernel.
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {     // 0 none, 1 wait_event (pending at enqueue), 2 wait_value (pending at enqueue)
        static hipEvent_t evs[256];
        if (rep == 0 && mode == 0) for (int i = 0; i < N; ++i) CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipMemset(sig, 0, 8));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, 400000LL, (unsigned*)nullptr, 0u);     // 4 ms
        hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, b, 200000LL, (unsigned*)nullptr, 0u);       // 2 ms
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 100LL, (unsigned*)nullptr, 0u);
            if (mode == 1) CK(hipEventRecord(evs[i], b));
        }
        if (mode == 2) CK(hipStreamWriteValue32(b, sig, 11u, 0));
        CK(hipEventRecord(t0, a));
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, cyc, (unsigned*)nullptr, 0u);
            if (mode == 1) CK(hipStreamWaitEvent(a, evs[i], 0));
            if (mode == 2) CK(hipStreamWaitValue32(a, sig, 11u, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, cyc, (unsigned*)nullptr, 0u);
        }
        CK(hipEventRecord(t1, a));
        CK(hipEventSynchronize(t1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, t0, t1));
        const char* nm[] = {"none", "wait_event", "wait_value"};
        printf("pending at enqueue: %-12s %7.2f us per pair of 5 us kernels\n", nm[mode], 1000.0 * ms / N);
    }
    // and the consumer side: a kernel on stream B behind hipStreamWaitValue32 for a flag a kernel on A stores -- latency from the store
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(sig, 0, 8));
        CK(hipDeviceSynchronize());
        CK(hipStreamWaitValue32(b, sig, 9u, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, 100LL, (unsigned*)nullptr, 0u);
        CK(hipEventRecord(t1, b));
        CK(hipEventRecord(t0, a));
        hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, a, 2000LL, sig, 9u);     // stores at its START, runs 20 us
        CK(hipEventSynchronize(t1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, t0, t1));
        printf("wait_value consumer: kernel on B done %.2f us after the record in front of the storing kernel on A\n", 1000.0 * ms);
        CK(hipDeviceSynchronize());
    }
    return 0;
}
