// Why is the hipGraph of the 512^2 iteration slower than its eager launches (162.9 vs 174.8 it/s in round 4, four rounds running)?
// A single-stream chain replays FASTER as a graph (tools/ubench/dispatch_rate.hip: 1.7 vs 2.8 us per node).  The iteration is not
// a chain: a main stream of ~170 launches forks work to a side stream ~25 times (1-3 launches each, joined a few launches later)
// and to a bulk stream ~5 times (batches of ~10 launches, joined at the end).  This program builds that shape out of spin kernels
// (durations of the real mix: most 5-20 us, twenty 100-500 us) and times it (a) as eager launches on three streams with events,
// (b) captured from exactly those launches into one hipGraph and replayed.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/graph_fork_join.hip -o tools/ubench/bin/graph_fork_join && tools/ubench/bin/graph_fork_join
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void spin_kernel(float* p, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x = p[i];
    for (int k = 0; k < spin; ++k) x = x * 1.0000001f + 1e-7f;
    p[i] = x;
}

struct Iter {
    hipStream_t s[3];
    std::vector<hipEvent_t> ev;
    float* buf[3];
    size_t nev = 0;
    hipEvent_t event() { if (nev == ev.size()) { hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev.push_back(e); } return ev[nev++]; }
    void k(int st, int spin, int blocks) { hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s[st], buf[st], spin); }
    // one "iteration": returns the number of kernel launches
    int issue() {
        nev = 0;
        int n = 0;
        hipEvent_t bulk_done = nullptr;
        for (int i = 0; i < 170; ++i) {
            const bool big = (i % 9) == 4;                              // ~19 long launches
            k(0, big ? 12000 + 3000 * (i % 7) : 400 + 150 * (i % 5), big ? 1024 : 64); ++n;
            if (i % 7 == 3) {                                           // fork to the side stream, join 3 launches later
                hipEvent_t f = event(); CK(hipEventRecord(f, s[0])); CK(hipStreamWaitEvent(s[1], f, 0));
                for (int j = 0; j < 1 + (i % 3); ++j) { k(1, 600, 64); ++n; }
                hipEvent_t d = event(); CK(hipEventRecord(d, s[1]));
                k(0, 500, 64); k(0, 500, 64); k(0, 500, 64); n += 3;
                CK(hipStreamWaitEvent(s[0], d, 0));
            }
            if (i >= 100 && i % 14 == 2) {                              // bulk batches (the weight gradients): joined at the end
                hipEvent_t f = event(); CK(hipEventRecord(f, s[0])); CK(hipStreamWaitEvent(s[2], f, 0));
                for (int j = 0; j < 10; ++j) { k(2, (j % 3 == 0) ? 9000 : 700, (j % 3 == 0) ? 512 : 64); ++n; }
                bulk_done = event(); CK(hipEventRecord(bulk_done, s[2]));
            }
        }
        if (bulk_done) CK(hipStreamWaitEvent(s[0], bulk_done, 0));
        k(0, 400, 64); ++n;                                              // "Adam"
        return n;
    }
};

int main() {
    Iter it;
    for (int i = 0; i < 3; ++i) { CK(hipStreamCreateWithFlags(&it.s[i], hipStreamNonBlocking)); CK(hipMalloc(&it.buf[i], 1024 * 256 * 4)); CK(hipMemset(it.buf[i], 0, 1024 * 256 * 4)); }
    hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int rounds = 30;
    int n = 0;
    for (int r = 0; r < 5; ++r) n = it.issue();
    CK(hipStreamSynchronize(it.s[0]));
    CK(hipEventRecord(t0, it.s[0]));
    for (int r = 0; r < rounds; ++r) it.issue();
    CK(hipEventRecord(t1, it.s[0])); CK(hipStreamSynchronize(it.s[0]));
    float ms_eager; CK(hipEventElapsedTime(&ms_eager, t0, t1));
    // single stream, eager: the serial sum of the kernels
    hipStream_t keep1 = it.s[1], keep2 = it.s[2];
    it.s[1] = it.s[0]; it.s[2] = it.s[0];
    for (int r = 0; r < 2; ++r) it.issue();
    CK(hipStreamSynchronize(it.s[0]));
    CK(hipEventRecord(t0, it.s[0]));
    for (int r = 0; r < rounds; ++r) it.issue();
    CK(hipEventRecord(t1, it.s[0])); CK(hipStreamSynchronize(it.s[0]));
    float ms_serial; CK(hipEventElapsedTime(&ms_serial, t0, t1));
    it.s[1] = keep1; it.s[2] = keep2;
    // the same launches captured into ONE graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(it.s[0], hipStreamCaptureModeThreadLocal));
    it.issue();
    CK(hipStreamEndCapture(it.s[0], &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, it.s[0]));
    CK(hipStreamSynchronize(it.s[0]));
    CK(hipEventRecord(t0, it.s[0]));
    for (int r = 0; r < rounds; ++r) CK(hipGraphLaunch(ge, it.s[0]));
    CK(hipEventRecord(t1, it.s[0])); CK(hipStreamSynchronize(it.s[0]));
    float ms_graph; CK(hipEventElapsedTime(&ms_graph, t0, t1));
    printf("%d launches per iteration, %d rounds\n", n, rounds);
    printf("eager, one stream (serial sum)      : %8.3f ms per iteration\n", ms_serial / rounds);
    printf("eager, three streams + events       : %8.3f ms per iteration\n", ms_eager / rounds);
    printf("ONE hipGraph captured from the same : %8.3f ms per iteration   (graph / eager three streams = %.3f)\n", ms_graph / rounds, ms_graph / ms_eager);
    return 0;
}
