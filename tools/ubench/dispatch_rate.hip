// Dispatch-rate bound of "B independent fits on one GPU" (DESIGN 7.2b).
// A small-net fit is a dependent chain of ~270 launches of 5-15 us each.  Today B fits run as B hipGraphs on B
// streams; 8 snail fits reach 985 it/s = 2160 kernel nodes in 8.1 ms = 3.75 us per node.  Is that the command
// processor's dispatch rate (then only FEWER dispatches help: grouped launches, one launch for B instances), or is
// there headroom in the schedule?  This program measures, for a kernel of a given duration:
//   (1) one stream, eager launches                         -> us per launch of a dependent chain
//   (2) one hipGraph of N chained kernel nodes             -> us per node
//   (3) B graphs of N chained nodes on B streams at once   -> aggregate us per node
//   (4) one graph of N chained nodes with B x the blocks   -> us per node of the "grouped launch" form
// usage: dispatch_rate [nodes=270] [B=8] [rounds=20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// one block = 256 threads; `spin` dependent FMAs per thread set the kernel's duration (0: ~2 us launch floor)
__global__ __launch_bounds__(256) void chain_kernel(float* p, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float x = p[i];
    for (int k = 0; k < spin; ++k) x = x * 1.0000001f + 1e-7f;
    p[i] = x;
}

static float elapsed_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

static hipGraphExec_t capture_chain(hipStream_t st, float* buf, int nodes, int blocks, int spin) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int n = 0; n < nodes; ++n) hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, st, buf, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    return ge;       // (graphs are kept alive until exit)
}

int main(int argc, char** argv) {
    const int nodes = argc > 1 ? atoi(argv[1]) : 270;
    const int B = argc > 2 ? atoi(argv[2]) : 8;
    const int rounds = argc > 3 ? atoi(argv[3]) : 20;
    const int maxblocks = 64 * B;
    std::vector<hipStream_t> st(B);
    std::vector<float*> buf(B);
    for (int b = 0; b < B; ++b) {
        CK(hipStreamCreateWithFlags(&st[b], hipStreamNonBlocking));
        CK(hipMalloc(&buf[b], (size_t)maxblocks * 256 * sizeof(float)));
        CK(hipMemset(buf[b], 0, (size_t)maxblocks * 256 * sizeof(float)));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> done(B);
    for (int b = 0; b < B; ++b) CK(hipEventCreateWithFlags(&done[b], hipEventDisableTiming));

    printf("# nodes per chain %d, B %d, rounds %d; blocks per kernel 64 (x B in the grouped form)\n", nodes, B, rounds);
    const int spins[3] = {0, 2000, 8000};
    for (int si = 0; si < 3; ++si) {
        const int spin = spins[si];
        // kernel duration alone: 50 launches back to back, events around them, one stream (includes the launch gaps)
        // (1) eager, one stream
        for (int n = 0; n < 20; ++n) hipLaunchKernelGGL(chain_kernel, dim3(64), dim3(256), 0, st[0], buf[0], spin);
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < rounds; ++r)
            for (int n = 0; n < nodes; ++n) hipLaunchKernelGGL(chain_kernel, dim3(64), dim3(256), 0, st[0], buf[0], spin);
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        const float eager_us = elapsed_ms(e0, e1) * 1e3f / (rounds * nodes);

        // (2) one graph, one stream
        hipGraphExec_t g1 = capture_chain(st[0], buf[0], nodes, 64, spin);
        CK(hipGraphLaunch(g1, st[0]));
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < rounds; ++r) CK(hipGraphLaunch(g1, st[0]));
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        const float graph1_us = elapsed_ms(e0, e1) * 1e3f / (rounds * nodes);

        // (3) B graphs on B streams: fork from stream 0, join on stream 0 (as GraphedIteration.group does per round)
        std::vector<hipGraphExec_t> gb(B);
        for (int b = 0; b < B; ++b) gb[b] = capture_chain(st[b], buf[b], nodes, 64, spin);
        for (int b = 0; b < B; ++b) CK(hipGraphLaunch(gb[b], st[b]));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int b = 1; b < B; ++b) CK(hipStreamWaitEvent(st[b], e0, 0));
        for (int r = 0; r < rounds; ++r)
            for (int b = 0; b < B; ++b) CK(hipGraphLaunch(gb[b], st[b]));
        for (int b = 1; b < B; ++b) {
            CK(hipEventRecord(done[b], st[b]));
            CK(hipStreamWaitEvent(st[0], done[b], 0));
        }
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        const float ms_b = elapsed_ms(e0, e1);
        const float graphB_us = ms_b * 1e3f / ((float)rounds * nodes * B);

        // (4) one graph whose kernels carry B x the blocks (the grouped-launch form: same nodes, B x the work)
        hipGraphExec_t gg = capture_chain(st[0], buf[0], nodes, 64 * B, spin);
        CK(hipGraphLaunch(gg, st[0]));
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventRecord(e0, st[0]));
        for (int r = 0; r < rounds; ++r) CK(hipGraphLaunch(gg, st[0]));
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        const float ms_g = elapsed_ms(e0, e1);
        const float grouped_us = ms_g * 1e3f / (rounds * nodes);

        printf("spin %5d: eager %.2f us/launch | 1 graph %.2f us/node | %d graphs on %d streams %.2f us/node aggregate "
               "(%.2f ms per round of %d chains) | grouped x%d %.2f us/node (%.2f ms per round of %d chains)"
               " -> grouped / B-graphs time %.2f\n",
               spin, eager_us, graph1_us, B, B, graphB_us, ms_b / rounds, B, B, grouped_us, ms_g / rounds, B,
               (ms_g / rounds) / (ms_b / rounds));
    }
    return 0;
}
