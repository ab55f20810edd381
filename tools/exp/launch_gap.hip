// Dispatch cadence on one stream: what makes a kernel-to-kernel boundary cost ~6 us instead of ~0 in the training step's trace?
// Every kernel spins ~4 us (wall clock) in wave 0 of each workgroup; sequences of 2000 launches are timed with events.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/launch_gap tools/exp/launch_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

struct Big { const float* p[30]; int n[16]; };   // 304 bytes like GemmArgs

__device__ __forceinline__ void spin(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__global__ void __launch_bounds__(256) k_small(float* out, long ticks) { spin(ticks); if (ticks < 0) out[0] = 1.f; }
__global__ void __launch_bounds__(256) k_bigarg(Big b, float* out, long ticks) { spin(ticks); if (ticks < 0) out[0] = b.p[b.n[3] & 15][0]; }
template <int KB>
__global__ void __launch_bounds__(256) k_lds(float* out, long ticks) {
    __shared__ float s[KB * 256];
    s[threadIdx.x] = (float)ticks;
    __syncthreads();
    spin(ticks);
    if (ticks < 0) out[0] = s[(threadIdx.x * 7) % (KB * 256)];
}
__global__ void __launch_bounds__(256) k_vgpr(float* out, long ticks) {
    float acc[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) acc[i] = out[i + threadIdx.x] * (float)i;
    spin(ticks);
#pragma unroll
    for (int i = 0; i < 200; ++i) asm volatile("" : "+v"(acc[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 200; ++i) s += acc[i];
    if (ticks < 0 || s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_mfma(float* out, long ticks) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f a = {0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x4f32((float)threadIdx.x, 1.f, a, 0, 0, 0);
    spin(ticks);
    if (ticks < 0 || a[0] == 123.456f) out[0] = a[1];
}
__global__ void __launch_bounds__(256) k_write(float* out, long n) {      // writes n floats (dirty L2 lines at the kernel's end)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (float)i;
}

int main() {
    float* buf; hipMalloc(&buf, 256 << 20); hipMemset(buf, 0, 256 << 20);
    hipStream_t st; hipStreamCreate(&st);
    const long ticks = 400;     // wall_clock64 runs at 100 MHz: 4 us
    Big big{}; for (int i = 0; i < 30; ++i) big.p[i] = buf;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    using Fn = std::function<void()>;
    auto small = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 0, st, buf, ticks); }); };
    auto bigarg = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_bigarg, dim3(wgs), dim3(256), 0, st, big, buf, ticks); }); };
    auto lds20 = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_lds<20>, dim3(wgs), dim3(256), 0, st, buf, ticks); }); };
    auto lds60 = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_lds<60>, dim3(wgs), dim3(256), 0, st, buf, ticks); }); };
    auto dynlds = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), 60 * 1024, st, buf, ticks); }); };
    auto vgpr = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_vgpr, dim3(wgs), dim3(256), 0, st, buf, ticks); }); };
    auto mfma = [&](int wgs) { return Fn([=] { hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, st, buf, ticks); }); };
    auto wr = [&](long n) { return Fn([=] { hipLaunchKernelGGL(k_write, dim3(1024), dim3(256), 0, st, buf, n); }); };
    auto run = [&](const char* name, std::vector<Fn> seq) {
        const int reps = 2000 / (int)seq.size();
        for (int i = 0; i < 50; ++i) for (auto& f : seq) f();
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) for (auto& f : seq) f();
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / (reps * seq.size()));
    };
    for (int wgs : {64, 320, 2048}) {
        printf("-- %d workgroups of 256 threads, 4 us spin\n", wgs);
        run("small", {small(wgs)});
        run("bigarg (304 B)", {bigarg(wgs)});
        run("static lds 20 KB", {lds20(wgs)});
        run("static lds 60 KB", {lds60(wgs)});
        run("dynamic lds 60 KB", {dynlds(wgs)});
        run("200 vgprs", {vgpr(wgs)});
        run("mfma", {mfma(wgs)});
        run("small / lds60 alternating", {small(wgs), lds60(wgs)});
        run("small / vgpr alternating", {small(wgs), vgpr(wgs)});
        run("small / mfma alternating", {small(wgs), mfma(wgs)});
        run("lds20 / vgpr alternating", {lds20(wgs), vgpr(wgs)});
    }
    printf("-- writers (no spin): n floats written per launch\n");
    for (long n : {1L << 16, 1L << 20, 1L << 22, 1L << 24}) { char nm[64]; snprintf(nm, 64, "write %ld floats", n); run(nm, {wr(n)}); }
    for (long n : {1L << 16, 1L << 20, 1L << 22}) { char nm[64]; snprintf(nm, 64, "write %ld floats / small(64)", n); run(nm, {wr(n), small(64)}); }
    return 0;
}
