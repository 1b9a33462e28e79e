// Dev microbenchmark (gfx950): what does one "flag through LDS + workgroup barrier" hand-off cost eight waves of a workgroup,
// alone on a CU and beside a second workgroup that keeps the SIMDs busy?  hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, int busy_blocks, float *sink) {
    __shared__ int flags[64];
    __shared__ float4 pts[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 512) pts[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    if ((int)blockIdx.x % 2 == 1 && busy_blocks) {           // the neighbour: VALU + LDS work, no barriers
        float a = tid;
        for (int i = 0; i < iters * 40; i++) { float4 p = pts[(tid * 7 + i) & 1023]; a = a * p.x + p.y; a = a * a + p.z; a = a * 0.5f + p.w; }
        sink[blockIdx.x * 512 + tid] = a;
        return;
    }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int acc = 0;
    float accf = 0.f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {                     // barrier only
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (MODE == 1) {              // flag write + barrier + flag read
            if (lane == 0) flags[(i & 1) * 8 + wave] = i & 3;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            acc += flags[(i & 1) * 8 + (lane & 7)];
            acc = __builtin_amdgcn_readfirstlane(acc);
        } else if (MODE == 2) {              // dependent LDS read chain (latency of a 16-byte read)
            float4 p = pts[(acc + lane) & 1023];
            acc = (int)p.x & 1023; accf += p.y;
        } else if (MODE == 3) {              // one full round: particle reads + eval + flag + barrier + flag read + barrier + w_end read
            float4 p = pts[(acc + lane) & 1023], q = pts[(acc + lane * 3) & 1023];
            float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
            float l = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz);
            unsigned long long tb = __builtin_amdgcn_ballot_w64(l > 1e30f);
            if (lane == 0) flags[(i & 1) * 8 + wave] = tb != 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            int fv = flags[(i & 1) * 8 + (lane & 7)];
            unsigned fm = (unsigned)__builtin_amdgcn_ballot_w64(fv != 0) & 0xff;
            if (fm == 0 || wave == 0) { if (lane == 0) flags[32] = i; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            acc = __builtin_amdgcn_readfirstlane(flags[32]) & 1;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + tid] = acc + accf;
}

template <int MODE> void run(const char *name, int blocks, int busy) {
    unsigned long long *d; float *s;
    hipMalloc(&d, blocks * 8); hipMalloc(&s, blocks * 512 * 4);
    hipMemset(d, 0, blocks * 8);
    const int iters = 2000;
    k<MODE><<<blocks, 512>>>(d, iters, busy, s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; b++) if (h[b]) { sum += (double)h[b]; n++; }
    printf("%-44s blocks %4d busy-neighbour %d : %.1f shader cycles (s_memtime) per iteration\n", name, blocks, busy, sum / n / iters);
    hipFree(d); hipFree(s);
}

int main() {
    for (int busy = 0; busy < 2; busy++) {
        const int blocks = busy ? 512 : 256;
        run<0>("barrier only", blocks, busy);
        run<1>("flag write + barrier + flag read", blocks, busy);
        run<2>("dependent 16-byte LDS read", blocks, busy);
        run<3>("full round (2 barriers, 3 LDS trips)", blocks, busy);
    }
    run<0>("barrier only, two sweeping WGs per CU", 512, 0);
    run<1>("flag+barrier, two sweeping WGs per CU", 512, 0);
    run<3>("full round, two sweeping WGs per CU", 512, 0);
    return 0;
}
