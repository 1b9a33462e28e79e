// Dev microbenchmark (gfx950): how fast does ONE wave get through its instruction stream while the other seven waves of its workgroup
// wait at a barrier (the strain sweep's situation)? Cycles per instruction for independent / dependent VALU, SALU, taken branches,
// LDS round trips; alone on the CU and beside a workgroup whose eight waves keep the SIMDs busy.
//   hipcc --offload-arch=gfx950 -O3 -o lone_wave_issue lone_wave_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *out, int iters, int busy, float *sink, int prio) {
    __shared__ float4 pts[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 512) pts[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    if ((int)blockIdx.x % 2 == 1 && busy) {                  // the neighbour: VALU + LDS work on all eight waves
        float a = tid;
        for (int i = 0; i < iters * 60; i++) { float4 p = pts[(tid * 7 + i) & 1023]; a = a * p.x + p.y; a = a * a + p.z; a = a * 0.5f + p.w; }
        sink[blockIdx.x * 512 + tid] = a;
        return;
    }
    unsigned long long t0 = 0, t1 = 0;
    float v0 = tid, v1 = 1.5f, v2 = 2.5f, v3 = 3.5f, v4 = 4.5f, v5 = 5.5f, v6 = 6.5f, v7 = 7.5f;
    int s0 = 1, s1 = 2; unsigned long long m0 = 0;
    if (tid < 64) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int i = 0; i < iters; i++) {
            if (MODE == 0) {          // 64 independent VALU (8 chains of 8)
                asm volatile(REP4(REP4("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"))
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(1.0f));
            } else if (MODE == 1) {   // 64 dependent VALU
                asm volatile(REP64("v_add_f32 %0, %0, %1\n\t") : "+v"(v0) : "v"(1.0f));
            } else if (MODE == 2) {   // 64 SALU (dependent)
                asm volatile(REP64("s_add_i32 %0, %0, %1\n\t") : "+s"(s0) : "s"(s1) : "scc");
            } else if (MODE == 3) {   // 32 x (VALU, SALU) interleaved
                asm volatile(REP16(REP4("v_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1\n\t")) : "+v"(v0), "+s"(s0) : "v"(1.0f) : "scc");
            } else if (MODE == 4) {   // 16 x (3 VALU + taken branch)
                asm volatile(REP16("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\ts_branch 1f\n\ts_nop 0\n\ts_nop 0\n1:\n\t") : "+v"(v0) : "v"(1.0f));
            } else if (MODE == 5) {   // 16 dependent v_sqrt
                asm volatile(REP16("v_sqrt_f32 %0, %0\n\t") : "+v"(v0));
            } else if (MODE == 6) {   // 16 dependent LDS round trips (address from the data)
                float4 p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + (float)(i & 7);
                REP4(p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y;) REP4(p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y;)
                REP4(p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y;) p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y;
                p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y; p = pts[((int)v0 + tid) & 1023]; v0 = p.x * 0.0f + p.y;
            } else if (MODE == 8) {   // 8 x (4 VALU, 4 SALU): type switches every four instructions
                asm volatile(REP4("v_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\t"
                                  "v_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\ts_add_i32 %1, %1, 1\n\t")
                             : "+v"(v0), "+s"(s0) : "v"(1.0f) : "scc");
            } else if (MODE == 9) {   // 16 x (v_cmp -> sgpr pair, s_cmp_lg_u64, s_cbranch_scc1 not taken): the ballot idiom
                asm volatile(REP16("v_cmp_gt_f32 %1, %0, %2\n\ts_cmp_lg_u64 %1, 0\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %2\n\t1:\n\t") : "+v"(v0), "=s"(m0) : "v"(-1.0f) : "scc");
            } else if (MODE == 10) {  // 16 x (v_readfirstlane, s_cmp, s_cbranch not taken, v_add)
                asm volatile(REP16("v_readfirstlane_b32 %1, %0\n\ts_cmp_eq_u32 %1, 77\n\ts_cbranch_scc1 1f\n\tv_add_f32 %0, %0, %2\n\t1:\n\t") : "+v"(v0), "+s"(s0) : "v"(1.0f) : "scc");
            } else if (MODE == 11) {  // 16 x (ds_read_b128, 8 independent VALU, wait, 1 dependent VALU): is the LDS latency hidden?
                for (int u = 0; u < 16; u++) {
                    float4 p = pts[(tid + u + i) & 1023];
                    asm volatile(REP4("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t") : "+v"(v1), "+v"(v2) : "v"(1.0f));
                    v0 += p.x;
                }
            } else if (MODE == 12) {  // 16 x (s_waitcnt with nothing outstanding + v_add)
                asm volatile(REP16("s_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(v0) : "v"(1.0f));
            } else if (MODE == 13) {  // 16 x (v_cmp vcc, v_cndmask using vcc): VALU -> VALU through vcc
                asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_f32 %0, %0, %1\n\t") : "+v"(v0) : "v"(1.0f) : "vcc");
            } else if (MODE == 14) {  // 16 x (s_and_saveexec, v_add, s_or exec): exec-mask detours
                asm volatile(REP16("s_and_saveexec_b64 %1, %2\n\tv_add_f32 %0, %0, %3\n\ts_or_b64 exec, exec, %1\n\t") : "+v"(v0), "=s"(m0) : "s"(~0ull), "v"(1.0f) : "scc");
            } else if (MODE == 15 || MODE == 16 || MODE == 17 || MODE == 18) {
                // 4 x (NR independent 16-byte LDS reads back to back, then one use of all): are a lone wave's reads pipelined?
                // 15: 2 reads, lane-consecutive records; 16: 4 reads, consecutive; 17: 4 reads, scattered records (a hash of the lane); 18: 8 reads, scattered
                for (int u = 0; u < 4; u++) {
                    const int base = (MODE == 17 || MODE == 18) ? ((tid * 37 + u * 101 + i * 7) & 1023) : ((tid + u * 64 + i) & 1023);
                    float acc4 = 0.f;
                    constexpr int NR = MODE == 15 ? 2 : (MODE == 18 ? 8 : 4);
                    float4 p[NR];
#pragma unroll
                    for (int r = 0; r < NR; r++) p[r] = pts[(base + ((MODE == 17 || MODE == 18) ? r * 211 : r * 64)) & 1023];
#pragma unroll
                    for (int r = 0; r < NR; r++) acc4 += p[r].x;
                    v0 += acc4;
                }
            } else if (MODE == 7) {   // 16 x (v_cmp -> vcc -> s_cbranch_vccz not taken)
                asm volatile(REP16("v_cmp_gt_f32 vcc, %0, %1\n\ts_cbranch_vccnz 1f\n\tv_add_f32 %0, %0, %1\n\t1:\n\t") : "+v"(v0) : "v"(-1.0f) : "vcc");
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (prio) __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + tid] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + (float)m0;
}

template <int MODE> void run(const char *name, int n_instr, int blocks, int busy, int prio) {
    unsigned long long *d; float *s;
    (void)hipMalloc(&d, blocks * 8); (void)hipMalloc(&s, blocks * 512 * 4);
    (void)hipMemset(d, 0, blocks * 8);
    const int iters = 400;
    k<MODE><<<blocks, 512>>>(d, iters, busy, s, prio);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < blocks; b++) if (h[b]) { sum += (double)h[b]; n++; }
    printf("%-46s busy-neighbour %d prio %d : %7.1f cycles per block of %2d = %5.2f per instruction\n", name, busy, prio, sum / n / iters, n_instr, sum / n / iters / n_instr);
    (void)hipFree(d); (void)hipFree(s);
}

int main() {
    for (int busy = 0; busy < 1; busy++)
        for (int prio = 0; prio < 1; prio++) {
            const int blocks = busy ? 512 : 256;
            run<0>("64 independent v_add_f32", 64, blocks, busy, prio);
            run<1>("64 dependent v_add_f32", 64, blocks, busy, prio);
            run<2>("64 dependent s_add_i32", 64, blocks, busy, prio);
            run<3>("32 x (v_add, s_add)", 64, blocks, busy, prio);
            run<4>("16 x (3 v_add + taken s_branch)", 64, blocks, busy, prio);
            run<5>("16 dependent v_sqrt_f32", 16, blocks, busy, prio);
            run<6>("16 dependent 16-byte LDS round trips", 16, blocks, busy, prio);
            run<7>("16 x (v_cmp, s_cbranch_vccnz not taken, v_add)", 48, blocks, busy, prio);
            run<8>("8 x (4 v_add, 4 s_add)", 64, blocks, busy, prio);
            run<9>("16 x (v_cmp->sgpr, s_cmp_lg_u64, s_cbranch, v_add)", 64, blocks, busy, prio);
            run<10>("16 x (v_readfirstlane, s_cmp, s_cbranch, v_add)", 64, blocks, busy, prio);
            run<11>("16 x (ds_read_b128, 8 VALU, wait+use)", 16, blocks, busy, prio);
            run<12>("16 x (s_waitcnt idle, v_add)", 32, blocks, busy, prio);
            run<13>("16 x (v_cmp vcc, v_cndmask vcc, v_add)", 48, blocks, busy, prio);
            run<14>("16 x (s_and_saveexec, v_add, s_or exec)", 48, blocks, busy, prio);
            run<15>("4 x (2 independent b128 reads, consecutive lanes, use)", 4, blocks, busy, prio);
            run<16>("4 x (4 independent b128 reads, consecutive lanes, use)", 4, blocks, busy, prio);
            run<17>("4 x (4 independent b128 reads, scattered records, use)", 4, blocks, busy, prio);
            run<18>("4 x (8 independent b128 reads, scattered records, use)", 4, blocks, busy, prio);
        }
    return 0;
}
