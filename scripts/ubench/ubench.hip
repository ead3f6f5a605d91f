// Issue/latency microbenchmarks for one gfx950 wave (dev tool, not part of the product path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 512
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

__global__ void kb(unsigned long long *out, int *sink, int mode) {
    __shared__ int lds[256];
    int lane = threadIdx.x & 63;
    lds[threadIdx.x & 255] = lane;
    __syncthreads();
    int v = lane, w = lane * 3, s = mode, w2 = lane + 7;
    int *sinkp = sink + threadIdx.x;
    asm volatile("s_mov_b64 s[22:23], exec" ::: "s22", "s23");
    unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {          // dependent VALU chain, 32 per iter
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(w));) }
    } else if (mode == 1) {   // independent VALU pairs
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1" : "+v"(v), "+v"(w));) }
    } else if (mode == 2) {   // readfirstlane -> salu -> v_mov chain (3 instr)
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_mov_b32 %0, %1" : "+v"(v), "+s"(s) :: "scc");) }
    } else if (mode == 3) {   // dependent SALU chain
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");) }
    } else if (mode == 4) {   // taken branch each
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_branch 1f\n s_nop 0\n1:\n" ::: "memory");) }
    } else if (mode == 5) {   // DPP dependent scan step with 2 wait states
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_nop 1\n v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v));) }
    } else if (mode == 6) {   // lds read dependent chain (pointer chase)
        int a = (lane * 4) & 1020;
        for (int i = 0; i < N; ++i) { REP32(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 2, %0" : "+v"(a) :: "memory");) }
        v = a;
    } else if (mode == 7) {   // v_cmp -> v_cndmask chain
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(w) : "vcc");) }
    } else if (mode == 8) {   // v_cmp -> s_and_saveexec -> valu -> restore exec
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_eq_u32 vcc, 63, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_mov_b64 exec, s[20:21]" : "+v"(v) : "v"(lane) : "vcc", "s20", "s21");) }
    } else if (mode == 9) {   // ds_write then ds_read same address, waited (mailbox round trip within wave)
        int a = lane * 4;
        for (int i = 0; i < N; ++i) { REP32(asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a) : "memory");) }
    } else if (mode == 10) {  // v_readlane -> s_cmp -> s_cbranch not taken
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_readlane_b32 %1, %0, 3\n s_cmp_eq_u32 %1, -7\n s_cbranch_scc1 2f\n v_add_u32 %0, %0, 1\n2:" : "+v"(v), "+s"(s) :: "scc");) }
    } else if (mode == 11) {  // conditional branch taken via scc (s_cmp; s_cbranch_scc1 taken)
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 3f\n s_nop 0\n3:" :: "s"(s) : "scc");) }
    } else if (mode == 12) {  // v_min3 dependent
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_min3_i32 %0, %0, %1, %1" : "+v"(v) : "v"(w));) }
    } else if (mode == 13) {  // s_memtime cost
        unsigned long long tt;
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt));) }
        v += (int)tt;
    } else if (mode == 14) {  // s_barrier cost for whatever block size
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_barrier" ::: "memory");) }
    } else if (mode == 15) {  // ds_write lane63 only + waitcnt
        int a = lane * 4;
        for (int i = 0; i < N; ++i) { REP32(asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)" :: "v"(v), "v"(a) : "memory");) }
    } else if (mode == 16) {  // v_readfirstlane independent of previous (throughput)
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_readfirstlane_b32 %1, %0" : "+v"(v), "=s"(s));) }
    } else if (mode == 17) {  // salu->valu: s_add; v_add using sgpr
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, %1" : "+v"(v), "+s"(s) :: "scc");) }
    }
    else if (mode == 18) {  // v_cmp e64 -> sgpr pair -> v_addc carry-in from sgpr pair
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_le_i32_e64 s[20:21], %0, %1\n v_addc_co_u32_e64 %0, s[20:21], %0, %0, s[20:21]" : "+v"(v) : "v"(w) : "s20", "s21");) }
    } else if (mode == 19) {  // v_cmp -> vcc -> v_addc carry-in from vcc
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_le_i32_e32 vcc, %0, %1\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(v) : "v"(w) : "vcc");) }
    } else if (mode == 20) {  // v_readlane -> VALU reads that sgpr
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_readlane_b32 s20, %0, 5\n v_add_u32 %0, %0, s20" : "+v"(v) :: "s20");) }
    } else if (mode == 21) {  // v_cmp e64 sgpr pair -> v_cndmask with that pair
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_le_i32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v) : "v"(w) : "s20", "s21");) }
    } else if (mode == 22) {  // global store (lane-masked) issue cost
        for (int i = 0; i < N; ++i) { REP32(asm volatile("global_store_dword %0, %1, off" :: "v"(sinkp), "v"(v) : "memory");) }
    } else if (mode == 23) {  // s_and_saveexec + restore
        for (int i = 0; i < N; ++i) { REP32(asm volatile("s_and_saveexec_b64 s[20:21], s[22:23]\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]" : "+v"(v) :: "s20", "s21");) }
    } else if (mode == 24) {  // independent v_cmp e64 + addc pairs on 2 chains
        for (int i = 0; i < N; ++i) { REP32(asm volatile("v_cmp_le_i32_e64 s[20:21], %0, %2\n v_cmp_le_i32_e64 s[24:25], %1, %2\n v_addc_co_u32_e64 %0, s[20:21], %0, %0, s[20:21]\n v_addc_co_u32_e64 %1, s[24:25], %1, %1, s[24:25]" : "+v"(v), "+v"(w2) : "v"(w) : "s20", "s21", "s24", "s25");) }
    }
    else if (mode == 25) {
        asm volatile("v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v83, 0\n s_mov_b32 s40, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b32 s46, 0x3fffffff\n s_mov_b64 s[42:43], 0\n s_bitset1_b32 s43, 31\n v_mov_b32 v89, 0\n" ::: "v60","v61","v62","v63","v69","v70","v71","v72","v88","v89","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","s40","s42","s43","s44","s45","s46","s47","s48","s49","memory");
        for (int i = 0; i < N; ++i) { asm volatile(
    "v_bfrev_b32_e32 v70, -2\n s_nop 1\n v_mov_b32_dpp v70, v69 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    "s_waitcnt lgkmcnt(1)\n v_min3_i32 v70, v60, v70, s46\n v_cmp_ge_i32_e64 s[48:49], v71, v70\n v_addc_co_u32_e64 v72, s[48:49], v72, v72, s[48:49]\n"
    "v_min_i32_e32 v70, v70, v71\n v_cmp_ge_i32_e64 s[48:49], v88, v70\n v_min_i32_e32 v88, v70, v88\n v_add_u32_e32 v88, v89, v88\n v_min_u32_e32 v88, 2.0, v88\n"
    "v_addc_co_u32_e64 v75, s[48:49], v75, v75, s[48:49]\n"
    
    "v_add_u32_e32 v60, v76, v70\n v_min_u32_e32 v60, 2.0, v60\n v_mov_b32_dpp v61, v88 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_waitcnt lgkmcnt(0)\n"
    "v_add_u32_e32 v61, v62, v61\n v_add_u32_e32 v62, v77, v60\n v_cmp_le_i32_e64 s[48:49], v61, v62\n v_min3_i32 v71, v61, v62, v78\n v_add_u32_e32 v62, v63, v60\n"
    "v_mov_b32_e32 v60, s45\n ds_read_b64 v[60:61], v60 offset:24\n v_add_u32_e32 v63, v79, v88\n v_addc_co_u32_e64 v80, s[48:49], v80, v80, s[48:49]\n"
    "v_min3_i32 v88, v62, v63, v81\n v_cmp_le_i32_e64 s[48:49], v62, v63\n v_min3_i32 v69, v71, v88, s46\n v_addc_co_u32_e64 v82, s[48:49], v82, v82, s[48:49]\n"
    "s_nop 1\n v_min_i32_dpp v69, v69, v69 row_shr:1 row_mask:0xf bank_mask:0xf\n s_or_b32 s48, s40, 4\n v_readlane_b32 s48, v83, s48\n s_lshr_b32 s48, s48, 24\n s_min_u32 s48, s48, 3\n"
    "v_min_i32_dpp v69, v69, v69 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_i32_dpp v69, v69, v69 row_shr:4 row_mask:0xf bank_mask:0xf\n"
    "v_lshl_add_u32 v62, s48, 9, v84\n ds_read_b64 v[62:63], v62\n v_min_i32_dpp v69, v69, v69 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
    "v_min_i32_dpp v69, v69, v69 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n v_min_i32_dpp v69, v69, v69 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
 :: [gp] "s"(sink) : "v60","v61","v62","v63","v69","v70","v71","v72","v88","v89","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","s40","s42","s43","s44","s45","s46","s47","s48","s49","memory"); }
    }
    else if (mode == 26) {
        asm volatile("v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v83, 0\n s_mov_b32 s40, 0\n s_mov_b32 s44, 0\n s_mov_b32 s45, 0\n s_mov_b32 s46, 0x3fffffff\n s_mov_b64 s[42:43], 0\n s_bitset1_b32 s43, 31\n v_mov_b32 v89, 0\n" ::: "v60","v61","v62","v63","v69","v70","v71","v72","v88","v89","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","s40","s42","s43","s44","s45","s46","s47","s48","s49","memory");
        for (int i = 0; i < N; ++i) { asm volatile(
    "v_bfrev_b32_e32 v70, -2\n s_nop 1\n v_mov_b32_dpp v70, v69 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
    "s_waitcnt lgkmcnt(1)\n v_min3_i32 v70, v60, v70, s46\n v_cmp_ge_i32_e64 s[48:49], v71, v70\n v_addc_co_u32_e64 v72, s[48:49], v72, v72, s[48:49]\n"
    "v_min_i32_e32 v70, v70, v71\n v_cmp_ge_i32_e64 s[48:49], v88, v70\n v_min_i32_e32 v88, v70, v88\n v_add_u32_e32 v88, v89, v88\n v_min_u32_e32 v88, 2.0, v88\n"
    "v_addc_co_u32_e64 v75, s[48:49], v75, v75, s[48:49]\n"
    "s_and_saveexec_b64 s[48:49], s[42:43]\n s_cbranch_execz 9f\n s_or_b32 s47, s40, s44\n v_min_i32_e32 v86, v60, v69\n v_mov_b32_e32 v87, s47\n v_mov_b32_e32 v89, s47\n"
    "global_store_dwordx2 v85, v[86:87], %[gp] offset:32 sc1\n global_store_dwordx2 v85, v[88:89], %[gp] offset:40 sc1\n 9:\n s_or_b64 exec, exec, s[48:49]\n"
    "v_add_u32_e32 v60, v76, v70\n v_min_u32_e32 v60, 2.0, v60\n v_mov_b32_dpp v61, v88 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_waitcnt lgkmcnt(0)\n"
    "v_add_u32_e32 v61, v62, v61\n v_add_u32_e32 v62, v77, v60\n v_cmp_le_i32_e64 s[48:49], v61, v62\n v_min3_i32 v71, v61, v62, v78\n v_add_u32_e32 v62, v63, v60\n"
    "v_mov_b32_e32 v60, s45\n ds_read_b64 v[60:61], v60 offset:24\n v_add_u32_e32 v63, v79, v88\n v_addc_co_u32_e64 v80, s[48:49], v80, v80, s[48:49]\n"
    "v_min3_i32 v88, v62, v63, v81\n v_cmp_le_i32_e64 s[48:49], v62, v63\n v_min3_i32 v69, v71, v88, s46\n v_addc_co_u32_e64 v82, s[48:49], v82, v82, s[48:49]\n"
    "s_nop 1\n v_min_i32_dpp v69, v69, v69 row_shr:1 row_mask:0xf bank_mask:0xf\n s_or_b32 s48, s40, 4\n v_readlane_b32 s48, v83, s48\n s_lshr_b32 s48, s48, 24\n s_min_u32 s48, s48, 3\n"
    "v_min_i32_dpp v69, v69, v69 row_shr:2 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_min_i32_dpp v69, v69, v69 row_shr:4 row_mask:0xf bank_mask:0xf\n"
    "v_lshl_add_u32 v62, s48, 9, v84\n ds_read_b64 v[62:63], v62\n v_min_i32_dpp v69, v69, v69 row_shr:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n"
    "v_min_i32_dpp v69, v69, v69 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n v_min_i32_dpp v69, v69, v69 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
 :: [gp] "s"(sink) : "v60","v61","v62","v63","v69","v70","v71","v72","v88","v89","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","s40","s42","s43","s44","s45","s46","s47","s48","s49","memory"); }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x] = v + w + s + w2;
}

// cross-work-group hand-over latency: block A stores a sequence number (agent scope), block B polls it and answers
__global__ void kpp(unsigned long long *out, unsigned long long *mail, int partner, int iters) {
    if (blockIdx.x != 0 && (int)blockIdx.x != partner) return;
    if (threadIdx.x != 0) return;
    unsigned long long *a = mail, *b = mail + 32;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (blockIdx.x == 0) {
        for (int i = 1; i <= iters; ++i) {
            __hip_atomic_store(a, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
        }
        out[0] = __builtin_readcyclecounter() - t0;
    } else {
        for (int i = 1; i <= iters; ++i) {
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i) {}
            __hip_atomic_store(b, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main() {
    unsigned long long *d; int *sink; hipMalloc(&d, 64); hipMalloc(&sink, 4096);
    const char *names[] = {"dep v_add", "2 indep v_add (pair)", "readfirstlane+s_add+v_mov (3)", "dep s_add", "s_branch taken(+skip nop)",
        "s_nop1 + dpp min", "ds_read chase (+waitcnt+shl)", "v_cmp+v_cndmask (2)", "v_cmp+saveexec+v_add+restore (4)", "ds_write+ds_read+wait (3)",
        "readlane+s_cmp+cbranch nt+v_add (4)", "s_cmp+cbranch taken (2)", "dep v_min3", "s_memtime+wait", "s_barrier", "ds_write+wait", "indep readfirstlane", "s_add + v_add sgpr (2)", "v_cmp_e64 sgpr + v_addc sgpr (2)", "v_cmp vcc + v_addc vcc (2)", "v_readlane + v_add sgpr (2)", "v_cmp_e64 sgpr + v_cndmask sgpr (2)", "global_store_dword", "saveexec + v_add + restore (3)", "2x(v_cmp_e64) + 2x(v_addc) indep (4)", "interior DP row as compiled, no publish (x32 = one row)", "interior DP row as compiled, with publish (x32 = one row)"};
    for (int threads = 64; threads <= 64; threads += 256)
    for (int m = 0; m < 27; ++m) {
        hipLaunchKernelGGL(kb, dim3(1), dim3(threads), 0, 0, d, sink, m);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kb, dim3(1), dim3(threads), 0, 0, d, sink, m);
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz? report raw ticks per group
        printf("threads %3d mode %2d %-42s ticks/group %.2f\n", threads, m, names[m], (double)h / (N * 32.0)); fflush(stdout);
    }
    unsigned long long *mail; hipMalloc(&mail, 4096);
    for (int partner : {1, 2, 8, 9, 16, 64, 255}) {
        hipMemset(mail, 0, 4096);
        hipLaunchKernelGGL(kpp, dim3(256), dim3(64), 0, 0, d, mail, partner, 2000);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("ping-pong block 0 <-> block %3d: %.0f cycles per one-way hand-over (store sc1 -> polling load sc1)\n", partner, (double)h / 4000.0); fflush(stdout);
    }
    return 0;
}
