// Direct all-reduce of the data-parallel gradient message over peer-mapped buffers (reduce-scatter + all-gather on
// the xGMI mesh instead of a ring; no reference call site exists -- the reference is single-GPU).
//
// Every rank owns an exchange REGION in its own HBM, exported with hipIpcGetMemHandle and mapped by all peers:
//     inbox  [R][S]  floats   inbox[q]  = rank q's contribution to MY slice of the message (written by q)
//     result [R][S]  floats   result[q] = the reduced slice q (written by its owner q)
//     flagsA [R][W]  u32      flagsA[q][w] = epoch: part w of q's contribution has landed
//     flagsB [R][W]  u32      flagsB[q][w] = epoch: part w of reduced slice q has landed
//     status [16]    u32      [0] != 0: a wait gave up
// The message (n floats, the flat gradient buffer of include/umlh.h) is cut into R slices of S floats, a slice into W parts, one
// per workgroup.  Workgroup w of rank me:
//     1. writes part w of slice q of ITS message into inbox[me] of every peer q, then flagsA[me][w] there        (reduce-scatter)
//     2. waits for flagsA[q][w] of every peer, sums part w of its own slice over the ranks IN RANK ORDER (its own term comes from
//        its message) -- the same association on every owner, so all ranks end with bit-identical sums -- and writes the sum
//        into result[me] of EVERY rank (its own included), then flagsB[me][w] there                                 (all-gather)
//     3. waits for flagsB[q][w] of every rank and copies part w of every reduced slice into its message.
// Every dependency is between equal part indices: no grid-wide barrier, no workgroup waits for another workgroup of its own
// grid, and the flags are epoch-tagged (one epoch per call, never reset).  Per link and step a rank sends n/R floats twice:
// 2 x 256 KB for cfg2's 2 MB message on 8 GPUs = 3.4 us of wire time at 153 GB/s against 23 us for a ring.
// Remote data is stored write-through at system scope (sc0 sc1) and drained before the flag (a system-scope release store);
// inbox / result are read with system-scope loads, so a line that this GPU's L2 kept from the previous call is not served.
// UNMEASURED ON A MULTI-GPU NODE: built and tested with two processes on ONE GPU (tests/test_dp_gpu.py); RCCL stays the default
// transport, this one is attached explicitly (umlh_p2p_attach / HeadEngine.init_p2p / UMLH_DP_P2P=1 in bench.py).
#include "umlh_common.h"
#include <cstring>

constexpr int P2P_W = 64;            // workgroups (= parts per slice)
constexpr int P2P_MAX_RANKS = 8;
constexpr unsigned long long P2P_SPIN_TICKS = 3000000000ull;   // 30 s of s_memrealtime: a peer may simply be late (its loader, its step)

struct P2PArgs {
    unsigned char* region[P2P_MAX_RANKS];   // region[q] = rank q's region as mapped into this process
    float* msg;
    long long n, S, P;                       // message floats, slice floats (multiple of 4), part floats (multiple of 4)
    long long S_cap;                         // slice capacity of the region layout
    int R, me;
    unsigned epoch;
};

__host__ __device__ inline size_t p2p_off_inbox(long long S_cap, int R, int q) { (void)R; return (size_t)q * S_cap * 4; }
__host__ __device__ inline size_t p2p_off_result(long long S_cap, int R, int q) { return ((size_t)R + q) * S_cap * 4; }
__host__ __device__ inline size_t p2p_off_flagsA(long long S_cap, int R) { return (size_t)2 * R * S_cap * 4; }
__host__ __device__ inline size_t p2p_off_flagsB(long long S_cap, int R) { return p2p_off_flagsA(S_cap, R) + (size_t)R * P2P_W * 4; }
__host__ __device__ inline size_t p2p_off_status(long long S_cap, int R) { return p2p_off_flagsB(S_cap, R) + (size_t)R * P2P_W * 4; }
static size_t p2p_region_bytes(long long S_cap, int R) { return p2p_off_status(S_cap, R) + 64; }

__device__ __forceinline__ void st_sys_f32x4(float* p, f32x4v v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
// thread 0 of the workgroup: wait until flags[q * P2P_W + w] == epoch for every q in the mask; false = gave up (status set)
__device__ __forceinline__ bool p2p_wait(const unsigned* flags, int R, int skip, int w, unsigned epoch, unsigned* status, unsigned code) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int q = 0; q < R; ++q) {
        if (q == skip) continue;
        for (unsigned spin = 0;; ++spin) {
            if (__hip_atomic_load(flags + q * P2P_W + w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == epoch) break;
            if ((spin & 255u) == 255u) {
                if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
                if (__builtin_amdgcn_s_memrealtime() - t0 > P2P_SPIN_TICKS) {
                    unsigned expect = 0u;
                    (void)__hip_atomic_compare_exchange_strong(status, &expect, code | ((unsigned)q << 8), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return false;
                }
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
    __shared__ int ok_sh;
    const int w = (int)blockIdx.x, t = (int)threadIdx.x, R = a.R, me = a.me;
    unsigned char* mine = a.region[me];
    unsigned* status = reinterpret_cast<unsigned*>(mine + p2p_off_status(a.S_cap, R));
    const long long p0 = (long long)w * a.P, p1 = min(p0 + a.P, a.S);     // this workgroup's part of every slice
    auto msg4 = [&](long long i) -> f32x4v {                                // 4 message floats from index i, zeros past n
        f32x4v v = {0.f, 0.f, 0.f, 0.f};
        if (i + 4 <= a.n) v = *reinterpret_cast<const f32x4v*>(a.msg + i);
        else for (int j = 0; j < 4; ++j) if (i + j < a.n) v[j] = a.msg[i + j];
        return v;
    };
    // ---- 1. my contribution to every peer's slice ----
    for (int q = 0; q < R; ++q) {
        if (q == me) continue;
        float* dst = reinterpret_cast<float*>(a.region[q] + p2p_off_inbox(a.S_cap, R, me));
        for (long long i = p0 + 4 * t; i < p1; i += 1024) st_sys_f32x4(dst + i, msg4((long long)q * a.S + i));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0)
        for (int q = 0; q < R; ++q)
            if (q != me)
                __hip_atomic_store(reinterpret_cast<unsigned*>(a.region[q] + p2p_off_flagsA(a.S_cap, R)) + me * P2P_W + w, a.epoch, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- 2. reduce my slice's part in rank order, deliver it to everybody ----
    if (t == 0) ok_sh = p2p_wait(reinterpret_cast<const unsigned*>(mine + p2p_off_flagsA(a.S_cap, R)), R, me, w, a.epoch, status, 1u) ? 1 : 0;
    __syncthreads();
    if (!ok_sh) return;
    for (long long i = p0 + 4 * t; i < p1; i += 1024) {
        // all peers' terms are requested at once (one wait for the lot), then added in rank order.  The eight loads are
        // UNCONDITIONAL (absent ranks and the own slot read a valid dummy line of the own region): an asm load under a branch
        // lets the compiler copy its destination at the join, before the wait below -- stale registers
        f32x4v v[P2P_MAX_RANKS];
#pragma unroll
        for (int q = 0; q < P2P_MAX_RANKS; ++q) {
            const float* src = (q < R && q != me) ? reinterpret_cast<const float*>(mine + p2p_off_inbox(a.S_cap, R, q)) + i
                                                  : reinterpret_cast<const float*>(mine + p2p_off_inbox(a.S_cap, R, 0));
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v[q]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
        const f32x4v own = msg4((long long)me * a.S + i);
        f32x4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < P2P_MAX_RANKS; ++q)
            if (q < R) s += q == me ? own : v[q];
        for (int q = 0; q < R; ++q) st_sys_f32x4(reinterpret_cast<float*>(a.region[q] + p2p_off_result(a.S_cap, R, me)) + i, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0)
        for (int q = 0; q < R; ++q)
            __hip_atomic_store(reinterpret_cast<unsigned*>(a.region[q] + p2p_off_flagsB(a.S_cap, R)) + me * P2P_W + w, a.epoch, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- 3. all reduced slices -> my message ----
    __syncthreads();
    if (t == 0) ok_sh = p2p_wait(reinterpret_cast<const unsigned*>(mine + p2p_off_flagsB(a.S_cap, R)), R, -1, w, a.epoch, status, 2u) ? 1 : 0;
    __syncthreads();
    if (!ok_sh) return;
    for (long long i = p0 + 4 * t; i < p1; i += 1024) {
        f32x4v v[P2P_MAX_RANKS];
#pragma unroll
        for (int q = 0; q < P2P_MAX_RANKS; ++q) {
            const float* src = reinterpret_cast<const float*>(mine + p2p_off_result(a.S_cap, R, q < R ? q : 0)) + i;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v[q]) : "v"(src) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
        for (int q = 0; q < P2P_MAX_RANKS; ++q) {
            if (q >= R) continue;
            const long long o = (long long)q * a.S + i;
            if (o + 4 <= a.n) *reinterpret_cast<f32x4v*>(a.msg + o) = v[q];
            else for (int j = 0; j < 4; ++j) if (o + j < a.n) a.msg[o + j] = v[q][j];
        }
    }
}

extern "C" {

// bytes of one rank's exchange region for messages of up to n_max floats among n_ranks ranks
uint64_t umlh_p2p_region_bytes(int64_t n_max, int32_t n_ranks) {
    if (n_max < 1 || n_ranks < 1 || n_ranks > P2P_MAX_RANKS) return 0;
    const long long S_cap = (((n_max + n_ranks - 1) / n_ranks) + 4 * P2P_W - 1) / (4 * P2P_W) * (4 * P2P_W);
    return (uint64_t)p2p_region_bytes(S_cap, n_ranks);
}

// device allocation suited to peer access (fine-grained where the runtime offers it), zeroed; freed with umlh_p2p_free
int umlh_p2p_alloc(uint64_t bytes, void** out) {
    if (!out || bytes == 0) return (int)hipErrorInvalidValue;
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, (size_t)bytes); }
    if (e != hipSuccess) return (int)e;
    e = hipMemset(p, 0, (size_t)bytes);
    if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
    *out = p;
    return 0;
}
int umlh_p2p_free(void* p) { return p ? (int)hipFree(p) : 0; }
int umlh_p2p_export(void* p, void* handle64) {
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) return (int)e;
    static_assert(sizeof(h) == 64, "hipIpcMemHandle_t is 64 bytes");
    memcpy(handle64, &h, 64);
    return 0;
}
int umlh_p2p_open(const void* handle64, void** out) {
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, 64);
    return (int)hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess);
}
int umlh_p2p_close(void* p) { return p ? (int)hipIpcCloseMemHandle(p) : 0; }

int umlh_p2p_launch(void* const* regions, int n_ranks, int rank, float* msg, long long n, long long n_max, unsigned epoch, hipStream_t st) {
    if (!regions || n_ranks < 1 || n_ranks > P2P_MAX_RANKS || rank < 0 || rank >= n_ranks || n < 1 || n > n_max || epoch == 0) return (int)hipErrorInvalidValue;
    P2PArgs a;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < n_ranks; ++q) a.region[q] = static_cast<unsigned char*>(regions[q]);
    a.msg = msg; a.n = n; a.R = n_ranks; a.me = rank; a.epoch = epoch;
    a.S_cap = (((n_max + n_ranks - 1) / n_ranks) + 4 * P2P_W - 1) / (4 * P2P_W) * (4 * P2P_W);
    a.S = (((n + n_ranks - 1) / n_ranks) + 3) / 4 * 4;
    a.P = ((a.S + P2P_W - 1) / P2P_W + 3) / 4 * 4;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_W), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int umlh_p2p_status_offset(long long n_max, int n_ranks, unsigned long long* off) {
    if (!off || n_ranks < 1) return (int)hipErrorInvalidValue;
    const long long S_cap = (((n_max + n_ranks - 1) / n_ranks) + 4 * P2P_W - 1) / (4 * P2P_W) * (4 * P2P_W);
    *off = p2p_off_status(S_cap, n_ranks);
    return 0;
}

}  // extern "C"
