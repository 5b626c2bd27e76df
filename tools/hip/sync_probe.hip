// Flag hand-over latency between two workgroups of ONE persistent launch, same XCD and across XCDs, for the load / store
// encodings a hand-written protocol can choose from on gfx950 (the compiler only offers "agent": sc1 on both sides, which
// goes to the memory side; an XCD's workgroups share its L2 and could meet there):
//   ping-pong of a counter between thread 0 of workgroup A and thread 0 of workgroup B; one-way latency = round trip / 2.
//   After each hand-over the consumer also reads a 2 KB payload the producer has just rewritten with plain stores (the data a
//   stage leaves for the next one) and counts stale words - with the acquire (`buffer_inv`) the variant prescribes.
// build: hipcc --offload-arch=gfx950 -O2 -o sync_probe sync_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
// ---- loads
__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_sc0(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_plain(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_nt(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned ld_rmw_wg(unsigned* p) { return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned ld_rmw_agent(unsigned* p) { return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ---- stores
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_plain(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc0(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_rmw_wg(unsigned* p, unsigned v) { (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

enum { L_AGENT, L_SC0, L_PLAIN, L_NT, L_RMW_WG, L_RMW_AGENT };
enum { S_AGENT, S_PLAIN, S_SC0, S_RMW_WG };
enum { REL_AGENT, REL_WAIT };    // release before the flag store: agent fence (buffer_wbl2 sc1 + waits) | s_waitcnt vmcnt(0) only
enum { ACQ_AGENT, ACQ_INV_SC0, ACQ_NONE };  // acquire after the flag is seen: agent fence (buffer_inv sc1) | buffer_inv sc0 | nothing

template <int L> __device__ __forceinline__ unsigned ld(unsigned* p) {
    if (L == L_AGENT) return ld_agent(p);
    if (L == L_SC0) return ld_sc0(p);
    if (L == L_PLAIN) return ld_plain(p);
    if (L == L_NT) return ld_nt(p);
    if (L == L_RMW_WG) return ld_rmw_wg(p);
    return ld_rmw_agent(p);
}
template <int S> __device__ __forceinline__ void st(unsigned* p, unsigned v) {
    if (S == S_AGENT) st_agent(p, v);
    else if (S == S_PLAIN) st_plain(p, v);
    else if (S == S_SC0) st_sc0(p, v);
    else st_rmw_wg(p, v);
}
template <int R> __device__ __forceinline__ void rel() {
    if (R == REL_AGENT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int A> __device__ __forceinline__ void acq() {
    if (A == ACQ_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    else if (A == ACQ_INV_SC0) asm volatile("buffer_inv sc0" ::: "memory");
}

#define PAYLOAD 256  /* doubles */
struct Res { long long ticks; unsigned stale, timeouts, xa, xb; };

// workgroups wa and wb play; everybody else leaves.  flags: [0] A -> B, [32] B -> A (separate lines); payloads likewise
template <int L, int S, int R, int A>
__global__ void k_pingpong(unsigned* flags, double* payA, double* payB, int wa, int wb, int rounds, Res* res) {
    const int me = (int)blockIdx.x;
    if (me != wa && me != wb) return;
    const bool isA = me == wa;
    unsigned* f_out = flags + (isA ? 0 : 32);
    unsigned* f_in = flags + (isA ? 32 : 0);
    double* p_out = isA ? payA : payB;
    const double* p_in = isA ? payB : payA;
    __shared__ unsigned s_to;
    unsigned stale = 0, timeouts = 0;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; ++i) {
        if (isA) {  // A writes round i, then waits for B's echo of round i
            for (int k = threadIdx.x; k < PAYLOAD; k += blockDim.x) p_out[k] = (double)(i * 1000 + k);
            __syncthreads();
            if (threadIdx.x == 0) { rel<R>(); st<S>(f_out, (unsigned)i); }
        }
        if (threadIdx.x == 0) {
            unsigned spins = 0, to = 0;
            while (ld<L>(f_in) < (unsigned)i) { if (++spins > (1u << 22)) { to = 1; break; } }
            acq<A>();
            s_to = to;
        }
        __syncthreads();
        if (s_to) { ++timeouts; break; }
        for (int k = threadIdx.x; k < PAYLOAD; k += blockDim.x) if (p_in[k] != (double)(i * 1000 + k)) ++stale;
        if (!isA) {  // B echoes
            for (int k = threadIdx.x; k < PAYLOAD; k += blockDim.x) p_out[k] = (double)(i * 1000 + k);
            __syncthreads();
            if (threadIdx.x == 0) { rel<R>(); st<S>(f_out, (unsigned)i); }
        }
    }
    const long long t1 = wall_clock64();
    for (int o = 32; o > 0; o >>= 1) stale += __shfl_xor(stale, o);
    if (threadIdx.x == 0) {
        Res* r = res + (isA ? 0 : 1);
        r->ticks = t1 - t0; r->stale = stale; r->timeouts = timeouts; r->xa = xcc_id(); r->xb = 0;
    }
}
__global__ void k_xcc(unsigned* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <int L, int S, int R, int A>
static int run(const char* name, unsigned* flags, double* pa, double* pb, Res* dres, int wa, int wb, int grid) {
    const int rounds = 2000;
    CK(hipMemset(flags, 0, 64 * 4));
    CK(hipMemset(dres, 0, 2 * sizeof(Res)));
    k_pingpong<L, S, R, A><<<grid, 64>>>(flags, pa, pb, wa, wb, rounds, dres);
    CK(hipDeviceSynchronize());
    Res h[2];
    CK(hipMemcpy(h, dres, sizeof h, hipMemcpyDeviceToHost));
    printf("  %-58s wg %3d (xcc %u) <-> wg %3d (xcc %u): one way %.3f us   stale words %u   timeouts %u\n", name, wa, h[0].xa, wb, h[1].xa,
           h[0].ticks / 100.0 / rounds / 2.0, h[0].stale + h[1].stale, h[0].timeouts + h[1].timeouts);
    return 0;
}

int main() {
    const int grid = 256;
    unsigned *flags, *dx;
    double *pa, *pb;
    Res* dres;
    CK(hipMalloc(&flags, 64 * 4)); CK(hipMalloc(&dx, grid * 4));
    CK(hipMalloc(&pa, PAYLOAD * 8)); CK(hipMalloc(&pb, PAYLOAD * 8)); CK(hipMalloc(&dres, 2 * sizeof(Res)));
    k_xcc<<<grid, 64>>>(dx);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hx(grid);
    CK(hipMemcpy(hx.data(), dx, grid * 4, hipMemcpyDeviceToHost));
    int same = 0;
    for (int i = 0; i < grid; ++i) same += hx[i] == (unsigned)(i & 7) ? 1 : 0;
    printf("xcc id == blockIdx & 7 for %d of %d workgroups (first 16:", same, grid);
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
    printf(")\n");
    for (int pass = 0; pass < 2; ++pass) {
        const int wa = 0, wb = pass == 0 ? 8 : 1;
        printf("%s\n", pass == 0 ? "same XCD (by blockIdx & 7):" : "different XCDs:");
#define RUN(L, S, R, A) if (run<L, S, R, A>(#L " " #S " " #R " " #A, flags, pa, pb, dres, wa, wb, grid)) return 1
        RUN(L_AGENT, S_AGENT, REL_AGENT, ACQ_AGENT);
        RUN(L_AGENT, S_AGENT, REL_WAIT, ACQ_AGENT);
        RUN(L_SC0, S_PLAIN, REL_WAIT, ACQ_AGENT);
        RUN(L_SC0, S_SC0, REL_WAIT, ACQ_AGENT);
        RUN(L_SC0, S_SC0, REL_WAIT, ACQ_INV_SC0);
        RUN(L_SC0, S_SC0, REL_WAIT, ACQ_NONE);
        RUN(L_NT, S_PLAIN, REL_WAIT, ACQ_AGENT);
        RUN(L_PLAIN, S_PLAIN, REL_WAIT, ACQ_AGENT);
        RUN(L_RMW_WG, S_RMW_WG, REL_WAIT, ACQ_AGENT);
        RUN(L_RMW_WG, S_PLAIN, REL_WAIT, ACQ_AGENT);
        RUN(L_RMW_AGENT, S_AGENT, REL_WAIT, ACQ_AGENT);
        RUN(L_AGENT, S_RMW_WG, REL_WAIT, ACQ_AGENT);
        RUN(L_SC0, S_RMW_WG, REL_WAIT, ACQ_AGENT);
#undef RUN
    }
    return 0;
}
