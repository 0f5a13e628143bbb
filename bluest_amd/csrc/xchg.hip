// xchg.hip -- Part 5 of include/bluest_hip.h: one-shot all-reduce(SUM) of the small Phi record between the GPUs of one node by
// direct peer writes.  The record is 5-27 KB, so the exchange is LATENCY-bound: a ring (RCCL) pays
// 2(P-1) hops, this scheme pays one -- every rank writes its record into its slot on every peer (xGMI is point-to-point, all
// peers are written concurrently), then sums the P slots it received in RANK ORDER, so every rank ends with the bit-identical sum.
//
// Transport: 8-byte granules {32 payload bits | 32-bit tag}; an aligned 8-byte store is single-copy atomic, so a reader that
// finds the expected tag has the payload of THIS call -- no flags, no fences, nothing to reset (the tag is the call counter).
// Two mailbox sets alternate by call parity: a rank can only be one call ahead of a peer (it needs the peer's data of call c
// to finish call c, and the peer sends that only after finishing call c-1), so the set of call c-1 is free when call c+1 writes.
// Mailboxes live in FINE-GRAINED device memory (remote writes must become visible to a kernel that is already running) and are
// shared between the processes of a node with hipIpc handles.  Every wait is bounded: a time-out poisons the result with NaN and
// raises a sticky flag that bluest_xchg_status reports.
#include "common.hpp"

struct bluest_xchg_s {
    int world = 0, rank = 0;
    int64_t max_doubles = 0;
    size_t set_bytes = 0;                 // bytes of one mailbox set (world slots)
    void *local = nullptr;                // my mailboxes: [2 sets][world slots][max_doubles][2 granules]
    std::vector<void *> peer;             // peer[p] = base of rank p's mailboxes as mapped here (peer[rank] = local)
    void **d_peer = nullptr;              // the same table in device memory
    unsigned long long *d_state = nullptr;   // [0] = call counter (tag of the next call), [1] = sticky time-out flag
    bool connected = false;
};

__device__ __forceinline__ void store_sys(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long load_sys(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one workgroup; thread t owns the doubles t, t + blockDim, ...: it sends them to every peer and then sums what every peer sent
__global__ __launch_bounds__(1024) void k_xchg_allreduce(double *__restrict__ buf, int64_t n, int world, int rank, int64_t max_doubles,
                                                        size_t set_bytes, void *const *__restrict__ peers, void *local,
                                                        unsigned long long *__restrict__ state, long long max_polls)
{
    const unsigned long long call = state[0];
    const unsigned long long tag = (call % 0xfffffffeull) + 1ull;          // 1 .. 2^32-2: never 0 (0 = empty mailbox)
    const size_t set_off = (size_t)(call & 1ull) * set_bytes;
    const size_t slot_bytes = (size_t)max_doubles * 16;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(buf[i]);
        const unsigned long long g0 = (bits & 0xffffffffull) | (tag << 32), g1 = (bits >> 32) | (tag << 32);
        for (int q = 0; q < world; q++) {
            const int p = (rank + q) % world;                                   // spread the first writes over the links
            unsigned long long *dst = reinterpret_cast<unsigned long long *>((char *)peers[p] + set_off + (size_t)rank * slot_bytes) + 2 * i;
            store_sys(dst, g0);
            store_sys(dst + 1, g1);
        }
    }
    bool timed_out = false;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        double sum = 0.0;
        for (int p = 0; p < world; p++) {                                       // fixed order: identical bits on every rank
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>((const char *)local + set_off + (size_t)p * slot_bytes) + 2 * i;
            unsigned long long a = load_sys(src), b = load_sys(src + 1);
            long long polls = 0;
            while (((a >> 32) != tag || (b >> 32) != tag) && polls < max_polls) {
                __builtin_amdgcn_s_sleep(2);
                a = load_sys(src); b = load_sys(src + 1);
                polls++;
            }
            if ((a >> 32) != tag || (b >> 32) != tag) { timed_out = true; break; }
            sum += __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
        }
        buf[i] = timed_out ? NAN : sum;
    }
    if (timed_out) atomicExch(&state[1], 1ull);
    __syncthreads();
    if (threadIdx.x == 0) state[0] = call + 1ull;
}

extern "C" int bluest_xchg_create(bluest_xchg_t *x_out, int world, int rank, int64_t max_doubles, void *handle_out)
{
    if (!x_out || !handle_out) return fail(BLUEST_ERR_ARG, "null pointer");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(BLUEST_ERR_ARG, "world=%d rank=%d out of range", world, rank);
    if (max_doubles <= 0 || max_doubles > (1 << 20)) return fail(BLUEST_ERR_ARG, "max_doubles out of range");
    int rc = require_gpu(); if (rc) return rc;
    bluest_xchg_s *x = new bluest_xchg_s();
    x->world = world; x->rank = rank; x->max_doubles = max_doubles;
    x->set_bytes = (size_t)world * (size_t)max_doubles * 16;
    const size_t bytes = 2 * x->set_bytes;
    hipError_t e = hipExtMallocWithFlags(&x->local, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); delete x; return fail(BLUEST_ERR_HIP, "fine-grained mailbox allocation failed: %s", hipGetErrorString(e)); }
    e = hipMemset(x->local, 0, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&x->d_state, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(x->d_state, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void **)&x->d_peer, (size_t)world * sizeof(void *));
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, x->local);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (x->local) (void)hipFree(x->local);
        if (x->d_state) (void)hipFree(x->d_state);
        if (x->d_peer) (void)hipFree(x->d_peer);
        delete x;
        return fail(BLUEST_ERR_HIP, "mailbox set-up failed: %s", hipGetErrorString(e));
    }
    static_assert(sizeof(hipIpcMemHandle_t) <= BLUEST_XCHG_HANDLE_BYTES, "handle size");
    memset(handle_out, 0, BLUEST_XCHG_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    x->peer.assign(world, nullptr);
    x->peer[rank] = x->local;
    *x_out = x;
    return BLUEST_OK;
}

extern "C" int bluest_xchg_connect(bluest_xchg_t x, const void *all_handles)
{
    if (!x || !all_handles) return fail(BLUEST_ERR_ARG, "null pointer");
    if (x->connected) return fail(BLUEST_ERR_STATE, "already connected");
    for (int p = 0; p < x->world; p++) {
        if (p == x->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)all_handles + (size_t)p * BLUEST_XCHG_HANDLE_BYTES, sizeof(h));
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(BLUEST_ERR_HIP, "hipIpcOpenMemHandle for rank %d failed: %s", p, hipGetErrorString(e));
        }
        x->peer[p] = ptr;
    }
    HIP_TRY(hipMemcpy(x->d_peer, x->peer.data(), (size_t)x->world * sizeof(void *), hipMemcpyHostToDevice));
    x->connected = true;
    return BLUEST_OK;
}

extern "C" int bluest_xchg_allreduce_sum(bluest_xchg_t x, double *buf_dev, int64_t n_doubles, void *stream)
{
    if (!x || !buf_dev) return fail(BLUEST_ERR_ARG, "null pointer");
    if (!x->connected) return fail(BLUEST_ERR_STATE, "exchange not connected");
    if (n_doubles <= 0 || n_doubles > x->max_doubles) return fail(BLUEST_ERR_ARG, "n_doubles=%lld outside 1..%lld", (long long)n_doubles, (long long)x->max_doubles);
    const int threads = (int)std::min<int64_t>(1024, (n_doubles + 63) / 64 * 64);
    hipLaunchKernelGGL(k_xchg_allreduce, dim3(1), dim3(threads), 0, (hipStream_t)stream, buf_dev, n_doubles, x->world, x->rank,
                       x->max_doubles, x->set_bytes, (void *const *)x->d_peer, x->local, x->d_state, (long long)4000000);
    HIP_TRY(hipGetLastError());
    return BLUEST_OK;
}

extern "C" int bluest_xchg_status(bluest_xchg_t x, int64_t *calls, int *timed_out)
{
    if (!x) return fail(BLUEST_ERR_ARG, "null pointer");
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, x->d_state, sizeof(h), hipMemcpyDeviceToHost));
    if (calls) *calls = (int64_t)h[0];
    if (timed_out) *timed_out = h[1] ? 1 : 0;
    return BLUEST_OK;
}

extern "C" int bluest_xchg_destroy(bluest_xchg_t x)
{
    if (!x) return BLUEST_OK;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < x->world; p++)
        if (p != x->rank && x->peer.size() > (size_t)p && x->peer[p]) (void)hipIpcCloseMemHandle(x->peer[p]);
    if (x->local) (void)hipFree(x->local);
    if (x->d_state) (void)hipFree(x->d_state);
    if (x->d_peer) (void)hipFree(x->d_peer);
    (void)hipGetLastError();
    delete x;
    return BLUEST_OK;
}
