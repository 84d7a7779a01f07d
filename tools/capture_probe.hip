// capture_probe: which HIP runtime calls of ANOTHER thread break a thread-local stream capture on this ROCm?
//
// The round ring's graph form (hs_host.hip, hs_ed25519_ring_submit) captures a slot's round with
// hipStreamCaptureModeThreadLocal while other threads of the host keep calling the engine (key-set registration,
// context growth, synchronous verifies).  Round 5: tests/test_gpu_ring.py's registration-while-capturing test failed
// once in two runs with "operation failed due to a previous error during capture" - some call of the OTHER thread
// invalidated the capture.  This program measures every candidate by itself: thread A captures / instantiates /
// launches / destroys a small graph in a loop, thread B loops ONE runtime call for `ms` milliseconds; per candidate
// it prints how many of A's captures failed (and with what) and what B's own calls returned.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/capture_probe tools/capture_probe.hip -lpthread ; tools/capture_probe [ms]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

__global__ void k_touch(unsigned *p, unsigned n) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 2654435761u + i;
}

struct capture_side {
    hipStream_t st = nullptr;
    unsigned *d = nullptr, *h = nullptr;
    size_t ok = 0, failed = 0;
    std::map<std::string, size_t> why;      // "<call>: <error name>" -> count
    void note(const char *call, hipError_t e) { why[std::string(call) + ": " + hipGetErrorName(e)]++; }
    // one round of what ring_submit does under capture; returns false when the capture did not survive
    bool round(unsigned n) {
        hipGraph_t g = nullptr; hipGraphExec_t x = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { note("begin", e); (void)hipGetLastError(); return false; }
        hipError_t first = hipSuccess; const char *where = "";
        auto step = [&](const char *w, hipError_t r) { if (r != hipSuccess && first == hipSuccess) { first = r; where = w; } };
        step("memset", hipMemsetAsync(d, 0, 4 * n, st));
        step("h2d", hipMemcpyAsync(d, h, 4 * n, hipMemcpyHostToDevice, st));
        void *args[] = { &d, &n };
        step("launch", hipLaunchKernel((const void *)k_touch, dim3((n + 63) / 64), dim3(64), args, 0, st));
        step("d2h", hipMemcpyAsync(h, d, 4 * n, hipMemcpyDeviceToHost, st));
        e = hipStreamEndCapture(st, &g);
        if (first != hipSuccess || e != hipSuccess) {
            if (first != hipSuccess) note(where, first); else note("end", e);
            if (first != hipSuccess) note("end after a failed step", e);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            // what state is the stream left in, and what brings it back?
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            hipError_t q = hipStreamIsCapturing(st, &cs);
            why[std::string("after: hipStreamIsCapturing -> ") + hipGetErrorName(q) + " status " + std::to_string((int)cs)]++;
            (void)hipGetLastError();
            if (q != hipSuccess || cs != hipStreamCaptureStatusNone) {
                hipGraph_t g2 = nullptr;
                q = hipStreamEndCapture(st, &g2);
                if (g2) (void)hipGraphDestroy(g2);
                hipError_t q2 = hipStreamIsCapturing(st, &cs);
                why[std::string("second hipStreamEndCapture -> ") + hipGetErrorName(q) + ", then status " + hipGetErrorName(q2) + " " + std::to_string((int)cs)]++;
                (void)hipGetLastError();
            }
            q = hipStreamSynchronize(st);
            why[std::string("hipStreamSynchronize after the failure -> ") + hipGetErrorName(q)]++;
            (void)hipGetLastError();
            return false;
        }
        bool good = true;
        if ((e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0)) != hipSuccess) { note("instantiate", e); good = false; }
        if (good && (e = hipGraphLaunch(x, st)) != hipSuccess) { note("graph launch", e); good = false; }
        if ((e = hipStreamSynchronize(st)) != hipSuccess) { note("sync", e); good = false; }
        if (x) (void)hipGraphExecDestroy(x);
        (void)hipGraphDestroy(g);
        if (!good) (void)hipGetLastError();
        return good;
    }
};

struct candidate { const char *name; hipError_t (*op)(void *); };
struct other_side { hipStream_t st; void *d; void *h; };

static hipError_t op_none(void *) { std::this_thread::yield(); return hipSuccess; }
static hipError_t op_device_sync(void *) { return hipDeviceSynchronize(); }
static hipError_t op_device_sync_slow(void *) { std::this_thread::sleep_for(std::chrono::milliseconds(2)); return hipDeviceSynchronize(); }
static hipError_t op_malloc_free(void *) { void *p = nullptr; hipError_t e = hipMalloc(&p, 1 << 20); if (e == hipSuccess) e = hipFree(p); return e; }
static hipError_t op_host_malloc_free(void *) { void *p = nullptr; hipError_t e = hipHostMalloc(&p, 1 << 16, hipHostMallocPortable | hipHostMallocMapped); if (e == hipSuccess) e = hipHostFree(p); return e; }
static hipError_t op_stream_create_destroy(void *) { hipStream_t s; hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking); if (e == hipSuccess) e = hipStreamDestroy(s); return e; }
static hipError_t op_event_create_destroy(void *) { hipEvent_t v; hipError_t e = hipEventCreateWithFlags(&v, hipEventDisableTiming); if (e == hipSuccess) e = hipEventDestroy(v); return e; }
static hipError_t op_memcpy_legacy(void *p) { other_side *o = (other_side *)p; return hipMemcpy(o->d, o->h, 4096, hipMemcpyHostToDevice); }
static hipError_t op_memset_legacy(void *p) { other_side *o = (other_side *)p; return hipMemset(o->d, 0, 4096); }
static hipError_t op_stream_sync_legacy(void *) { return hipStreamSynchronize(nullptr); }
static hipError_t op_async_own_stream(void *p) {
    other_side *o = (other_side *)p;
    hipError_t e = hipMemcpyAsync(o->d, o->h, 4096, hipMemcpyHostToDevice, o->st);
    unsigned n = 1024; void *args[] = { &o->d, &n };
    if (e == hipSuccess) e = hipLaunchKernel((const void *)k_touch, dim3(16), dim3(64), args, 0, o->st);
    if (e == hipSuccess) e = hipStreamSynchronize(o->st);
    return e;
}
static hipError_t op_stream_query_own(void *p) { hipError_t e = hipStreamQuery(((other_side *)p)->st); return e == hipErrorNotReady ? hipSuccess : e; }
static hipError_t op_set_device(void *) { return hipSetDevice(0); }
static hipError_t op_mem_get_info(void *) { size_t a, b; return hipMemGetInfo(&a, &b); }
static hipError_t op_memcpy_to_symbol_async(void *p);
__device__ unsigned g_sym[4];
static hipError_t op_memcpy_to_symbol_async(void *p) {
    other_side *o = (other_side *)p; unsigned v[4] = {1, 2, 3, 4};
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sym), v, sizeof v, 0, hipMemcpyHostToDevice, o->st);
    if (e == hipSuccess) e = hipStreamSynchronize(o->st);
    return e;
}

static int run_candidate(const candidate &c, int ms, bool firstc) {
    if (hipSetDevice(0) != hipSuccess) { fprintf(stderr, "no device\n"); return 2; }
    capture_side A; other_side B;
    const unsigned n = 1 << 16;
    if (hipStreamCreateWithFlags(&A.st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&A.d, 4 * n) != hipSuccess ||
        hipHostMalloc((void **)&A.h, 4 * n, hipHostMallocPortable) != hipSuccess) return 2;
    memset(A.h, 1, 4 * n);
    if (hipStreamCreateWithFlags(&B.st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&B.d, 1 << 16) != hipSuccess ||
        hipHostMalloc(&B.h, 1 << 16, hipHostMallocPortable) != hipSuccess) return 2;
    std::atomic<bool> stop{false};
    size_t calls = 0; std::map<std::string, size_t> berr;
    std::thread tb([&] {
        (void)hipSetDevice(0);
        while (!stop.load(std::memory_order_relaxed)) {
            const hipError_t e = c.op(&B);
            calls++;
            if (e != hipSuccess) { berr[hipGetErrorName(e)]++; (void)hipGetLastError(); }
        }
    });
    const auto t0 = std::chrono::steady_clock::now();
    unsigned r = 0;
    while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() < ms) {
        if (A.round(64 * (1 + (r * 7) % 1023))) A.ok++; else A.failed++;
        r++;
    }
    stop.store(true); tb.join();
    printf("%s {\"other_thread\": \"%s\", \"its_calls\": %zu, \"its_errors\": {", firstc ? " " : ",", c.name, calls);
    bool f = true; for (auto &kv : berr) { printf("%s\"%s\": %zu", f ? "" : ", ", kv.first.c_str(), kv.second); f = false; }
    printf("}, \"captures_ok\": %zu, \"captures_failed\": %zu, \"failures\": {", A.ok, A.failed);
    f = true; for (auto &kv : A.why) { printf("%s\"%s\": %zu", f ? "" : ", ", kv.first.c_str(), kv.second); f = false; }
    printf("}}\n"); fflush(stdout);
    return 0;
}

// capture_probe [ms]            every candidate, each in a process of its own (a call that crashes the runtime is a result)
// capture_probe ms index first  one candidate (what the parent runs)
int main(int argc, char **argv) {
    const int ms = argc > 1 ? atoi(argv[1]) : 1500;
    const candidate cands[] = {
        {"nothing (control)", op_none}, {"hipMalloc + hipFree", op_malloc_free},
        {"hipHostMalloc + hipHostFree", op_host_malloc_free}, {"hipStreamCreate + Destroy", op_stream_create_destroy},
        {"hipEventCreate + Destroy", op_event_create_destroy}, {"hipMemcpy (legacy stream)", op_memcpy_legacy},
        {"hipMemset (legacy stream)", op_memset_legacy}, {"hipStreamSynchronize(NULL)", op_stream_sync_legacy},
        {"async copy + kernel + sync on its own stream", op_async_own_stream}, {"hipStreamQuery (own stream)", op_stream_query_own},
        {"hipSetDevice", op_set_device}, {"hipMemGetInfo", op_mem_get_info}, {"hipMemcpyToSymbolAsync (own stream)", op_memcpy_to_symbol_async},
        {"hipDeviceSynchronize every 2 ms", op_device_sync_slow}, {"hipDeviceSynchronize", op_device_sync},
    };
    const int ncand = (int)(sizeof cands / sizeof cands[0]);
    if (argc > 3) { const int i = atoi(argv[2]); return i >= 0 && i < ncand ? run_candidate(cands[i], ms, atoi(argv[3]) != 0) : 2; }
    int version = 0; (void)hipRuntimeGetVersion(&version);
    printf("{\"hip_runtime_version\": %d, \"ms_per_candidate\": %d, \"capture_mode\": \"hipStreamCaptureModeThreadLocal\", \"candidates\": [\n", version, ms);
    fflush(stdout);
    for (int i = 0; i < ncand; i++) {
        char cmd[512];
        snprintf(cmd, sizeof cmd, "%s %d %d %d", argv[0], ms, i, i == 0);
        const int rc = system(cmd);
        if (rc != 0) {
            printf("%s {\"other_thread\": \"%s\", \"process\": \"died: wait status %d (a crash inside the HIP runtime)\", \"its_calls\": 0, \"its_errors\": {}, "
                   "\"captures_ok\": 0, \"captures_failed\": 0, \"failures\": {}}\n", i ? "," : " ", cands[i].name, rc);
            fflush(stdout);
        }
    }
    printf("]}\n");
    return 0;
}
