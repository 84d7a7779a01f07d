"""ctypes binding of libhs_ed25519.so (the C-ABI in include/hs_ed25519.h).

The library is built in-tree by ``__graft_entry__.build()``.  Loading fails
loudly: there is no CPU path.
"""
import ctypes
import os
import threading

# one hardware queue per stream context (HIP's default of 4 multiplexes the pool / ring slots);
# read by the HIP runtime when it initialises, i.e. at the first device call in the process
# (a process that shares its GPU with co-tenants - HS_ED25519_TENANTS, include/hs_ed25519.h - takes its share of them)
try:
    _tenants = max(1, min(8, int(os.environ.get("HS_ED25519_TENANTS", "1"))))
except ValueError:
    _tenants = 1
def hw_queue_budget(tenants):
    """hs_ed25519_hw_queue_budget(tenants) restated (hs_ed25519.hip: 16 for a process alone; for co-tenants narrow pool +
    wide pool + 1, the process's default stream: 11 / 6 / 4 for 2 / 4 / 8 tenants; tests hold the two equal).
    The HIP runtime reads GPU_MAX_HW_QUEUES when it initialises - in a Python process at torch's first device call - and the
    library itself no longer touches the environment (round 5): the HOST exports the variable before any thread starts.  This
    wrapper is the host here, and import time is its main()."""
    t = max(1, min(8, int(tenants)))
    if t == 1:
        return 16
    parts = max(1, 8 // t)
    return 2 * parts + max(1, 4 // t) + 1


if os.environ.get("HS_ED25519_KEEP_ENV") != "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(hw_queue_budget(_tenants)))

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HS_ED25519_LIB") or os.path.join(_HERE, "libhs_ed25519.so")   # override: A/B measurements

BATCH = 0
STRICT = 1
BATCH_RLC = 2          # dalek's verify_batch as computed: one random linear combination, per-signature fallback
RLC_NOT_RUN = 2        # hs_ed25519_ctx_rlc_last: the RLC stage has not run on this context
EXCHANGE_HOST, EXCHANGE_RCCL = 0, 1
KEYS_REPLICATE, KEYS_BY_OWNER = 0, 1


class EngineError(RuntimeError):
    """Engine-level failure (no GPU, HIP error, bad argument).  Callers must treat
    it as a verification failure - never as success."""


class PoolBusy(EngineError):
    """hs_ed25519_submit: HS_ED25519_E_BUSY - every pool context is in flight; wait on a ticket and retry."""


_lib = None
_lock = threading.Lock()
_inited = False


def lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise EngineError(
                        "libhs_ed25519.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                        % LIB_PATH)
                l = ctypes.CDLL(LIB_PATH)
                vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
                l.hs_ed25519_init.argtypes = [ctypes.c_int, sz]
                l.hs_ed25519_init.restype = ctypes.c_int
                l.hs_ed25519_shutdown.argtypes = []
                l.hs_ed25519_shutdown.restype = None
                l.hs_ed25519_last_error.argtypes = []
                l.hs_ed25519_last_error.restype = ctypes.c_char_p
                l.hs_ed25519_dominant_kernel.argtypes = []
                l.hs_ed25519_dominant_kernel.restype = ctypes.c_char_p
                l.hs_ed25519_dominant_kernel_for.argtypes = [sz]
                l.hs_ed25519_dominant_kernel_for.restype = ctypes.c_char_p
                l.hs_ed25519_cached_kernel_for.argtypes = [sz]
                l.hs_ed25519_cached_kernel_for.restype = ctypes.c_char_p
                l.hs_ed25519_batches_kernel_for.argtypes = [vp, sz]
                l.hs_ed25519_batches_kernel_for.restype = ctypes.c_char_p
                l.hs_ed25519_verify_shared.argtypes = [vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_shared.restype = ctypes.c_int
                l.hs_ed25519_verify_multi.argtypes = [vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_multi.restype = ctypes.c_int
                l.hs_ed25519_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), sz]
                l.hs_ed25519_ctx_create.restype = ctypes.c_int
                l.hs_ed25519_ctx_create_partition.argtypes = [ctypes.POINTER(ctypes.c_void_p), sz, ctypes.c_uint, ctypes.c_uint]
                l.hs_ed25519_ctx_create_partition.restype = ctypes.c_int
                l.hs_ed25519_ctx_stream.argtypes = [vp]
                l.hs_ed25519_ctx_stream.restype = ctypes.c_void_p
                l.hs_ed25519_ctx_destroy.argtypes = [vp]
                l.hs_ed25519_ctx_destroy.restype = None
                l.hs_ed25519_verify_device.argtypes = [vp, vp, sz, vp, vp, sz, u32, vp, vp, vp]
                l.hs_ed25519_verify_device.restype = ctypes.c_int
                l.hs_ed25519_ctx_set_profiling.argtypes = [vp, sz]
                l.hs_ed25519_ctx_set_profiling.restype = ctypes.c_int
                l.hs_ed25519_ctx_set_profiling_stride.argtypes = [vp, sz]
                l.hs_ed25519_ctx_set_profiling_stride.restype = ctypes.c_int
                l.hs_ed25519_ctx_kernel_times.argtypes = [vp, ctypes.POINTER(ctypes.c_double),
                                                          ctypes.POINTER(ctypes.c_double), ctypes.POINTER(sz)]
                l.hs_ed25519_ctx_kernel_times.restype = ctypes.c_int
                u64 = ctypes.c_uint64
                l.hs_ed25519_verify_tc.argtypes = [u64, vp, vp, vp, sz, vp]
                l.hs_ed25519_verify_tc.restype = ctypes.c_int
                l.hs_ed25519_verify_tc_cached.argtypes = [vp, u64, vp, vp, vp, sz, vp]
                l.hs_ed25519_verify_tc_cached.restype = ctypes.c_int
                l.hs_ed25519_verify_device_tc.argtypes = [vp, u64, vp, vp, vp, sz, vp, vp, vp]
                l.hs_ed25519_verify_device_tc.restype = ctypes.c_int
                l.hs_ed25519_verify_device_tc_cached.argtypes = [vp, vp, u64, vp, vp, vp, sz, vp, vp, vp]
                l.hs_ed25519_verify_device_tc_cached.restype = ctypes.c_int
                l.hs_ed25519_verify_device_batches.argtypes = [vp, vp, sz, vp]
                l.hs_ed25519_verify_device_batches.restype = ctypes.c_int
                l.hs_ed25519_verify_device_batches_cached.argtypes = [vp, vp, vp, sz, vp]
                l.hs_ed25519_verify_device_batches_cached.restype = ctypes.c_int
                l.hs_ed25519_submit.argtypes = [ctypes.POINTER(ctypes.c_void_p), vp, vp, sz, vp, vp, sz, u32, u32]
                l.hs_ed25519_submit.restype = ctypes.c_int
                l.hs_ed25519_ticket_ready.argtypes = [vp]
                l.hs_ed25519_ticket_ready.restype = ctypes.c_int
                l.hs_ed25519_ticket_device_slot.argtypes = [vp]
                l.hs_ed25519_ticket_device_slot.restype = ctypes.c_int
                l.hs_ed25519_ticket_wait.argtypes = [vp, vp]
                l.hs_ed25519_ticket_wait.restype = ctypes.c_int
                l.hs_ed25519_set_tenancy.argtypes = [ctypes.c_uint, ctypes.c_uint]
                l.hs_ed25519_set_tenancy.restype = ctypes.c_int
                l.hs_ed25519_host_alloc.argtypes = [sz]
                l.hs_ed25519_host_alloc.restype = ctypes.c_void_p
                l.hs_ed25519_host_free.argtypes = [vp]
                l.hs_ed25519_host_free.restype = None
                l.hs_ed25519_shard_rccl_ranks.argtypes = []
                l.hs_ed25519_shard_rccl_ranks.restype = ctypes.c_int
                l.hs_ed25519_verify_wire_shared.argtypes = [vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_wire_shared.restype = ctypes.c_int
                l.hs_ed25519_verify_wire_tc.argtypes = [u64, vp, sz, vp]
                l.hs_ed25519_verify_wire_tc.restype = ctypes.c_int
                l.hs_ed25519_wire_decode_device.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
                l.hs_ed25519_wire_decode_device.restype = ctypes.c_int
                l.hs_ed25519_sign_batch.argtypes = [vp, vp, sz, sz, vp]
                l.hs_ed25519_sign_batch.restype = ctypes.c_int
                l.hs_ed25519_sign_device.argtypes = [vp, vp, vp, sz, sz, vp, vp]
                l.hs_ed25519_sign_device.restype = ctypes.c_int
                l.hs_ed25519_public_keys.argtypes = [vp, sz, vp]
                l.hs_ed25519_public_keys.restype = ctypes.c_int
                l.hs_ed25519_ctx_set_kernel.argtypes = [vp, ctypes.c_int]
                l.hs_ed25519_ctx_set_kernel.restype = ctypes.c_int
                l.hs_ed25519_ctx_set_concurrent.argtypes = [vp, ctypes.c_int]
                l.hs_ed25519_ctx_set_concurrent.restype = ctypes.c_int
                l.hs_ed25519_init_devices.argtypes = [ctypes.POINTER(ctypes.c_int), sz, sz]
                l.hs_ed25519_init_devices.restype = ctypes.c_int
                l.hs_ed25519_device_count.argtypes = []
                l.hs_ed25519_device_count.restype = ctypes.c_int
                l.hs_ed25519_min_gpu_batch.argtypes = [ctypes.c_int]
                l.hs_ed25519_min_gpu_batch.restype = sz
                l.hs_ed25519_set_rlc_key.argtypes = [vp]
                l.hs_ed25519_set_rlc_key.restype = ctypes.c_int
                l.hs_ed25519_rlc_min_batch.argtypes = []
                l.hs_ed25519_rlc_min_batch.restype = sz
                l.hs_ed25519_ctx_create_on.argtypes = [ctypes.POINTER(ctypes.c_void_p), sz, ctypes.c_int]
                l.hs_ed25519_ctx_create_on.restype = ctypes.c_int
                l.hs_ed25519_ctx_device.argtypes = [vp]
                l.hs_ed25519_ctx_device.restype = ctypes.c_int
                l.hs_ed25519_ctx_set_rlc_min.argtypes = [vp, sz]
                l.hs_ed25519_ctx_set_rlc_min.restype = ctypes.c_int
                l.hs_ed25519_ctx_rlc_last.argtypes = [vp, ctypes.POINTER(sz)]
                l.hs_ed25519_ctx_rlc_last.restype = ctypes.c_int
                l.hs_ed25519_keyset_create_on.argtypes = [ctypes.POINTER(ctypes.c_void_p), vp, sz, ctypes.c_int]
                l.hs_ed25519_keyset_create_on.restype = ctypes.c_int
                l.hs_ed25519_shard_plan.argtypes = [sz, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(sz), ctypes.POINTER(sz)]
                l.hs_ed25519_shard_plan.restype = ctypes.c_int
                l.hs_ed25519_set_shard_exchange.argtypes = [ctypes.c_int]
                l.hs_ed25519_set_shard_exchange.restype = ctypes.c_int
                l.hs_ed25519_shard_reset.argtypes = []
                l.hs_ed25519_shard_reset.restype = None
                l.hs_ed25519_verify_shared_sharded.argtypes = [vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_shared_sharded.restype = ctypes.c_int
                l.hs_ed25519_verify_multi_sharded.argtypes = [vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_multi_sharded.restype = ctypes.c_int
                l.hs_ed25519_keyset_sharded_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), vp, sz, ctypes.c_int]
                l.hs_ed25519_keyset_sharded_create.restype = ctypes.c_int
                l.hs_ed25519_keyset_sharded_destroy.argtypes = [vp]
                l.hs_ed25519_keyset_sharded_destroy.restype = None
                l.hs_ed25519_keyset_sharded_bytes_per_device.argtypes = [vp, ctypes.c_int]
                l.hs_ed25519_keyset_sharded_bytes_per_device.restype = sz
                l.hs_ed25519_verify_shared_sharded_cached.argtypes = [vp, vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_shared_sharded_cached.restype = ctypes.c_int
                l.hs_ed25519_verify_multi_sharded_cached.argtypes = [vp, vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_multi_sharded_cached.restype = ctypes.c_int
                l.hs_ed25519_keyset_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), vp, sz]
                l.hs_ed25519_keyset_create.restype = ctypes.c_int
                l.hs_ed25519_keyset_destroy.argtypes = [vp]
                l.hs_ed25519_keyset_destroy.restype = None
                l.hs_ed25519_keyset_size.argtypes = [vp]
                l.hs_ed25519_keyset_size.restype = sz
                l.hs_ed25519_keyset_bytes.argtypes = [sz]
                l.hs_ed25519_keyset_bytes.restype = sz
                l.hs_ed25519_verify_shared_cached.argtypes = [vp, vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_shared_cached.restype = ctypes.c_int
                l.hs_ed25519_verify_multi_cached.argtypes = [vp, vp, vp, vp, sz, u32, vp]
                l.hs_ed25519_verify_multi_cached.restype = ctypes.c_int
                l.hs_ed25519_verify_device_cached.argtypes = [vp, vp, vp, sz, vp, vp, sz, u32, vp, vp, vp]
                l.hs_ed25519_verify_device_cached.restype = ctypes.c_int
                pp = ctypes.POINTER(ctypes.c_void_p)
                l.hs_ed25519_ring_create.argtypes = [pp, sz, ctypes.c_uint, vp, u32]
                l.hs_ed25519_ring_create.restype = ctypes.c_int
                l.hs_ed25519_ring_destroy.argtypes = [vp]
                l.hs_ed25519_ring_destroy.restype = None
                l.hs_ed25519_ring_begin.argtypes = [vp, ctypes.c_uint, sz, sz, pp, pp, pp]
                l.hs_ed25519_ring_begin.restype = ctypes.c_int
                l.hs_ed25519_ring_submit.argtypes = [vp, ctypes.c_uint, u32]
                l.hs_ed25519_ring_submit.restype = ctypes.c_int
                l.hs_ed25519_ring_poll.argtypes = [vp, ctypes.c_uint]
                l.hs_ed25519_ring_poll.restype = ctypes.c_int
                l.hs_ed25519_ring_wait.argtypes = [vp, ctypes.c_uint, vp]
                l.hs_ed25519_ring_wait.restype = ctypes.c_int
                l.hs_ed25519_ring_stats.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(sz)]
                l.hs_ed25519_ring_stats.restype = ctypes.c_int
                l.hs_ed25519_shard_msm_min.restype = sz
                l.hs_ed25519_set_shard_msm_min.argtypes = [sz]
                l.hs_ed25519_shard_msm_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
                l.hs_ed25519_shard_rccl_fallbacks.argtypes = []
                l.hs_ed25519_shard_rccl_fallbacks.restype = ctypes.c_ulonglong
                l.hs_ed25519_hw_queue_budget.argtypes = [ctypes.c_uint]
                l.hs_ed25519_hw_queue_budget.restype = ctypes.c_uint
                l.hs_ed25519_abi_version.argtypes = []
                l.hs_ed25519_abi_version.restype = ctypes.c_uint32
                l.hs_ed25519_config_dump.argtypes = [ctypes.c_char_p, sz]
                l.hs_ed25519_config_dump.restype = sz
                l.hs_ed25519_autotune.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
                l.hs_ed25519_autotune.restype = ctypes.c_int
                l.hs_ed25519_ring_slot_ctx.argtypes = [vp, ctypes.c_uint]
                l.hs_ed25519_ring_slot_ctx.restype = vp
                _lib = l
    return _lib


def last_error():
    return lib().hs_ed25519_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc < 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, last_error()))
    return rc


def init(device=None, max_batch=8192):
    """hs_ed25519_init.  device defaults to LOCAL_RANK (one process per GPU) or 0."""
    global _inited
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(lib().hs_ed25519_init(int(device), int(max_batch)), "hs_ed25519_init")
    _inited = True


def init_devices(ordinals, max_batch=8192):
    """hs_ed25519_init_devices: several GPUs behind this one process; entry s of `ordinals` becomes
    device slot s (slot 0 = the primary device).  The same ordinal may repeat (virtual devices)."""
    global _inited
    arr = (ctypes.c_int * len(ordinals))(*[int(o) for o in ordinals])
    check(lib().hs_ed25519_init_devices(arr, len(ordinals), int(max_batch)), "hs_ed25519_init_devices")
    _inited = True


def shutdown():
    global _inited
    lib().hs_ed25519_shard_reset()
    lib().hs_ed25519_shutdown()
    _inited = False


def device_count():
    return lib().hs_ed25519_device_count()


def shard_plan(n, nshards, shard):
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib().hs_ed25519_shard_plan(int(n), int(nshards), int(shard), ctypes.byref(lo), ctypes.byref(hi)), "hs_ed25519_shard_plan")
    return lo.value, hi.value


def set_tenancy(tenants, index):
    """This process is tenant `index` of `tenants` on its GPU (before init; include/hs_ed25519.h)."""
    check(lib().hs_ed25519_set_tenancy(int(tenants), int(index)), "hs_ed25519_set_tenancy")


class PinnedBuffer:
    """hs_ed25519_host_alloc memory holding `data`: host calls on buffers that all live in such memory are read in
    place by the kernels (no staging copy).  .addr is what to pass as a pointer."""

    def __init__(self, data):
        ensure_init()
        self.n = len(data)
        self.addr = lib().hs_ed25519_host_alloc(max(1, self.n))
        if not self.addr:
            raise EngineError("hs_ed25519_host_alloc failed: " + last_error())
        ctypes.memmove(self.addr, bytes(data), self.n)

    def close(self):
        if getattr(self, "addr", None):
            lib().hs_ed25519_host_free(self.addr)
            self.addr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_loadgen = None


def _loadgen_lib():
    global _loadgen
    if _loadgen is None:
        path = os.path.join(os.path.dirname(_HERE), "tools", "libhs_loadgen.so")
        if not os.path.exists(path):
            raise EngineError("tools/libhs_loadgen.so is missing: run __graft_entry__.build()")
        lib()                                  # the product library first: the tool resolves its symbols against it
        g = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        g.hs_loadgen.argtypes = [vp, vp, sz, vp, vp, sz, sz, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_double,
                                 ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        g.hs_loadgen.restype = ctypes.c_int
        _loadgen = g
    return _loadgen


def loadgen(ks, digests, digest_stride, pks, sigs, n, nbatches, mode, threads, inflight, batches_per_launch, seconds):
    """tools/libhs_loadgen.so (a measurement aid over the public C ABI, not part of the product library): native
    host threads drive the host-pointer API.  Pointers are raw addresses.  Returns (calls, p50_ms, elapsed_s)."""
    _loadgen_lib()
    calls, p50, el = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
    rc = _loadgen.hs_loadgen(ks, digests, digest_stride, pks, sigs, n, nbatches, mode, threads, inflight, batches_per_launch, seconds,
                             ctypes.byref(calls), ctypes.byref(p50), ctypes.byref(el))
    if rc != 0:
        raise EngineError("hs_loadgen failed (%d): %s" % (rc, last_error()))
    return calls.value, p50.value, el.value


def callers_bench(validators=10000, timeouts=200):
    """tools/libhs_loadgen.so hs_callers_bench: the C++ twin's VoteIngest and VerifiedQCCache at N validators (a measurement
    aid like loadgen).  Returns a dict of milliseconds and counts."""
    loadgen_lib = _loadgen_lib()
    out = (ctypes.c_double * 6)()
    loadgen_lib.hs_callers_bench.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_double)]
    loadgen_lib.hs_callers_bench.restype = ctypes.c_int
    rc = loadgen_lib.hs_callers_bench(int(validators), int(timeouts), out)
    if rc != 0:
        raise EngineError("hs_callers_bench failed (%d): %s" % (rc, last_error()))
    return {"vote_ingest_ms": out[0], "vote_ingest_engine_calls": int(out[1]), "timeout_storm_ms": out[2], "qc_cache_misses": int(out[3]),
            "qc_cache_hits": int(out[4]), "qc_verify_ms_uncached": out[5]}


def set_rlc_key(key):
    """Pin the coefficient key of BATCH_RLC (reproducible runs); None = back to a fresh OS secret + per-call counters."""
    if key is None:
        check(lib().hs_ed25519_set_rlc_key(None), "hs_ed25519_set_rlc_key")
        return
    buf, keep = _buf(key)
    check(lib().hs_ed25519_set_rlc_key(buf), "hs_ed25519_set_rlc_key")


def shard_msm_stats():
    """(sharded BATCH_RLC calls that ran the one-combination stage, how many of them it accepted outright)"""
    a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    check(lib().hs_ed25519_shard_msm_stats(ctypes.byref(a), ctypes.byref(b)), "hs_ed25519_shard_msm_stats")
    return a.value, b.value


def shard_rccl_fallbacks():
    """sharded calls that asked for the RCCL exchange and went through pinned host memory because it could not be set up"""
    return int(lib().hs_ed25519_shard_rccl_fallbacks())


def verify_packed_sharded(digests, pks, sigs, n, mode, shared, want_bitmap=False):
    """hs_ed25519_verify_shared_sharded / _multi_sharded: one call over every device slot."""
    n = int(n)
    if len(pks) != 32 * n or len(sigs) != 64 * n or len(digests) != (32 if shared else 32 * n):
        raise ValueError("packed buffer sizes do not match n=%d" % n)
    ensure_init()
    if n == 0:
        return True, (b"" if want_bitmap else None)
    bm = (ctypes.c_ubyte * ((n + 7) // 8))() if want_bitmap else None
    pd, k1 = _buf(digests)
    pp, k2 = _buf(pks)
    ps, k3 = _buf(sigs)
    fn = lib().hs_ed25519_verify_shared_sharded if shared else lib().hs_ed25519_verify_multi_sharded
    rc = check(fn(pd, pp, ps, n, mode, ctypes.cast(bm, ctypes.c_void_p) if bm is not None else None),
               "hs_ed25519_verify_sharded")
    del k1, k2, k3
    return rc == 1, (bytes(bm) if bm is not None else None)


class ShardedKeySet:
    """hs_ed25519_keyset_sharded: the committee key cache over every device slot, replicated or
    partitioned by key owner (include/hs_ed25519.h)."""

    UNKNOWN = 0xFFFFFFFF

    def __init__(self, pks, placement=KEYS_REPLICATE):
        ensure_init()
        pks = [bytes(p) for p in pks]
        self.index = {}
        for i, p in enumerate(pks):
            self.index.setdefault(p, i)
        self.n = len(pks)
        self._h = ctypes.c_void_p()
        buf, keep = _buf(b"".join(pks))
        check(lib().hs_ed25519_keyset_sharded_create(ctypes.byref(self._h), buf, self.n, int(placement)),
              "hs_ed25519_keyset_sharded_create")
        del keep

    def bytes_per_device(self, slot):
        return lib().hs_ed25519_keyset_sharded_bytes_per_device(self._h, int(slot))

    def verify_packed(self, digests, key_idx_packed, sigs, n, mode, shared, want_bitmap=False):
        n = int(n)
        if n == 0:
            return True, (b"" if want_bitmap else None)
        bm = (ctypes.c_ubyte * ((n + 7) // 8))() if want_bitmap else None
        pd, k1 = _buf(digests)
        pi, k2 = _buf(key_idx_packed)
        ps, k3 = _buf(sigs)
        fn = lib().hs_ed25519_verify_shared_sharded_cached if shared else lib().hs_ed25519_verify_multi_sharded_cached
        rc = check(fn(self._h, pd, pi, ps, n, mode, ctypes.cast(bm, ctypes.c_void_p) if bm is not None else None),
                   "hs_ed25519_verify_sharded_cached")
        del k1, k2, k3
        return rc == 1, (bytes(bm) if bm is not None else None)

    def close(self):
        if getattr(self, "_h", None):
            lib().hs_ed25519_keyset_sharded_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ensure_init():
    if not _inited:
        init()


# DeviceContext.set_kernel: 0 = by batch size; 1 = one lane per signature (cold: k_prepare + k_verify;
# key-cache calls: as 3); 3 = fused / two-lane kernels at any size; 4 = the pairs form of the fused kernel
# in verify_batches (two signatures per quad; elsewhere it behaves as 3).
KERNEL_AUTO, KERNEL_LANE, KERNEL_FUSED, KERNEL_FUSED_PAIRS = 0, 1, 3, 4
KERNEL_CACHED_C = 5          # key cache: chains + batched finish by compression at any size (cold calls: as KERNEL_AUTO)
KERNEL_OCT, KERNEL_OCT_SPLIT = 6, 7   # cold single calls of <= 8 signatures per SIMD: k_verify_fused8 / k_verify_fused8h whatever the autotune chose
ABI_VERSION = 0x00050000     # HS_ED25519_ABI_VERSION of include/hs_ed25519_core.h this binding was written against (test: == the header's)


def abi_version():
    """hs_ed25519_abi_version(): what the loaded library was built from (no GPU needed)."""
    return int(lib().hs_ed25519_abi_version())


def config_dump():
    """hs_ed25519_config_dump() parsed: every value the library resolved at init (pools, thresholds, env overrides, autotune)."""
    import json
    l = lib()
    need = l.hs_ed25519_config_dump(None, 0)
    buf = ctypes.create_string_buffer(need + 1)
    l.hs_ed25519_config_dump(buf, need + 1)
    return json.loads(buf.value.decode())


def autotune(slot=0):
    """hs_ed25519_autotune(slot): (pick, {"k_verify_fused8": ms, "k_verify_fused8h": ms}) measured now on that device slot."""
    ensure_init()
    ms = (ctypes.c_double * 2)()
    rc = lib().hs_ed25519_autotune(int(slot), ms)
    if rc < 0:
        raise EngineError("hs_ed25519_autotune: %s" % last_error())
    return ("k_verify_fused8h" if rc else "k_verify_fused8"), {"k_verify_fused8": ms[0], "k_verify_fused8h": ms[1]}


def _buf(b):
    """bytes-like -> (c_void_p, keepalive object)."""
    mv = memoryview(b).cast("B")
    arr = (ctypes.c_char * max(1, len(mv))).from_buffer_copy(bytes(mv) or b"\0")
    return ctypes.cast(arr, ctypes.c_void_p), arr


def verify_packed(digests, pks, sigs, n, mode, shared, want_bitmap=False):
    """hs_ed25519_verify_shared / _multi on packed byte strings.
    Returns (all_ok: bool, bitmap: bytes | None)."""
    n = int(n)
    if len(pks) != 32 * n or len(sigs) != 64 * n or len(digests) != (32 if shared else 32 * n):
        raise ValueError("packed buffer sizes do not match n=%d" % n)
    ensure_init()
    if n == 0:
        return True, (b"" if want_bitmap else None)
    bm = (ctypes.c_ubyte * ((n + 7) // 8))() if want_bitmap else None
    pd, k1 = _buf(digests)
    pp, k2 = _buf(pks)
    ps, k3 = _buf(sigs)
    fn = lib().hs_ed25519_verify_shared if shared else lib().hs_ed25519_verify_multi
    rc = check(fn(pd, pp, ps, n, mode, ctypes.cast(bm, ctypes.c_void_p) if bm is not None else None),
               "hs_ed25519_verify")
    del k1, k2, k3
    return rc == 1, (bytes(bm) if bm is not None else None)


def verify_tc_packed(round_, hq_rounds, pks, sigs, keyset=None):
    """hs_ed25519_verify_tc[_cached]: TC::verify with the per-vote digests
    SHA-512(round || high_qc_round)[..32] built on the device.  Returns (all_ok, bitmap)."""
    import struct
    n = len(hq_rounds)
    if len(sigs) != 64 * n:
        raise ValueError("packed buffer sizes do not match n=%d" % n)
    ensure_init()
    if n == 0:
        return True, b""
    bm = (ctypes.c_ubyte * ((n + 7) // 8))()
    ph, k1 = _buf(struct.pack("<%dQ" % n, *hq_rounds))
    ps, k2 = _buf(sigs)
    if keyset is None:
        if len(pks) != 32 * n:
            raise ValueError("pks must be n*32 bytes")
        pp, k3 = _buf(pks)
        rc = lib().hs_ed25519_verify_tc(int(round_), ph, pp, ps, n, ctypes.cast(bm, ctypes.c_void_p))
    else:
        packed, _ = keyset.indices([pks[32 * i:32 * i + 32] for i in range(n)] if isinstance(pks, (bytes, bytearray)) else pks)
        pp, k3 = _buf(packed)
        rc = lib().hs_ed25519_verify_tc_cached(keyset.handle, int(round_), ph, pp, ps, n, ctypes.cast(bm, ctypes.c_void_p))
    check(rc, "hs_ed25519_verify_tc")
    del k1, k2, k3
    return rc == 1, bytes(bm)


def verify_wire(frames, n, digest=None, mode=BATCH, tc_round=None):
    """hs_ed25519_verify_wire_shared / _tc: QC (116-byte) or TC (124-byte) vote elements as they sit in the
    received bincode frame.  Returns (all_ok, bitmap)."""
    ensure_init()
    if n == 0:
        return True, b""
    bm = (ctypes.c_ubyte * ((n + 7) // 8))()
    pf, k1 = _buf(frames)
    if tc_round is None:
        pd, k2 = _buf(digest)
        rc = lib().hs_ed25519_verify_wire_shared(pd, pf, n, mode, ctypes.cast(bm, ctypes.c_void_p))
    else:
        rc = lib().hs_ed25519_verify_wire_tc(int(tc_round), pf, n, ctypes.cast(bm, ctypes.c_void_p))
    check(rc, "hs_ed25519_verify_wire")
    return rc == 1, bytes(bm)


def sign_batch(secrets, digests, n, shared):
    """hs_ed25519_sign_batch: n x 64-byte secrets (seed || pk), digests n x 32 or one -> n x 64 signature bytes."""
    ensure_init()
    out = (ctypes.c_ubyte * (64 * max(1, n)))()
    ps, k1 = _buf(secrets)
    pd, k2 = _buf(digests)
    check(lib().hs_ed25519_sign_batch(ps, pd, 0 if shared else 32, n, ctypes.cast(out, ctypes.c_void_p)), "hs_ed25519_sign_batch")
    return bytes(out)[:64 * n]


def public_keys(seeds, n):
    ensure_init()
    out = (ctypes.c_ubyte * (32 * max(1, n)))()
    ps, k1 = _buf(seeds)
    check(lib().hs_ed25519_public_keys(ps, n, ctypes.cast(out, ctypes.c_void_p)), "hs_ed25519_public_keys")
    return bytes(out)[:32 * n]


class Batch(ctypes.Structure):
    """hs_ed25519_batch: one entry of hs_ed25519_verify_device_batches (device pointers)."""
    _fields_ = [("d_digests", ctypes.c_void_p), ("digest_stride", ctypes.c_size_t), ("d_pks", ctypes.c_void_p),
                ("d_sigs", ctypes.c_void_p), ("n", ctypes.c_size_t), ("mode", ctypes.c_uint32),
                ("d_bitmap", ctypes.c_void_p), ("d_nreject", ctypes.c_void_p)]


def batch_array(entries):
    """[(d_digests, stride, d_pks, d_sigs, n, mode, d_bitmap, d_nreject), ...] -> ctypes array of Batch"""
    arr = (Batch * len(entries))()
    for a, e in zip(arr, entries):
        a.d_digests, a.digest_stride, a.d_pks, a.d_sigs, a.n, a.mode, a.d_bitmap, a.d_nreject = e
    return arr


SUBMIT_MORE = 1
E_BUSY = -5


class Ticket:
    """hs_ed25519_submit / hs_ed25519_ticket_wait: an asynchronous host-pointer verification.  `keys` = packed key
    bytes, or (with `keyset`) a sequence of key indices.  Raises PoolBusy when every pool context is in flight."""

    def __init__(self, digests, keys, sigs, n, mode=STRICT, shared=False, keyset=None, more=False):
        ensure_init()
        if keyset is not None and not isinstance(keys, (bytes, bytearray)):
            import struct
            keys = struct.pack("<%dI" % n, *keys)
        self.n = n
        self._keyset = keyset           # keeps the key set alive as long as the ticket (hs_ed25519_keyset_destroy waits for uncollected tickets)
        self._h = ctypes.c_void_p()
        rc = lib().hs_ed25519_submit(ctypes.byref(self._h), keyset.handle if keyset is not None else None, bytes(digests),
                                     0 if shared else 32, bytes(keys), bytes(sigs), n, mode, SUBMIT_MORE if more else 0)
        if rc == E_BUSY:
            raise PoolBusy("every pool context is in flight: wait on a ticket first")
        check(rc, "hs_ed25519_submit")

    def ready(self):
        if not self._h:
            raise EngineError("ticket already waited on")
        rc = lib().hs_ed25519_ticket_ready(self._h)
        check(rc, "hs_ed25519_ticket_ready")
        return bool(rc)

    def device_slot(self):
        return check(lib().hs_ed25519_ticket_device_slot(self._h), "hs_ed25519_ticket_device_slot")

    def wait(self):
        """-> (verdict: bool, accept bitmap bytes); consumes the ticket"""
        if not self._h:
            raise EngineError("ticket already waited on")
        bm = ctypes.create_string_buffer(max(1, (self.n + 7) // 8))
        h, self._h = self._h, ctypes.c_void_p()
        rc = lib().hs_ed25519_ticket_wait(h, bm)
        check(rc, "hs_ed25519_ticket_wait")
        return rc == 1, bm.raw[:(self.n + 7) // 8]

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().hs_ed25519_ticket_wait(self._h, None)
            except Exception:
                pass


class KeySet:
    """hs_ed25519_keyset: the committee key cache.  Built once per epoch from the committee's
    public keys (consensus/src/config.rs:39-43); per-key window tables live in HBM and every
    later verification of those keys skips the doublings and both decompressions.
    `index` maps key bytes -> position, the lookup the Rust shim does next to the stake map."""

    UNKNOWN = 0xFFFFFFFF

    def __init__(self, pks, slot=0):
        ensure_init()
        pks = [bytes(p) for p in pks]
        if not pks or any(len(p) != 32 for p in pks):
            raise ValueError("KeySet needs a non-empty list of 32-byte keys")
        self.index = {}
        for i, p in enumerate(pks):
            self.index.setdefault(p, i)
        self.n = len(pks)
        self._h = ctypes.c_void_p()
        buf, keep = _buf(b"".join(pks))
        check(lib().hs_ed25519_keyset_create_on(ctypes.byref(self._h), buf, self.n, int(slot)), "hs_ed25519_keyset_create")
        del keep

    @property
    def handle(self):
        return self._h

    def hbm_bytes(self):
        return lib().hs_ed25519_keyset_bytes(self.n)

    def indices(self, pks):
        """uint32 little-endian packed indices for a sequence of key byte strings."""
        import struct
        idx = [self.index.get(bytes(p), self.UNKNOWN) for p in pks]
        return struct.pack("<%dI" % len(idx), *idx), idx

    def verify_packed(self, digests, key_idx_packed, sigs, n, mode, shared, want_bitmap=False):
        n = int(n)
        if len(key_idx_packed) != 4 * n or len(sigs) != 64 * n or len(digests) != (32 if shared else 32 * n):
            raise ValueError("packed buffer sizes do not match n=%d" % n)
        if n == 0:
            return True, (b"" if want_bitmap else None)
        bm = (ctypes.c_ubyte * ((n + 7) // 8))() if want_bitmap else None
        pd, k1 = _buf(digests)
        pi, k2 = _buf(key_idx_packed)
        ps, k3 = _buf(sigs)
        fn = lib().hs_ed25519_verify_shared_cached if shared else lib().hs_ed25519_verify_multi_cached
        rc = check(fn(self._h, pd, pi, ps, n, mode, ctypes.cast(bm, ctypes.c_void_p) if bm is not None else None),
                   "hs_ed25519_verify_cached")
        del k1, k2, k3
        return rc == 1, (bytes(bm) if bm is not None else None)

    def close(self):
        if getattr(self, "_h", None):
            lib().hs_ed25519_keyset_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceContext:
    """hs_ed25519_ctx: workspace + stream for device-resident batches.  Pointers are raw
    HIP device addresses (e.g. ``torch.Tensor.data_ptr()``)."""

    def __init__(self, max_batch=1024, partition=None, slot=0):
        """partition = (part, nparts): the context's own stream only runs on that slice of the
        chip's CUs (pass stream=0 to verify() to use it).  slot: device slot (hs_ed25519_init_devices)."""
        ensure_init()
        self._h = ctypes.c_void_p()
        if partition is None:
            check(lib().hs_ed25519_ctx_create_on(ctypes.byref(self._h), int(max_batch), int(slot)), "hs_ed25519_ctx_create")
        else:
            check(lib().hs_ed25519_ctx_create_partition(ctypes.byref(self._h), int(max_batch), int(partition[0]),
                                                        int(partition[1])), "hs_ed25519_ctx_create_partition")

    @property
    def stream(self):
        return lib().hs_ed25519_ctx_stream(self._h)

    def verify(self, d_digests, digest_stride, d_pks, d_sigs, n, mode, d_bitmap, d_nreject=0, stream=0):
        check(lib().hs_ed25519_verify_device(self._h, d_digests, digest_stride, d_pks, d_sigs, int(n), mode,
                                             d_bitmap, d_nreject or None, stream or None),
              "hs_ed25519_verify_device")

    def verify_batches(self, batches, stream=0):
        """hs_ed25519_verify_device_batches: `batches` = a ctypes array from batch_array()."""
        check(lib().hs_ed25519_verify_device_batches(self._h, ctypes.cast(batches, ctypes.c_void_p), len(batches), stream or None),
              "hs_ed25519_verify_device_batches")

    def verify_batches_cached(self, keyset, batches, stream=0):
        """hs_ed25519_verify_device_batches_cached: d_pks of every entry = the batch's uint32 key indices."""
        check(lib().hs_ed25519_verify_device_batches_cached(self._h, keyset.handle, ctypes.cast(batches, ctypes.c_void_p), len(batches),
                                                            stream or None), "hs_ed25519_verify_device_batches_cached")

    def verify_cached(self, keyset, d_digests, digest_stride, d_key_idx, d_sigs, n, mode, d_bitmap, d_nreject=0, stream=0):
        check(lib().hs_ed25519_verify_device_cached(self._h, keyset.handle, d_digests, digest_stride, d_key_idx, d_sigs,
                                                    int(n), mode, d_bitmap, d_nreject or None, stream or None),
              "hs_ed25519_verify_device_cached")

    def verify_tc(self, round_, d_hq_rounds, d_pks, d_sigs, n, d_bitmap, d_nreject=0, stream=0, keyset=None, d_key_idx=0):
        """hs_ed25519_verify_device_tc[_cached]: TC::verify shape, per-vote digests built on the device."""
        if keyset is None:
            check(lib().hs_ed25519_verify_device_tc(self._h, int(round_), d_hq_rounds, d_pks, d_sigs, int(n), d_bitmap,
                                                    d_nreject or None, stream or None), "hs_ed25519_verify_device_tc")
        else:
            check(lib().hs_ed25519_verify_device_tc_cached(self._h, keyset.handle, int(round_), d_hq_rounds, d_key_idx, d_sigs,
                                                           int(n), d_bitmap, d_nreject or None, stream or None),
                  "hs_ed25519_verify_device_tc_cached")

    def set_kernel(self, which):
        """hs_ed25519_ctx_set_kernel; returns the previous choice."""
        return check(lib().hs_ed25519_ctx_set_kernel(self._h, int(which)), "hs_ed25519_ctx_set_kernel")

    def set_concurrent(self, on=True):
        """hs_ed25519_ctx_set_concurrent: this context is one of several the caller keeps busy on the GPU - throughput
        forms (four lanes per signature, no side stream) instead of the lone call's latency forms.  Returns the previous setting."""
        return check(lib().hs_ed25519_ctx_set_concurrent(self._h, 1 if on else 0), "hs_ed25519_ctx_set_concurrent")

    def set_rlc_min(self, n):
        """Run the RLC stage of BATCH_RLC calls from n signatures up on this context (0 = default)."""
        check(lib().hs_ed25519_ctx_set_rlc_min(self._h, int(n)), "hs_ed25519_ctx_set_rlc_min")

    def rlc_last(self):
        """(verdict of the RLC stage of the last BATCH_RLC call: 1 accepted outright, 0 fell through to the
        per-signature kernels, RLC_NOT_RUN (2) not run; calls that ran it).  Synchronises the context's stream."""
        c = ctypes.c_size_t()
        r = check(lib().hs_ed25519_ctx_rlc_last(self._h, ctypes.byref(c)), "hs_ed25519_ctx_rlc_last")
        return r, c.value

    def set_profiling(self, slots, every=1):
        check(lib().hs_ed25519_ctx_set_profiling(self._h, int(slots)), "hs_ed25519_ctx_set_profiling")
        check(lib().hs_ed25519_ctx_set_profiling_stride(self._h, int(every)), "hs_ed25519_ctx_set_profiling_stride")

    def kernel_times(self):
        """(prepare_ms_sum, verify_ms_sum, launches) over the recorded calls; synchronise first."""
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_size_t()
        check(lib().hs_ed25519_ctx_kernel_times(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
              "hs_ed25519_ctx_kernel_times")
        return a.value, b.value, c.value

    def close(self):
        if getattr(self, "_h", None):
            lib().hs_ed25519_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


RING_KINDS = {"auto": 0, "graph": 1, "stream-ops": 2, "zero-copy": 3}


class RoundRing:
    """hs_ed25519_ring: streamed rounds (BASELINE config 5).  `slots` rounds in flight on their own
    streams; kind = "graph" (pinned H2D -> kernels -> D2H as one hipGraph launch), "stream-ops"
    (the same, issued one by one), "zero-copy" (kernels read the pinned slot directly) or "auto".

        dg, keys, sg = ring.begin(slot, n, shared=True)   # numpy uint8 views of PINNED memory
        dg[:] = ...; keys[:] = ...; sg[:] = ...            # write the packed tuples in place
        ring.submit(slot, mode)                            # asynchronous
        ok, bitmap = ring.wait(slot)

    With a key set `keys` is n little-endian uint32 indices (4n bytes), else n*32 key bytes."""

    def __init__(self, slot_batch, slots=4, keyset=None, kind="auto"):
        ensure_init()
        self._h = ctypes.c_void_p()
        self.keyset = keyset            # keeps the key set alive as long as the ring
        self.slot_batch, self.slots = int(slot_batch), int(slots)
        self._n = [0] * self.slots
        check(lib().hs_ed25519_ring_create(ctypes.byref(self._h), self.slot_batch, self.slots,
                                           keyset.handle if keyset is not None else None,
                                           RING_KINDS[kind]), "hs_ed25519_ring_create")

    def begin(self, slot, n, shared):
        import numpy as np
        d, k, s = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        check(lib().hs_ed25519_ring_begin(self._h, int(slot), int(n), 0 if shared else 32, ctypes.byref(d),
                                          ctypes.byref(k), ctypes.byref(s)), "hs_ed25519_ring_begin")
        self._n[slot] = int(n)

        def view(p, nbytes):
            return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_ubyte)), shape=(nbytes,))
        return (view(d, 32 if shared else 32 * n), view(k, (4 if self.keyset is not None else 32) * n), view(s, 64 * n))

    def submit(self, slot, mode):
        check(lib().hs_ed25519_ring_submit(self._h, int(slot), int(mode)), "hs_ed25519_ring_submit")

    def poll(self, slot):
        return check(lib().hs_ed25519_ring_poll(self._h, int(slot)), "hs_ed25519_ring_poll") == 1

    def wait(self, slot, want_bitmap=True):
        n = self._n[slot]
        bm = (ctypes.c_ubyte * ((n + 7) // 8))() if want_bitmap else None
        rc = check(lib().hs_ed25519_ring_wait(self._h, int(slot), ctypes.cast(bm, ctypes.c_void_p) if bm is not None else None),
                   "hs_ed25519_ring_wait")
        return rc == 1, (bytes(bm) if bm is not None else None)

    def slot_ctx(self, slot):
        """hs_ed25519_ring_slot_ctx: the slot's context, BORROWED (closing the returned object does nothing) - for the
        per-context knobs and read-outs (set_rlc_min, rlc_last, set_concurrent) while the slot has nothing in flight."""
        h = lib().hs_ed25519_ring_slot_ctx(self._h, int(slot))
        if not h:
            raise EngineError("hs_ed25519_ring_slot_ctx: " + last_error())
        c = DeviceContext.__new__(DeviceContext)
        c._h = ctypes.c_void_p(h)
        c.close = lambda: None
        return c

    def stats(self):
        """(graph captures / updates, rounds submitted)"""
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().hs_ed25519_ring_stats(self._h, ctypes.byref(a), ctypes.byref(b)), "hs_ed25519_ring_stats")
        return a.value, b.value

    def close(self):
        if getattr(self, "_h", None):
            lib().hs_ed25519_ring_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
