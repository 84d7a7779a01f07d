"""Sharding one large committee's signatures over the GPUs of a node.

SURVEY.md 8(e): the units (signatures) are independent, so the batch is cut into
contiguous index ranges, one per rank (one process per GPU).  No data-path collective
is needed for the work itself; the single exchange step the north star names is an
all-gather of the per-shard accept bitmaps so every rank - whichever one the host polls -
holds the full bitmap and the global verdict.  With ``torch.distributed`` backend
"nccl" that all-gather IS RCCL over xGMI; at ~1 KB per GPU it is latency-bound.

Shard size is rounded up to a multiple of 64 signatures so that every shard's bitmap is
whole 64-bit wavefront ballots and the gathered slices concatenate bit-exactly.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous range [lo, hi) of signature indices owned by `rank`, and the
    per-rank padded shard size S (multiple of 64; every rank uses the same S)."""
    per = (n + world - 1) // world
    S = (per + 63) // 64 * 64 if n else 64
    lo = min(n, rank * S)
    hi = min(n, lo + S)
    return lo, hi, S


_gather_out = {}     # (device, bytes) -> reusable output buffer: one QC per call, no allocation on the path


def gather_bitmaps(local_bitmap, group=None):
    """All-gather equal-sized uint8 bitmap slices -> concatenated uint8 tensor
    (world * len(local_bitmap) bytes) on every rank.  The returned tensor is reused by the next
    call with the same shape on the same device: copy it to keep it."""
    world = dist.get_world_size(group)
    if local_bitmap.is_cuda and dist.get_backend(group) == "gloo":
        # ranks sharing one GPU (a rehearsal of the multi-GPU path on fewer devices; RCCL refuses two ranks per
        # device): the <= 1 KB slices travel through host memory and gloo
        return gather_bitmaps(local_bitmap.cpu(), group).to(local_bitmap.device)
    key = (local_bitmap.device, world * local_bitmap.numel())
    out = _gather_out.get(key)
    if out is None:
        out = _gather_out[key] = torch.empty(key[1], dtype=torch.uint8, device=local_bitmap.device)
    dist.all_gather_into_tensor(out, local_bitmap.contiguous(), group=group)
    return out


def verify_sharded(verify_shard, n, group=None, device=None):
    """Verify signatures [0, n) cooperatively.

    verify_shard(lo, hi, S) -> uint8 tensor of S // 8 bytes on `device`: the accept
    bitmap of signatures [lo, hi) (bit j = signature lo + j; bits past hi - lo zero).
    Returns (all_ok: bool, bitmap: uint8 tensor of ceil(n / 8) bytes), identical on
    every rank.  `bitmap` aliases a buffer the next call of the same shape overwrites.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi, S = shard_bounds(n, world, rank)
    local = verify_shard(lo, hi, S)
    if local.numel() != S // 8:
        raise ValueError("verify_shard must return S/8 = %d bytes, got %d" % (S // 8, local.numel()))
    full = gather_bitmaps(local, group)
    nbytes = (n + 7) // 8
    bitmap = full[:nbytes]
    # all accepted <=> the first n bits are all ones.  One device-to-host copy (and one
    # synchronisation) of the few KB, checked on the host.
    if n == 0:
        return True, bitmap
    host = bitmap.cpu().numpy()
    whole = n // 8
    ok = bool((host[:whole] == 0xFF).all())
    if n % 8:
        ok = ok and int(host[whole]) == (1 << (n % 8)) - 1
    return ok, bitmap


class GpuShardVerifier:
    """verify_shard callable for device-resident inputs: slices the packed tensors and
    runs this rank's range through the C-ABI device entry point on the current stream."""

    def __init__(self, digests, digest_stride, pks, sigs, mode, max_batch=None, keyset=None, key_idx=None, first=0):
        """keyset / key_idx (int32 device tensor, one index per signature): verify through the
        committee key cache instead of by key bytes (`pks` may then be None).
        `first`: the global index of the first signature the tensors hold - a rank normally holds only its own shard
        (signatures [lo, hi) of the call), so first = lo.  (Until round 3 the call indexed the tensors with GLOBAL
        indices, which is only right for rank 0 or for tensors holding the whole batch: the first rehearsal of a second
        rank found it.)"""
        from . import _lib
        self._lib = _lib
        self.digests, self.digest_stride, self.pks, self.sigs, self.mode = digests, digest_stride, pks, sigs, mode
        self.keyset, self.key_idx, self.first = keyset, key_idx, int(first)
        self.ctx = _lib.DeviceContext(max_batch or 1024)
        # the kernels run on the context's own stream; consumers on torch's current stream (the
        # all-gather) are ordered behind it explicitly - a raw handle of 0 would mean "the
        # context's stream" to the C-ABI, not torch's null stream
        self._stream = torch.cuda.ExternalStream(self.ctx.stream, device=sigs.device)
        self._buf, self._buf_key = None, None

    def __call__(self, lo, hi, S):
        dev = self.sigs.device
        # the kernels write whole 64-bit words for [lo, hi); the tail up to S stays zero from here
        if self._buf is None or self._buf_key != (lo, hi, S):
            self._buf, self._buf_key = torch.zeros(S // 8, dtype=torch.uint8, device=dev), (lo, hi, S)
        bitmap = self._buf
        n = hi - lo
        if n > 0:
            lo -= self.first                         # position inside the tensors this rank holds
            held = self.sigs.numel() // 64
            if lo < 0 or lo + n > held:
                raise ValueError("shard [%d, %d) is outside the signatures this verifier holds [%d, %d)"
                                 % (lo + self.first, lo + self.first + n, self.first, self.first + held))
            dg = self.digests if self.digest_stride == 0 else self.digests[32 * lo:]
            cur = torch.cuda.current_stream(dev)
            self._stream.wait_stream(cur)            # inputs / the zeroed bitmap were produced on `cur`
            stream = self.ctx.stream
            if self.keyset is not None:
                self.ctx.verify_cached(self.keyset, dg.data_ptr(), self.digest_stride, self.key_idx[lo:].data_ptr(),
                                       self.sigs[64 * lo:].data_ptr(), n, self.mode, bitmap.data_ptr(), 0, stream)
            else:
                self.ctx.verify(dg.data_ptr(), self.digest_stride, self.pks[32 * lo:].data_ptr(),
                                self.sigs[64 * lo:].data_ptr(), n, self.mode, bitmap.data_ptr(), 0, stream)
            cur.wait_stream(self._stream)            # whoever reads the bitmap on `cur` sees the result
        return bitmap
