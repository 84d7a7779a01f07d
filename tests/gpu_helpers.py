"""Device-pointer test helper: runs packed byte strings through a DeviceContext (so that a test can
force a kernel variant or the RLC threshold per context) and returns per-signature accept bits."""
import struct

import torch

from conftest import bits
from hotstuff_amd import _lib


class DeviceRunner:
    def __init__(self, max_batch=1024, kernel=_lib.KERNEL_AUTO, rlc_min=None, slot=0):
        _lib.ensure_init()
        self.ctx = _lib.DeviceContext(max_batch, slot=slot)
        self.dev = torch.device("cuda", _lib.lib().hs_ed25519_ctx_device(self.ctx._h))
        if kernel != _lib.KERNEL_AUTO:
            self.ctx.set_kernel(kernel)
        if rlc_min is not None:
            self.ctx.set_rlc_min(rlc_min)

    def _t(self, b):
        return torch.frombuffer(bytearray(b) if len(b) else bytearray(16), dtype=torch.uint8).to(self.dev)

    def _finish(self, bm, nr, n):
        torch.cuda.current_stream(self.dev).synchronize()
        st = torch.cuda.ExternalStream(self.ctx.stream, device=self.dev)
        st.synchronize()
        raw = bytes(bm.cpu().tolist())
        assert all(b == 0 for b in raw[(n + 7) // 8:]), "bits past n must be zero"
        if n % 8:
            assert raw[n // 8] >> (n % 8) == 0, "bits past n must be zero"
        b = bits(raw, n)
        assert int(nr.item()) == sum(1 for x in b if not x), "reject counter disagrees with the bitmap"
        return b

    def bits(self, dg, pk, sg, n, mode, shared, keyset=None, keys=None):
        """accept bits of n packed tuples; with `keyset`, `keys` is the list of key byte strings."""
        d_dg, d_sg = self._t(dg), self._t(sg)
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=self.dev)
        nr = torch.zeros(1, dtype=torch.int32, device=self.dev)
        torch.cuda.synchronize(self.dev)
        if keyset is None:
            d_pk = self._t(pk)
            torch.cuda.synchronize(self.dev)
            self.ctx.verify(d_dg.data_ptr(), 0 if shared else 32, d_pk.data_ptr(), d_sg.data_ptr(), n, mode,
                            bm.data_ptr(), nr.data_ptr(), 0)
        else:
            packed, _ = keyset.indices(keys)
            d_ki = self._t(packed).view(torch.int32)
            torch.cuda.synchronize(self.dev)
            self.ctx.verify_cached(keyset, d_dg.data_ptr(), 0 if shared else 32, d_ki.data_ptr(), d_sg.data_ptr(), n, mode,
                                   bm.data_ptr(), nr.data_ptr(), 0)
        return self._finish(bm, nr, n)

    def tc_bits(self, round_, hq_rounds, pk, sg, keyset=None, keys=None):
        n = len(hq_rounds)
        d_hq = self._t(struct.pack("<%dQ" % n, *hq_rounds))
        d_sg = self._t(sg)
        bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=self.dev)
        nr = torch.zeros(1, dtype=torch.int32, device=self.dev)
        if keyset is None:
            d_pk = self._t(pk)
            torch.cuda.synchronize(self.dev)
            self.ctx.verify_tc(round_, d_hq.data_ptr(), d_pk.data_ptr(), d_sg.data_ptr(), n, bm.data_ptr(), nr.data_ptr(), 0)
        else:
            packed, _ = keyset.indices(keys)
            d_ki = self._t(packed).view(torch.int32)
            torch.cuda.synchronize(self.dev)
            self.ctx.verify_tc(round_, d_hq.data_ptr(), 0, d_sg.data_ptr(), n, bm.data_ptr(), nr.data_ptr(), 0,
                               keyset=keyset, d_key_idx=d_ki.data_ptr())
        return self._finish(bm, nr, n)

    def close(self):
        self.ctx.close()
