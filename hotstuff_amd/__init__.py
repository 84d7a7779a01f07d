"""hotstuff_amd - MI355X-native batch Ed25519 verification for asonnino/hotstuff.

One hot path only (BASELINE.json north_star): what sits under
``crypto::Signature::verify`` / ``Signature::verify_batch``
(/root/reference/crypto/src/lib.rs:200-219) and hence ``QC::verify`` /
``TC::verify`` (consensus/src/messages.rs:180-198, 290-315).

  hotstuff_amd.csrc      HIP kernels + C-ABI (include/hs_ed25519.h) -> libhs_ed25519.so
  hotstuff_amd.crypto    Digest / PublicKey / Signature mirror of the reference types
  hotstuff_amd.messages  QC / TC / Committee call-site mirror
  hotstuff_amd.sharding  multi-GPU sharding of one large committee (RCCL all-gather)
  hotstuff_amd.ingest    callers either side of the path: batched vote ingest, verified-QC cache
  hotstuff_amd._lib      ctypes binding incl. KeySet (committee key cache) and DeviceContext

There is no CPU fallback: without the HIP library and a GPU every verify raises.
"""
from .crypto import CryptoError, Digest, PublicKey, Signature  # noqa: F401
