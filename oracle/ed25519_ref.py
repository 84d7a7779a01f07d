"""Big-integer ground truth for the Ed25519 verification hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``hotstuff_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may ("oracle").

What it restates (the arithmetic is NOT in /root/reference; it lives in the
un-vendored crate ``ed25519-dalek = 1.0.1`` pulled in by
``crypto/Cargo.toml:10``), anchored on the reference's call sites:

* ``Signature::verify``        crypto/src/lib.rs:200-204  -> ``verify_strict``
* ``Signature::verify_batch``  crypto/src/lib.rs:206-219  -> ``verify_batch``
* ``Signature::new``           crypto/src/lib.rs:185-191  -> ``sign`` (RFC 8032 5.1.6)
* ``QC::verify`` / ``TC::verify`` consensus/src/messages.rs:180-198, 290-315

dalek 1.0.1 behaviours restated here ([UPSTREAM-RECALL]):

* s accepted iff s < L;
* y decoded from the low 255 bits WITHOUT a canonical check (y in [p, 2^255)
  is accepted and reduced); x = 0 with sign bit 1 is accepted;
* the hash input is the ORIGINAL bytes of R and A;
* single verify compares POINTS: [k](-A) + [s]B == decompress(R), cofactorless;
* strict additionally rejects small-order R or A ([8]P == identity);
* batch verify accepts iff  [-sum z_i s_i]B + sum [z_i]R_i + sum [z_i k_i]A_i
  is the identity, z_i random 128-bit, no cofactor multiplication.

Parity status: pinned on honest / tampered signatures through RFC 8032 and
OpenSSL (tests/golden); the edge classes (small order, non-canonical, s >= L)
are "parity unpinned" by the reference - it holds no vectors for them.
"""
import hashlib
import os

P = 2 ** 255 - 19
L = 2 ** 252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)

# Base point: y = 4/5, x even ("positive").
_BY = (4 * pow(5, P - 2, P)) % P


def _recover_x(y, sign):
    """RFC 8032 5.1.3 restated with dalek's leniency: returns None if off curve."""
    u = (y * y - 1) % P
    v = (D * y * y + 1) % P
    # candidate root x = u v^3 (u v^7)^((p-5)/8)
    x = (u * pow(v, 3, P) * pow(u * pow(v, 7, P) % P, (P - 5) // 8, P)) % P
    vxx = (v * x * x) % P
    if vxx == u:
        pass
    elif vxx == (-u) % P:
        x = (x * SQRT_M1) % P
    else:
        return None
    if x & 1:  # dalek: make x non-negative first ...
        x = P - x
    if sign:   # ... then conditionally negate; -0 == 0 is accepted
        x = (P - x) % P
    return x


BX = _recover_x(_BY, 0)
B = (BX, _BY, 1, (BX * _BY) % P)
IDENT = (0, 1, 1, 0)


def pt_add(p, q):
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q
    a = ((y1 - x1) * (y2 - x2)) % P
    b = ((y1 + x1) * (y2 + x2)) % P
    c = (2 * D * t1 * t2) % P
    d = (2 * z1 * z2) % P
    e, f, g, h = b - a, d - c, d + c, b + a
    return ((e * f) % P, (g * h) % P, (f * g) % P, (e * h) % P)


def pt_neg(p):
    x, y, z, t = p
    return ((-x) % P, y, z, (-t) % P)


def pt_mul(k, p):
    q = IDENT
    while k > 0:
        if k & 1:
            q = pt_add(q, p)
        p = pt_add(p, p)
        k >>= 1
    return q


def pt_eq(p, q):
    x1, y1, z1, _ = p
    x2, y2, z2, _ = q
    return (x1 * z2 - x2 * z1) % P == 0 and (y1 * z2 - y2 * z1) % P == 0


def pt_is_identity(p):
    return pt_eq(p, IDENT)


def pt_is_small_order(p):
    return pt_is_identity(pt_mul(8, p))


def decompress(b):
    """CompressedEdwardsY::decompress: 32 bytes -> extended point or None."""
    n = int.from_bytes(b, "little")
    sign = n >> 255
    y = (n & ((1 << 255) - 1)) % P      # no canonical check
    x = _recover_x(y, sign)
    if x is None:
        return None
    return (x, y, 1, (x * y) % P)


def compress(p):
    x, y, z, _ = p
    zi = pow(z, P - 2, P)
    x, y = (x * zi) % P, (y * zi) % P
    return int(y | ((x & 1) << 255)).to_bytes(32, "little")


def sha512_modl(*parts):
    h = hashlib.sha512()
    for x in parts:
        h.update(x)
    return int.from_bytes(h.digest(), "little") % L


# ---------------------------------------------------------------- signing
def secret_expand(seed):
    h = hashlib.sha512(seed).digest()
    a = int.from_bytes(h[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a, h[32:]


def public_key(seed):
    a, _ = secret_expand(seed)
    return compress(pt_mul(a, B))


def sign(seed, msg):
    """RFC 8032 5.1.6 (what dalek's Keypair::sign computes; lib.rs:185-191)."""
    a, prefix = secret_expand(seed)
    A = compress(pt_mul(a, B))
    r = sha512_modl(prefix, msg)
    R = compress(pt_mul(r, B))
    k = sha512_modl(R, A, msg)
    s = (r + k * a) % L
    return R + int(s).to_bytes(32, "little")


# ---------------------------------------------------------------- verification
def _parse(pk, sig):
    """Signature::from_bytes + PublicKey::from_bytes (lib.rs:201-202, 215-216).

    Returns (A, Rbytes, s) or None on a format error.
    """
    if len(pk) != 32 or len(sig) != 64:
        return None
    s = int.from_bytes(sig[32:], "little")
    if sig[63] & 0xE0:        # ed25519 1.x pre-filter (subsumed by s < L)
        return None
    if s >= L:                # dalek check_scalar
        return None
    A = decompress(pk)
    if A is None:
        return None
    return A, sig[:32], s


def verify_strict(msg, pk, sig):
    """dalek PublicKey::verify_strict as reached from crypto/src/lib.rs:200-204."""
    parsed = _parse(pk, sig)
    if parsed is None:
        return False
    A, Rb, s = parsed
    R = decompress(Rb)
    if R is None:
        return False
    if pt_is_small_order(R) or pt_is_small_order(A):
        return False
    k = sha512_modl(Rb, pk, msg)
    Rp = pt_add(pt_mul(k, pt_neg(A)), pt_mul(s, B))
    return pt_eq(Rp, R)


def verify_cofactorless(msg, pk, sig):
    """Per-signature verdict of the equation verify_batch sums (no strict extras).

    A batch made only of signatures for which this is True is always accepted
    by dalek::verify_batch (SURVEY.md 8a A4-note class ii).
    """
    parsed = _parse(pk, sig)
    if parsed is None:
        return False
    A, Rb, s = parsed
    R = decompress(Rb)
    if R is None:
        return False
    k = sha512_modl(Rb, pk, msg)
    Rp = pt_add(pt_mul(k, pt_neg(A)), pt_mul(s, B))
    return pt_eq(Rp, R)


def verify_cofactored(msg, pk, sig):
    """[8]([s]B - [k]A - R) == O.  A batch with any signature failing THIS is
    rejected by dalek::verify_batch w.p. 1 - 2^-125 (class i)."""
    parsed = _parse(pk, sig)
    if parsed is None:
        return False
    A, Rb, s = parsed
    R = decompress(Rb)
    if R is None:
        return False
    k = sha512_modl(Rb, pk, msg)
    Rp = pt_add(pt_add(pt_mul(k, pt_neg(A)), pt_mul(s, B)), pt_neg(R))
    return pt_is_identity(pt_mul(8, Rp))


def verify_batch(msgs, pks, sigs, rng=None):
    """dalek::verify_batch (random-linear-combination, cofactorless) as reached
    from crypto/src/lib.rs:206-219.  ``rng(nbytes)`` supplies the z_i entropy
    (dalek: merlin transcript finalised with thread_rng()); default os.urandom.
    """
    if not (len(msgs) == len(pks) == len(sigs)):
        return False
    rng = rng or os.urandom
    parsed = []
    for pk, sig in zip(pks, sigs):   # lib.rs:213-217: per-vote format checks, early return
        q = _parse(pk, sig)
        if q is None:
            return False
        parsed.append(q)
    acc = IDENT
    bcoef = 0
    for (A, Rb, s), pk, msg in zip(parsed, pks, msgs):
        k = sha512_modl(Rb, pk, msg)
        z = int.from_bytes(rng(16), "little")
        R = decompress(Rb)
        if R is None:                # optional_multiscalar_mul -> None -> Err
            return False
        bcoef = (bcoef + z * s) % L
        acc = pt_add(acc, pt_mul(z, R))
        acc = pt_add(acc, pt_mul((z * k) % L, A))
    acc = pt_add(acc, pt_mul((-bcoef) % L, B))
    return pt_is_identity(acc)


def verify_batch_shared(digest, votes, rng=None):
    """Signature::verify_batch(&digest, votes) crypto/src/lib.rs:206-219:
    one shared 32-byte message for every (pk, sig) pair; empty -> Ok."""
    pks = [v[0] for v in votes]
    sigs = [v[1] for v in votes]
    return verify_batch([digest] * len(votes), pks, sigs, rng)


# ---------------------------------------------------------------- torsion helpers (edge vectors)
def eight_torsion():
    """The 8 points of order dividing 8, generated from one order-8 point."""
    # find a point of order 8: take any point Q, T = [L]Q has order | 8.
    y = 2
    while True:
        x = _recover_x(y, 0)
        if x is not None:
            Q = (x, y, 1, (x * y) % P)
            T = pt_mul(L, Q)
            if not pt_is_identity(pt_mul(4, T)):
                break
        y += 1
    pts, cur = [], IDENT
    for _ in range(8):
        pts.append(cur)
        cur = pt_add(cur, T)
    return pts
