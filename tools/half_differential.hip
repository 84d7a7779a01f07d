// half_differential.hip - the half-size scalar reduction (halfscalar.h: Lehmer steps from a double-precision quotient estimate, a
// checked result with a silent (k, 1) fallback) on the DEVICE, at scale, against three independent referees (VERDICT r4 weak #1 /
// "Next round" #3: the CPU tests had seen ~10^4 scalars; the GPU verifies 10^8 signatures a second):
//   1. the same header compiled for the host (what tests/emu runs): the device's (u, v) must be bit-identical for every k;
//   2. the predicate the verification relies on, recomputed here with plain 64-bit limb arithmetic that shares no code with the
//      header:  v odd, 0 < v < L, |u| < 2^253,  u = v k (mod 8L) - by a 512-bit product and a shift-subtract remainder;
//   3. the sizes: max(bitlen |u|, bitlen v), the number of 4-bit windows the cold kernels will walk, and how often the pair is the
//      long fallback (k, 1) although k is not short.
// Scalars: N random (a 64-bit mixer, 252 bits at a time, rejected above L) + the adversarial families below (k = 0 mod 8 and mod 2^j, k next to 0 and
// to L, long runs of ones / zeros / alternating bits, floor(8L / m) +- 1 - a huge second quotient, past the 32-bit step -, rational
// multiples floor(L a / b), near-golden-ratio k - the longest Euclid runs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I hotstuff_amd/csrc tools/half_differential.hip -o tools/half_differential -lpthread
//   tools/half_differential [random_count = 10000000] > profiles/r5_half_differential.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "halfscalar.h"
using namespace hs;

typedef unsigned __int128 u128;
struct scal { u32 w[8]; };
struct res { u32 u[8], v[8], u_neg, nwin; };

__host__ __device__ static void one(res &r, const u32 k[8]) {
    half_pair h; half_scalar(h, k);
    u32 ru[8], rv[8];
    sc_recode16(ru, h.u); sc_recode16(rv, h.v);
    const int a = sc_recode16_len(ru), b = sc_recode16_len(rv);
    for (int j = 0; j < 8; j++) { r.u[j] = h.u[j]; r.v[j] = h.v[j]; }
    r.u_neg = h.u_neg; r.nwin = (u32)(a > b ? a : b);
}
// every scalar comes from the host's list (the hash tables of sha512_sc.h live in device constant memory: a hipcc host build
// cannot draw scalars through them)
__global__ void k_half(res *out, const scal *ks, u32 n) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k[8];
    for (int j = 0; j < 8; j++) k[j] = ks[i].w[j];
    one(out[i], k);
}

// ---- independent arithmetic: little-endian 64-bit limbs ----
static const uint64_t L4[4] = { 0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0, 0x1000000000000000ull };
static void words_to_limbs(uint64_t o[4], const u32 w[8]) { for (int i = 0; i < 4; i++) o[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32); }
static int cmp(const uint64_t *a, const uint64_t *b, int n) { for (int i = n - 1; i >= 0; i--) { if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1; } return 0; }
static int bitlen4(const uint64_t a[4]) { for (int i = 3; i >= 0; i--) if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]); return 0; }
// (v k -+ |u|) mod 8L == 0 ?   v k is a 512-bit product; the remainder by shift-subtract, 5-limb accumulator (8L < 2^256)
static bool congruent_mod_8L(const uint64_t u[4], bool u_neg, const uint64_t v[4], const uint64_t k[4]) {
    uint64_t p[9] = {0};
    for (int i = 0; i < 4; i++) { u128 c = 0; for (int j = 0; j < 4; j++) { c += (u128)v[i] * k[j] + p[i + j]; p[i + j] = (uint64_t)c; c >>= 64; } p[i + 4] += (uint64_t)c; }
    uint64_t M[5] = {0};       // 8L
    { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)L4[i] << 3; M[i] = (uint64_t)c; c >>= 64; } M[4] = (uint64_t)c; }
    auto rem = [&](const uint64_t *x, int limbs, uint64_t r[5]) {
        for (int i = 0; i < 5; i++) r[i] = 0;
        for (int b = 64 * limbs - 1; b >= 0; b--) {
            for (int i = 4; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
            r[0] = (r[0] << 1) | ((x[b / 64] >> (b % 64)) & 1);
            if (cmp(r, M, 5) >= 0) { u128 br = 0; for (int i = 0; i < 5; i++) { u128 t = (u128)r[i] - M[i] - (uint64_t)br; r[i] = (uint64_t)t; br = (t >> 64) & 1; } }
        }
    };
    uint64_t rp[5], ru[5], u5[4] = { u[0], u[1], u[2], u[3] };
    rem(p, 8, rp); rem(u5, 4, ru);
    if (u_neg) {      // -|u| mod 8L
        bool zero = true; for (int i = 0; i < 5; i++) zero = zero && ru[i] == 0;
        if (!zero) { u128 br = 0; for (int i = 0; i < 5; i++) { u128 t = (u128)M[i] - ru[i] - (uint64_t)br; ru[i] = (uint64_t)t; br = (t >> 64) & 1; } }
    }
    return cmp(rp, ru, 5) == 0;
}

struct tally {
    unsigned long long n = 0, device_ne_host = 0, predicate_fail = 0, long_fallback = 0, already_short = 0, v_is_one = 0;
    int max_bits_u = 0, max_bits_v = 0, max_bits_short = 0;      // max_bits_short: over the pairs that are NOT the long fallback
    unsigned long long bits_hist[257] = {0}, nwin_hist[66] = {0};
    char first_bad[512] = "";
    void add(const tally &o) {
        n += o.n; device_ne_host += o.device_ne_host; predicate_fail += o.predicate_fail; long_fallback += o.long_fallback;
        already_short += o.already_short; v_is_one += o.v_is_one;
        if (o.max_bits_u > max_bits_u) max_bits_u = o.max_bits_u;
        if (o.max_bits_v > max_bits_v) max_bits_v = o.max_bits_v;
        if (o.max_bits_short > max_bits_short) max_bits_short = o.max_bits_short;
        for (int i = 0; i < 257; i++) bits_hist[i] += o.bits_hist[i];
        for (int i = 0; i < 66; i++) nwin_hist[i] += o.nwin_hist[i];
        if (!first_bad[0] && o.first_bad[0]) memcpy(first_bad, o.first_bad, sizeof first_bad);
    }
};
static void check_one(tally &t, const u32 kw[8], const res &g) {
    res h; one(h, kw);
    t.n++;
    const bool same = memcmp(&h, &g, sizeof(res)) == 0;
    if (!same) t.device_ne_host++;
    uint64_t u[4], v[4], k[4];
    words_to_limbs(u, g.u); words_to_limbs(v, g.v); words_to_limbs(k, kw);
    const int bu = bitlen4(u), bv = bitlen4(v), bk = bitlen4(k);
    const bool pred = (v[0] & 1) && cmp(v, L4, 4) < 0 && bu <= 253 && congruent_mod_8L(u, g.u_neg != 0, v, k);
    if (!pred) t.predicate_fail++;
    if ((!same || !pred) && !t.first_bad[0]) {
        int o = snprintf(t.first_bad, sizeof t.first_bad, "k=");
        for (int j = 7; j >= 0; j--) o += snprintf(t.first_bad + o, sizeof t.first_bad - o, "%08x", kw[j]);
        o += snprintf(t.first_bad + o, sizeof t.first_bad - o, " same=%d pred=%d u_neg=%u u=", (int)same, (int)pred, g.u_neg);
        for (int j = 7; j >= 0; j--) o += snprintf(t.first_bad + o, sizeof t.first_bad - o, "%08x", g.u[j]);
        o += snprintf(t.first_bad + o, sizeof t.first_bad - o, " v=");
        for (int j = 7; j >= 0; j--) o += snprintf(t.first_bad + o, sizeof t.first_bad - o, "%08x", g.v[j]);
    }
    const bool plain = g.u_neg == 0 && bv == 1 && v[0] == 1 && cmp(u, k, 4) == 0;      // the pair (k, 1)
    if (plain) { t.v_is_one++; if (bk > HS_HALF_TARGET_BITS) t.long_fallback++; else t.already_short++; }
    const int bits = bu > bv ? bu : bv;
    if (bu > t.max_bits_u) t.max_bits_u = bu;
    if (bv > t.max_bits_v) t.max_bits_v = bv;
    if (!(plain && bk > HS_HALF_TARGET_BITS) && bits > t.max_bits_short) t.max_bits_short = bits;
    t.bits_hist[bits]++;
    t.nwin_hist[g.nwin > 65 ? 65 : g.nwin]++;
}

// ---- adversarial scalars (all < L) ----
static void limbs_to_scal(scal &s, const uint64_t a[4]) { for (int i = 0; i < 4; i++) { s.w[2 * i] = (u32)a[i]; s.w[2 * i + 1] = (u32)(a[i] >> 32); } }
static void push_mod_L(std::vector<scal> &out, uint64_t a[4]) {      // reduce by subtraction (inputs are < 16 L)
    while (cmp(a, L4, 4) >= 0) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - L4[i] - (uint64_t)br; a[i] = (uint64_t)t; br = (t >> 64) & 1; } }
    scal s; limbs_to_scal(s, a); out.push_back(s);
}
static void div_small(uint64_t q[5], const uint64_t x[5], uint64_t m) { u128 r = 0; for (int i = 4; i >= 0; i--) { u128 c = (r << 64) | x[i]; q[i] = (uint64_t)(c / m); r = c % m; } }
static void adversarial(std::vector<scal> &out, size_t want) {
    uint64_t s = 0x243f6a8885a308d3ull;
    auto rnd = [&]() { s += 0x9e3779b97f4a7c15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); };
    auto rand_lt_L = [&](uint64_t a[4]) { for (int i = 0; i < 4; i++) a[i] = rnd(); a[3] &= 0x0fffffffffffffffull; if (cmp(a, L4, 4) >= 0) a[3] &= 0x07ffffffffffffffull; };
    uint64_t M8[5] = {0};
    { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)L4[i] << 3; M8[i] = (uint64_t)c; c >>= 64; } M8[4] = (uint64_t)c; }
    const size_t per = want / 10;
    // 1. k = 0 mod 8, and mod 2^j
    for (size_t i = 0; i < per; i++) { uint64_t a[4]; rand_lt_L(a); const int j = 3 + (int)(i % 248); for (int b = 0; b < j; b++) a[b / 64] &= ~(1ull << (b % 64)); push_mod_L(out, a); }
    // 2. next to L and next to 0
    for (size_t i = 0; i < per; i++) { uint64_t a[4] = { L4[0], L4[1], L4[2], L4[3] }; uint64_t d = i < per / 2 ? 1 + i : 8 * (1 + rnd() % 100000); u128 br = d; for (int q = 0; q < 4; q++) { u128 t = (u128)a[q] - (uint64_t)br; a[q] = (uint64_t)t; br = (t >> 64) & 1; } push_mod_L(out, a); }
    for (size_t i = 0; i < per; i++) { uint64_t a[4] = { i < per / 2 ? i : rnd() >> (rnd() % 64), 0, 0, 0 }; if (i % 7 == 3) { a[1] = rnd() >> (rnd() % 64); } push_mod_L(out, a); }
    // 3. powers of two and their neighbours
    for (size_t i = 0; i < per / 2; i++) { uint64_t a[4] = {0, 0, 0, 0}; const int j = (int)(i % 252); a[j / 64] = 1ull << (j % 64); const int d = (int)((i / 252) % 3) - 1;
        if (d > 0) { u128 c = 1; for (int q = 0; q < 4; q++) { c += a[q]; a[q] = (uint64_t)c; c >>= 64; } }
        if (d < 0 && j) { u128 br = 1; for (int q = 0; q < 4; q++) { u128 t = (u128)a[q] - (uint64_t)br; a[q] = (uint64_t)t; br = (t >> 64) & 1; } }
        push_mod_L(out, a); }
    // 4. long runs: (2^a - 1) << b, alternating bits, runs inside random material
    for (size_t i = 0; i < per; i++) {
        uint64_t a[4] = {0, 0, 0, 0};
        const int len = 8 + (int)(rnd() % 240), sh = (int)(rnd() % (252 - (len > 251 ? 251 : len) + 1));
        if (i % 3 == 0) { for (int b = sh; b < sh + len && b < 252; b++) a[b / 64] |= 1ull << (b % 64); }
        else if (i % 3 == 1) { rand_lt_L(a); for (int b = sh; b < sh + len && b < 252; b++) a[b / 64] &= ~(1ull << (b % 64)); }
        else { for (int q = 0; q < 4; q++) a[q] = (i & 4) ? 0xaaaaaaaaaaaaaaaaull : 0x5555555555555555ull; a[3] &= 0x0fffffffffffffffull; for (int b = 0; b < sh; b++) a[b / 64] ^= (rnd() & 1) << (b % 64); }
        push_mod_L(out, a);
    }
    // 5. floor(8L / m) + d: the first quotient is m, the second one huge (beyond the 32-bit step of the inner loop)
    for (size_t i = 0; i < per; i++) { const uint64_t m = 9 + (i / 3 < 60000 ? i / 3 : rnd() % (1ull << (20 + rnd() % 40))); uint64_t q[5]; div_small(q, M8, m);
        const int d = (int)(i % 3) - 1; uint64_t a[4] = { q[0], q[1], q[2], q[3] };
        if (d > 0) { u128 c = 1; for (int t = 0; t < 4; t++) { c += a[t]; a[t] = (uint64_t)c; c >>= 64; } }
        if (d < 0) { u128 br = 1; for (int t = 0; t < 4; t++) { u128 x = (u128)a[t] - (uint64_t)br; a[t] = (uint64_t)x; br = (x >> 64) & 1; } }
        push_mod_L(out, a); }
    // 6. floor(8L a / b) mod L for small a / b: short continued fractions
    for (size_t i = 0; i < per; i++) { const uint64_t b = 2 + rnd() % 4096, a_ = 1 + rnd() % (b - 1); uint64_t q[5], t5[5]; div_small(q, M8, b);
        u128 c = 0; for (int t = 0; t < 5; t++) { c += (u128)q[t] * a_; t5[t] = (uint64_t)c; c >>= 64; }
        uint64_t a[4] = { t5[0], t5[1], t5[2], t5[3] & 0x7fffffffffffffffull }; push_mod_L(out, a); }
    // 7. near the golden ratio: k ~ 8L / phi^j - all quotients 1, the longest Euclid runs - with small perturbations
    for (size_t i = 0; out.size() < want; i++) {
        // 1 / phi = 0.6180339887498948482...: multiply 8L by the 64-bit fraction j times
        uint64_t x[5] = { M8[0], M8[1], M8[2], M8[3], M8[4] };
        const int j = 1 + (int)(i % 6);
        for (int r = 0; r < j; r++) { u128 c = 0; uint64_t y[6] = {0}; for (int t = 0; t < 5; t++) { c += (u128)x[t] * 0x9e3779b97f4a7c15ull; y[t] = (uint64_t)c; c >>= 64; } y[5] = (uint64_t)c; for (int t = 0; t < 5; t++) x[t] = y[t + 1]; }
        uint64_t a[4] = { x[0] ^ (rnd() >> (8 + rnd() % 56)), x[1], x[2], x[3] & 0x7fffffffffffffffull };
        push_mod_L(out, a);
    }
    out.resize(want);
}

int main(int argc, char **argv) {
    const unsigned long long n_random = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull;
    const size_t n_adv = argc > 2 ? strtoull(argv[2], nullptr, 10) : 100000;
    const bool host_only = argc > 3 && !strcmp(argv[3], "host");      // no device: the host twin stands in (checks this tool's own arithmetic on a box without a GPU)
    const u32 chunk = 1u << 20;
    const unsigned threads = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    res *d = nullptr; scal *dk = nullptr;
    if (!host_only && (hipMalloc((void **)&d, (size_t)chunk * sizeof(res)) != hipSuccess || hipMalloc((void **)&dk, (size_t)chunk * sizeof(scal)) != hipSuccess)) { fprintf(stderr, "hipMalloc failed\n"); return 3; }
    std::vector<res> g(chunk);
    std::vector<scal> adv, rnd_k(chunk);
    adversarial(adv, n_adv);
    uint64_t rs = 0x13198a2e03707344ull;
    auto draw = [&](scal &o) {
        for (;;) {
            uint64_t a[4];
            for (int i = 0; i < 4; i++) { rs += 0x9e3779b97f4a7c15ull; uint64_t z = rs; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; a[i] = z ^ (z >> 31); }
            a[3] &= 0x1fffffffffffffffull;
            if (cmp(a, L4, 4) < 0) { limbs_to_scal(o, a); return; }
        }
    };
    tally total[2];
    for (int part = 0; part < 2; part++) {
        const unsigned long long count = part == 0 ? n_random : adv.size();
        for (unsigned long long first = 0; first < count; first += chunk) {
            const u32 n = (u32)std::min<unsigned long long>(chunk, count - first);
            if (part == 0) for (u32 i = 0; i < n; i++) draw(rnd_k[i]);
            const scal *src = part == 0 ? rnd_k.data() : adv.data() + first;
            if (host_only) {
                for (u32 i = 0; i < n; i++) one(g[i], src[i].w);
            } else {
                if (hipMemcpy(dk, src, (size_t)n * sizeof(scal), hipMemcpyHostToDevice) != hipSuccess) return 3;
                hipLaunchKernelGGL(k_half, dim3((n + 63) / 64), dim3(64), 0, 0, d, dk, n);
                if (hipMemcpy(g.data(), d, (size_t)n * sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "copy failed: %s\n", hipGetErrorString(hipGetLastError())); return 3; }
            }
            std::vector<tally> ts(threads);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < threads; t++)
                th.emplace_back([&, t]() {
                    for (u32 i = (u32)((unsigned long long)n * t / threads); i < (u32)((unsigned long long)n * (t + 1) / threads); i++) {
                        check_one(ts[t], src[i].w, g[i]);
                    }
                });
            for (auto &x : th) x.join();
            for (auto &x : ts) total[part].add(x);
        }
    }
    printf("{\n \"tool\": \"tools/half_differential.hip\", \"target_bits\": %d, \"device\": %s,\n", HS_HALF_TARGET_BITS, host_only ? "false" : "true");
    for (int part = 0; part < 2; part++) {
        const tally &t = total[part];
        printf(" \"%s\": {\"scalars\": %llu, \"device_differs_from_host_twin\": %llu, \"predicate_failures\": %llu,\n"
               "  \"long_fallback_pairs_k_1\": %llu, \"fallback_rate\": %.3e, \"already_short_k\": %llu,\n"
               "  \"max_bits_abs_u\": %d, \"max_bits_v\": %d, \"max_bits_of_pairs_that_are_not_the_long_fallback\": %d,\n  \"bits_histogram\": {",
               part ? "adversarial" : "random", t.n, t.device_ne_host, t.predicate_fail, t.long_fallback, t.n ? (double)t.long_fallback / (double)t.n : 0.0,
               t.already_short, t.max_bits_u, t.max_bits_v, t.max_bits_short);
        bool firstp = true;
        for (int i = 0; i < 257; i++) if (t.bits_hist[i]) { printf("%s\"%d\": %llu", firstp ? "" : ", ", i, t.bits_hist[i]); firstp = false; }
        printf("},\n  \"four_bit_windows_histogram\": {");
        firstp = true;
        for (int i = 0; i < 66; i++) if (t.nwin_hist[i]) { printf("%s\"%d\": %llu", firstp ? "" : ", ", i, t.nwin_hist[i]); firstp = false; }
        printf("},\n  \"first_bad\": \"%s\"}%s\n", t.first_bad, part ? "" : ",");
    }
    printf("}\n");
    const bool bad = total[0].device_ne_host || total[0].predicate_fail || total[1].device_ne_host || total[1].predicate_fail;
    return bad ? 2 : 0;
}
