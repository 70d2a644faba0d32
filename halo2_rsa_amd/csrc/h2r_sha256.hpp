// The caller's side of the path: RSASignatureVerifier::verify_pkcs1v15_signature (reference src/lib.rs:183-246) hashes the
// signed message with SHA-256, reverses the 32 digest bytes and composes them eight at a time into the four 64-bit limbs
// `hashed_msg` that RSAChip::verify_pkcs1v15_signature compares with the low limbs of sig^e mod n (src/chip.rs:141-144).
//
//   sha256_kernel   one lane per message (messages are ragged: [off[e], off[e+1]) of one byte buffer, or fixed-length):
//                   FIPS 180-4 SHA-256 with the sixteen-word schedule window and the eight working variables in registers,
//                   all 64 rounds unrolled (rotations = v_alignbit_b32), whole blocks fetched as big-endian words when the
//                   message start is 4-byte aligned.  Writes per element
//                     digest   32 bytes, the order sha2 / decompose_digest_to_bytes yield them (what the reference returns
//                              as `hashed_bytes`, src/lib.rs:243-244)
//                     hashed   4 x u64, limb i = sum_j hashed_bytes_reversed[8i + j] * 2^(8j)   (src/lib.rs:213-239)
//                     region   the step's flat stream: the 32 reversed byte cells, then the 32 running limb values of the
//                              mul_add chain (src/lib.rs:227-236), 8 bytes each -- 288 bytes.
//                   Integer ALU bound: ~2,900 VALU instructions per 64-byte block and lane; the bytes moved are noise next to
//                   the 1.2 MB of op-trace the same element gets from the record kernel.
#pragma once

#include "h2r_kernels.hpp"

namespace h2r {

constexpr u32 HM_BYTES_OFF = 0, HM_RUN_OFF = 32, HM_REGION = 288;   // == H2R_HASHED_MSG_STREAM_BYTES

__device__ __forceinline__ u32 sha_rotr(u32 x, u32 n) { return __builtin_amdgcn_alignbit(x, x, n); }

#define H2R_SHA_K(X) \
    X(0x428a2f98) X(0x71374491) X(0xb5c0fbcf) X(0xe9b5dba5) X(0x3956c25b) X(0x59f111f1) X(0x923f82a4) X(0xab1c5ed5) \
    X(0xd807aa98) X(0x12835b01) X(0x243185be) X(0x550c7dc3) X(0x72be5d74) X(0x80deb1fe) X(0x9bdc06a7) X(0xc19bf174) \
    X(0xe49b69c1) X(0xefbe4786) X(0x0fc19dc6) X(0x240ca1cc) X(0x2de92c6f) X(0x4a7484aa) X(0x5cb0a9dc) X(0x76f988da) \
    X(0x983e5152) X(0xa831c66d) X(0xb00327c8) X(0xbf597fc7) X(0xc6e00bf3) X(0xd5a79147) X(0x06ca6351) X(0x14292967) \
    X(0x27b70a85) X(0x2e1b2138) X(0x4d2c6dfc) X(0x53380d13) X(0x650a7354) X(0x766a0abb) X(0x81c2c92e) X(0x92722c85) \
    X(0xa2bfe8a1) X(0xa81a664b) X(0xc24b8b70) X(0xc76c51a3) X(0xd192e819) X(0xd6990624) X(0xf40e3585) X(0x106aa070) \
    X(0x19a4c116) X(0x1e376c08) X(0x2748774c) X(0x34b0bcb5) X(0x391c0cb3) X(0x4ed8aa4a) X(0x5b9cca4f) X(0x682e6ff3) \
    X(0x748f82ee) X(0x78a5636f) X(0x84c87814) X(0x8cc70208) X(0x90befffa) X(0xa4506ceb) X(0xbef9a3f7) X(0xc67178f2)

// one compression: state += F(state, w[0..15]); w is consumed (it becomes the schedule window)
__device__ __forceinline__ void sha256_compress(u32 (&h)[8], u32 (&w)[16]) {
    constexpr u32 K[64] = {
#define X(v) v##u,
        H2R_SHA_K(X)
#undef X
    };
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int t = 0; t < 64; ++t) {
        if (t >= 16) {
            const u32 w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
            const u32 s0 = sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3);
            const u32 s1 = sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10);
            w[t & 15] = w[t & 15] + s0 + w[(t + 9) & 15] + s1;
        }
        const u32 S1 = sha_rotr(e, 6) ^ sha_rotr(e, 11) ^ sha_rotr(e, 25);
        const u32 ch = (e & f) ^ (~e & g);
        const u32 t1 = hh + S1 + ch + K[t] + w[t & 15];
        const u32 S0 = sha_rotr(a, 2) ^ sha_rotr(a, 13) ^ sha_rotr(a, 22);
        const u32 mj = (a & b) ^ (a & c) ^ (b & c);
        const u32 t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// digest, hashed-message limbs and the composition stream of element e from its final state
__device__ __forceinline__ void sha256_outputs(const Sha256Args &a, u64 e, const u32 (&h)[8]) {
    if (a.digest) {
        uint4 *d = reinterpret_cast<uint4 *>(a.digest + e * 32);
        d[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
        d[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
    }
    // hashed_bytes.reverse() (src/lib.rs:213) makes byte k the digest's byte 31 - k, so limb i = sum_j byte[8i + j] 2^(8j)
    // (:225-237) is the big-endian pair (h[6 - 2i], h[7 - 2i])
    u64 limb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) limb[i] = ((u64)h[6 - 2 * i] << 32) | h[7 - 2 * i];
    if (a.hashed) {
        ulonglong2 *ho = reinterpret_cast<ulonglong2 *>(a.hashed + e * 4);
        ho[0] = make_ulonglong2(limb[0], limb[1]); ho[1] = make_ulonglong2(limb[2], limb[3]);
    }
    if (a.region) {
        u8 *r = a.region + e * a.region_stride;
        // the reversed byte cells, little-endian inside each limb: the 32 bytes ARE the four limbs' bytes in memory order
        pst16(r + HM_BYTES_OFF, limb[0], limb[1]); pst16(r + HM_BYTES_OFF + 16, limb[2], limb[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {   // limb_val after byte j and after byte j + 1
                const u64 lo = j == 6 ? limb[i] & 0x00ffffffffffffffull : limb[i] & ((1ull << (8 * (j + 1))) - 1);
                const u64 hi = j == 6 ? limb[i] : limb[i] & ((1ull << (8 * (j + 2))) - 1);
                pst16(r + HM_RUN_OFF + 8 * (8 * i + j), lo, hi);
            }
        }
    }
}

#ifdef H2R_TU_API   // a plain (non-template) kernel: defined in the one translation unit that launches it
__global__ __launch_bounds__(64) void sha256_kernel(Sha256Args a) {
    const u64 e = (u64)blockIdx.x * 64 + threadIdx.x;
    if (e >= a.batch) return;
    u64 beg, len;
    if (a.off) { beg = a.off[e]; const u64 end = a.off[e + 1]; len = end >= beg ? end - beg : 0; }
    else { beg = e * a.fixed_len; len = a.fixed_len; }
    const u8 *m = a.msgs + beg;   // never dereferenced when len == 0
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const u64 n_blocks = (len + 9 + 63) / 64;
    const bool aligned = (reinterpret_cast<u64>(m) & 3) == 0;
    u32 w[16];
    for (u64 blk = 0; blk < n_blocks; ++blk) {
        const u64 p0 = blk * 64;
        if (p0 + 64 <= len && aligned) {                 // a whole block of message bytes, word loads
            const u32 *m4 = reinterpret_cast<const u32 *>(m + p0);
#pragma unroll
            for (int t = 0; t < 16; ++t) w[t] = __builtin_bswap32(m4[t]);
        } else {                                         // bytes, the 0x80 terminator, zero fill, the bit length
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                u32 v = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 p = p0 + 4 * t + q;
                    const u32 byte = p < len ? m[p] : (p == len ? 0x80u : 0u);
                    v = (v << 8) | byte;
                }
                w[t] = v;
            }
            if (blk == n_blocks - 1) { const u64 bits = len * 8; w[14] = (u32)(bits >> 32); w[15] = (u32)bits; }
        }
        sha256_compress(h, w);
    }
    sha256_outputs(a, e, h);
}
#endif

// ---- the same hash as a ROLE of the step launch (step_kernel, h2r_kernels.hpp) -------------------------------------------------------
// The register budget there is the chain role's (80 VGPRs; the fully unrolled sha256_kernel takes 94), so the rounds run as four
// passes of sixteen unrolled rounds -- the schedule window stays in registers with static indices, the round constants come from
// constant memory -- about 45 VGPRs.  The role is one latency-bound wave per 64 messages next to five busy workgroups per CU: it
// raises its wave priority, because the chain role of the same launch may wait for its limbs (Sha256Args::done).
__constant__ u32 SHA256_K[64] = {
#define X(v) v##u,
    H2R_SHA_K(X)
#undef X
};
template <int NT>
__device__ void sha256_role(const Sha256Args &a, u32 blk, u32 *) {
    const u32 tid = threadIdx.x;
    const u64 e = (u64)blk * NT + tid;
    if (e >= a.batch) return;
    __builtin_amdgcn_s_setprio(3);
    u64 beg, len;
    if (a.off) { beg = a.off[e]; const u64 end = a.off[e + 1]; len = end >= beg ? end - beg : 0; }
    else { beg = e * a.fixed_len; len = a.fixed_len; }
    const u8 *m = a.msgs + beg;
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const u64 n_blocks = (len + 9 + 63) / 64;
    const bool aligned = (reinterpret_cast<u64>(m) & 3) == 0;
    u32 w[16];
    for (u64 bk = 0; bk < n_blocks; ++bk) {
        const u64 p0 = bk * 64;
        if (p0 + 64 <= len && aligned) {
            const u32 *m4 = reinterpret_cast<const u32 *>(m + p0);
#pragma unroll
            for (int t = 0; t < 16; ++t) w[t] = __builtin_bswap32(m4[t]);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                u32 v = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 p = p0 + 4 * t + q;
                    const u32 byte = p < len ? m[p] : (p == len ? 0x80u : 0u);
                    v = (v << 8) | byte;
                }
                w[t] = v;
            }
            if (bk == n_blocks - 1) { const u64 bits = len * 8; w[14] = (u32)(bits >> 32); w[15] = (u32)bits; }
        }
        u32 va = h[0], vb = h[1], vc = h[2], vd = h[3], ve = h[4], vf = h[5], vg = h[6], vh = h[7];
#pragma unroll 1
        for (u32 g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (g) {
                    const u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
                    w[i] += (sha_rotr(w15, 7) ^ sha_rotr(w15, 18) ^ (w15 >> 3)) + w[(i + 9) & 15] + (sha_rotr(w2, 17) ^ sha_rotr(w2, 19) ^ (w2 >> 10));
                }
                const u32 t1 = vh + (sha_rotr(ve, 6) ^ sha_rotr(ve, 11) ^ sha_rotr(ve, 25)) + ((ve & vf) ^ (~ve & vg)) + SHA256_K[16 * g + i] + w[i];
                const u32 t2 = (sha_rotr(va, 2) ^ sha_rotr(va, 13) ^ sha_rotr(va, 22)) + ((va & vb) ^ (va & vc) ^ (vb & vc));
                vh = vg; vg = vf; vf = ve; ve = vd + t1; vd = vc; vc = vb; vb = va; va = t1 + t2;
            }
        }
        h[0] += va; h[1] += vb; h[2] += vc; h[3] += vd; h[4] += ve; h[5] += vf; h[6] += vg; h[7] += vh;
    }
    sha256_outputs(a, e, h);
    if (a.done) {   // the chain role of the same launch consumes the limbs: publish them, then count (release / agent scope)
        __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace h2r
