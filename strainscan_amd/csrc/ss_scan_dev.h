// Device helpers shared by the scan kernels (flat table: ss_scan.hip, minimizer buckets: ss_mini.hip).
#pragma once
#include "ss_common.h"

namespace ss { namespace dev {

constexpr int SCAN_THREADS = 256;
constexpr int PPT = 16;                        // k-mer start positions per thread per tile
constexpr int TILE = SCAN_THREADS * PPT;       // bytes of the base stream per tile

// ---------------------------------------------------------------------------------------------
// 16 ASCII bases (4 dwords) -> 32 bits of 2-bit codes (base i at bits 2i) + 16 invalid flags.
// SWAR: no per-byte loop, no LDS lookup table.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v)
{   // 0x80 in every byte of v that is zero (exact form, no borrow artefacts)
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}

__device__ __forceinline__ void encode4(uint32_t w, uint32_t &code8, uint32_t &bad4)
{
    uint32_t x = (w & 0xDFDFDFDFu) ^ 0x41414141u;           // A->00 C->02 G->06 T->15 (either case)
    uint32_t ok = zero_bytes(x) | zero_bytes(x ^ 0x02020202u) | zero_bytes(x ^ 0x06060606u) |
                  zero_bytes(x ^ 0x15151515u);
    bad4 = (((ok ^ 0x80808080u) >> 7) * 0x01020408u) >> 24; // bit i = byte i is not ACGT
    uint32_t c = (w >> 1) & 0x03030303u;                    // (ascii >> 1) & 3 per byte
    code8 = (c * 0x01041040u) >> 24;                        // pack the four 2-bit fields
}

__device__ __forceinline__ void encode16(const uint32_t w[4], uint32_t &code, uint32_t &inv)
{
    uint32_t c0, c1, c2, c3, b0, b1, b2, b3;
    encode4(w[0], c0, b0);
    encode4(w[1], c1, b1);
    encode4(w[2], c2, b2);
    encode4(w[3], c3, b3);
    code = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    inv = b0 | (b1 << 4) | (b2 << 8) | (b3 << 12);
}

// 16 bytes at `off` of the base stream; bytes at or beyond n read as '\n'.
template <bool ALIGNED>
__device__ __forceinline__ void load16(const uint8_t *__restrict__ bases, uint64_t off, uint64_t n,
                                       uint32_t w[4])
{
    if (ALIGNED && off + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(bases + off);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                uint64_t p = off + d * 4 + b;
                uint32_t c = (p < n) ? bases[p] : 0x0Au;
                x |= c << (8 * b);
            }
            w[d] = x;
        }
    }
}


}}  // namespace ss::dev
