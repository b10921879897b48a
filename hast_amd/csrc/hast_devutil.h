// hast_devutil.h -- small gfx950 device helpers shared by the classify kernel (hast_kernels.hip) and the k-mer
// counting kernels (kc_kernels.hip): ASCII -> 2-bit packing, byte-class tests on whole dwords, window extraction.
#pragma once
#include "hast_common.h"

namespace hast {

__device__ __forceinline__ uint32_t pack4(uint32_t x) {
    // four ASCII bytes (first base = lowest byte) -> 8 bits, first base in the top pair
    uint32_t t = (x >> 1) & 0x03030303u;
    return (t * 0x40100401u) >> 24;
}
__device__ __forceinline__ uint32_t has_byte_N(uint32_t x) {
    uint32_t y = x ^ 0x4E4E4E4Eu;                       // 'N' -> 0
    return (y - 0x01010101u) & ~y & 0x80808080u;        // != 0 iff some byte of y is 0
}
// 4 ASCII bytes -> 4 bits (first base = bit 3): 1 where the byte is not one of 'A','C','G','T'
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {            // 0x80 in every zero byte, exact
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
}
__device__ __forceinline__ uint32_t not_acgt4(uint32_t x) {
    const uint32_t ok = zero_bytes(x ^ 0x41414141u) | zero_bytes(x ^ 0x43434343u) | zero_bytes(x ^ 0x47474747u) |
                        zero_bytes(x ^ 0x54545454u);
    const uint32_t t = (~ok & 0x80808080u) >> 7;                         // bits 0,8,16,24
    return ((t * 0x08040201u) >> 24) & 0xFu;                             // -> bits 3,2,1,0
}
// bases [p, p+n) of a packed read (n <= 31), right-aligned
__device__ __forceinline__ uint64_t window_bits(const unsigned long long *words, uint32_t p, uint32_t shift_out) {
    const unsigned long long w0 = words[p >> 5], w1 = words[(p >> 5) + 1];
    const uint32_t sh = (p & 31) * 2;
    const unsigned long long x = (w0 << sh) | ((w1 >> 1) >> (63 - sh));       // sh == 0 safe
    return x >> shift_out;
}
// as not_acgt4, but both cases of a/c/g/t are bases (clearing bit 5 maps exactly a,c,g,t onto A,C,G,T)
__device__ __forceinline__ uint32_t not_acgt4_anycase(uint32_t x) { return not_acgt4(x & 0xDFDFDFDFu); }

}  // namespace hast
