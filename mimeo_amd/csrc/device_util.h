// device_util.h — plane access and HOXD70 scoring shared by the extension kernels.
#pragma once
#include "common.h"

namespace mimeo {

// 12 care bits of the 19-bit window of the 12of19 seed 1110100110010101111: offsets {0,1,2,4,7,8,11,13,15,16,17,18}
__device__ __forceinline__ uint32_t pext12(uint32_t x) {
    return (x & 0x7u) | ((x >> 1) & 0x8u) | ((x >> 3) & 0x30u) | ((x >> 5) & 0x40u) | ((x >> 6) & 0x80u) |
           ((x >> 7) & 0xF00u);
}

struct Win32 { uint32_t lo, hi, nm, sv; };
struct Win64 { uint64_t lo, hi, nm, sv; };

// 32 consecutive bases of every plane starting at (possibly negative, padded) base index s:
// two 16-byte loads, normally from the same cache line
__device__ __forceinline__ Win32 win32(const StrandView &v, int32_t s) {
    const int32_t w = s >> 5;
    const uint32_t b = (uint32_t)s & 31u;
    const uint4 a = v.pw[w], c = v.pw[w + 1];
    Win32 r;
    r.lo = (uint32_t)((((uint64_t)c.x << 32) | a.x) >> b);
    r.hi = (uint32_t)((((uint64_t)c.y << 32) | a.y) >> b);
    r.nm = (uint32_t)((((uint64_t)c.z << 32) | a.z) >> b);
    r.sv = (uint32_t)((((uint64_t)c.w << 32) | a.w) >> b);
    return r;
}
// 64 consecutive bases of every plane
__device__ __forceinline__ Win64 win64(const StrandView &v, int32_t s) {
    const int32_t w = s >> 5;
    const uint32_t b = (uint32_t)s & 31u;
    const uint4 a = v.pw[w], c = v.pw[w + 1], e = v.pw[w + 2];
    Win64 r;
    uint64_t l;
    l = ((uint64_t)c.x << 32) | a.x; r.lo = b ? (l >> b) | ((uint64_t)e.x << (64 - b)) : l;
    l = ((uint64_t)c.y << 32) | a.y; r.hi = b ? (l >> b) | ((uint64_t)e.y << (64 - b)) : l;
    l = ((uint64_t)c.z << 32) | a.z; r.nm = b ? (l >> b) | ((uint64_t)e.z << (64 - b)) : l;
    l = ((uint64_t)c.w << 32) | a.w; r.sv = b ? (l >> b) | ((uint64_t)e.w << (64 - b)) : l;
    return r;
}
// 32 bits of a plain (non-interleaved) plane
__device__ __forceinline__ uint32_t get32(const uint32_t *__restrict__ pl, int32_t s) {
    int32_t w = s >> 5;
    uint32_t b = (uint32_t)s & 31u;
    uint64_t v = (uint64_t)pl[w] | ((uint64_t)pl[w + 1] << 32);
    return (uint32_t)(v >> b);
}
// seed-start validity of 32 starts in the role the view was made for
__device__ __forceinline__ uint32_t seedvalid32(const StrandView &v, int32_t s, uint32_t sv_from_window) {
    return v.svt ? get32(v.svt, s) : sv_from_window;
}
// the three base bits at one position
struct Base1 { uint32_t lo, hi, nm; };
__device__ __forceinline__ Base1 base_at(const StrandView &v, int32_t s) {
    const uint4 a = v.pw[s >> 5];
    const uint32_t b = (uint32_t)s & 31u;
    return Base1{(a.x >> b) & 1u, (a.y >> b) & 1u, (a.z >> b) & 1u};
}

// HOXD70 + N = -100 (lastz fill_score) from the difference planes: dl/dh = xor of the lo/hi
// planes, cg = target base is C or G, nn = either base is N.
__device__ __forceinline__ int32_t sub_score(uint32_t dl, uint32_t dh, uint32_t cg, uint32_t nn) {
    uint32_t tb = dl ? 0x83858E8Eu : 0xE1E1645Bu;  // {-114,-114,-123,-125} : {91,100,-31,-31}
    uint32_t sh = 24u - (((dh << 1) | cg) << 3);
    int32_t s = ((int32_t)(tb << sh)) >> 24;
    return nn ? -100 : s;
}

// substitution score and match flag of target base pt against query base pq
__device__ __forceinline__ int32_t pair_score(const StrandView &T, const StrandView &Q, int32_t pt, int32_t pq,
                                              bool *is_match) {
    const Base1 a = base_at(T, pt), b = base_at(Q, pq);
    const uint32_t dl = a.lo ^ b.lo, dh = a.hi ^ b.hi, nn = a.nm | b.nm;
    *is_match = !(dl | dh | nn);
    return sub_score(dl, dh, a.lo ^ a.hi, nn);
}

}  // namespace mimeo
