// device_util.h — plane access and HOXD70 scoring shared by the extension kernels.
#pragma once
#include "common.h"

namespace mimeo {

// 32 / 64 consecutive plane bits starting at (possibly negative, padded) base index s
__device__ __forceinline__ uint32_t get32(const uint32_t *__restrict__ pl, int32_t s) {
    int32_t w = s >> 5;
    uint32_t b = (uint32_t)s & 31u;
    uint64_t v = (uint64_t)pl[w] | ((uint64_t)pl[w + 1] << 32);
    return (uint32_t)(v >> b);
}
__device__ __forceinline__ uint64_t get64(const uint32_t *__restrict__ pl, int32_t s) {
    int32_t w = s >> 5;
    uint32_t b = (uint32_t)s & 31u;
    uint64_t lo = (uint64_t)pl[w] | ((uint64_t)pl[w + 1] << 32);
    uint64_t hi = pl[w + 2];
    return b ? (lo >> b) | (hi << (64 - b)) : lo;
}
__device__ __forceinline__ uint32_t getbit(const uint32_t *__restrict__ pl, int32_t s) {
    return (pl[s >> 5] >> ((uint32_t)s & 31u)) & 1u;
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
    uint32_t tlo = getbit(T.lo, pt), thi = getbit(T.hi, pt);
    uint32_t dl = tlo ^ getbit(Q.lo, pq), dh = thi ^ getbit(Q.hi, pq);
    uint32_t nn = getbit(T.nm, pt) | getbit(Q.nm, pq);
    *is_match = !(dl | dh | nn);
    return sub_score(dl, dh, tlo ^ thi, nn);
}

}  // namespace mimeo
