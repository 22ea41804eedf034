// dmz_wave.h -- wave64 reductions on DPP (device code only).
//
// Four row_shr steps leave each 16-lane row's result in its lane 15 (out-of-row sources read as 0: the identity
// of +, of unsigned max and of max over non-negative ints), the four row results are combined on the scalar
// unit: ~8 VALU instructions and no LDS traffic, where six __shfl_xor steps cost six ds_bpermute round trips
// plus their address arithmetic.  Results are wave-uniform (SGPR).
#pragma once

#define DMZ_DPP_SHR0(v, n) __builtin_amdgcn_update_dpp(0, (v), 0x110 + (n), 0xf, 0xf, true)

namespace dmzwave {

__device__ __forceinline__ unsigned max_u32(unsigned x) {
  int v = (int)x;
  unsigned o;
  o = (unsigned)DMZ_DPP_SHR0(v, 1), v = (int)((unsigned)v > o ? (unsigned)v : o);
  o = (unsigned)DMZ_DPP_SHR0(v, 2), v = (int)((unsigned)v > o ? (unsigned)v : o);
  o = (unsigned)DMZ_DPP_SHR0(v, 4), v = (int)((unsigned)v > o ? (unsigned)v : o);
  o = (unsigned)DMZ_DPP_SHR0(v, 8), v = (int)((unsigned)v > o ? (unsigned)v : o);
  const unsigned a = (unsigned)__builtin_amdgcn_readlane(v, 15), b = (unsigned)__builtin_amdgcn_readlane(v, 31);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane(v, 47), d = (unsigned)__builtin_amdgcn_readlane(v, 63);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// minimum of unsigned values: the maximum of the complements
__device__ __forceinline__ unsigned min_u32(unsigned x) { return ~max_u32(~x); }

__device__ __forceinline__ int sum_i32(int v) {
  v += DMZ_DPP_SHR0(v, 1);
  v += DMZ_DPP_SHR0(v, 2);
  v += DMZ_DPP_SHR0(v, 4);
  v += DMZ_DPP_SHR0(v, 8);
  return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) +
         __builtin_amdgcn_readlane(v, 63);
}

// minimum of 64-bit keys (hi, lo) as two 32-bit reductions: the smallest hi, then the smallest lo among its holders
__device__ __forceinline__ unsigned long long min_u64(unsigned hi, unsigned lo) {
  const unsigned mh = min_u32(hi);
  const unsigned ml = min_u32(hi == mh ? lo : 0xffffffffu);
  return ((unsigned long long)mh << 32) | ml;
}

// inclusive prefix sum over the 64 lanes: four row_shr steps scan each 16-lane row, row_bcast:15 carries a row's total
// into the next row (rows 1 and 3), row_bcast:31 the first half's total into rows 2 and 3
__device__ __forceinline__ int inclusive_scan_i32(int v) {
  v += DMZ_DPP_SHR0(v, 1);
  v += DMZ_DPP_SHR0(v, 2);
  v += DMZ_DPP_SHR0(v, 4);
  v += DMZ_DPP_SHR0(v, 8);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

}  // namespace dmzwave
