// Kernel arguments read where they are used.
#pragma once
#include <hip/hip_runtime.h>

namespace xrt {

// A record among the kernel's arguments, read from the argument segment WHERE IT IS USED.
// The compiler loads by-value arguments in the entry block; what the tail of a ray pass needs
// (screen, apertures, plot, the array pointers of the image and of the global beam: some 200
// SGPRs) then waits through the whole pass in VGPR lanes -- v_writelane at the head, v_readlane
// at the tail, ~400 VALU slots per wave of kernels that are bound by their VALU issue
// (profiles/r06_sgpr_late_ab.txt). So the kernels with a tail take ONE record of arguments
// (offset 0 of the segment), never name the tail's members, and the consumers read them through
// the segment pointer + offsetof behind an empty asm: a new value to the compiler, scalar loads
// issued at the point of use. (Taking the address of a by-value argument instead would make the
// compiler keep a copy of the whole record in scratch memory.)
template <class T>
__device__ __forceinline__ const T& kernarg_at(unsigned off) {
  typedef const T __attribute__((address_space(4))) * KernArgPtr;
  const unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + off;
  // (the same in every lane; said so explicitly: behind divergent control flow the compiler
  // may hold it in a VGPR)
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  asm volatile("" : "+s"(lo), "+s"(hi));
  return *(const T*)(KernArgPtr)(((unsigned long long)hi << 32) | lo);
}

}  // namespace xrt
