// sg_lds_dma.h -- loads that write LDS directly (gfx950 global_load_lds_dwordx4) and the waits that order them.
//
// The load has no register destination: lane i of the instruction moves 16 bytes from its own global address to
// LDS at (wave-uniform base in M0) + 16 i.  The compiler does not count these loads, so the wave orders them itself:
// loads retire in issue order, so after s_waitcnt vmcnt(n) all but the newest n vector-memory operations of the wave
// are complete (stores issued in between are counted too, which only makes the wait longer than needed), and the
// wave may read the LDS bytes they wrote.  Before a slot is loaded again the wave waits for the LDS reads it issued
// on the slot (s_waitcnt lgkmcnt(0)).  A wave must not end with such a load outstanding.
#pragma once

namespace sg {

template <int N>
__device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wait_lgkm0()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// waits until at most min(ahead, A) * PER of the wave's vector-memory operations are outstanding
template <int PER, int A>
__device__ __forceinline__ void wait_groups_ahead(int ahead)
{
    if constexpr (A == 0) {
        wait_vm<0>();
    } else {
        if (ahead >= A)
            wait_vm<A * PER>();
        else
            wait_groups_ahead<PER, A - 1>(ahead);
    }
}

// 16 bytes per active lane from base + voff (bytes) to LDS at lds_dst + 16 * lane; lds_dst and base wave-uniform.
// M0 belongs to the compiler outside the statement, so it is saved and restored.
// (The non-temporal form of the load, `nt`, was measured on the blur and extrema kernels: their own times stay the
// same or grow by 3-10 %, and the kernels that read the same planes afterwards slow down several-fold; not used.)
__device__ __forceinline__ void load16_to_lds(unsigned int lds_dst, unsigned long long base, unsigned int voff)
{
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_dst), "v"(voff), "s"(base)
                 : "memory");
}

} // namespace sg
