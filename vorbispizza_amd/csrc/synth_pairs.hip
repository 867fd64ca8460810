// synth_pairs_kernel -- the stereo fast path's kernel for the CHANNEL PAIRS of streams with 4, 6, 8, ... channels whose coupling steps
// join the channels two by two: synth_dual.hip compiled with VPZ_DUAL_PAIRS=1 (the comment at its head says what changes --
// addressing only; per channel the arithmetic is the stereo kernel's, operation for operation, which is synth_kernel's).
// Mapping.cs:166-195 (a coupling step touches its two channels and no other), Residue2.cs:42-51 (the [bin][C] vector read by
// columns), StreamDecoder.cs:515-638 (PCM rows of a planar buffer, columns of an interleaved one).
#define VPZ_DUAL_PAIRS 1
#include "synth_dual.hip"
