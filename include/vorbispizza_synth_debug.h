/*
 * vorbispizza_synth_debug.h -- TEST-ONLY entry points of libvorbispizza_synth.so.
 *
 * Not part of the surface a C# host binds (INTEGRATION.md lists only vorbispizza_synth.h): these calls exist so
 * that the parity tests can read back the INTEGER intermediates of the device path and compare them index for
 * index with the reference's arithmetic (north_star: "the Huffman/indexing stage bit-exact").
 */
#ifndef VORBISPIZZA_SYNTH_DEBUG_H
#define VORBISPIZZA_SYNTH_DEBUG_H

#include "vorbispizza_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Runs the Floor1 device code of vpz_decoder_synth on `n_records` channel records and returns its integers:
 *   floor1_unwrap_kernel -- Floor1.UnwrapPosts (Floor1.cs:270-353) and the choice of the posts a line is drawn to
 *                           (Floor1.cs:236-252);
 *   floor1_render_kernel -- RenderLineMulti (Floor1.cs:372-397) as one inverse-dB table index per bin: the same
 *                           render_floor_indices device function the fused synthesis kernel runs in LDS.
 * Inputs (host memory): posts[rec*64 + i] raw posts as Unpack left them, post_counts[rec] (0 => no curve),
 * record_floor[rec] index into the decoder's floor table (type-1 floors only), record_long[rec] != 0 => the
 * record belongs to a block_size1 block.
 * Outputs (host memory, each may be NULL):
 *   curve_out[rec * (block_size1/2) + bin]   table index of every bin below the record's blocksize/2 (the device
 *                                            clamps to 0..255 where the reference would index outside its table);
 *   final_y_out[rec*64 + i]                  finalY[i] * multiplier, i < x_count (what Apply passes to the render);
 *   step_flags_out[rec*64 + i]               stepFlags[i];
 *   active_count_out[rec]                    posts a line is drawn to (post 0 + flagged posts).
 * Rows of records with post_counts == 0 are left untouched. */
int vpz_debug_floor1_indices(vpz_decoder *dec, int64_t n_records, const int16_t *posts, const uint8_t *post_counts,
                             const uint8_t *record_floor, const uint8_t *record_long, uint8_t *curve_out,
                             int16_t *final_y_out, uint8_t *step_flags_out, uint8_t *active_count_out);

#ifdef __cplusplus
}
#endif
#endif /* VORBISPIZZA_SYNTH_DEBUG_H */
