/*
 * oracle/vorbis_synth_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, scalar, -ffp-contract=off) of the PCM-synthesis path of
 * TechPizzaDev/VorbisPizza (C#): inverse MDCT, Floor1 render, inverse coupling, window +
 * overlap-add state machine, clip + interleave.  Each function cites the reference file:line
 * it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it;
 * the product path (vorbispizza_amd/csrc) never links or calls anything in here.
 *
 * PARITY PINNING: "parity unpinned" by reference golden vectors -- the reference's tests hold no
 * float golden vectors for this path (they compare live against libvorbisfile at +-2 LSB s16,
 * NVorbis.Tests/AssetTest.cs:131-161) and the C# code cannot be executed in this pipeline.
 * What pins the restatement instead: closed-form float64 IMDCT, TDAC reconstruction, output
 * symmetries, integer DDA vs closed form, and the fixtures' known sample counts (tests/).
 */
#ifndef VORBIS_SYNTH_ORACLE_H
#define VORBIS_SYNTH_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Utils.cs:19-42 ---- */
int orc_ilog(int x);
uint32_t orc_bit_reverse(uint32_t n, int bits);

/* ---- Mdct.cs:21-66 (MdctImpl ctor) ---- */
typedef struct orc_mdct {
    int n, ld;
    float *a, *b, *c;      /* n/2, n/2, n/4 floats */
    uint16_t *bitrev;      /* n/8 entries */
} orc_mdct;

orc_mdct *orc_mdct_create(int n);
void orc_mdct_destroy(orc_mdct *m);
/* Mdct.cs:77-419: buffer has n floats (first n/2 = spectrum in, all n = PCM out), buf2 >= n/2 scratch */
void orc_mdct_reverse(const orc_mdct *m, float *buffer, float *buf2);
/* convenience: count rows, spectra [count][n/2] -> out [count][n] */
void orc_mdct_reverse_batch(int n, long count, const float *spectra, float *out);

/* ---- BlocksizeDerivedCache.cs:14-36 : slope has `half` (= blockSize/2) entries ---- */
void orc_window_slope(int half, float *slope);

/* ---- PacketInfo.cs:3-15, Mode.cs:30-66 ---- */
typedef struct orc_packet_info {
    int32_t Length;
    int32_t LeftUseSize1;
    int32_t LeftStart, LeftEnd;
    int32_t RightStart, RightEnd;
} orc_packet_info;
void orc_get_packet_info(int size0, int size1, int block_flag, int prev_flag, int next_flag,
                         orc_packet_info *info);

/* ---- Mapping.cs:198-269.  vector_form=1 follows the Vector<T> branch (:205-233, what runs on
 * SIMD hardware; differs from the scalar branch only in the sign of zero), 0 the scalar branch. */
void orc_apply_coupling(float *mag, float *ang, int n, int vector_form);

/* ---- Residue2.cs:42-51 : src [half][channels] -> dst planar, channel c at c*stride ---- */
void orc_residue2_deinterleave(const float *src, int half, int channels, float *dst, int stride);

/* ---- Floor1.cs static part (ctor :108-154) ---- */
typedef struct orc_floor1 {
    int count;              /* xList length (<= 65) */
    int multiplier, range;
    int xlist[65], lneigh[65], hneigh[65], sortidx[65];
} orc_floor1;
/* builds lNeigh/hNeigh/sortIdx from xList exactly as Floor1.cs:108-149; returns 0, or -1 on duplicate X */
int orc_floor1_init(orc_floor1 *f, const int *xlist, int count, int multiplier);
/* Floor1.cs:270-353: posts[0..post_count) raw -> final Y (in place), step_flags[64] */
void orc_floor1_unwrap_posts(const orc_floor1 *f, int *posts, int post_count, uint8_t *step_flags);
/* Floor1.cs:222-268: whole Apply (unwrap + render * residue[0..block_size/2)); mutates posts */
void orc_floor1_apply(const orc_floor1 *f, int *posts, int post_count, int block_size, float *residue);
/* render only, given final Y + step flags (for tests of the GPU render stage) */
void orc_floor1_render(const orc_floor1 *f, const int *final_y, const uint8_t *step_flags,
                       int post_count, int n, float *residue);
/* the same walk, but the table INDEX of every bin instead of the product: out_y[0..n), unclamped */
void orc_floor1_render_indices(const orc_floor1 *f, const int *final_y, const uint8_t *step_flags,
                               int post_count, int n, int *out_y);
const float *orc_floor1_inverse_db_table(void);

/* ---- Floor0.cs (LSP floor; "virtually unused", no fixture) ----
 * ctor tables :82-111 (bark map per block size, w map) and Apply :164-225.  coeff[order] and amp are
 * what Unpack (:113-162) leaves in Floor0.Data.  Two reference quirks are kept: the bark map leaves bin
 * n-1 at 0 (:88-93 loop bound), and the w map is indexed by the BARK value but sized by the block
 * (:99-109), so the reference throws when bark_map_size > n/2 -- the oracle then returns -1. */
typedef struct orc_floor0 {
    int order, rate, bark_map_size, amp_bits, amp_ofs;
} orc_floor0;
/* residue[0..block_size/2) *= curve; mutates coeff (coeff[j] = 2 cos(coeff[j]), :181-184).
 * returns 0, or -1 where the reference would throw IndexOutOfRangeException */
int orc_floor0_apply(const orc_floor0 *f, float *coeff, float amp, int block_size, float *residue);
/* the bark map of :82-95 for n = block_size/2 (n+1 entries) */
void orc_floor0_bark_map(const orc_floor0 *f, int n, int *map);

/* ---- Utils.cs:44-58 ---- */
float orc_clip_value(float v, int *clipped);

/* ---- Mapping.cs:166-195 (DecodePacket tail) for one packet.
 * buffer: planar, channel c at c*stride, first block_size/2 floats = decoded residue on entry,
 * first block_size floats = PCM (pre-window) on exit.  posts: [channels][64] raw posts,
 * post_count[channels] (0 => ExecuteChannel false).  floor_of_channel[channels] indexes floors[].
 * coupling steps applied in reverse order. */
void orc_mapping_synth(int channels, int block_size, float *buffer, int stride,
                       const orc_floor1 *floors, const int *floor_of_channel,
                       int *posts, const int *post_count,
                       const uint8_t *coupling_mag, const uint8_t *coupling_ang, int coupling_steps,
                       int coupling_vector_form);

/* ---- StreamDecoder.cs decode half: state (:45-49), ReadNextPacket (:640-694),
 * OverlapBuffers (:764-791), Read (:418-498), StoreInterleaved/StoreContiguous (:515-638). ---- */
typedef struct orc_stream orc_stream;
/* state as after ProcessHeaderPackets (StreamDecoder.cs:165-168): position 0, known */
orc_stream *orc_stream_create(int channels, int size0, int size1);
void orc_stream_destroy(orc_stream *s);
void orc_stream_reset(orc_stream *s);                     /* ResetDecoder :357-369 */
/* packet.IsResync seen by DecodeNextPacket (:718-722): `_hasPosition = false` */
void orc_stream_mark_resync(orc_stream *s);
/* StreamDecoder.SeekTo (:817-880) once the packet provider is positioned; see the definition */
typedef int (*orc_read_next_packet_fn)(void *user);
int orc_stream_seek_to(orc_stream *s, int64_t sample_position, int64_t provider_pos, int64_t max_granule_count,
                       orc_read_next_packet_fn read_next_packet, void *user);
/* buffer the decoder should fill for the next packet (planar, stride = size1) */
float *orc_stream_next_buffer(orc_stream *s);
/* ReadNextPacket with an already decoded packet in orc_stream_next_buffer().
 * decoded=0 models DecodeNextPacket returning null.  eos_flag: EndOfStreamFlags.PacketFlag.
 * granule = packet.GranulePosition (-1 if none).  Returns 1 if a packet was accepted. */
int orc_stream_read_next_packet(orc_stream *s, int decoded, const orc_packet_info *info,
                                int64_t granule, int eos_flag);
/* number of samples currently readable (_prevPacketEnd - _prevPacketStart) */
int orc_stream_available(const orc_stream *s);
/* drain after a failed EOS packet (:451-455) */
void orc_stream_drain_eos(orc_stream *s);
/* copy `count` readable samples out (count <= available) and advance.
 * interleave=1: dst[i*channels+ch] (StoreInterleaved); 0: dst[ch*channel_stride+offset+i]. */
void orc_stream_store(orc_stream *s, float *dst, long offset, int count, long channel_stride,
                      int interleave, int clip);
int orc_stream_has_clipped(const orc_stream *s);
int64_t orc_stream_position(const orc_stream *s);

/* ---- whole-path batch drivers used by parity tests and the cpu_baseline timing ----
 * One stream, `frames` packets, planar spectra: frame f channel c at
 * spectra[(f*channels + c)*(size1/2) ...] (only first blocksize/2 used).  flags[f] bit0 block_flag,
 * bit1 prev, bit2 next.  Output planar pcm[c*pcm_stride + t]; returns samples per channel. */
long orc_synth_stream_planar(int channels, int size0, int size1, long frames, const uint8_t *flags,
                             const float *spectra, float *pcm, long pcm_stride, int clip);

/* One stream's packets through the restated Mapping.DecodePacket tail + StreamDecoder, in C, with the Mdct tables warm:
 * bench.py's CPU baseline of the fused workloads.  See the definition for the argument layout. */
long orc_synth_stream_floored(int channels, int size0, int size1, long n_packets, const uint8_t *flags,
                              const uint8_t *mapping, const int64_t *granule, const int64_t *residue_offset,
                              const float *residue, const orc_floor1 *floors, const int *floor_of_channel,
                              const uint8_t *coupling_mag, const uint8_t *coupling_ang, const int *coupling_off,
                              const int16_t *posts, const uint8_t *post_count, float *pcm, long pcm_stride, int clip);

#ifdef __cplusplus
}
#endif
#endif
