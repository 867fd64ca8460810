/*
 * vorbispizza_front.h -- C entry points of the C++ CPU front end, libvorbispizza_host.so.
 *
 * This is the part the real host (the C# VorbisReader / StreamDecoder) keeps on the CPU: Ogg page sync +
 * CRC + lacing -> packets (Ogg/PageReaderBase.cs, Ogg/PacketProvider.cs:427-560), the three Vorbis
 * headers (StreamDecoder.cs:213-353: codebooks, floors, residues, mappings, modes) and, per audio
 * packet, the bit-serial entropy decode up to Mapping.cs:163 (mode bits, Floor1.Unpack, Residue0/1/2
 * decode).  It produces exactly the arrays vpz_decoder_synth takes, so that real .ogg files can be pushed
 * through the GPU back end without a .NET host.  Integer / Huffman stage only: bit-exact by construction,
 * no sample arithmetic beyond the codebook vector accumulation the reference does in Residue*.cs.
 */
#ifndef VORBISPIZZA_FRONT_H
#define VORBISPIZZA_FRONT_H

#include <stdint.h>

#include "vorbispizza_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VPZH_OK             0
#define VPZH_E_INVALID_DATA (-1)  /* InvalidDataException in the reference */
#define VPZH_E_UNSUPPORTED  (-2)
#define VPZH_E_ARG          (-3)
#define VPZH_E_NO_STREAM    (-4)  /* the container has no logical stream of that index (FindNextStream() == false) */

typedef struct vpzh_stream vpzh_stream;

/* Parses the container and the three header packets of the first logical stream in `data`. */
int  vpzh_open_memory(const uint8_t *data, uint64_t size, vpzh_stream **out);
/* The same for the `stream_index`-th logical stream, counted by beginning-of-stream pages: chained files
 * (VorbisReader.FindNextStream / SwitchStreams, VorbisReader.cs:191-217) are decoded stream by stream, each with its
 * own setup headers.  VPZH_E_NO_STREAM when there is none. */
int  vpzh_open_memory_stream(const uint8_t *data, uint64_t size, int32_t stream_index, vpzh_stream **out);
void vpzh_close(vpzh_stream *s);
const char *vpzh_last_error(vpzh_stream *s);

typedef struct vpzh_info {
    int32_t channels, sample_rate, block_size0, block_size1;
    int32_t floor_count, residue_count, mapping_count, mode_count, codebook_count;
    int64_t audio_packets;      /* packets after the three headers */
    int64_t last_granule;       /* granule position of the last page == total samples per channel */
    int64_t residue_floats;     /* floats vpzh_decode_all writes to `residue` */
    int32_t pages, bad_crc_pages;
    int32_t stream_serial;      /* IStreamDecoder.StreamSerial */
    int32_t reserved;
} vpzh_info;
int vpzh_get_info(vpzh_stream *s, vpzh_info *info);

/* setup products the synthesis back end needs (vpz_stream_config) */
int vpzh_get_floor_type(vpzh_stream *s, int index);   /* 0 or 1 */
int vpzh_get_floor1(vpzh_stream *s, int index, vpz_floor1_config *out);
int vpzh_get_floor0(vpzh_stream *s, int index, vpz_floor0_config *out);
int vpzh_max_floor0_order(vpzh_stream *s);            /* 0 when the stream has no type-0 floor */
int vpzh_get_mapping(vpzh_stream *s, int index, vpz_mapping_config *out);
int vpzh_get_residue_type(vpzh_stream *s, int index);

/* Seeking (PacketProvider.SeekTo / GetGranuleCount, Ogg/PacketProvider.cs:35-160).  Positions are counted
 * samples: the first audio packet only primes the overlap, every later packet adds PacketInfo.SampleCount.
 * vpzh_seek returns the index of the PRE-ROLL packet (decode from there) and how many samples of the packet
 * after it lie before `sample_position`; VPZH_E_ARG when the position is outside the stream
 * (SeekOutOfRangeException). */
int64_t vpzh_total_samples(vpzh_stream *s);
int vpzh_seek(vpzh_stream *s, int64_t sample_position, int64_t *first_packet, int64_t *roll_forward);

/* Entropy-decodes every audio packet.  packets[audio_packets], residue[residue_floats],
 * posts[audio_packets*channels*64], post_counts[audio_packets*channels].  `stream_id` is written to
 * vpz_packet.stream, `residue_base` is added to every residue_offset.  Packets whose first bit is set
 * or that fail GetPacketInfo get VPZ_PKT_NOT_DECODED (StreamDecoder.cs:728, 750-761); packets completed on a page
 * that was found after lost sync (bad checksum, garbage between pages) or out of sequence get VPZ_PKT_RESYNC
 * (VorbisPacket.IsResync; StreamDecoder.cs:718-722 then picks the position up again). */
int vpzh_decode_all(vpzh_stream *s, int32_t stream_id, int64_t residue_base, vpz_packet *packets,
                    float *residue, int16_t *posts, uint8_t *post_counts);

/* Same for the packet range [first, first+count); *residue_floats_used receives the floats written.
 * `residue` must hold channels * block_size1/2 * count floats in the worst case. */
int vpzh_decode_range(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                      vpz_packet *packets, float *residue, int16_t *posts, uint8_t *post_counts,
                      int64_t *residue_floats_used);

/* With the Floor0.Data of type-0 floor channels: f0_amp[count*channels], f0_coeff[count*channels*f0_stride]
 * (f0_stride >= vpzh_max_floor0_order); pass NULL when the stream has no type-0 floor. */
int vpzh_decode_range_ex(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                         vpz_packet *packets, float *residue, int16_t *posts, uint8_t *post_counts,
                         int64_t *residue_floats_used, float *f0_amp, float *f0_coeff, int32_t f0_stride);

/* The residue as 16-bit integers.  A residue value is a sum of at most one codebook value per cascade stage; libvorbis' residue books
 * are integer lattices, so for its streams every value is a (small) integer, the same in float32 and in int16 -- and half the bytes on
 * their way to the device (vpz_decoder_set_residue_format(VPZ_RESIDUE_I16)).  vpzh_residue_is_integral: 1 when the setup header
 * guarantees it (every residue value book holds integers only, the worst-case sum stays below 2^15), else 0.
 * vpzh_decode_range_i16: vpzh_decode_range_ex with `residue` as int16 values (offsets and counts in VALUES); VPZH_E_ARG for a stream
 * whose residue is not integral. */
int vpzh_residue_is_integral(vpzh_stream *s);
int vpzh_decode_range_i16(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                          vpz_packet *packets, int16_t *residue, int16_t *posts, uint8_t *post_counts,
                          int64_t *residue_values_used, float *f0_amp, float *f0_coeff, int32_t f0_stride);

/* Many containers at once -- what a host that transcodes a library of files does: opens and entropy-decodes `n` in-memory
 * containers (first logical stream of each) on `threads` host threads (0: vpzh_default_threads), one stream at a
 * time per thread, the reference's model of one decoder per stream.  Stream k writes its packets at packets + packet_base[k]
 * (records of posts / post_counts at packet_base[k] * channels), its residue at residue + residue_base[k]; its packets carry
 * stream id stream_id0 + k and residue offsets relative to residue + residue_origin (the start of the buffer a later
 * vpz_decoder_synth call is given).  The caller sizes the slices from a vpzh_get_info of each distinct file and says how much
 * room each has (packet_room[k] packets, residue_room[k] floats): a container that holds more is refused (VPZH_E_ARG, nothing of
 * it written; the other streams are still decoded).  All streams must have `channels` channels -- the post records of the batch are
 * laid out for that count: a container with another one is refused the same way (done[k] = -1).  threads 0: vpzh_default_threads().  failed_packets (may be
 * NULL): packets whose decode failed (see vpzh_decode_failures).  Type-0 floor data is not collected here. */
int vpzh_decode_many(int32_t n, int32_t channels, const uint8_t *const *data, const uint64_t *size, int32_t threads, int32_t stream_id0,
                     const int64_t *packet_base, const int64_t *packet_room, const int64_t *residue_base,
                     const int64_t *residue_room, int64_t residue_origin, vpz_packet *packets, float *residue, int16_t *posts,
                     uint8_t *post_counts, int64_t *failed_packets);

/* The CPUs this process may use -- its affinity mask, capped by the container's CPU-time quota (cgroup cpu.max) -- divided by
 * LOCAL_WORLD_SIZE (processes per node under torchrun), >= 1. */
int vpzh_default_threads(void);

/* The same with a progress report, for a caller that hands finished streams on (to vpz_decoder_synth) while the rest is still
 * being decoded: done[k] (caller-zeroed, n entries) becomes 1 once stream k's slices are complete, -1 if it was refused or
 * failed to open; streams are taken in index order, one per thread at a time.  The flags are stored with release semantics --
 * read them from another thread, see non-zero, then read the slices. */
int vpzh_decode_many_progress(int32_t n, int32_t channels, const uint8_t *const *data, const uint64_t *size, int32_t threads, int32_t stream_id0,
                              const int64_t *packet_base, const int64_t *packet_room, const int64_t *residue_base,
                              const int64_t *residue_room, int64_t residue_origin, vpz_packet *packets, float *residue,
                              int16_t *posts, uint8_t *post_counts, int64_t *failed_packets, int32_t *done);

/* Packets of the LAST vpzh_decode_range* / vpzh_decode_all call whose entropy decode failed the way the reference's
 * DecodeNextPacket throws (InvalidDataException "Unused mode index.", a residue vector overrunning its block, ...).
 * Such a packet costs only itself, like the reference's exception: it is handed over with VPZ_PKT_NOT_DECODED and
 * WITHOUT its EOS flag (the reference has not executed `_eosFound |= isEndOfStream` when the exception leaves), the
 * rest of the range decodes normally and the call returns VPZH_OK.  Returns the number of such packets;
 * *first_failed_packet = index of the first one relative to `first` (-1 if none); vpzh_last_error has its text. */
int64_t vpzh_decode_failures(vpzh_stream *s, int64_t *first_failed_packet);

#ifdef __cplusplus
}
#endif
#endif
