/*
 * vorbispizza_reader.h -- C entry points (libvorbispizza_host.so) of the host-side mirror of VorbisReader / StreamDecoder.Read
 * (VorbisReader.cs:232-253, StreamDecoder.cs:407-498) built above the C ABI.  See vorbis_reader.cpp.
 */
#ifndef VORBISPIZZA_READER_H
#define VORBISPIZZA_READER_H

#include <stdint.h>

#include "vorbispizza_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vpzr_reader vpzr_reader;

/* `new VorbisReader(stream); Initialize()` on an in-memory .ogg; decoding runs on `ctx`'s GPU. */
int  vpzr_open_memory(vpz_context *ctx, const uint8_t *data, uint64_t size, vpzr_reader **out);
void vpzr_close(vpzr_reader *r);
const char *vpzr_last_error(vpzr_reader *r);

/* Logical streams of the container (VorbisReader.Streams / FindNextStream / SwitchStreams, VorbisReader.cs:191-217):
 * the reader opens the first one; vpzr_find_next_stream looks for one more (a chained file starts it after the
 * previous stream's last page) and returns 1 when it added one; vpzr_switch_streams makes stream `index` the one
 * every other call below talks to and returns 1 when its channel count or sample rate differ from the previous
 * one's, 0 when they do not, VPZ_E_INVALID_ARG for a bad index.  Every stream keeps its own decoder and position. */
int     vpzr_find_next_stream(vpzr_reader *r);
int     vpzr_stream_count(vpzr_reader *r);
int     vpzr_switch_streams(vpzr_reader *r, int index);
int     vpzr_stream_serial(vpzr_reader *r);        /* IStreamDecoder.StreamSerial of the current stream */

int     vpzr_channels(vpzr_reader *r);             /* IVorbisReader.Channels */
int     vpzr_sample_rate(vpzr_reader *r);          /* IVorbisReader.SampleRate */
int64_t vpzr_sample_position(vpzr_reader *r);      /* samples per channel handed out so far */
int     vpzr_is_end_of_stream(vpzr_reader *r);
int     vpzr_has_clipped(vpzr_reader *r);          /* IStreamDecoder.HasClipped */
int     vpzr_set_clip_samples(vpzr_reader *r, int clip);      /* ClipSamples; default true (VorbisReader.cs:71) */
int     vpzr_set_batch_packets(vpzr_reader *r, int packets);  /* packets synthesised per GPU call (default 128) */
/* Sample format of the read calls, fixed before the first read: float32 (default) or the 16-bit samples the reference's
 * tests derive from them, `(int)(x * 32768f)` clamped (AssetTest.cs:131-132) -- converted on the GPU. */
#define VPZR_FORMAT_F32 0
#define VPZR_FORMAT_S16 1
int     vpzr_set_sample_format(vpzr_reader *r, int format);

/* `SeekTo(long samplePosition, SeekOrigin seekOrigin)` (StreamDecoder.cs:815-881) and `TotalSamples`.
 * Positions are counted samples per channel (see vpzh_seek).  VPZ_E_INVALID_ARG outside the stream
 * (SeekOutOfRangeException / ArgumentOutOfRangeException) or when the pre-roll cannot be read. */
#define VPZR_SEEK_BEGIN   0
#define VPZR_SEEK_CURRENT 1
#define VPZR_SEEK_END     2
int     vpzr_seek_to(vpzr_reader *r, int64_t sample_position, int origin);
int64_t vpzr_total_samples(vpzr_reader *r);

/* `ReadSamples(Span<float> buffer)`: interleaved, returns samples per channel, at most one packet's
 * worth per call, 0 at the end of the stream.  *status receives a VPZ_* code. */
int64_t vpzr_read_samples(vpzr_reader *r, float *buffer, int64_t buffer_len, int *status);
/* The same call delivering 16-bit samples (VPZR_FORMAT_S16 readers only). */
int64_t vpzr_read_samples_s16(vpzr_reader *r, int16_t *buffer, int64_t buffer_len, int *status);
/* `ReadSamples(Span<float> buffer, int samplesToRead, int channelStride)`: planar. */
int64_t vpzr_read_samples_planar(vpzr_reader *r, float *buffer, int64_t buffer_len, int64_t samples_to_read,
                                 int64_t channel_stride, int *status);

#ifdef __cplusplus
}
#endif
#endif
