/*
 * vorbispizza_multi.h -- in-process multi-device dispatcher (libvorbispizza_host.so): ONE host process decodes a library of
 * Ogg/Vorbis containers on several MI355X at once.
 *
 * The reference's host is one process: a VorbisReader holds N StreamDecoders (VorbisReader.cs:56-85), each stream is
 * independent of every other one.  This is that shape above the C ABI: the streams are partitioned
 * contiguously over the devices (stream k of n -> device k * n_devices / n, the rule of vorbispizza_amd/sharding.py:
 * shard_range); every device has its own host thread, its own vpz_context(s) and HIP stream(s), its own share of the
 * entropy-decode threads, its own vpz_decoder per setup header -- and nothing is exchanged between devices: no collective, no
 * peer copy, xGMI stays idle.  The per-device pipeline is the one bench.py's end-to-end leg measures: containers are opened and
 * entropy-decoded on host threads straight into page-locked batch arrays (one stream at a time per thread, the reference's
 * model), sub-batches of streams go to the GPU as one host-memory vpz_decoder_synth call each, issued while the following
 * sub-batches are still being decoded, by `contexts_per_device` issuing threads in turn so that the upload of one call
 * overlaps the download of the previous one.
 *
 * Everything a C# host would P/Invoke is cdecl / POD / int status, like vorbispizza_synth.h.
 */
#ifndef VORBISPIZZA_MULTI_H
#define VORBISPIZZA_MULTI_H

#include <stdint.h>

#include "vorbispizza_front.h"
#include "vorbispizza_synth.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VPZM_OK            0
#define VPZM_E_ARG        (-1)
#define VPZM_E_DEVICE     (-2)   /* a context could not be created on one of the devices; text via vpzm_last_error */
#define VPZM_E_NOMEM      (-3)
/* per-stream status (vpzm_stream_result.status): VPZM_OK, or */
#define VPZM_E_OPEN       (-10)  /* the container could not be opened (vpzh_open_memory failed) */
#define VPZM_E_CAPACITY   (-11)  /* pcm_capacity[k] is smaller than the stream's sample count -- the announced one (vpzh_total_samples), or what a damaged stream that lost its end-of-stream trim really produces */
#define VPZM_E_SYNTH      (-12)  /* the stream's synthesis failed (after the call that held it failed, every member is synthesised alone: its own outcome) */
#define VPZM_E_SETUP      (-13)  /* a setup the back end cannot represent (e.g. a Floor1 with more than 64 posts) */

typedef struct vpzm_dispatcher vpzm_dispatcher;

typedef struct vpzm_options {
    int32_t host_threads;         /* entropy-decode threads over ALL devices (0: vpzh_default_threads() plus one per context -- the issuing threads mostly wait); each device gets its share */
    int32_t streams_per_call;     /* streams per vpz_decoder_synth call (0: 16); a call also closes at 64 Mi residue values, so whole songs ride in fewer per call */
    int32_t contexts_per_device;  /* contexts -- HIP streams, issuing threads -- that take a device's calls in turn (0: 4 when the device has 8 or more host threads, else 2) */
    int32_t clip_samples;         /* StreamDecoder.ClipSamples (VorbisReader sets it to true, VorbisReader.cs:71) */
    int32_t slots_per_device;     /* sub-batches in flight per device: decoded or being decoded ahead of their synth call (0: 4 * contexts + 4; about 70 MB of page-locked memory each for stereo streams of a few seconds -- with fewer the entropy decode waits for synth calls) */
    int32_t float_residue;        /* 0 (default): the residue of streams whose setup header guarantees 16-bit integers (vpzh_residue_is_integral: every libvorbis stream) crosses the host link as int16 -- the same values at half the bytes; non-zero: always float32 */
    int32_t reserved[2];
} vpzm_options;

/* One context group per entry of device_ids (an id may appear more than once: several groups on one GPU, which is how a
 * one-GPU box rehearses the N-device path).  opt may be NULL (all defaults). */
int  vpzm_create(const int32_t *device_ids, int32_t n_devices, const vpzm_options *opt, vpzm_dispatcher **out);
void vpzm_destroy(vpzm_dispatcher *m);
const char *vpzm_last_error(vpzm_dispatcher *m);
int  vpzm_device_count(vpzm_dispatcher *m);

typedef struct vpzm_stream_result {
    int32_t status;        /* VPZM_OK or a per-stream VPZM_E_* */
    int32_t device_slot;   /* index into device_ids of the group that decoded the stream */
    int32_t channels;
    int32_t sample_rate;
    int64_t samples;       /* samples per channel written at pcm_out + pcm_offset[k] */
    int64_t packets;       /* audio packets of the stream */
    int64_t skipped_packets;  /* packets the window check skipped (StreamDecoder.cs:777-778) or whose entropy decode failed */
} vpzm_stream_result;

typedef struct vpzm_stats {
    double wall_s;                /* the whole call */
    double device_wall_s[16];     /* per device group (the first 16): its thread's wall time */
    double device_decode_s[16];   /* ... until its last stream was entropy-decoded */
    double device_synth_s[16];    /* ... summed time inside vpz_decoder_synth (over its issuing threads) */
    int64_t device_streams[16];
    int64_t device_samples[16];   /* samples x channels produced */
    int32_t threads_per_device;
    int32_t pinned_mib;           /* page-locked host memory the dispatcher's slots hold after this call, all groups, in MiB: it grows to
                                     slots_per_device sub-batches of the largest size seen per group and is kept until vpzm_destroy */
} vpzm_stats;

/* Decodes containers 0..n-1 (first logical stream of each) to interleaved PCM in host memory: stream k's sample s of
 * channel c at pcm_out[pcm_offset[k] + s * channels + c] -- float32 for VPZ_OUT_INTERLEAVED, int16 for
 * VPZ_OUT_INTERLEAVED_S16 (`(int)(x * 32768f)` clamped, AssetTest.cs:131-132); pcm_offset counts elements of that type.
 * pcm_capacity[k]: samples per channel the stream's area holds (vpzh_total_samples of the file, or more).  pcm_out should
 * be page-locked (vpz_host_alloc) for the link's full rate; any host memory works.
 * Streams may differ in setup headers, channel counts and block sizes: streams of one setup ride in the same calls.
 * results[n] (required) receives every stream's outcome; a stream that fails costs only itself.  Returns VPZM_OK when the
 * job ran (look at the per-stream statuses), VPZM_E_ARG for bad arguments.  A dispatcher runs one call at a time: calls from
 * several host threads are safe and take it in turn (wall_s then counts from the moment the call got the dispatcher);
 * vpzm_destroy must not race a call. */
int vpzm_decode_library(vpzm_dispatcher *m, int32_t n, const uint8_t *const *data, const uint64_t *size, int32_t out_layout,
                        void *pcm_out, const int64_t *pcm_offset, const int64_t *pcm_capacity, vpzm_stream_result *results,
                        vpzm_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
