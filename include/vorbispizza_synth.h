/*
 * vorbispizza_synth.h -- C ABI of the MI355X-native Vorbis PCM-synthesis back end.
 *
 * This is the drop-in boundary for ONE hot path of TechPizzaDev/VorbisPizza (C#): everything that
 * happens to a packet after its CPU entropy decode (Mapping.cs:109-163) until PCM leaves
 * StreamDecoder.Read:  inverse coupling -> Floor1 curve x residue -> inverse MDCT ->
 * window + overlap-add -> clip -> interleaved / planar store.
 *
 * The reference has no FFI seam for this path (the callee types are `internal`); each entry point
 * below names the reference call it replaces.  A C# host binds these with
 * [DllImport("vorbispizza_synth", CallingConvention = CallingConvention.Cdecl)] exactly as
 * NVorbis.Tests/Bindings/Vorbisfile.cs:43-107 binds libvorbisfile.
 *
 * Conventions (mirroring that binding): cdecl, POD structs with sequential layout, `int` status
 * returns (0 = OK, < 0 = error), caller-owned I/O buffers valid for the duration of the call, opaque
 * handles owned by the caller (SafeHandle on the C# side).  The library never calls back into the
 * host and never throws.  A context / decoder is used from one thread at a time; distinct contexts
 * (one per GPU) may run concurrently.  There is NO CPU fallback: every compute entry point fails
 * with VPZ_E_NO_DEVICE / VPZ_E_HIP when no gfx950 device is usable.
 */
#ifndef VORBISPIZZA_SYNTH_H
#define VORBISPIZZA_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPZ_ABI_VERSION 6   /* 2: s16 output layouts, VPZ_PKT_RESYNC, vpz_decoder_set_position; structs unchanged
                               3: vpz_decoder_synth takes the extents of its input buffers (residue_floats, n_records) and
                                  reports a window mismatch per packet (vpz_decoder_last_packet_status) instead of failing
                                  the batch; structs unchanged
                               4: vpz_mapping_config carries the residue's support (residue_begin / residue_end, what
                                  Residue0.cs:122-125 clamps every decode to): the kernels neither load nor de-couple nor
                                  floor-multiply the bins the setup header says are zero.  The struct grew by 16 bytes
                               5: vpz_decoder_set_residue_format: the residue may be handed over as 16-bit integers
                               6: vpz_decoder_set_host_threads: the host that runs several decoders at once gives each its share of
                                  the host's threads.  Nothing else changed: a v5 caller works unchanged */

/* ---- status codes (negative like the OV_* codes, Vorbisfile.cs:10-24) ---- */
#define VPZ_OK                 0
#define VPZ_E_INVALID_ARG     (-1)  /* null pointer, bad size, bad index */
#define VPZ_E_UNSUPPORTED     (-2)  /* block size / layout this build has no kernel for */
#define VPZ_E_HIP             (-3)  /* HIP runtime error; text via vpz_context_last_error */
#define VPZ_E_NOMEM           (-4)
#define VPZ_E_WINDOW_MISMATCH (-5)  /* PER-PACKET status (vpz_decoder_last_packet_status), never the return value of
                                       vpz_decoder_synth: the previous tail is longer than the packet's window slope,
                                       where StreamDecoder.cs:777-778 would throw out of that packet's Read.  Like that
                                       exception the condition costs only the offending packet: it is skipped, the
                                       stream state is untouched, and the rest of the batch -- other streams included --
                                       is synthesised */
#define VPZ_E_NO_DEVICE       (-6)
#define VPZ_E_CAPACITY        (-7)  /* output buffer too small for the samples produced */

/* ---- where caller pointers live ---- */
#define VPZ_MEM_HOST   0   /* host memory (what a pinned C# array is); call is synchronous */
#define VPZ_MEM_DEVICE 1   /* device memory on the context's GPU; call is asynchronous on the
                              context stream, order with vpz_context_synchronize.  The library never
                              looks at other streams: whoever produced the input buffers on another
                              HIP stream finishes (or synchronises) that work before the call, and
                              consumers of the output wait for vpz_context_synchronize or an event on
                              vpz_context_stream */

typedef struct vpz_context vpz_context;   /* one GPU + one HIP stream + tables */
typedef struct vpz_decoder vpz_decoder;   /* synthesis state of a group of streams */

int         vpz_abi_version(void);
const char *vpz_error_string(int status);

/* Number of usable devices (0 when there is no GPU); never fails. */
int vpz_device_count(void);

/* Context = what `Mdct._setupCache` (Mdct.cs:13) and `StreamDecoder._blockSizeCache`
 * (StreamDecoder.cs:29,226-229) are in the reference: immutable per-block-size tables, plus the
 * device stream the work runs on. */
int  vpz_context_create(int device_id, vpz_context **out);
void vpz_context_destroy(vpz_context *ctx);
int  vpz_context_synchronize(vpz_context *ctx);
const char *vpz_context_last_error(vpz_context *ctx);
/* hipStream_t of the context (for callers that time with their own events) */
void *vpz_context_stream(vpz_context *ctx);
/* HIP-event timer on the context stream: start .. stop brackets whatever was enqueued between. */
int vpz_context_timer_start(vpz_context *ctx);
int vpz_context_timer_stop(vpz_context *ctx, float *elapsed_ms);

/* Device memory helpers so a host without its own HIP binding can keep batches resident. */
int vpz_device_alloc(vpz_context *ctx, uint64_t bytes, void **dev_ptr);
int vpz_device_free(vpz_context *ctx, void *dev_ptr);
/* Page-locked host memory (ABI v4): VPZ_MEM_HOST calls copy from / to it at the link's full rate and asynchronously; any
 * other host memory works too, through the runtime's staging.  Usable from every context of the process.  These two calls
 * are the exception to "one thread at a time per context": they write nothing of the context (no error text either -- the
 * status is all there is) and may run while another thread is inside a call on it. */
int vpz_host_alloc(vpz_context *ctx, uint64_t bytes, void **host_ptr);
int vpz_host_free(vpz_context *ctx, void *host_ptr);
int vpz_memcpy_h2d(vpz_context *ctx, void *dev_dst, const void *host_src, uint64_t bytes);
int vpz_memcpy_d2h(vpz_context *ctx, void *host_dst, const void *dev_src, uint64_t bytes);

/* ------------------------------------------------------------------------------------------
 * vpz_imdct_batch  ==  `Mdct.Reverse(samples, buf2, n)` (Mdct.cs:15-19) applied to `count` rows.
 * spectra: [count][n/2] float32 (the first half of each reference `samples` span);
 * out:     [count][n]   float32 (the whole span after the call).  n is a power of two,
 * 64 <= n <= 8192 (Vorbis block sizes).  n = 64/128 reproduce the reference's literal (non-IMDCT)
 * output, quirk q1 of SURVEY.md 7.2.  `mode`: VPZ_IMDCT_FAST uses the wavefront FFT factorisation
 * (<= 1e-5 abs error for |PCM| <= 1; every n from 256 up -- 64 and 128 fall back to EXACT);
 * VPZ_IMDCT_EXACT runs the reference's own butterfly schedule (Mdct.cs:98-414) and is bit-identical
 * to it.
 * ------------------------------------------------------------------------------------------ */
#define VPZ_IMDCT_FAST  0
#define VPZ_IMDCT_EXACT 1
int vpz_imdct_batch(vpz_context *ctx, int n, int64_t count, const float *spectra, float *out,
                    int mem_space, int mode);

/* ------------------------------------------------------------------------------------------
 * Stream configuration: the setup-header products the synthesis path reads.
 * ------------------------------------------------------------------------------------------ */
#define VPZ_MAX_FLOOR1_POSTS 65   /* array bound of vpz_floor1_config.x_list (the specification's maximum) */
#define VPZ_POSTS_STRIDE     64   /* `Data.Posts = new int[64]`, Floor1.cs:17: the reference cannot hold a 65th post
                                     (its Unpack would throw), so x_count <= 64 is what every entry point accepts */
#define VPZ_MAX_CHANNELS     255
#define VPZ_MAX_COUPLING     256

typedef struct vpz_floor1_config {        /* Floor1.cs:30-31 (`_xList`, `_multiplier`) */
    int32_t x_count;                      /* length of _xList, 2..VPZ_POSTS_STRIDE */
    int32_t multiplier;                   /* _multiplier, 1..4 (range = {256,128,86,64}) */
    int32_t x_list[VPZ_MAX_FLOOR1_POSTS]; /* _xList in bitstream order, distinct values; x_list[0] must be 0 and
                                             x_list[1] > 0 as Floor1.cs:96-97 constructs them (0 and 1 << rangeBits);
                                             the library derives _lNeigh/_hNeigh/_sortIdx as Floor1.cs:108-149 does */
} vpz_floor1_config;

typedef struct vpz_floor0_config {        /* Floor0.cs:29-35 (LSP floor, "virtually unused") */
    int32_t order;                        /* _order, 1..255 */
    int32_t rate;                         /* _rate */
    int32_t bark_map_size;                /* _bark_map_size */
    int32_t amp_bits, amp_ofs;            /* _ampBits, _ampOfs */
} vpz_floor0_config;

typedef struct vpz_mapping_config {       /* Mapping.cs:11-15 */
    int32_t coupling_steps;
    uint8_t coupling_magnitude[VPZ_MAX_COUPLING];
    uint8_t coupling_angle[VPZ_MAX_COUPLING];
    uint8_t channel_floor[VPZ_MAX_CHANNELS + 1]; /* _submapFloor[_mux[ch]] per channel */
    /* The residue's SUPPORT, per block size ([0]: block_size0, [1]: block_size1), in bins of one channel: every packet of
     * this mapping has residue[bin] == +0.0 for bin < residue_begin or bin >= residue_end, in every channel.  It is a
     * setup-header product: a residue decodes into [min(_begin, n/2), min(_end, n/2)) and nowhere else (Residue0.cs:122-125;
     * the buffer was cleared before, Mapping.cs:117), so the host fills in the smallest begin and the largest end over the
     * mapping's submaps (`_submapResidue`, Mapping.cs:13) -- for a type-2 residue, whose vector interleaves the submap's
     * channels (Residue2.cs:31-34), its [begin, end) divided by that channel count, rounded outward.  The library then
     * does not load, de-couple or floor-multiply bins beyond the support (it works in steps of blocksize/16 bins); what
     * the buffer holds there is not looked at.  residue_end == 0 means "not stated": the whole block.  Values above
     * blocksize/2 are clamped; begin > end is VPZ_E_INVALID_ARG.  VPZ_PKT_NO_FLOOR packets have no mapping: whole block. */
    int32_t residue_begin[2];
    int32_t residue_end[2];
} vpz_mapping_config;

typedef struct vpz_stream_config {
    int32_t channels;                     /* StreamDecoder._channels */
    int32_t block_size0, block_size1;     /* BlockSizes.Size0 / Size1 */
    int32_t floor_count;
    const vpz_floor1_config *floors;
    int32_t mapping_count;
    const vpz_mapping_config *mappings;
    int32_t clip_samples;                 /* StreamDecoder.ClipSamples (StreamDecoder.cs:993) */
    /* floor types: NULL = every floor is type 1.  Otherwise floor_types[i] in {0, 1} for each of the
     * floor_count floors; floors[i] is read for type 1 and floors0[i] for type 0 (same index). */
    const uint8_t *floor_types;
    const vpz_floor0_config *floors0;
} vpz_stream_config;

/* One audio packet as the CPU stage leaves it at Mapping.cs:163. */
#define VPZ_PKT_BLOCK_FLAG   0x01  /* Mode._blockFlag */
#define VPZ_PKT_PREV_FLAG    0x02  /* Mode.cs:39 first bit  (long blocks only) */
#define VPZ_PKT_NEXT_FLAG    0x04  /* Mode.cs:39 second bit (long blocks only) */
#define VPZ_PKT_EOS          0x08  /* packet.IsEndOfStream -> EndOfStreamFlags.PacketFlag */
#define VPZ_PKT_NOT_DECODED  0x10  /* DecodeNextPacket returned null (StreamDecoder.cs:758-761) */
#define VPZ_PKT_INTERLEAVED  0x20  /* residue is the Residue2 vector [n/2][channels]
                                      (Residue2.cs:31-34) instead of planar [channels][n/2] */
#define VPZ_PKT_NO_FLOOR     0x40  /* residue already is the floored spectrum: skip coupling and
                                      Floor1 (boundary variant "between Mapping.cs:187 and :188") */
#define VPZ_PKT_RESYNC       0x80  /* packet.IsResync (StreamDecoder.cs:718-722): the container lost sync before
                                      this packet, so `_hasPosition = false` -- the stream picks its position up
                                      again from the next packet that carries a granule position (:459-463).
                                      Applies to not-decoded packets as well (the check precedes the type bit) */
typedef struct vpz_packet {
    int32_t stream;          /* 0 .. n_streams-1 */
    uint8_t flags;           /* VPZ_PKT_* */
    uint8_t mapping;         /* index into vpz_stream_config.mappings (Mode._mapping) */
    uint16_t reserved;
    int64_t granule;         /* packet.GranulePosition, -1 if none (StreamDecoder.cs:744) */
    int64_t residue_offset;  /* float index of this packet's residue in the `residue` buffer:
                                channels * blocksize/2 floats */
} vpz_packet;

/* Output placement == the two public Read overloads (Contracts/IStreamDecoder.cs:126,151). */
#define VPZ_OUT_INTERLEAVED 0   /* StoreInterleaved: dst[i*channels + ch]            */
#define VPZ_OUT_PLANAR      1   /* StoreContiguous : dst[ch*channel_stride + off + i] */
/* The same two placements with 16-bit samples: `pcm_out` points to int16_t, every sample is what the reference's own
 * tests compute from the float -- `(int)(x * 32768f)` clamped to the short range (NVorbis.Tests/AssetTest.cs:131-132),
 * applied after the optional clip -- and stream_out_offset / stream_out_capacity / channel_stride count int16
 * elements.  Fused into the store: half the PCM bytes leave the GPU. */
#define VPZ_OUT_INTERLEAVED_S16 2
#define VPZ_OUT_PLANAR_S16      3

int  vpz_decoder_create(vpz_context *ctx, const vpz_stream_config *cfg, int32_t n_streams,
                        vpz_decoder **out);
void vpz_decoder_destroy(vpz_decoder *dec);
/* `StreamDecoder.ResetDecoder` (StreamDecoder.cs:357-369) for one stream; stream < 0 = all. */
int  vpz_decoder_reset(vpz_decoder *dec, int32_t stream);

/* ------------------------------------------------------------------------------------------
 * vpz_decoder_synth == for each packet, in order:  Mapping.DecodePacket tail (Mapping.cs:166-195)
 * + StreamDecoder.ReadNextPacket / OverlapBuffers (StreamDecoder.cs:640-694, 764-791) + the
 * Store* call of Read (:474-489), i.e. everything `while (idx == 0)` does for that packet.
 *
 * packets[n_packets]: packets of one stream must appear in stream order; streams may interleave.
 * residue: float32, addressed by vpz_packet.residue_offset; residue_floats = floats readable at `residue`.  A packet
 *   whose residue (channels * blocksize/2 floats from its offset) reaches beyond that is an error
 *   (VPZ_E_INVALID_ARG, nothing is synthesised, no state changes) -- the reference's arguments are Span<float>,
 *   bounds-checked the same way (Mapping.cs:98).
 * posts / post_counts: per packet p and channel c, record r = p*channels + c:
 *   post_counts[r] = Floor1.Data.PostCount (0 => ExecuteChannel false, channel outputs zeros),
 *   posts[r*64 + i] = Floor1.Data.Posts[i] as `Unpack` left them (raw, before UnwrapPosts).
 *   n_records = records readable at both (n_records >= n_packets * channels, else VPZ_E_INVALID_ARG).
 *   Both may be NULL (n_records 0) when every packet has VPZ_PKT_NO_FLOOR.
 * pcm_out: stream s writes at pcm_out + stream_out_offset[s] (float index; NULL offsets = all 0),
 *   interleaved [sample][channel] or planar with `channel_stride` floats between channels.
 *   At most stream_out_capacity samples per channel are written per stream (vpz_decoder_set_stream_capacities: per stream).
 * samples_written[n_streams]: samples per channel produced by this call (host memory, always).
 *   The count is final when the call returns even in VPZ_MEM_DEVICE mode (it is computed by the
 *   host-side state machine); the PCM itself is ready after vpz_context_synchronize.
 * Returns VPZ_OK when the batch was taken.  Conditions that cost ONE packet its Read in the reference (the window
 * check of StreamDecoder.cs:777-778) do not fail the call: see vpz_decoder_last_packet_status.
 * ------------------------------------------------------------------------------------------ */
int vpz_decoder_synth(vpz_decoder *dec, int64_t n_packets, const vpz_packet *packets,
                      const float *residue, int64_t residue_floats,
                      const int16_t *posts, const uint8_t *post_counts, int64_t n_records,
                      int mem_space,
                      void *pcm_out, /* float32, or int16 for the _S16 layouts */ const int64_t *stream_out_offset, int64_t stream_out_capacity,
                      int out_layout, int64_t channel_stride,
                      int64_t *samples_written);

/* Status of each packet of the LAST vpz_decoder_synth call, in packet order: VPZ_OK, or VPZ_E_WINDOW_MISMATCH for a
 * packet the state machine skipped where `OverlapBuffers` would have thrown (StreamDecoder.cs:777-778: that one Read
 * fails, the decoder state stays as it was, the next Read goes on with the next packet).  Copies
 * min(n_packets, capacity) entries; `out` may be NULL with capacity 0.  n_not_ok (may be NULL) receives how many
 * packets of the call are not VPZ_OK -- 0 is the common case and needs no per-packet look. */
int vpz_decoder_last_packet_status(vpz_decoder *dec, int32_t *out, int64_t capacity, int64_t *n_not_ok);

/* Floor 0 data of the NEXT vpz_decoder_synth call (Floor0.Data after Unpack, Floor0.cs:113-162), for the
 * channel records whose floor is type 0: amp[rec] = Data.Amp (0 => ExecuteChannel false; the caller also
 * sets post_counts[rec] = Amp != 0), coeff[rec*coeff_stride + j] = Data.Coeff[j], j < order.  Pointers
 * live in `mem_space` like the other inputs of that call and are consumed by it.  Where the reference
 * indexes its w map out of range (bark_map_size > blocksize/2, Floor0.cs:99-109,185-190 -- it throws) the
 * value 2*cos(pi*k/bark_map_size) is computed instead. */
int vpz_decoder_set_floor0_data(vpz_decoder *dec, const float *amp, const float *coeff, int32_t coeff_stride);

/* Samples per channel each packet of the LAST vpz_decoder_synth call contributed, in packet order
 * (what one `Read` of the reference returns for that packet: it hands out at most one packet's worth per
 * call, StreamDecoder.cs:436).  0 for skipped / ignored packets.  Copies min(n_packets, capacity). */
int vpz_decoder_last_packet_samples(vpz_decoder *dec, int32_t *out, int64_t capacity);

/* `StreamDecoder.HasClipped` (StreamDecoder.cs:1001); synchronises the context. */
int vpz_decoder_has_clipped(vpz_decoder *dec, int32_t stream, int32_t *has_clipped);
/* `_currentPosition` after the last synth call (StreamDecoder.cs:493) */
int vpz_decoder_position(vpz_decoder *dec, int32_t stream, int64_t *sample_position);
/* `_currentPosition = value; _hasPosition = true` -- what StreamDecoder.SeekTo does around its pre-roll
 * (StreamDecoder.cs:851-852, 879-880) and what the SamplePosition setter amounts to.  A seek is, on this side of
 * the boundary: vpz_decoder_reset(stream); vpz_decoder_set_position(stream, <kept value>) so that the granule
 * pick-up stays off; synth of the pre-roll packet and the target packet; vpz_decoder_set_position(stream,
 * target + samples still undelivered).  INTEGRATION.md spells the sequence out. */
int vpz_decoder_set_position(vpz_decoder *dec, int32_t stream, int64_t sample_position);

/* The element type of `residue` in the synth calls that follow (until changed).  VPZ_RESIDUE_F32 (the default): float32, what
 * Residue*.Decode leaves in its buffer.  VPZ_RESIDUE_I16: the same values as 16-bit integers -- exact whenever every value is an
 * integer of that range, which the entropy decoder can tell from the setup header (libvorbis' residue books are integer
 * lattices; vpzh_residue_is_integral in vorbispizza_front.h) -- at half the bytes over the host link; the device widens them
 * to float32 before anything else looks at them.  `residue` then points to int16_t; residue_offset, residue_floats count
 * VALUES as before. */
#define VPZ_RESIDUE_F32 0
#define VPZ_RESIDUE_I16 1
int vpz_decoder_set_residue_format(vpz_decoder *dec, int32_t format);

/* Output areas of different sizes in one batch (files of one encoder setting differ in length): capacity[s] tightens
 * stream_out_capacity for stream s in the synth calls that follow -- stream s may produce min(stream_out_capacity, capacity[s])
 * samples per channel, a call that would produce more fails with VPZ_E_CAPACITY before anything is written or any state
 * changes.  n must be the decoder's stream count; (NULL, 0) removes the per-stream bounds again. */
int vpz_decoder_set_stream_capacities(vpz_decoder *dec, const int64_t *capacity, int32_t n);

/* Host threads the integer half of this decoder's synth calls may use (ABI v6): the state machine of a large batch -- the
 * restatement of StreamDecoder.ReadNextPacket / Read, packet by packet -- is split over a small pool of host threads owned by the
 * decoder's context.  n = 0 (the default): the CPUs the process may run on (affinity mask, divided by LOCAL_WORLD_SIZE), at most
 * 16; n = 1: no pool, everything on the calling thread; n > 1: that many (at most 16).  A host that drives several contexts at
 * once (one thread per context, like the reference's one StreamDecoder per thread, VorbisReader.cs:56-85) gives each decoder
 * its share here -- otherwise every context would start a pool of its own for the whole machine.  The result of a synth call
 * does not depend on the value. */
int vpz_decoder_set_host_threads(vpz_decoder *dec, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* VORBISPIZZA_SYNTH_H */
