// Host-side mirror of the reference's public read path -- `VorbisReader.ReadSamples` ->
// `StreamDecoder.Read` (VorbisReader.cs:232-253, StreamDecoder.cs:407-498) -- written in C++ above the
// C ABI, because no .NET toolchain exists in this pipeline.  It keeps exactly what the C# host keeps
// (container + entropy decode, vorbis_front.cpp) and hands every batch of packets to
// vpz_decoder_synth; the PCM a call returns obeys the reference's contract:
//   * interleaved `L R L R` for Read(buffer), planar with `channelStride` for Read(buffer, n, stride);
//   * the return value is samples per channel;
//   * at most ONE packet's worth of samples per call (`while (idx == 0)`, StreamDecoder.cs:436);
//   * ClipSamples defaults to true as VorbisReader.ProcessNewStream sets it (VorbisReader.cs:71).
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/vorbispizza_front.h"
#include "../../include/vorbispizza_reader.h"

// One logical stream of the container: what a StreamDecoder is in the reference.
struct Sub {
    vpz_context *ctx = nullptr;
    vpzh_stream *front = nullptr;
    vpz_decoder *dec = nullptr;
    vpzh_info info{};
    std::string error;
    bool clip = true;
    int batch = 128;
    int64_t next_packet = 0;          // first packet not yet synthesised
    // PCM of the current batch, interleaved [sample][channel], and how it splits into packets
    bool s16 = false;                 // samples leave the GPU as int16 (VPZ_OUT_INTERLEAVED_S16) instead of float32
    std::vector<float> pcm;
    std::vector<int16_t> pcm16;
    std::vector<int32_t> packet_samples;
    size_t cur_packet = 0;            // index into packet_samples
    int64_t cur_offset = 0;           // samples of the batch already handed out (start of cur_packet + consumed)
    int32_t cur_remaining = 0;        // samples left in the current packet
    int64_t position = 0;
    bool ended = false;
    bool opened = false;              // headers parsed: every read / seek entry point refuses a reader that is not
    // a packet of the current batch whose entropy decode threw (the reference's exception out of
    // DecodeNextPacket): surfaced once by the Read that reaches it, the packets around it are unaffected
    int64_t failed_packet = -1;
    std::string failed_text;
    std::vector<int32_t> packet_status;  // per packet of the batch, empty when all are VPZ_OK
    // scratch for the front end
    std::vector<vpz_packet> packets;
    std::vector<float> residue;
    std::vector<int16_t> residue16;  // ABI v5: the batch's residue as 16-bit integers, for a stream whose setup header guarantees them
    std::vector<int16_t> posts;
    std::vector<uint8_t> counts;
    std::vector<float> f0_amp, f0_coeff;

    int fail(int status, const char *what)
    {
        error = what;
        if (ctx && status == VPZ_E_HIP) error += std::string(": ") + vpz_context_last_error(ctx);
        return status;
    }

    int ensure_decoder()
    {
        if (dec) return VPZ_OK;
        std::vector<vpz_floor1_config> floors(info.floor_count);
        std::vector<vpz_floor0_config> floors0(info.floor_count);
        std::vector<uint8_t> floor_types(info.floor_count, 1);
        std::vector<vpz_mapping_config> mappings(info.mapping_count);
        for (int i = 0; i < info.floor_count; ++i) {
            floor_types[i] = (uint8_t)vpzh_get_floor_type(front, i);
            const int rc0 = floor_types[i] == 0 ? vpzh_get_floor0(front, i, &floors0[i]) : vpzh_get_floor1(front, i, &floors[i]);
            if (rc0 != VPZH_OK) return fail(VPZ_E_UNSUPPORTED, "floor cannot be represented");
        }
        for (int i = 0; i < info.mapping_count; ++i) vpzh_get_mapping(front, i, &mappings[i]);
        vpz_stream_config cfg{};
        cfg.channels = info.channels;
        cfg.block_size0 = info.block_size0;
        cfg.block_size1 = info.block_size1;
        cfg.floor_count = info.floor_count;
        cfg.floors = floors.data();
        cfg.mapping_count = info.mapping_count;
        cfg.mappings = mappings.data();
        cfg.clip_samples = clip ? 1 : 0;
        cfg.floor_types = floor_types.data();
        cfg.floors0 = floors0.data();
        int rc = vpz_decoder_create(ctx, &cfg, 1, &dec);
        if (rc != VPZ_OK) return fail(rc, vpz_context_last_error(ctx));
        return VPZ_OK;
    }

    // decode + synthesise the next batch of packets; returns VPZ_OK, sets `ended` when none are left
    int refill()
    {
        pcm.clear();
        pcm16.clear();
        packet_samples.clear();
        cur_packet = 0;
        cur_offset = 0;
        cur_remaining = 0;
        failed_packet = -1;
        packet_status.clear();
        if (next_packet >= info.audio_packets) { ended = true; return VPZ_OK; }
        int rc = ensure_decoder();
        if (rc != VPZ_OK) return rc;
        const int64_t n = std::min<int64_t>(batch, info.audio_packets - next_packet);
        const int C = info.channels;
        const int half1 = info.block_size1 / 2;
        packets.resize((size_t)n);
        const bool i16 = vpzh_residue_is_integral(front) != 0;  // (half the bytes over the link, the same values)
        if (i16) residue16.assign((size_t)n * C * half1, 0);
        else residue.assign((size_t)n * C * half1, 0.f);
        posts.assign((size_t)n * C * 64, 0);
        counts.assign((size_t)n * C, 0);
        int64_t used = 0;
        const int f0_stride = vpzh_max_floor0_order(front);
        if (f0_stride > 0) {
            f0_amp.assign((size_t)n * C, 0.f);
            f0_coeff.assign((size_t)n * C * f0_stride, 0.f);
        }
        const int drc = i16 ? vpzh_decode_range_i16(front, next_packet, n, 0, 0, packets.data(), residue16.data(), posts.data(), counts.data(),
                                                    &used, f0_stride ? f0_amp.data() : nullptr, f0_stride ? f0_coeff.data() : nullptr, f0_stride)
                            : vpzh_decode_range_ex(front, next_packet, n, 0, 0, packets.data(), residue.data(), posts.data(), counts.data(),
                                                   &used, f0_stride ? f0_amp.data() : nullptr, f0_stride ? f0_coeff.data() : nullptr, f0_stride);
        if (drc != VPZH_OK) return fail(VPZ_E_INVALID_ARG, vpzh_last_error(front));
        if ((rc = vpz_decoder_set_residue_format(dec, i16 ? VPZ_RESIDUE_I16 : VPZ_RESIDUE_F32)) != VPZ_OK)
            return fail(rc, vpz_context_last_error(ctx));
        {   // a packet that threw is consumed and changes nothing (StreamDecoder.cs:696-762: the exception leaves
            // DecodeNextPacket before any state is touched); the front end handed it over as "not decoded"
            int64_t first_failed = -1;
            if (vpzh_decode_failures(front, &first_failed) > 0) {
                failed_packet = first_failed;
                failed_text = vpzh_last_error(front);
            }
        }
        if (f0_stride > 0) vpz_decoder_set_floor0_data(dec, f0_amp.data(), f0_coeff.data(), f0_stride);
        next_packet += n;
        const int64_t cap = n * half1 + info.block_size1;
        if (s16) pcm16.assign((size_t)cap * C, 0);
        else pcm.assign((size_t)cap * C, 0.f);
        int64_t written = 0;
        rc = vpz_decoder_synth(dec, n, packets.data(), i16 ? reinterpret_cast<const float *>(residue16.data()) : residue.data(),
                               (int64_t)(i16 ? residue16.size() : residue.size()), posts.data(), counts.data(),
                               (int64_t)counts.size(), VPZ_MEM_HOST,
                               s16 ? static_cast<void *>(pcm16.data()) : static_cast<void *>(pcm.data()), nullptr, cap,
                               s16 ? VPZ_OUT_INTERLEAVED_S16 : VPZ_OUT_INTERLEAVED, 0, &written);
        if (rc != VPZ_OK) return fail(rc, vpz_context_last_error(ctx));
        packet_samples.resize((size_t)n);
        vpz_decoder_last_packet_samples(dec, packet_samples.data(), n);
        // a window mismatch (the reference's OverlapBuffers exception, StreamDecoder.cs:777-778) costs only that packet's
        // Read: the Read that reaches the packet reports it, once, and the next one goes on behind it
        int64_t not_ok = 0;
        vpz_decoder_last_packet_status(dec, nullptr, 0, &not_ok);
        packet_status.clear();
        if (not_ok > 0) {
            packet_status.resize((size_t)n);
            vpz_decoder_last_packet_status(dec, packet_status.data(), n, nullptr);
        }
        if (s16) pcm16.resize((size_t)written * C);
        else pcm.resize((size_t)written * C);
        return VPZ_OK;
    }

    // StreamDecoder.Read core (:418-498); T = float, or int16_t when the reader delivers 16-bit samples
    template <typename T>
    int64_t read(T *buffer, int64_t buffer_len, int64_t samples_to_read, int64_t channel_stride, bool interleave,
                 int *status)
    {
        const std::vector<T> &pcm = batch_pcm(static_cast<T *>(nullptr));
        const int C = info.channels;
        *status = VPZ_OK;
        if (!opened || C < 1) { *status = fail(VPZ_E_INVALID_ARG, "the reader has no open stream"); return 0; }
        if (buffer_len % C != 0) { *status = fail(VPZ_E_INVALID_ARG, "Length must be a multiple of Channels."); return 0; }
        if (samples_to_read < 0 || buffer_len < samples_to_read * C) { *status = fail(VPZ_E_INVALID_ARG, "The buffer is too small for the requested amount."); return 0; }
        // StoreContiguous slices `buffer[ch * channelStride + offset ..]` (StreamDecoder.cs:594-638): a stride that puts
        // the last channel's samples outside the span throws there (ArgumentOutOfRangeException)
        if (!interleave && (channel_stride < 0 || (int64_t)(C - 1) * channel_stride + samples_to_read > buffer_len)) {
            *status = fail(VPZ_E_INVALID_ARG, "channelStride puts the last channel outside the buffer.");
            return 0;
        }
        int64_t idx = 0;
        while (idx == 0) {
            if (cur_remaining == 0) {
                // next packet that exists; packets that emit nothing simply loop on, like ReadNextPacket
                bool have = false;
                while (!have) {
                    if (cur_packet < packet_samples.size()) {
                        if ((int64_t)cur_packet == failed_packet) {  // the reference's Read throws here, once
                            failed_packet = -1;
                            ++cur_packet;
                            *status = fail(VPZ_E_INVALID_ARG, failed_text.c_str());
                            return idx;
                        }
                        if (!packet_status.empty() && packet_status[cur_packet] != VPZ_OK) {  // OverlapBuffers threw
                            const int st = packet_status[cur_packet];
                            packet_status[cur_packet] = VPZ_OK;
                            ++cur_packet;
                            *status = fail(st, "the packet's previous tail is longer than its window slope "
                                               "(StreamDecoder.cs:777-778); the packet was skipped");
                            return idx;
                        }
                        cur_remaining = packet_samples[cur_packet++];
                        have = true;
                    } else {
                        if (ended) return idx;
                        int rc = refill();
                        if (rc != VPZ_OK) { *status = rc; return idx; }
                        if (ended) return idx;
                    }
                }
                if (cur_remaining == 0) continue;
            }
            const int64_t copy_len = std::min<int64_t>(samples_to_read - idx, cur_remaining);
            if (copy_len <= 0) break;
            const T *src = pcm.data() + (size_t)cur_offset * C;
            if (interleave) {
                memcpy(buffer + idx * C, src, sizeof(T) * (size_t)(copy_len * C));
            } else {
                for (int ch = 0; ch < C; ++ch)
                    for (int64_t i = 0; i < copy_len; ++i) buffer[ch * channel_stride + idx + i] = src[i * C + ch];
            }
            idx += copy_len;
            cur_offset += copy_len;
            cur_remaining -= (int32_t)copy_len;
            position += copy_len;
        }
        return idx;
    }
    const std::vector<float> &batch_pcm(float *) const { return pcm; }
    const std::vector<int16_t> &batch_pcm(int16_t *) const { return pcm16; }

    ~Sub()
    {
        if (dec) vpz_decoder_destroy(dec);
        if (front) vpzh_close(front);
    }
};

// VorbisReader (VorbisReader.cs): the logical streams found so far (`Streams`), the one the convenience members
// forward to (`_streamDecoder`), and the container bytes FindNextStream keeps scanning.
struct vpzr_reader {
    vpz_context *ctx = nullptr;
    std::vector<uint8_t> data;
    std::vector<std::unique_ptr<Sub>> subs;
    Sub *cur = nullptr;
    std::string open_error;

    // opens logical stream number `index` of the container; VPZH_* status
    int open_stream(int index, std::unique_ptr<Sub> &out)
    {
        std::unique_ptr<Sub> s(new Sub());
        s->ctx = ctx;
        const int rc = vpzh_open_memory_stream(data.data(), data.size(), index, &s->front);
        if (rc != VPZH_OK) {
            open_error = s->front ? vpzh_last_error(s->front) : "could not load the specified container";
            return rc;
        }
        vpzh_get_info(s->front, &s->info);
        s->opened = s->info.channels >= 1;
        out = std::move(s);
        return VPZH_OK;
    }
};

extern "C" {

int vpzr_open_memory(vpz_context *ctx, const uint8_t *data, uint64_t size, vpzr_reader **out)
{
    if (!ctx || !data || !out) return VPZ_E_INVALID_ARG;
    *out = nullptr;
    std::unique_ptr<vpzr_reader> r(new vpzr_reader());
    r->ctx = ctx;
    r->data.assign(data, data + size);
    std::unique_ptr<Sub> first;
    const int rc = r->open_stream(0, first);
    if (rc != VPZH_OK) {
        // the handle only carries the error text; every other entry point refuses it
        *out = r.release();
        return rc == VPZH_E_UNSUPPORTED ? VPZ_E_UNSUPPORTED : VPZ_E_INVALID_ARG;
    }
    r->subs.push_back(std::move(first));
    r->cur = r->subs[0].get();
    *out = r.release();
    return VPZ_OK;
}

void vpzr_close(vpzr_reader *r) { delete r; }

// VorbisReader.FindNextStream (VorbisReader.cs:191-194): looks for one more logical stream in the container (a
// chained file starts it after the previous stream's last page) and adds it to the list.  1: found, 0: none.
int vpzr_find_next_stream(vpzr_reader *r)
{
    if (!r || !r->cur) return 0;
    std::unique_ptr<Sub> next;
    if (r->open_stream((int)r->subs.size(), next) != VPZH_OK) return 0;
    next->clip = r->cur->clip;
    next->batch = r->cur->batch;
    next->s16 = r->cur->s16;
    r->subs.push_back(std::move(next));
    return 1;
}

int vpzr_stream_count(vpzr_reader *r) { return r ? (int)r->subs.size() : 0; }

// VorbisReader.SwitchStreams (VorbisReader.cs:197-217): 1 when channels or sample rate differ from the stream that
// was current (the caller has to re-open its output), 0 otherwise, VPZ_E_INVALID_ARG for a bad index.
int vpzr_switch_streams(vpzr_reader *r, int index)
{
    if (!r || !r->cur || index < 0 || index >= (int)r->subs.size()) return VPZ_E_INVALID_ARG;
    Sub *next = r->subs[(size_t)index].get(), *old = r->cur;
    if (next == old) return 0;
    if (!next->dec) next->clip = old->clip;  // "carry-through the clipping setting"
    r->cur = next;
    return (next->info.channels != old->info.channels || next->info.sample_rate != old->info.sample_rate) ? 1 : 0;
}

int vpzr_stream_serial(vpzr_reader *r) { return r && r->cur ? r->cur->info.stream_serial : 0; }

const char *vpzr_last_error(vpzr_reader *R) { return !R ? "" : (R->cur ? R->cur->error.c_str() : R->open_error.c_str()); }
int vpzr_channels(vpzr_reader *R) { Sub *r = R ? R->cur : nullptr; return r && r->opened ? r->info.channels : 0; }
int vpzr_sample_rate(vpzr_reader *R) { Sub *r = R ? R->cur : nullptr; return r && r->opened ? r->info.sample_rate : 0; }
int64_t vpzr_sample_position(vpzr_reader *R) { Sub *r = R ? R->cur : nullptr; return r ? r->position : 0; }
int vpzr_is_end_of_stream(vpzr_reader *R) { Sub *r = R ? R->cur : nullptr; return r ? (r->ended && r->cur_remaining == 0 && r->cur_packet >= r->packet_samples.size()) : 1; }

int64_t vpzr_total_samples(vpzr_reader *R) { Sub *r = R ? R->cur : nullptr; return r && r->opened ? vpzh_total_samples(r->front) : 0; }

// StreamDecoder.SeekTo(long, SeekOrigin) (StreamDecoder.cs:815-881)
int vpzr_seek_to(vpzr_reader *R, int64_t sample_position, int origin)
{
    Sub *r = R ? R->cur : nullptr;
    if (!r) return VPZ_E_INVALID_ARG;
    if (!r->opened) return r->fail(VPZ_E_INVALID_ARG, "the reader has no open stream");
    if (sample_position < 0) return r->fail(VPZ_E_INVALID_ARG, "samplePosition");  // ArgumentOutOfRangeException
    switch (origin) {
    case VPZR_SEEK_BEGIN: break;
    case VPZR_SEEK_CURRENT: sample_position = r->position - sample_position; break;               // :838 (as written there)
    case VPZR_SEEK_END: sample_position = vpzh_total_samples(r->front) - sample_position; break;  // :842
    default: return r->fail(VPZ_E_INVALID_ARG, "seekOrigin");
    }
    int64_t pre = 0, roll = 0;
    if (vpzh_seek(r->front, sample_position, &pre, &roll) != VPZH_OK)
        return r->fail(VPZ_E_INVALID_ARG, "The requested seek position extends beyond the stream.");
    int rc = r->ensure_decoder();
    if (rc != VPZ_OK) return rc;
    // ResetDecoder(); _hasPosition = true (the position VALUE is whatever it was, :851-852)
    int64_t kept = 0;
    vpz_decoder_position(r->dec, 0, &kept);
    vpz_decoder_reset(r->dec, 0);
    vpz_decoder_set_position(r->dec, 0, kept);
    r->ended = false;
    r->next_packet = pre;
    // the pre-roll packet and the target packet, as one two-packet call (:855-876)
    const int batch = r->batch;
    r->batch = 2;
    rc = r->refill();
    r->batch = batch;
    if (rc != VPZ_OK) return rc;
    if (r->packet_samples.size() < 2) {  // PreRollPacketException
        r->ended = true;
        return r->fail(VPZ_E_INVALID_ARG, "Could not read pre-roll packet. Try seeking again prior to reading more samples.");
    }
    // _prevPacketStart += rollForward; _currentPosition = samplePosition (:879-880)
    const int32_t have = r->packet_samples[1];
    const int32_t skip = (int32_t)std::min<int64_t>(roll, have);
    r->cur_packet = 2;
    r->cur_offset = r->packet_samples[0] + skip;
    r->cur_remaining = have - skip;
    r->position = sample_position;
    vpz_decoder_set_position(r->dec, 0, sample_position + r->cur_remaining);
    return VPZ_OK;
}

int vpzr_set_clip_samples(vpzr_reader *R, int clip)
{
    Sub *r = R ? R->cur : nullptr;
    if (!r) return VPZ_E_INVALID_ARG;
    if (r->dec) return VPZ_E_INVALID_ARG;  // fixed once decoding has started
    r->clip = clip != 0;
    return VPZ_OK;
}

int vpzr_set_sample_format(vpzr_reader *R, int format)
{
    Sub *r = R ? R->cur : nullptr;
    if (!r || (format != VPZR_FORMAT_F32 && format != VPZR_FORMAT_S16)) return VPZ_E_INVALID_ARG;
    if (r->dec) return VPZ_E_INVALID_ARG;  // fixed once decoding has started
    r->s16 = format == VPZR_FORMAT_S16;
    return VPZ_OK;
}

int vpzr_set_batch_packets(vpzr_reader *R, int packets)
{
    Sub *r = R ? R->cur : nullptr;
    if (!r || packets < 1) return VPZ_E_INVALID_ARG;
    r->batch = packets;
    return VPZ_OK;
}

int vpzr_has_clipped(vpzr_reader *R)
{
    Sub *r = R ? R->cur : nullptr;
    int32_t v = 0;
    if (r && r->dec) vpz_decoder_has_clipped(r->dec, 0, &v);
    return v;
}

int64_t vpzr_read_samples(vpzr_reader *R, float *buffer, int64_t buffer_len, int *status)
{
    Sub *r = R ? R->cur : nullptr;
    int st = VPZ_OK;
    if (!r || (!buffer && buffer_len) || buffer_len < 0) { if (status) *status = VPZ_E_INVALID_ARG; return 0; }
    if (!r->opened) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader has no open stream"); return 0; }
    if (r->s16) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader delivers 16-bit samples (vpzr_set_sample_format)"); return 0; }
    const int C = r->info.channels;
    const int64_t count = buffer_len - buffer_len % C;  // VorbisReader.cs:235
    int64_t n = count == 0 ? 0 : r->read(buffer, count, count / C, 0, true, &st);
    if (status) *status = st;
    return n;
}

int64_t vpzr_read_samples_s16(vpzr_reader *R, int16_t *buffer, int64_t buffer_len, int *status)
{
    Sub *r = R ? R->cur : nullptr;
    int st = VPZ_OK;
    if (!r || (!buffer && buffer_len) || buffer_len < 0) { if (status) *status = VPZ_E_INVALID_ARG; return 0; }
    if (!r->opened) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader has no open stream"); return 0; }
    if (!r->s16) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader delivers float samples (vpzr_set_sample_format)"); return 0; }
    const int C = r->info.channels;
    const int64_t count = buffer_len - buffer_len % C;
    int64_t n = count == 0 ? 0 : r->read(buffer, count, count / C, 0, true, &st);
    if (status) *status = st;
    return n;
}

int64_t vpzr_read_samples_planar(vpzr_reader *R, float *buffer, int64_t buffer_len, int64_t samples_to_read,
                                 int64_t channel_stride, int *status)
{
    Sub *r = R ? R->cur : nullptr;
    int st = VPZ_OK;
    if (!r || (!buffer && buffer_len) || buffer_len < 0) { if (status) *status = VPZ_E_INVALID_ARG; return 0; }
    if (!r->opened) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader has no open stream"); return 0; }
    if (r->s16) { if (status) *status = r->fail(VPZ_E_INVALID_ARG, "the reader delivers 16-bit samples (vpzr_set_sample_format)"); return 0; }
    const int C = r->info.channels;
    const int64_t count = buffer_len - buffer_len % C;  // VorbisReader.cs:246
    int64_t n = count == 0 ? 0 : r->read(buffer, count, samples_to_read, channel_stride, false, &st);
    if (status) *status = st;
    return n;
}

}  // extern "C"
