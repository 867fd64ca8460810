// In-process multi-device dispatcher (include/vorbispizza_multi.h): one host process, several MI355X.
//
// The reference's host is ONE process in which a VorbisReader holds N independent StreamDecoders (VorbisReader.cs:56-85);
// SURVEY.md section 8e prescribes "one host thread + one HIP stream per device; per-device context and tables", streams
// partitioned contiguously, no collective.  This file is that: vpzm_decode_library shards a library of containers over the
// device groups (shard_range's rule), and every group runs, on its own threads, the pipeline
//
//   open (setup headers)  ->  entropy decode into page-locked batch arrays  ->  vpz_decoder_synth (host memory)  ->  PCM
//        host threads              host threads, one stream each                 `contexts_per_device` issuing threads
//
// with sub-batches of `streams_per_call` streams of one setup header per synth call, `slots_per_device` sub-batches in
// flight.  Nothing is shared between groups but the caller's arrays; the C ABI below them is used exactly as any other host
// would use it (one thread at a time per context / decoder).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vorbispizza_multi.h"

namespace {

using Clock = std::chrono::steady_clock;
double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// what a vpz_decoder is created from: the setup-header products of a stream (StreamDecoder.cs:213-353)
struct Setup {
    vpzh_info info{};
    std::vector<vpz_floor1_config> floors;
    std::vector<vpz_floor0_config> floors0;
    std::vector<uint8_t> floor_types;
    std::vector<vpz_mapping_config> mappings;
    int f0_stride = 0;
    bool integral = false;  // every residue value is an integer of 16 bits: the residue travels as int16 (half the link bytes)

    bool load(vpzh_stream *h)
    {
        if (vpzh_get_info(h, &info) != VPZH_OK) return false;
        floors.assign((size_t)info.floor_count, vpz_floor1_config{});
        floors0.assign((size_t)info.floor_count, vpz_floor0_config{});
        floor_types.assign((size_t)info.floor_count, 1);
        mappings.assign((size_t)info.mapping_count, vpz_mapping_config{});
        for (int i = 0; i < info.floor_count; ++i) {
            const int t = vpzh_get_floor_type(h, i);
            floor_types[i] = (uint8_t)t;
            if ((t == 0 ? vpzh_get_floor0(h, i, &floors0[i]) : vpzh_get_floor1(h, i, &floors[i])) != VPZH_OK) return false;
        }
        for (int i = 0; i < info.mapping_count; ++i)
            if (vpzh_get_mapping(h, i, &mappings[i]) != VPZH_OK) return false;
        f0_stride = vpzh_max_floor0_order(h);
        integral = vpzh_residue_is_integral(h) != 0;
        return true;
    }
    // the same decoder serves two streams iff everything it was created from is the same
    bool same(const Setup &o) const
    {
        if (info.channels != o.info.channels || info.block_size0 != o.info.block_size0 || info.block_size1 != o.info.block_size1 ||
            floors.size() != o.floors.size() || mappings.size() != o.mappings.size() || floor_types != o.floor_types || integral != o.integral)
            return false;
        for (size_t i = 0; i < floors.size(); ++i) {
            if (floor_types[i] == 0) {
                if (memcmp(&floors0[i], &o.floors0[i], sizeof floors0[i]) != 0) return false;
            } else {
                const vpz_floor1_config &a = floors[i], &b = o.floors[i];
                if (a.x_count != b.x_count || a.multiplier != b.multiplier ||
                    memcmp(a.x_list, b.x_list, sizeof(int32_t) * (size_t)std::max(0, a.x_count)) != 0)
                    return false;
            }
        }
        for (size_t i = 0; i < mappings.size(); ++i)
            if (memcmp(&mappings[i], &o.mappings[i], sizeof mappings[i]) != 0) return false;
        return true;
    }
};

struct Lane {  // one context (HIP stream) of a device group and the decoders that live on it
    vpz_context *ctx = nullptr;
    std::vector<std::pair<std::shared_ptr<Setup>, vpz_decoder *>> decs;
};

struct Slot {  // page-locked batch arrays of one sub-batch in flight
    vpz_packet *packets = nullptr;
    float *residue = nullptr;            // (cap_residue floats; holds int16 values for a sub-batch whose residue travels as int16)
    int16_t *posts = nullptr;
    uint8_t *counts = nullptr;
    float *f0_amp = nullptr, *f0_coeff = nullptr;
    size_t cap_packets = 0, cap_residue = 0, cap_posts = 0, cap_counts = 0, cap_f0 = 0, cap_f0c = 0;
};

struct Group {
    int device = 0;
    std::vector<Lane> lanes;
    std::vector<Slot> slots;
};

template <class T>
bool grow(vpz_context *ctx, T *&p, size_t &cap, size_t need)
{
    if (need <= cap) return true;
    if (p) vpz_host_free(ctx, p);
    p = nullptr;
    cap = 0;
    void *q = nullptr;
    const size_t want = need + need / 4 + 64;
    if (vpz_host_alloc(ctx, (uint64_t)(want * sizeof(T)), &q) != VPZ_OK) return false;
    p = static_cast<T *>(q);
    cap = want;
    return true;
}

}  // namespace

struct vpzm_dispatcher {
    std::vector<Group> groups;
    vpzm_options opt{};
    std::string error;
    std::mutex err_mu;
    std::mutex call_mu;  // vpzm_decode_library holds it: calls from several host threads take the dispatcher in turn
    void fail(const std::string &what)
    {
        std::lock_guard<std::mutex> g(err_mu);
        if (error.empty()) error = what;
    }
};

namespace {

constexpr int kNoDecoder = 1;  // (synth_sub: no decoder for the sub-batch's setup; not a VPZ_* status, those are <= 0)
constexpr int64_t kCallValues = (int64_t)64 << 20;  // residue values of one synth call (256 MiB as float32), see plan_wave

struct Job {  // one stream of the library inside its group
    int32_t k = 0;  // index in the caller's arrays
    vpzh_stream *h = nullptr;
    std::shared_ptr<Setup> own;  // its setup-header products, loaded when it is opened
    int setup = -1;
    int64_t packets = 0, residue_floats = 0, total_samples = 0;
    int32_t status = VPZM_OK;
    bool finished = false;  // its PCM has been written (or it has its own failure status): what an aborted run leaves alone
};

struct Sub {  // streams of one setup that ride in one vpz_decoder_synth call
    int setup = 0;
    std::shared_ptr<Setup> st;           // (its own reference: `setups` grows under the group's mutex while sub-batches are worked on outside it)
    std::vector<int> members;            // indices into jobs
    std::vector<int64_t> pbase, rbase;   // where each member's packets / residue start in the slot's arrays
    int64_t n_packets = 0, res_floats = 0;
    int decoded = 0;                     // members whose entropy decode is complete (under the group's mutex)
    bool prepped = false, prepping = false, synth_done = false;
};

// One device group's share of a vpzm_decode_library call.  A streaming pipeline, every stage on the group's own threads and all
// of them overlapped:
//   open      a container is walked and its three headers parsed (setup cache: vorbis_front.cpp) -- `threads` workers, in
//             stream order;
//   plan      when a WAVE of consecutive streams (4 sub-batches' worth) is open, its streams are grouped by setup header and
//             cut into sub-batches (streams of one setup share a decoder and ride in the same synth calls);
//   decode    the workers entropy-decode a sub-batch's streams, one stream each, straight into the page-locked arrays of the
//             sub-batch's SLOT (slots_per_device of them: sub-batch b takes slot b mod slots once sub-batch b - slots has been
//             synthesised) -- a worker with no decode work opens the next container instead, so the first synth call is under
//             way a few milliseconds into the job;
//   synth     `contexts_per_device` issuing threads take the decoded sub-batches in order, one host-memory vpz_decoder_synth call
//             each: the upload of one overlaps the download of the other's.
struct GroupRun {
    vpzm_dispatcher *m;
    Group &G;
    int slot_index;
    int32_t lo, hi;  // the group's streams [lo, hi)
    const uint8_t *const *data;
    const uint64_t *size;
    int32_t out_layout;
    void *pcm_out;
    const int64_t *pcm_offset, *pcm_capacity;
    vpzm_stream_result *results;
    int threads;
    double t_wall = 0, t_decode = 0, t_synth = 0;
    Clock::time_point t_begin = Clock::now();
    int64_t samples_total = 0;
    const bool profile = getenv("VPZM_PROFILE") != nullptr;

    std::vector<Job> jobs;
    std::vector<std::shared_ptr<Setup>> setups;
    std::deque<Sub> subs;                    // (a deque: sub-batches are appended while others are in flight)
    std::vector<std::pair<int, int>> tasks;  // (sub, member) in the order they are decoded
    std::vector<int> wave_left;              // streams of each wave still to be opened
    int waves_planned = 0;
    std::mutex mu;
    std::condition_variable cv;
    size_t next_open = 0, next_task = 0, next_synth = 0;

    GroupRun(vpzm_dispatcher *m_, Group &g, int si, int32_t lo_, int32_t hi_, const uint8_t *const *d, const uint64_t *sz, int32_t layout,
             void *out, const int64_t *off, const int64_t *cap, vpzm_stream_result *res, int thr)
        : m(m_), G(g), slot_index(si), lo(lo_), hi(hi_), data(d), size(sz), out_layout(layout), pcm_out(out), pcm_offset(off),
          pcm_capacity(cap), results(res), threads(thr)
    {
        lane_host_threads = std::max(1, std::min(8, thr / std::max(1, (int)G.lanes.size())));
    }

    int wave_size() const { return 4 * m->opt.streams_per_call; }
    bool all_planned() const { return waves_planned == (int)wave_left.size(); }

    // ---- open: the container walked (pages, CRC, lacing), the three headers parsed, the setup products read out
    void open_one(size_t i)
    {
        Job &J = jobs[i];
        J.k = lo + (int32_t)i;
        vpzm_stream_result &R = results[J.k];
        R = vpzm_stream_result{};
        R.device_slot = slot_index;
        try {
            if (vpzh_open_memory(data[J.k], size[J.k], &J.h) != VPZH_OK) {
                J.status = VPZM_E_OPEN;
                if (J.h) vpzh_close(J.h);
                J.h = nullptr;
                return;
            }
            vpzh_info info{};
            vpzh_get_info(J.h, &info);
            J.packets = info.audio_packets;
            J.residue_floats = info.residue_floats;
            J.total_samples = vpzh_total_samples(J.h);
            R.channels = info.channels;
            R.sample_rate = info.sample_rate;
            R.packets = info.audio_packets;
            if (J.total_samples > pcm_capacity[J.k]) { J.status = VPZM_E_CAPACITY; return; }
            if (J.packets > 0) {
                J.own = std::make_shared<Setup>();
                if (!J.own->load(J.h)) J.status = VPZM_E_SETUP;
            }
        } catch (...) {
            J.status = VPZM_E_OPEN;
        }
    }

    // ---- plan (under `mu`): the streams of one wave grouped by setup header, every group cut into sub-batches
    void plan_wave(int w)
    {
        const int S = m->opt.streams_per_call;
        const size_t a = (size_t)w * (size_t)wave_size(), b = std::min(jobs.size(), a + (size_t)wave_size());
        std::vector<std::vector<int>> by_setup(setups.size());
        for (size_t i = a; i < b; ++i) {
            Job &J = jobs[i];
            if (J.status != VPZM_OK || !J.h || J.packets == 0 || !J.own) {  // (no audio packets: 0 samples, nothing to do)
                ++skipped_streams;
                continue;
            }
            int at = -1;
            for (size_t q = 0; q < setups.size(); ++q)
                if (setups[q]->same(*J.own)) { at = (int)q; break; }
            if (at < 0) {
                at = (int)setups.size();
                setups.push_back(J.own);
                by_setup.emplace_back();
            }
            J.own.reset();
            J.setup = at;
            by_setup[(size_t)at].push_back((int)i);
        }
        std::vector<Sub> fresh;
        // a synth call holds up to streams_per_call streams of one setup and up to kCallValues residue values (a library of whole
        // songs would otherwise ask for page-locked slots of gigabytes each): a long stream rides with fewer others, or alone
        int64_t budget = kCallValues;
        if (const char *e = getenv("VPZM_MAX_CALL_VALUES"))  // (tests: small calls)
            if (atoll(e) > 0) budget = atoll(e);
        for (size_t q = 0; q < by_setup.size(); ++q) {
            const std::vector<int> &v = by_setup[q];
            Sub sb;
            auto flush = [&] {
                if (!sb.members.empty()) fresh.push_back(std::move(sb));
                sb = Sub();
            };
            for (size_t j = 0; j < v.size(); ++j) {
                const Job &J = jobs[(size_t)v[j]];
                if (!sb.members.empty() && ((int)sb.members.size() >= S || sb.res_floats + J.residue_floats > budget)) flush();
                if (sb.members.empty()) {
                    sb.setup = (int)q;
                    sb.st = setups[q];
                }
                sb.members.push_back(v[j]);
                sb.pbase.push_back(sb.n_packets);
                sb.rbase.push_back(sb.res_floats);
                sb.n_packets += J.packets;
                sb.res_floats += J.residue_floats;
            }
            flush();
        }
        std::sort(fresh.begin(), fresh.end(), [](const Sub &x, const Sub &y) { return x.members[0] < y.members[0]; });
        for (Sub &sb : fresh) {
            const int bi = (int)subs.size();
            for (size_t j = 0; j < sb.members.size(); ++j) tasks.emplace_back(bi, (int)j);
            subs.push_back(std::move(sb));
        }
        ++waves_planned;
    }

    // the residue of a sub-batch travels as int16 when its setup header guarantees integers (and the caller has not asked for floats)
    bool use_i16(const Setup &st) const { return st.integral && !m->opt.float_residue; }

    // (the functions below run with `mu` RELEASED: they get their sub-batch by reference -- a deque's elements stay where they are, but
    // indexing `subs` / `setups` while plan_wave appends to them is a race)
    bool prep_slot(size_t b, Sub &sb)  // (one thread prepares a given sub-batch)
    {
        Slot &sl = G.slots[b % G.slots.size()];
        const Setup &st = *sb.st;
        const size_t C = (size_t)st.info.channels, rec = (size_t)sb.n_packets * C;
        vpz_context *ctx = G.lanes[0].ctx;
        // (the slot's residue array is float-typed: int16 values take half the elements)
        const size_t res_elems = use_i16(st) ? ((size_t)sb.res_floats + 1) / 2 : (size_t)sb.res_floats;
        bool ok = grow(ctx, sl.packets, sl.cap_packets, (size_t)sb.n_packets) && grow(ctx, sl.residue, sl.cap_residue, res_elems) &&
                  grow(ctx, sl.posts, sl.cap_posts, rec * 64) && grow(ctx, sl.counts, sl.cap_counts, rec);
        if (ok && st.f0_stride > 0)
            ok = grow(ctx, sl.f0_amp, sl.cap_f0, rec) && grow(ctx, sl.f0_coeff, sl.cap_f0c, rec * (size_t)st.f0_stride);
        return ok;
    }

    void decode_member(size_t b, Sub &sb, int j)
    {
        Job &J = jobs[(size_t)sb.members[(size_t)j]];
        if (J.status == VPZM_OK) {
            Slot &sl = G.slots[b % G.slots.size()];
            const Setup &st = *sb.st;
            const size_t C = (size_t)st.info.channels;
            const int64_t pb = sb.pbase[(size_t)j], rb = sb.rbase[(size_t)j];
            int rc = VPZH_E_ARG;
            try {
                float *amp_at = st.f0_stride ? sl.f0_amp + (size_t)pb * C : nullptr;
                float *coeff_at = st.f0_stride ? sl.f0_coeff + (size_t)pb * C * (size_t)st.f0_stride : nullptr;
                if (use_i16(st))
                    rc = vpzh_decode_range_i16(J.h, 0, J.packets, j, rb, sl.packets + pb, reinterpret_cast<int16_t *>(sl.residue) + rb,
                                               sl.posts + (size_t)pb * 64 * C, sl.counts + (size_t)pb * C, nullptr, amp_at, coeff_at, st.f0_stride);
                else
                    rc = vpzh_decode_range_ex(J.h, 0, J.packets, j, rb, sl.packets + pb, sl.residue + rb, sl.posts + (size_t)pb * 64 * C,
                                              sl.counts + (size_t)pb * C, nullptr, amp_at, coeff_at, st.f0_stride);
                if (rc == VPZH_OK) results[J.k].skipped_packets += vpzh_decode_failures(J.h, nullptr);
            } catch (...) {
                rc = VPZH_E_INVALID_DATA;
            }
            if (rc != VPZH_OK) J.status = VPZM_E_OPEN;
        }
        if (J.h) vpzh_close(J.h);  // (the container's packets are not needed any more)
        J.h = nullptr;
    }

    // ---- the workers: decode what can be decoded, else open the next container, else wait
    void worker()
    {
        const size_t B = G.slots.size();
        // containers open but not yet decoded hold their packets in memory: no more than a few waves ahead
        const size_t open_ahead = (size_t)wave_size() * 3;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            if (aborted) return;
            if (next_task < tasks.size()) {
                const size_t b = (size_t)tasks[next_task].first;
                Sub &sb = subs[b];
                if (sb.prepped) {
                    const int j = tasks[next_task++].second;
                    lk.unlock();
                    decode_member(b, sb, j);
                    lk.lock();
                    if (++sb.decoded == (int)sb.members.size()) cv.notify_all();
                    continue;
                }
                if (!sb.prepping && (b < B || subs[b - B].synth_done)) {  // its slot is free: get the arrays ready
                    sb.prepping = true;
                    lk.unlock();
                    const bool ok = prep_slot(b, sb);
                    lk.lock();
                    if (!ok) {
                        for (int mi : sb.members) jobs[(size_t)mi].status = VPZM_E_SYNTH;
                        m->fail("vpzm_decode_library: page-locked batch arrays could not be allocated");
                    }
                    sb.prepped = true;
                    cv.notify_all();
                    continue;
                }
            }
            const size_t undecoded = next_open - std::min(next_open, done_decoding());
            if (next_open < jobs.size() && undecoded < open_ahead + (size_t)threads) {
                const size_t i = next_open++;
                lk.unlock();
                open_one(i);
                lk.lock();
                const int w = (int)(i / (size_t)wave_size());
                if (--wave_left[(size_t)w] == 0) {
                    plan_wave(w);
                    cv.notify_all();
                }
                continue;
            }
            if (all_planned() && next_task >= tasks.size()) {
                t_decode = std::max(t_decode, seconds_since(t_begin));
                return;
            }
            cv.wait(lk);
        }
    }
    // streams whose decode task has been handed out (under `mu`): what the open-ahead limit is measured against
    size_t done_decoding() const { return std::min(next_task + skipped_streams, jobs.size()); }
    size_t skipped_streams = 0;  // (streams that never become a decode task: failed to open, no packets, area too small)
    // An exception on one of the run's threads (std::bad_alloc out of a vector in plan_wave / synth_sub, ...) must neither leave
    // the thread (std::terminate inside a C ABI that promises statuses) nor leave the others waiting for a counter that will never
    // move: the run is ABORTED -- every loop looks at the flag --, and the streams without a result get VPZM_E_SYNTH
    bool aborted = false;  // (under `mu`)
    void abort_run(const char *what) noexcept
    {
        try {
            std::lock_guard<std::mutex> lk(mu);
            aborted = true;
        } catch (...) {
        }
        try {
            m->fail(what);
        } catch (...) {
        }
        cv.notify_all();
    }
    template <class F>
    void guarded(F &&body) noexcept
    {
        try {
            body();
        } catch (const std::bad_alloc &) {
            abort_run("vpzm_decode_library: out of host memory on a pipeline thread");
        } catch (...) {
            abort_run("vpzm_decode_library: a pipeline thread failed");
        }
    }
    int lane_host_threads = 1;   // vpz_decoder_set_host_threads of every lane's decoders: the device's threads / its contexts

    vpz_decoder *decoder_for(Lane &L, const std::shared_ptr<Setup> &st)
    {
        for (auto &p : L.decs)
            if (p.first->same(*st)) return p.second;
        vpz_stream_config cfg{};
        cfg.channels = st->info.channels;
        cfg.block_size0 = st->info.block_size0;
        cfg.block_size1 = st->info.block_size1;
        cfg.floor_count = (int32_t)st->floors.size();
        cfg.floors = st->floors.data();
        cfg.mapping_count = (int32_t)st->mappings.size();
        cfg.mappings = st->mappings.data();
        cfg.clip_samples = m->opt.clip_samples;
        cfg.floor_types = st->floor_types.data();
        cfg.floors0 = st->floors0.data();
        vpz_decoder *dec = nullptr;
        if (vpz_decoder_create(L.ctx, &cfg, m->opt.streams_per_call, &dec) != VPZ_OK) {
            m->fail(std::string("vpz_decoder_create: ") + vpz_context_last_error(L.ctx));
            return nullptr;
        }
        // The integer half of a synth call may fork over a pool of the decoder's context (batches of whole songs do): every lane its
        // share of the device's host threads, not a pool for the whole machine each -- the decode workers need the CPUs
        (void)vpz_decoder_set_host_threads(dec, lane_host_threads);
        // (a library of files from many encoders: a context keeps the decoders of the last few setups, not of every one it has seen)
        constexpr size_t kDecodersPerContext = 8;
        if (L.decs.size() >= kDecodersPerContext) {
            vpz_decoder_destroy(L.decs.front().second);
            L.decs.erase(L.decs.begin());
        }
        L.decs.emplace_back(st, dec);
        return dec;
    }

    // ---- synth: one issuing thread per context takes the decoded sub-batches in order
    void issuer(Lane &L)
    {
        for (;;) {
            size_t b;
            Sub *sp = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (aborted) return;
                    if (next_synth < subs.size()) break;
                    if (all_planned()) return;
                    cv.wait(lk);
                }
                b = next_synth++;
                cv.wait(lk, [&] { return aborted || subs[b].decoded == (int)subs[b].members.size(); });
                if (aborted) return;
                sp = &subs[b];
            }
            synth_sub(L, b, *sp);
        }
    }

    void synth_sub(Lane &L, size_t b, Sub &sb)
    {
        const int S = m->opt.streams_per_call;
        std::vector<int64_t> offs((size_t)S), written((size_t)S), caps((size_t)S, 0);
        std::vector<int32_t> status;
        Slot &sl = G.slots[b % G.slots.size()];
        const Setup &st = *sb.st;
        const int C = st.info.channels;
        bool any = false, all_ok = true;
        int64_t cap = 0, base = INT64_MAX;
        // (the call sees the sub-batch's part of the caller's PCM array: a host-memory call mirrors its output extent on the
        // device, so the offsets handed over start at the sub-batch's lowest one)
        for (size_t j = 0; j < sb.members.size(); ++j) base = std::min(base, pcm_offset[jobs[(size_t)sb.members[j]].k]);
        for (size_t j = 0; j < sb.members.size(); ++j) {
            const Job &J = jobs[(size_t)sb.members[j]];
            offs[j] = pcm_offset[J.k] - base;
            if (J.status == VPZM_OK) {
                any = true;
                // (every stream has its own area: files of one encoder setting share a setup header and differ in length)
                caps[j] = pcm_capacity[J.k];
                cap = std::max(cap, caps[j]);
            } else {
                all_ok = false;
            }
        }
        const size_t elem = out_layout == VPZ_OUT_INTERLEAVED_S16 ? sizeof(int16_t) : sizeof(float);
        void *out_at = static_cast<char *>(pcm_out) + elem * (size_t)base;
        int64_t n_pk = sb.n_packets;
        if (any && !all_ok) {
            // a member whose container did not decode leaves its packets out (rare): the arrays are re-packed stream by stream,
            // stream ids kept, so that vpz_decoder_synth sees only decoded packets; their residue stays where it is
            int64_t w = 0;
            for (size_t j = 0; j < sb.members.size(); ++j) {
                const Job &J = jobs[(size_t)sb.members[j]];
                if (J.status != VPZM_OK) continue;
                const int64_t pb = sb.pbase[j];
                if (w != pb) {
                    memmove(sl.packets + w, sl.packets + pb, sizeof(vpz_packet) * (size_t)J.packets);
                    memmove(sl.posts + (size_t)w * 64 * C, sl.posts + (size_t)pb * 64 * C, sizeof(int16_t) * 64 * (size_t)C * (size_t)J.packets);
                    memmove(sl.counts + (size_t)w * C, sl.counts + (size_t)pb * C, (size_t)C * (size_t)J.packets);
                    if (st.f0_stride) {
                        memmove(sl.f0_amp + (size_t)w * C, sl.f0_amp + (size_t)pb * C, sizeof(float) * (size_t)C * (size_t)J.packets);
                        memmove(sl.f0_coeff + (size_t)w * C * st.f0_stride, sl.f0_coeff + (size_t)pb * C * st.f0_stride,
                                sizeof(float) * (size_t)C * st.f0_stride * (size_t)J.packets);
                    }
                }
                w += J.packets;
            }
            n_pk = w;
        }
        int rc = VPZ_OK;
        std::vector<int> member_rc(sb.members.size(), VPZ_OK);  // (a sub-batch is one synth call; after a failed one, a call per member)
        const auto t0 = Clock::now();
        static const bool no_synth = getenv("VPZM_NO_SYNTH") != nullptr;  // (diagnosis: the decode side of the pipeline alone)
        if (any && n_pk > 0 && !no_synth) {
            vpz_decoder *dec = decoder_for(L, sb.st);
            std::vector<int64_t> wr((size_t)S);
            // one synth call over the slot's packets [p0, p0 + n): the records, the Floor0 data and the statuses move with p0, the
            // residue offsets are the slot's
            auto call = [&](int64_t p0, int64_t n) -> int {
                int r = dec ? VPZ_OK : VPZ_E_NOMEM;
                // the decoder is re-used for new streams: back to what a StreamDecoder is after ProcessHeaderPackets
                // (`_currentPosition = 0; _hasPosition = true`, StreamDecoder.cs:165-168) -- a bare reset would leave the position to be
                // picked up from the first granule the way a seek does (:459-463), which moves the EOS trim (:658-666)
                if (r == VPZ_OK) r = vpz_decoder_reset(dec, -1);
                for (int sidx = 0; sidx < S && r == VPZ_OK; ++sidx) r = vpz_decoder_set_position(dec, sidx, 0);
                if (r == VPZ_OK && st.f0_stride > 0)
                    r = vpz_decoder_set_floor0_data(dec, sl.f0_amp + (size_t)p0 * C, sl.f0_coeff + (size_t)p0 * C * st.f0_stride, st.f0_stride);
                if (r == VPZ_OK) r = vpz_decoder_set_residue_format(dec, use_i16(st) ? VPZ_RESIDUE_I16 : VPZ_RESIDUE_F32);
                if (r == VPZ_OK) r = vpz_decoder_set_stream_capacities(dec, caps.data(), S);
                if (r == VPZ_OK)
                    r = vpz_decoder_synth(dec, n, sl.packets + p0, sl.residue, sb.res_floats, sl.posts + (size_t)p0 * 64 * C,
                                          sl.counts + (size_t)p0 * C, n * C, VPZ_MEM_HOST, out_at, offs.data(), cap, out_layout, 0, wr.data());
                if (r != VPZ_OK) m->fail(std::string("vpz_decoder_synth: ") + (dec ? vpz_context_last_error(L.ctx) : "no decoder"));
                int64_t not_ok = 0;
                if (r == VPZ_OK && vpz_decoder_last_packet_status(dec, nullptr, 0, &not_ok) == VPZ_OK && not_ok > 0) {
                    status.assign((size_t)n, 0);
                    vpz_decoder_last_packet_status(dec, status.data(), n, nullptr);
                    for (int64_t p = 0; p < n; ++p)
                        if (status[(size_t)p] != VPZ_OK) {
                            const int32_t sid = sl.packets[p0 + p].stream;
                            if (sid >= 0 && (size_t)sid < sb.members.size()) results[jobs[(size_t)sb.members[(size_t)sid]].k].skipped_packets += 1;
                        }
                }
                return r;
            };
            // (VPZM_FAIL_BATCH_CALLS=1, tests: every sub-batch's call counts as failed, so that the member-by-member path runs)
            const char *fb = getenv("VPZM_FAIL_BATCH_CALLS");
            const bool fail_batch = fb && atoi(fb) != 0;
            rc = !dec ? kNoDecoder : fail_batch ? VPZ_E_CAPACITY : call(0, n_pk);
            if (rc == VPZ_OK) {
                written = wr;
            } else if (rc == kNoDecoder) {
                // vpz_decoder_create refused the setup (its text is in vpzm_last_error): no member of it can be synthesised
                std::fill(member_rc.begin(), member_rc.end(), kNoDecoder);
            } else {
                // "a stream that fails costs only itself": whatever one member's packets did to the call, the others get a call
                // of their own (the packets lie member by member)
                for (int64_t p = 0; p < n_pk;) {
                    const int32_t sid = sl.packets[p].stream;
                    int64_t q = p;
                    while (q < n_pk && sl.packets[q].stream == sid) ++q;
                    if (sid >= 0 && (size_t)sid < sb.members.size()) {
                        member_rc[(size_t)sid] = q - p == n_pk && !fail_batch ? rc : call(p, q - p);
                        if (member_rc[(size_t)sid] == VPZ_OK) written[(size_t)sid] = wr[(size_t)sid];
                    }
                    p = q;
                }
            }
        }
        const double dt = seconds_since(t0);
        if (profile)
            fprintf(stderr, "[vpzm] group %d: sub-batch %zu synthesised at %.2f ms (call %.2f ms, %lld packets)\n", slot_index, b,
                    seconds_since(t_begin) * 1e3, dt * 1e3, (long long)n_pk);
        std::lock_guard<std::mutex> lk(mu);
        t_synth += dt;
        for (size_t j = 0; j < sb.members.size(); ++j) {
            Job &J = jobs[(size_t)sb.members[j]];
            if (J.status != VPZM_OK) continue;
            J.finished = true;
            if (member_rc[j] != VPZ_OK) {
                J.status = member_rc[j] == kNoDecoder ? VPZM_E_SETUP : member_rc[j] == VPZ_E_CAPACITY ? VPZM_E_CAPACITY : VPZM_E_SYNTH;
                continue;
            }
            results[J.k].samples = written[j];
            samples_total += written[j] * C;
        }
        sb.synth_done = true;
        cv.notify_all();
    }

    // run() for a thread of its own: whatever it throws before its pipeline stands (the job table's allocation) becomes the
    // statuses of the group's streams, never an exception out of the thread
    void run_guarded() noexcept
    {
        try {
            run();
        } catch (...) {
            for (int32_t k = lo; k < hi; ++k) {
                results[k] = vpzm_stream_result{};
                results[k].device_slot = slot_index;
                results[k].status = VPZM_E_SYNTH;
            }
            try {
                m->fail("vpzm_decode_library: a device group could not set its pipeline up (out of host memory)");
            } catch (...) {
            }
        }
    }

    void run()
    {
        t_begin = Clock::now();
        jobs.resize((size_t)(hi - lo));
        const size_t W = (size_t)wave_size();
        wave_left.assign((jobs.size() + W - 1) / W, 0);
        for (size_t w = 0; w < wave_left.size(); ++w) wave_left[w] = (int)(std::min(jobs.size(), (w + 1) * W) - w * W);
        std::vector<std::thread> pool;
        int workers = 0;
        try {
            for (int t = 0; t < threads; ++t) {
                pool.emplace_back([this] { guarded([this] { worker(); }); });
                ++workers;
            }
        } catch (...) {
        }
        if (workers == 0) {
            // no thread could be started: everything in order on this one
            guarded([this] {
                for (size_t i = 0; i < jobs.size(); ++i) open_one(i);
                for (int w = 0; w < (int)wave_left.size(); ++w) plan_wave(w);
                for (size_t b = 0; b < subs.size(); ++b) {
                    if (!prep_slot(b, subs[b])) {
                        for (int mi : subs[b].members) jobs[(size_t)mi].status = VPZM_E_SYNTH;
                        m->fail("vpzm_decode_library: page-locked batch arrays could not be allocated");
                    }
                    for (size_t j = 0; j < subs[b].members.size(); ++j) decode_member(b, subs[b], (int)j);
                    subs[b].decoded = (int)subs[b].members.size();
                    synth_sub(G.lanes[0], b, subs[b]);
                }
            });
            t_decode = seconds_since(t_begin);
        } else {
            try {
                for (size_t l = 1; l < G.lanes.size(); ++l) pool.emplace_back([this, l] { guarded([this, l] { issuer(G.lanes[l]); }); });
            } catch (...) {  // (fewer issuing threads: this thread's one drains every sub-batch)
            }
            guarded([this] { issuer(G.lanes[0]); });
        }
        for (std::thread &t : pool) t.join();
        for (Job &J : jobs) {
            if (J.h) vpzh_close(J.h);
            J.h = nullptr;
            if (aborted && J.status == VPZM_OK && !J.finished) {  // (an aborted run: no PCM, no count -- the stream has no result)
                J.status = VPZM_E_SYNTH;
                results[J.k].samples = 0;
            }
            results[J.k].status = J.status;
        }
        t_wall = seconds_since(t_begin);
    }
};

}  // namespace

extern "C" {

int vpzm_create(const int32_t *device_ids, int32_t n_devices, const vpzm_options *opt, vpzm_dispatcher **out)
{
    if (!device_ids || n_devices < 1 || n_devices > 64 || !out) return VPZM_E_ARG;
    *out = nullptr;
    std::unique_ptr<vpzm_dispatcher> m(new (std::nothrow) vpzm_dispatcher());
    if (!m) return VPZM_E_NOMEM;
    if (opt) m->opt = *opt;
    const bool threads_by_default = m->opt.host_threads <= 0;
    if (threads_by_default) m->opt.host_threads = vpzh_default_threads();
    if (m->opt.streams_per_call <= 0) m->opt.streams_per_call = 16;
    // (one GPU, 16 CPUs, 1 024 streams, 16-bit PCM, slots = 4 * contexts + 4: 2 contexts 90 ms, 3: 84 ms, 4: 78 ms -- a host-memory
    // synth call is upload, kernels, download in a row, and only other contexts' calls fill the link's other direction and the
    // device meanwhile.  Every context is an issuing thread that waits in hipStreamSynchronize, so few host threads get few)
    if (m->opt.contexts_per_device <= 0) m->opt.contexts_per_device = m->opt.host_threads / n_devices >= 8 ? 4 : 2;
    if (m->opt.contexts_per_device > 8) m->opt.contexts_per_device = 8;
    // (one GPU, 16 CPUs, 1 024 streams, 16-bit PCM: 6 slots 114 ms, 12 slots 101 ms, 24 slots 104 ms -- with few slots the decode of
    // sub-batch b + slots waits for the synth call of sub-batch b)
    // (the default number of decode threads: the CPUs the process may use PLUS one per issuing thread -- an issuer spends its time waiting
    // for its stream, and on 16 CPUs 20 decode threads beat 16: 74-76 ms against 76-83 for the 1 024-stream job)
    if (threads_by_default) m->opt.host_threads += n_devices * m->opt.contexts_per_device;
    if (m->opt.slots_per_device <= 0) m->opt.slots_per_device = 4 * m->opt.contexts_per_device + 4;
    if (m->opt.slots_per_device < m->opt.contexts_per_device + 1) m->opt.slots_per_device = m->opt.contexts_per_device + 1;
    m->groups.resize((size_t)n_devices);
    int rc = VPZM_OK;
    for (int d = 0; d < n_devices && rc == VPZM_OK; ++d) {
        Group &G = m->groups[(size_t)d];
        G.device = device_ids[d];
        G.lanes.resize((size_t)m->opt.contexts_per_device);
        G.slots.resize((size_t)m->opt.slots_per_device);
        for (Lane &L : G.lanes)
            if (vpz_context_create(G.device, &L.ctx) != VPZ_OK) {
                rc = VPZM_E_DEVICE;
                break;
            }
    }
    if (rc != VPZM_OK) {
        vpzm_destroy(m.release());
        return rc;
    }
    *out = m.release();
    return VPZM_OK;
}

void vpzm_destroy(vpzm_dispatcher *m)
{
    if (!m) return;
    for (Group &G : m->groups) {
        vpz_context *ctx0 = G.lanes.empty() ? nullptr : G.lanes[0].ctx;
        for (Slot &s : G.slots) {
            if (!ctx0) break;
            vpz_host_free(ctx0, s.packets);
            vpz_host_free(ctx0, s.residue);
            vpz_host_free(ctx0, s.posts);
            vpz_host_free(ctx0, s.counts);
            vpz_host_free(ctx0, s.f0_amp);
            vpz_host_free(ctx0, s.f0_coeff);
        }
        for (Lane &L : G.lanes) {
            for (auto &p : L.decs) vpz_decoder_destroy(p.second);
            if (L.ctx) vpz_context_destroy(L.ctx);
        }
    }
    delete m;
}

const char *vpzm_last_error(vpzm_dispatcher *m) { return m ? m->error.c_str() : "null dispatcher"; }
int vpzm_device_count(vpzm_dispatcher *m) { return m ? (int)m->groups.size() : 0; }

int vpzm_decode_library(vpzm_dispatcher *m, int32_t n, const uint8_t *const *data, const uint64_t *size, int32_t out_layout,
                        void *pcm_out, const int64_t *pcm_offset, const int64_t *pcm_capacity, vpzm_stream_result *results,
                        vpzm_stats *stats)
{
    if (!m || n < 0 || !results || (n > 0 && (!data || !size || !pcm_out || !pcm_offset || !pcm_capacity))) return VPZM_E_ARG;
    if (out_layout != VPZ_OUT_INTERLEAVED && out_layout != VPZ_OUT_INTERLEAVED_S16) return VPZM_E_ARG;
    for (int32_t k = 0; k < n; ++k)
        if (!data[k] || pcm_offset[k] < 0 || pcm_capacity[k] < 0) return VPZM_E_ARG;
    if (stats) *stats = vpzm_stats{};
    // (slots, contexts and decoder caches belong to one call at a time: a second caller waits here, it is not refused)
    std::lock_guard<std::mutex> one_call(m->call_mu);
    m->error.clear();
    const auto t0 = Clock::now();
    const int D = (int)m->groups.size();
    const int per_device = std::max(1, m->opt.host_threads / D);
    std::vector<std::unique_ptr<GroupRun>> runs;
    std::vector<std::thread> threads;
    try {
        for (int d = 0; d < D; ++d) {
            // shard_range (vorbispizza_amd/sharding.py): contiguous, sizes differ by at most one
            const int32_t lo = (int32_t)((int64_t)n * d / D), hi = (int32_t)((int64_t)n * (d + 1) / D);
            runs.emplace_back(new GroupRun(m, m->groups[(size_t)d], d, lo, hi, data, size, out_layout, pcm_out, pcm_offset,
                                           pcm_capacity, results, per_device));
        }
        for (int d = 1; d < D; ++d) threads.emplace_back([&runs, d] { runs[(size_t)d]->run_guarded(); });
        runs[0]->run_guarded();
    } catch (const std::bad_alloc &) {
        for (std::thread &t : threads) t.join();
        return VPZM_E_NOMEM;
    } catch (...) {
        for (std::thread &t : threads) t.join();
        m->fail("vpzm_decode_library: a device thread could not be started");
        return VPZM_E_NOMEM;
    }
    for (std::thread &t : threads) t.join();
    if (stats) {
        stats->wall_s = seconds_since(t0);
        stats->threads_per_device = per_device;
        size_t pinned = 0;
        for (const Group &G : m->groups)
            for (const Slot &sl : G.slots)
                pinned += sl.cap_packets * sizeof(vpz_packet) + sl.cap_residue * sizeof(float) + sl.cap_posts * sizeof(int16_t) + sl.cap_counts +
                          (sl.cap_f0 + sl.cap_f0c) * sizeof(float);
        stats->pinned_mib = (int32_t)std::min<size_t>(pinned >> 20, 0x7fffffff);
        for (int d = 0; d < D && d < 16; ++d) {
            stats->device_wall_s[d] = runs[(size_t)d]->t_wall;
            stats->device_decode_s[d] = runs[(size_t)d]->t_decode;
            stats->device_synth_s[d] = runs[(size_t)d]->t_synth;
            stats->device_streams[d] = runs[(size_t)d]->hi - runs[(size_t)d]->lo;
            stats->device_samples[d] = runs[(size_t)d]->samples_total;
        }
    }
    return VPZM_OK;
}

}  // extern "C"
