// synth_big_kernel -- the fused synthesis kernel for decoders whose LONG block is 4096 or 8192 samples (Mdct.cs:15-19 and
// StreamDecoder.cs:226-229 take any power of two from 64 to 8192; BlocksizeDerivedCache.cs:14-36, Mode.cs:30-66).  Until
// round 5 such streams went through three passes over HBM (generic_floor_kernel -> gathered IMDCT -> generic_ola_kernel ->
// generic_save_state_kernel: 32-36 bytes per sample for 8 algorithmic ones).  Here, as in synth_kernel:
//   one wavefront owns a RUN of consecutive blocks of ONE channel of one stream; per block it loads the spectrum (16-byte
//   pieces, the next block's in flight), renders the Floor1 curve into its LDS row and multiplies (Floor1.cs:222-268,
//   372-397), transforms (imdct8192_wave / imdct4096_wave, or the wave transform of the stream's short size, imdct_core.hpp),
//   windows and overlap-adds against the previous block's tail kept in LDS (StreamDecoder.cs:764-791), clips and stores
//   (StreamDecoder.cs:515-638, Utils.cs:44-58).  HBM traffic: the spectrum in, the PCM out.
// Inverse coupling and the Residue2 de-interleave (Mapping.cs:166-172, Residue2.cs:42-51) run as the separate
// coupling_tile_kernel pass into a planar temp in front of it (streams of these sizes are rare; the common uncoupled or
// already-floored packet reads the caller's vector directly), type-0 floors as floor0_apply_kernel on that temp.
// Arithmetic per sample is that of the three-pass path (the same wave transforms, ola(), table[index] * residue): the
// two give the same bits (tests/test_synth_gpu.py, VPZ_NO_BIG=1 selects the old path).
//
// LDS is sized at run time (the tables of the two block sizes, then per wave: h of the long size, its tail, the run's
// descriptors): 4 waves per workgroup and 78 KB for 4096 (two workgroups per CU).  An 8192 decoder keeps the TAIL in global
// memory instead (SynthArgs.big_tail, 8 KB per wave, written at the end of a block and read back one block later: it never
// leaves the L2): with h, tail and descriptors in LDS a CU held five waves, with h alone it holds eight -- two per SIMD, what
// the registers allow -- and this kernel is bound by the latencies of its dependent LDS round trips, i.e. by its occupancy.
#include <cstdlib>
#include <type_traits>

#include "imdct_core.hpp"
#include "synth_common.hpp"
#include "synth_desc.hpp"
#include "vpz_internal.hpp"

namespace vpz {

constexpr int kBigRunMax = kMaxRunLengthBig;
// 8192 decoders: seven waves of 17.4 KB beside 38 KB of tables -- the transform's 2048-entry twiddle table and the read half of its
// level-2 table are in LDS too (read from global memory they cost the pass some twenty trips to the L2, one behind the other:
// tools/kbench_slow_paths.py (c), 0.404 ms with eight waves and both in global memory, 0.376 with seven and the twiddles here)
#ifndef VPZ_BIG_WAVES_8192
#define VPZ_BIG_WAVES_8192 7
#endif
constexpr int kBigWaves8192 = VPZ_BIG_WAVES_8192;
constexpr int kBigTw8192 = kBigWaves8192 <= 7 ? 2048 + 512 : 0;  // entries of the 8192 twiddle tables kept in LDS: tw, and the half of w2 that is read
// wavefronts per workgroup: what the LDS holds -- 4 x 12.8 KB beside the tables for 4096 (two workgroups per CU), 8 x 17.4 KB for
// 8192 (one; the tails are in global memory)
__host__ __device__ inline int big_waves(int size1) { return size1 == 8192 ? kBigWaves8192 : 4; }

struct BigLayout {
    int tab_long, tab_short;  // float2 offsets of the two table sets (equal when the sizes are)
    int n_long, n_short;      // float2 entries staged for each
    int db;                   // byte offset of the inverse dB table (256 floats)
    int wave0;                // byte offset of wave 0's area
    int h_floats, tail_floats;
    int per_wave;             // bytes per wave
    int total;                // bytes
};

// table entries (float2) a block size keeps in LDS: 8192 leaves its two big twiddle tables in global memory
__host__ __device__ inline int big_table_entries(int n) { return n == 8192 ? 1088 : (n == 4096 ? kFast4096TableCount : kFastTableCount); }
__host__ __device__ inline int big_table_source(int n) { return n == 8192 ? kFast8192TwABOffset : 0; }

__host__ __device__ inline BigLayout big_layout(int size0, int size1)
{
    BigLayout L;
    L.n_long = big_table_entries(size1);
    L.n_short = size0 == size1 ? 0 : big_table_entries(size0);
    L.tab_long = 0;
    L.tab_short = size0 == size1 ? 0 : L.n_long;
    L.db = 0;
    L.wave0 = 0;  // (the tables are static arrays in front of the dynamic part)
    L.h_floats = size1 / 2;
    L.tail_floats = size1 == 8192 ? 0 : size1 / 4;  // (8192: SynthArgs.big_tail)
    L.per_wave = 4 * (L.h_floats + L.tail_floats) + 16 * 2 * (kBigRunMax + 1);
    L.total = L.wave0 + big_waves(size1) * L.per_wave;
    return L;
}

// the four bins of one LDS word of the rendered curve times four consecutive spectrum values
__device__ __forceinline__ void floor4(float4 &v, uint32_t w, const float *s_db)
{
    const float t0 = s_db[w & 0xFFu], t1 = s_db[(w >> 8) & 0xFFu], t2 = s_db[(w >> 16) & 0xFFu], t3 = s_db[w >> 24];
    v.x *= t0;
    v.y *= t1;
    v.z *= t2;
    v.w *= t3;
}

// kTailG: the long block is 8192 samples -- eight waves, the tails in global memory, 8.5 KB of long tables
template <bool kHasFloor, bool kS16, bool kTailG>
__global__ __launch_bounds__(kTailG ? 64 * kBigWaves8192 : 320, 1) void synth_big_kernel(SynthArgs a)
{
    using out_t = typename std::conditional<kS16, int16_t, float>::type;
    // the tables of the two block sizes: static (4096's set: tw 1024 | twAB 512 | twBC 64 | w 512; 8192 keeps twAB 512 | twBC 64 | w1 512
    // here and its two big tables in global memory), the waves' areas behind them in the dynamic part
    __shared__ __attribute__((aligned(16))) float2 s_tab_long[kTailG ? 1088 + kBigTw8192 : kFast4096TableCount];
    __shared__ __attribute__((aligned(16))) float2 s_tab_short[kFastTableCount];  // (the short size is at most 2048, or the long one)
    __shared__ float s_db[256];
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const BigLayout L = big_layout(a.size0, a.size1);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = a.channels;
    const int n_waves = (int)(blockDim.x >> 6), n_threads = (int)blockDim.x;
    const int item = blockIdx.x * n_waves + wave;
    const bool active = item < a.n_runs * C;
    const int run_idx = active ? item / C : 0;
    const int ch = active ? item - run_idx * C : 0;
    RunDesc run = a.runs[run_idx];

    for (int i = threadIdx.x; i < L.n_long; i += n_threads) s_tab_long[i] = a.tw_long[big_table_source(a.size1) + i];
    if (kTailG)
        for (int i = threadIdx.x; i < kBigTw8192; i += n_threads)
            s_tab_long[1088 + i] = a.tw_long[i < 2048 ? kFast8192TwOffset + i : kFast8192W2Offset + (i - 2048)];
    for (int i = threadIdx.x; i < L.n_short; i += n_threads) s_tab_short[i] = a.tw_short[big_table_source(a.size0) + i];
    if (threadIdx.x < 256) s_db[threadIdx.x] = kHasFloor ? a.inv_db[threadIdx.x] : 0.0f;
    __syncthreads();
    if (!active) return;

    float *hcur = reinterpret_cast<float *>(s_raw + L.wave0 + wave * L.per_wave);
    // (kTailG: this wave's own 8 KB of a.big_tail -- written and read back by this wave only, a block apart)
    float *tail = kTailG ? a.big_tail + (size_t)item * (a.size1 >> 2) : hcur + L.h_floats;
    uint4 *s_desc = reinterpret_cast<uint4 *>(hcur + L.h_floats + L.tail_floats);
    // the tail's stores of one block and its loads of the next are different lanes' (mirrored indices): ordered by a release / acquire
    // pair at workgroup scope -- a wait for the stores; the CU's cache is the same for both
    auto tail_written = [&]() { if (kTailG) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); };
    auto tail_wanted = [&]() { if (kTailG) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); };
    const int half1 = a.size1 >> 1;
    int prev_n4 = 0;

    auto size_of = [&](uint32_t flags) -> int { return (flags & kFrameLong) ? a.size1 : a.size0; };
    // the table set of a block size: the long set, or the short one
    auto tables_of = [&](int n) -> const float2 * { return n == a.size1 ? s_tab_long : s_tab_short; };

    // ---- the run's explicit descriptors into LDS
    const int fi0 = (run.pre_kind == kPreRecompute && run.count > 0) ? -1 : 0;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.frames + (run.first + fi0));
        const int n16 = (run.count - fi0) * 2;
        for (int i = lane; i < n16; i += 64) s_desc[i] = src[i];
    }
    __builtin_amdgcn_wave_barrier();
    int cc_run = 0;  // lane i: active posts of this wave's channel in the run's i-th staged frame
    if (kHasFloor && a.ccount != nullptr && lane < run.count - fi0) cc_run = a.ccount[(int)s_desc[2 * lane + 1].x + ch];
    auto frame_at = [&](int fi) -> FrameDesc {
        const uint4 lo = s_desc[(fi - fi0) * 2], hi = s_desc[(fi - fi0) * 2 + 1];
        FrameDesc fd;
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(lo.x), w1 = __builtin_amdgcn_readfirstlane(lo.y);
        const uint32_t w2 = __builtin_amdgcn_readfirstlane(lo.z), w3 = __builtin_amdgcn_readfirstlane(lo.w);
        const uint32_t w4 = __builtin_amdgcn_readfirstlane(hi.x), w5 = __builtin_amdgcn_readfirstlane(hi.y);
        const uint32_t w6 = __builtin_amdgcn_readfirstlane(hi.z), w7 = __builtin_amdgcn_readfirstlane(hi.w);
        fd.spec_off = (int64_t)(((uint64_t)w1 << 32) | w0);
        fd.out_off = (int64_t)(((uint64_t)w3 << 32) | w2);
        fd.rec = (int32_t)w4;
        fd.left_start = (uint16_t)(w5 & 0xFFFF);
        fd.packet_len = (uint16_t)(w5 >> 16);
        fd.prev_end = (uint16_t)(w6 & 0xFFFF);
        fd.out_count = (uint16_t)(w6 >> 16);
        fd.flags = w7;
        return fd;
    };

    if (run.pre_kind == kPreState) {
        const float *st = a.state_h + (size_t)run.state_slot * a.state_slot_floats + ((size_t)run.stream * C + ch) * half1;
        prev_n4 = run.prev_long ? (a.size1 >> 2) : (a.size0 >> 2);
        for (int i = lane; i < prev_n4; i += 64) tail[i] = st[i];
        tail_written();
    }
    out_t *out_base = reinterpret_cast<out_t *>(a.out) + (a.stream_out_off ? a.stream_out_off[run.stream] : 0);
    {
        const uint64_t ob = reinterpret_cast<uint64_t>(out_base);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ob), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ob >> 32));
        out_base = reinterpret_cast<out_t *>(((uint64_t)hi << 32) | lo);
    }
    float clip_peak = 0.0f;

    // The input of a frame: 16 registers of 16 bytes.  An 8192 block: X[2m] | X[2m + 1] = spectrum values 8k' .. 8k' + 7,
    // k' = lane + 64 m (imdct8192_wave's lo / hi); a 4096 block: X[m] = values 4k' .. 4k' + 3; smaller blocks: the eight
    // value pairs of load_spectrum in X[0..3].  ex: the channel has something to transform (ExecuteChannel, Mapping.cs:185).
    auto prefetch = [&](const FrameDesc &fd, int slot, bool valid, float4 (&X)[16], int &cp, int &cnt, bool &ex) {
        cnt = valid ? __builtin_amdgcn_readlane(cc_run, slot) : 0;
        ex = valid && (a.ccount == nullptr || (fd.flags & kFrameNoFloor) || cnt != 0);
        const int n = size_of(fd.flags);
        const VPZ_GLOBAL float *src = ex ? (const VPZ_GLOBAL float *)a.spec + fd.spec_off + (int64_t)ch * (n >> 1) : (const VPZ_GLOBAL float *)a.inv_db;
        int l = lane;
        asm volatile("" : "+v"(l));
        if (ex && n == 8192) {
            const VPZ_GLOBAL float4 *s4 = (const VPZ_GLOBAL float4 *)src;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                X[2 * m] = s4[2 * (l + 64 * m)];
                X[2 * m + 1] = s4[2 * (l + 64 * m) + 1];
            }
        } else if (ex && n == 4096) {
            const VPZ_GLOBAL float4 *s4 = (const VPZ_GLOBAL float4 *)src;
#pragma unroll
            for (int m = 0; m < 8; ++m) X[m] = s4[l + 64 * m];
#pragma unroll
            for (int m = 8; m < 16; ++m) X[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const VPZ_GLOBAL float2 *s2 = (const VPZ_GLOBAL float2 *)src;
            const int lpb = ex ? (n >> 5) : 1;
            const int k0 = l & (lpb - 1);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float2 p0 = s2[k0 + lpb * (2 * m)], p1 = s2[k0 + lpb * (2 * m + 1)];
                X[m] = make_float4(p0.x, p0.y, p1.x, p1.y);
            }
#pragma unroll
            for (int m = 4; m < 16; ++m) X[m] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (kHasFloor) {
            const bool floored = ex && !(fd.flags & kFrameNoFloor) && a.cposts != nullptr;
            const VPZ_GLOBAL int32_t *cpp = floored ? (const VPZ_GLOBAL int32_t *)a.cposts : (const VPZ_GLOBAL int32_t *)a.inv_db;
            cp = cpp[(size_t)(floored ? fd.rec + ch : 0) * 64 + l];
        }
    };

    float4 Xc[16];
    int cpcur = 0, cntcur = 0;
    bool excur = false;
    FrameDesc fd_next = frame_at(fi0);
    prefetch(fd_next, 0, run.count > 0 && !(fd_next.flags & kFrameDrain), Xc, cpcur, cntcur, excur);
#pragma unroll
    for (int m = 0; m < 16; ++m) asm volatile("" ::"v"(Xc[m].x), "v"(Xc[m].y), "v"(Xc[m].z), "v"(Xc[m].w));
    if (kHasFloor) asm volatile("" ::"v"(cpcur));

    const int iters = run.count - fi0;
    int fi = fi0;
    for (int it = 0; it < iters; ++it) {
        const FrameDesc fd = fd_next;
        const int fin = fi + 1;
        float4 Xn[16];
        int cpnext = 0, cntnext = 0;
        bool exnext = false;
        {
            const bool has_next = fin < run.count;
            fd_next = frame_at(has_next ? fin : fi);
            prefetch(fd_next, fin - fi0, has_next && !(fd_next.flags & kFrameDrain), Xn, cpnext, cntnext, exnext);
        }
        const bool drain = fd.flags & kFrameDrain;
        const int nblk = size_of(fd.flags);
        const int n4 = nblk >> 2;
        if (!drain) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float2 *h2 = reinterpret_cast<float2 *>(hcur);
            const bool floored = kHasFloor && excur && !(fd.flags & kFrameNoFloor) && a.cposts != nullptr;
            if (floored) {
                // the curve of this channel's record as one table index per bin in the wave's row (Floor1.cs:236-262, 372-397)
                const int n = nblk >> 1;
                uint8_t *row = reinterpret_cast<uint8_t *>(hcur);
                int *aux = reinterpret_cast<int *>(hcur) + (n >> 2);
                const int cp = ln < cntcur ? cpcur : 0;
                if (n > 1024 || !render_floor_indices_fast(row, aux, n, n, cp, cntcur, ln)) {
                    if (n > 1024) render_floor_indices<128>(row, aux, n, n, cp, cntcur, ln);
                    else render_floor_indices<32>(row, aux, n, n, cp, cntcur, ln);
                }
                __builtin_amdgcn_wave_barrier();
                const uint32_t *row32 = reinterpret_cast<const uint32_t *>(row);
                if (nblk == 8192) {
                    uint32_t w[16];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        w[2 * m] = row32[2 * (ln + 64 * m)];
                        w[2 * m + 1] = row32[2 * (ln + 64 * m) + 1];
                    }
#pragma unroll
                    for (int m = 0; m < 16; ++m) floor4(Xc[m], w[m], s_db);
                } else if (nblk == 4096) {
                    uint32_t w[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) w[m] = row32[ln + 64 * m];
#pragma unroll
                    for (int m = 0; m < 8; ++m) floor4(Xc[m], w[m], s_db);
                } else {
                    uint32_t fy[4];
                    load_floor_indices(fy, row, nblk >> 5, ln);
                    float2 x[8];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        x[2 * m] = make_float2(Xc[m].x, Xc[m].y);
                        x[2 * m + 1] = make_float2(Xc[m].z, Xc[m].w);
                    }
                    apply_floor(x, fy, s_db, true);
#pragma unroll
                    for (int m = 0; m < 4; ++m) Xc[m] = make_float4(x[2 * m].x, x[2 * m].y, x[2 * m + 1].x, x[2 * m + 1].y);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (!excur) {
                // Mapping.cs:190-194: a silent channel's block is all +0.0
                for (int i = ln; i < 2 * n4; i += 64) hcur[i] = 0.0f;
            } else if (nblk == 8192) {
                const float2 *tl = tables_of(8192);  // [twAB 512][twBC 64][w1 512]
                float4 lo[8], hi[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) { lo[m] = Xc[2 * m]; hi[m] = Xc[2 * m + 1]; }
                const float2 *gt = (nblk == a.size1 ? a.tw_long : a.tw_short);
                // (8192 is the long size, or both: the long set -- with the twiddle table behind it where that is staged)
                const float2 *tw = (kTailG && kBigTw8192 > 0) ? tl + 1088 : gt + kFast8192TwOffset;
#ifdef VPZ_BIG_W2_GLOBAL  // (A/B builds: the level-2 table read from global memory although its half is staged)
                const float2 *w2 = gt + kFast8192W2Offset;
#else
                const float2 *w2 = (kTailG && kBigTw8192 > 0) ? tl + 1088 + 2048 : gt + kFast8192W2Offset;
#endif
                imdct8192_wave(lo, hi, h2, tw, w2, tl + 576, tl, tl + 512, ln);
            } else if (nblk == 4096) {
                const float2 *tl = tables_of(4096);
                float4 xa[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) xa[m] = Xc[m];
                imdct4096_wave(xa, h2, tl + kFast4096TwOffset, tl + kFast4096TwABOffset, tl + kFast4096TwBCOffset, tl + kFast4096WOffset, ln);
            } else {
                const float2 *tl = tables_of(nblk);
                float2 x[8];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    x[2 * m] = make_float2(Xc[m].x, Xc[m].y);
                    x[2 * m + 1] = make_float2(Xc[m].z, Xc[m].w);
                }
                if (nblk == 2048) imdct2048_wave(x, h2, tl + kFastTwOffset, tl + kFastTwABOffset, tl + kFastTwBCOffset, ln);
                else if (nblk == 1024) imdct_mid_wave<4>(x, h2, tl + kFastTwOffset, tl + kFastTwABOffset, tl + kFastTwBCOffset, ln);
                else if (nblk == 512) imdct_mid_wave<2>(x, h2, tl + kFastTwOffset, tl + kFastTwABOffset, tl + kFastTwBCOffset, ln);
                else imdct256_wave8(x, h2, tl + kFastTwOffset, tl + kFastTwBCOffset, ln);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the next frame's input has to be here before this frame's stores are issued (vmcnt counts both, in order)
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("" ::"v"(Xn[m].x), "v"(Xn[m].y), "v"(Xn[m].z), "v"(Xn[m].w));
        if (kHasFloor) asm volatile("" ::"v"(cpnext));

        // ---- window + overlap-add + clip + store (StreamDecoder.cs:782-789, 515-638)
        if (fi >= 0 && fd.out_count > 0) {
            tail_wanted();
            const bool ilv = a.interleaved != 0;
            out_t *dst = ilv ? out_base + fd.out_off * C + ch : out_base + (int64_t)ch * a.channel_stride + fd.out_off;
            const int64_t ostep = ilv ? C : 1;
            const VPZ_GLOBAL float *slope = (const VPZ_GLOBAL float *)(((fd.flags & kFrameSlope1) || a.size0 == a.size1) ? a.slope1 : a.slope0);
            const int plen = fd.packet_len;
            const bool vec = !drain && ((fd.out_count | fd.left_start | plen | fd.prev_end) & 3) == 0 &&
                             (ilv || (reinterpret_cast<uintptr_t>(dst) & (kS16 ? 7 : 15)) == 0);
            int lv = lane;
            asm volatile("" : "+v"(lv));
            if (vec) {
                const float4 *h4 = reinterpret_cast<const float4 *>(hcur);
                const float4 *t4 = reinterpret_cast<const float4 *>(tail);
                const VPZ_GLOBAL float4 *t4g = (const VPZ_GLOBAL float4 *)t4;  // (kTailG)
                const VPZ_GLOBAL float4 *s4 = (const VPZ_GLOBAL float4 *)slope;
                const int cnt4 = fd.out_count >> 2;
                const int nr = (cnt4 + 63) >> 6;
                const int pn4 = prev_n4;
                // The window values (and, for 8192 decoders, the tail) come from global memory: a round that asks for its values and
                // uses them at once waits a whole trip to the L2 -- sixteen trips in a row for an 8192 block.  Four rounds' loads are
                // issued together, then their arithmetic and stores (lanes past the end clamp their reads and skip the store).
#ifndef VPZ_BIG_OLA_ROUNDS
#define VPZ_BIG_OLA_ROUNDS 4
#endif
                constexpr int kU = VPZ_BIG_OLA_ROUNDS;
                for (int r0 = 0; r0 < nr; r0 += kU) {
                    float4 hv[kU], pv[kU], wl[kU], wr[kU];
                    bool in_[kU], live_[kU], pc_[kU];
                    Y4Map mc_[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int g = lv + 64 * (r0 + u);
                        const bool live = g < cnt4;
                        const int i = (live ? g : cnt4 - 1) << 2;
                        const Y4Map mc = map_y4(fd.left_start + i, n4);
                        const bool in = i < plen;
                        const int ii = in ? i : 0;
                        const int q = fd.prev_end + ii;  // in [2 pn4, 4 pn4) whenever `in`
                        const bool pc = q >= 3 * pn4;
                        int pidx = (pc ? (4 * pn4 - 4 - q) : (q - 2 * pn4)) >> 2;
                        pidx = in ? pidx : 0;
                        const int ridx = in ? ((plen - 4 - ii) >> 2) : 0;
                        hv[u] = h4[mc.idx4];
                        if (kTailG) {  // (one base in scalar registers + a 32-bit offset per lane)
                            const VPZ_GLOBAL char *tb = (const VPZ_GLOBAL char *)t4g;
                            pv[u] = *(const VPZ_GLOBAL float4 *)(tb + (uint32_t)pidx * 16u);
                        } else {
                            pv[u] = t4[pidx];
                        }
                        wl[u] = s4[ii >> 2];
                        wr[u] = s4[ridx];
                        in_[u] = in;
                        live_[u] = live;
                        pc_[u] = pc;
                        mc_[u] = mc;
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int g = lv + 64 * (r0 + u);
                        const float4 v = apply_y4(hv[u], mc_[u].rev, mc_[u].neg);
                        const float4 t = apply_y4(pv[u], pc_[u], false);
                        float o0 = ola(v.x, wl[u].x, t.x, wr[u].w);
                        float o1 = ola(v.y, wl[u].y, t.y, wr[u].z);
                        float o2 = ola(v.z, wl[u].z, t.z, wr[u].y);
                        float o3 = ola(v.w, wl[u].w, t.w, wr[u].x);
                        o0 = in_[u] ? o0 : v.x;
                        o1 = in_[u] ? o1 : v.y;
                        o2 = in_[u] ? o2 : v.z;
                        o3 = in_[u] ? o3 : v.w;
                        if (a.clip) clip_group(o0, o1, o2, o3, clip_peak);
                        if (live_[u]) {
                            if (ilv) {
                                out_t *d = dst + (int64_t)(4 * g) * ostep;
                                store_pcm(d, kS16 ? (out_t)to_s16(o0) : (out_t)o0);
                                store_pcm(d + ostep, kS16 ? (out_t)to_s16(o1) : (out_t)o1);
                                store_pcm(d + 2 * ostep, kS16 ? (out_t)to_s16(o2) : (out_t)o2);
                                store_pcm(d + 3 * ostep, kS16 ? (out_t)to_s16(o3) : (out_t)o3);
                            } else if (kS16) {
                                store_nt(reinterpret_cast<uint2 *>(dst) + g, pack_s16(o0, o1), pack_s16(o2, o3));
                            } else {
                                store_pcm4(reinterpret_cast<float4 *>(dst) + g, make_float4(o0, o1, o2, o3));
                            }
                        }
                    }
                }
            } else {
                for (int i = lv; i < fd.out_count; i += 64) {
                    float v;
                    if (drain) {
                        v = tail_at(tail, fd.prev_end + i, prev_n4);
                    } else {
                        v = y_from_h(hcur, fd.left_start + i, n4);
                        if (i < plen) {
                            const float t = tail_at(tail, fd.prev_end + i, prev_n4);
                            v = ola(v, slope[i], t, slope[plen - 1 - i]);
                        }
                    }
                    if (a.clip) v = clip_track(v, clip_peak);
                    store_pcm(dst + i * ostep, kS16 ? (out_t)to_s16(v) : (out_t)v);
                }
            }
        }
        if (!drain) {
            // keep what a later block can overlap with: y[N/2 .. N) lives in the upper half of h
            __builtin_amdgcn_wave_barrier();
            if (n4 >= 256) {
                const float4 *s4 = reinterpret_cast<const float4 *>(hcur + n4);
                float4 *d4 = reinterpret_cast<float4 *>(tail);
                if (kTailG) {
                    int lt = lane;
                    asm volatile("" : "+v"(lt));  // (no 64-bit address of these stores may be kept in registers across the frame loop)
                    const VPZ_GLOBAL char *db = (const VPZ_GLOBAL char *)d4;
                    for (int i = lt; i < (n4 >> 2); i += 64) *(VPZ_GLOBAL float4 *)(db + (uint32_t)i * 16u) = s4[i];
                } else {
                    for (int i = lane; i < (n4 >> 2); i += 64) d4[i] = s4[i];
                }
            } else {
                for (int i = lane; i < n4; i += 64) tail[i] = hcur[n4 + i];
            }
            tail_written();
            prev_n4 = n4;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) Xc[m] = Xn[m];
        cpcur = cpnext;
        cntcur = cntnext;
        excur = exnext;
        fi = fin;
    }

    if ((run.flags & kRunSaveState) && prev_n4 > 0) {
        float *st = a.state_h + (size_t)(run.state_slot ^ 1) * a.state_slot_floats + ((size_t)run.stream * C + ch) * half1;
        tail_wanted();
        for (int i = lane; i < prev_n4; i += 64) st[i] = tail[i];
    }
    if (a.clip && __any(clip_peak > 0.99999994f) && lane == 0) atomicMax(&a.clipped[run.stream], run.clip_epoch);
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
bool synth_big_supported(int size0, int size1)
{
    auto fused = [](int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192; };
    // (a 4096 short block beside 8192 long ones would need a second table set of 17 KB in LDS: that pair keeps the three-pass path)
    return (size1 == 4096 || size1 == 8192) && fused(size0) && (size0 <= 2048 || size0 == size1);
}

// The kernel's dynamic LDS goes beyond the 64 KB a launch may ask for by default: the limit is raised to what the largest pair needs --
// ONE value whatever the decoder, so that contexts launching from several host threads (the dispatcher's lanes) never lower it under each other
template <typename K>
static hipError_t big_prepare(K kernel, int lds, bool tail_g)
{
    const int most = tail_g ? big_layout(8192, 8192).total : big_layout(4096, 4096).total;  // (per instantiation: its largest pair)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds > most ? lds : most);
}

hipError_t launch_synth_big(const SynthArgs &args, bool has_floor, hipStream_t stream)
{
    const long items = (long)args.n_runs * args.channels;
    if (items <= 0) return hipSuccess;
    const int waves = big_waves(args.size1);
    const int grid = (int)((items + waves - 1) / waves);
    const int lds = big_layout(args.size0, args.size1).total;
    hipError_t e = hipSuccess;
    const bool tail_g = args.size1 == 8192;
    if (tail_g && !args.big_tail) return hipErrorInvalidValue;
#define VPZ_LAUNCH_BIG(F, S)                                                                                  \
    do {                                                                                                      \
        if (tail_g) {                                                                                         \
            e = big_prepare(synth_big_kernel<F, S, true>, lds, true);                                               \
            if (e == hipSuccess) hipLaunchKernelGGL((synth_big_kernel<F, S, true>), dim3(grid), dim3(64 * waves), lds, stream, args); \
        } else {                                                                                              \
            e = big_prepare(synth_big_kernel<F, S, false>, lds, false);                                              \
            if (e == hipSuccess) hipLaunchKernelGGL((synth_big_kernel<F, S, false>), dim3(grid), dim3(64 * waves), lds, stream, args); \
        }                                                                                                     \
    } while (0)
    if (has_floor) {
        if (args.s16) VPZ_LAUNCH_BIG(true, true);
        else VPZ_LAUNCH_BIG(true, false);
    } else {
        if (args.s16) VPZ_LAUNCH_BIG(false, true);
        else VPZ_LAUNCH_BIG(false, false);
    }
#undef VPZ_LAUNCH_BIG
    return e != hipSuccess ? e : hipGetLastError();
}

// floats of SynthArgs.big_tail a launch of n_items (runs x channels) needs: a quarter block per item for an 8192 decoder, else none
int64_t synth_big_tail_floats(int size1, int64_t n_items) { return size1 == 8192 ? n_items * (size1 >> 2) : 0; }

// wavefronts (= channel-blocks in flight) the chip keeps resident under synth_big_kernel
int synth_big_resident_waves(bool has_floor, int num_cu, int size0, int size1)
{
    const int lds = big_layout(size0, size1).total;
    int per_cu = 0;
    const int waves = big_waves(size1);
    hipError_t e;
    if (size1 == 8192) {
        e = has_floor ? big_prepare(synth_big_kernel<true, false, true>, lds, true) : big_prepare(synth_big_kernel<false, false, true>, lds, true);
        if (e == hipSuccess)
            e = has_floor ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_big_kernel<true, false, true>, 64 * waves, lds)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_big_kernel<false, false, true>, 64 * waves, lds);
    } else {
        e = has_floor ? big_prepare(synth_big_kernel<true, false, false>, lds, false) : big_prepare(synth_big_kernel<false, false, false>, lds, false);
        if (e == hipSuccess)
            e = has_floor ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_big_kernel<true, false, false>, 64 * waves, lds)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, synth_big_kernel<false, false, false>, 64 * waves, lds);
    }
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return num_cu * per_cu * waves;
}

}  // namespace vpz
