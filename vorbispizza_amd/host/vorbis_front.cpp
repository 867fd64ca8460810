// C++ CPU front end: Ogg demux + Vorbis setup + per-packet entropy decode.
// Mirrors the reference's CPU stage so real .ogg files can be pushed through vpz_decoder_synth; every
// function cites the C# it follows.  Bit-serial / integer work only.
#include "../../include/vorbispizza_front.h"

#include <algorithm>
#include <atomic>
#include <thread>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <sched.h>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef VPZH_STATS
extern long g_vpzh_stats[64];  // diagnostic builds only (tools/front_prof.cpp)
#endif

namespace {

struct InvalidData : std::runtime_error {
    explicit InvalidData(const char *w) : std::runtime_error(w) {}
};
struct Unsupported : std::runtime_error {
    explicit Unsupported(const char *w) : std::runtime_error(w) {}
};

int ilog(int x)  // Utils.cs:19-28
{
    int cnt = 0;
    while (x > 0) { ++cnt; x >>= 1; }
    return cnt;
}

uint32_t bit_reverse32(uint32_t n)  // Utils.cs:30-42
{
    n = ((n & 0xAAAAAAAAu) >> 1) | ((n & 0x55555555u) << 1);
    n = ((n & 0xCCCCCCCCu) >> 2) | ((n & 0x33333333u) << 2);
    n = ((n & 0xF0F0F0F0u) >> 4) | ((n & 0x0F0F0F0Fu) << 4);
    n = ((n & 0xFF00FF00u) >> 8) | ((n & 0x00FF00FFu) << 8);
    return (n >> 16) | (n << 16);
}

float vorbis_float32(uint32_t bits)  // Utils.cs:92-105
{
    int sign = (int32_t)bits >> 31;
    int exponent = (int)((bits & 0x7fe00000u) >> 21) - 788;
    float mantissa = (float)((((int)(bits & 0x1fffffu)) ^ sign) + (sign & 1));
    return std::scalbn(mantissa, exponent);  // MathF.ScaleB
}

// ---------------------------------------------------------------------------------------------
// VorbisPacket bit reader (VorbisPacket.cs:157-292): LSB-first; reads past the end return the
// truncated value; only SkipBits past the end raises IsShort.
// ---------------------------------------------------------------------------------------------
struct BitReader {
    const uint8_t *data = nullptr;
    int64_t total_bits = 0, pos = 0;
    bool is_short = false;

    void init(const uint8_t *d, size_t bytes) { data = d; total_bits = (int64_t)bytes * 8; pos = 0; is_short = false; }
    int64_t remaining() const { return total_bits - pos; }

    uint64_t try_peek(int count, int *bits_read) const
    {
        int64_t rem = remaining();
        int n = (int)(rem < count ? rem : count);
        *bits_read = n;
        if (n <= 0) { *bits_read = 0; return 0; }
        const int64_t byte = pos >> 3;
        const int shift = (int)(pos & 7);
        if (n <= 56 && byte + 8 <= (total_bits >> 3)) {  // common case: one unaligned 8-byte load covers it
            uint64_t w;
            memcpy(&w, data + byte, 8);
            return (w >> shift) & (((uint64_t)1 << n) - 1);
        }
        // near the end of the packet, or more than 56 bits: gather up to 9 bytes
        const int need_bytes = (shift + n + 7) >> 3;
        unsigned __int128 acc = 0;
        for (int i = 0; i < need_bytes; ++i) acc |= (unsigned __int128)data[byte + i] << (8 * i);
        acc >>= shift;
        uint64_t v = (uint64_t)acc;
        if (n < 64) v &= ((uint64_t)1 << n) - 1;
        return v;
    }
    int skip(int count)  // SkipBits / SkipExtraBits
    {
        if (count <= 0) return 0;
        int64_t rem = remaining();
        if (rem >= count) { pos += count; return count; }
        pos = total_bits;
        is_short = true;
        return (int)rem;
    }
    uint64_t read_bits(int count)
    {
        int n;
        uint64_t v = try_peek(count, &n);
        pos += n;
        return v;
    }
    bool read_bit() { return read_bits(1) != 0; }
};

// ---------------------------------------------------------------------------------------------
// Codebook.cs + Huffman.cs
// ---------------------------------------------------------------------------------------------
struct HuffNode {
    int value = 0, length = 0, bits = 0, mask = 0;
};

struct Codebook {
    int dimensions = 0, entries = 0, map_type = 0, max_bits = 0, prefix_bits = 0;
    std::vector<int> lengths;
    std::vector<float> lookup;
    // the same values as 16-bit integers, when every one of them is one (bit for bit: no fraction, no -0.0, |v| < 2^15) -- else empty;
    // entry_l1: the largest sum of magnitudes over an entry's dimensions (what one vector can add to ONE bin at most, Residue0's quirk included)
    std::vector<int16_t> lookup_i16;
    double entry_l1 = 0.0;
    template <class T> const T *values() const;
    // prefix table of the first `prefix_bits` bits of a code (Huffman.cs:24-105), one 32-bit word per entry:
    // value << 6 | length, 0 = the code is longer than the table (overflow list).  4 KiB per book instead of the 16 KiB of
    // a table of nodes: the dozen books a packet touches stay in the L1 cache together.
    std::vector<uint32_t> prefix;
    std::vector<HuffNode> overflow;

    void read(BitReader &p)
    {
        if (p.read_bits(24) != 0x564342u) throw InvalidData("Book header had invalid signature!");
        dimensions = (int)p.read_bits(16);
        entries = (int)p.read_bits(24);
        lengths.assign(entries, 0);
        init_tree(p);
        map_type = (int)p.read_bits(4);
        init_lookup(p);
    }

    void init_tree(BitReader &p)  // Codebook.cs:44-145
    {
        bool sparse;
        int total = 0, max_len;
        if (p.read_bit()) {
            int len = (int)p.read_bits(5) + 1;
            for (int i = 0; i < entries;) {
                int cnt = (int)p.read_bits(ilog(entries - i));
                // codeword lengths above 32 index past the reference's `available` table (an exception there); with entries left
                // and nothing but empty lengths to come -- a header that ran out of bits reads zeros -- Codebook.cs:60-66 counts `len`
                // up until it wraps: no stream gets out of that loop with a usable book
                if (len > 32) throw InvalidData("codeword length above 32");
                while (--cnt >= 0) {
                    if (i >= entries) throw InvalidData("ordered codebook overruns its entry count");
                    lengths[i++] = len;
                }
                ++len;
            }
            sparse = false;
            max_len = len;
        } else {
            max_len = -1;
            sparse = p.read_bit();
            for (int i = 0; i < entries; i++) {
                if (!sparse || p.read_bit()) {
                    lengths[i] = (int)p.read_bits(5) + 1;
                    ++total;
                } else {
                    lengths[i] = -1;
                }
                if (lengths[i] > max_len) max_len = lengths[i];
            }
        }
        if (max_len <= -1) { max_bits = 0; return; }  // Huffman.Empty
        max_bits = max_len;

        // ComputeCodewords (Codebook.cs:147-218), always in the non-sparse arrangement: the sparse
        // one only changes how the same (symbol, length, code) triples are stored
        std::vector<int> codes(entries, 0);
        if (!compute_codewords(codes)) throw InvalidData("over-specified Huffman tree");
        generate_table(codes);
        (void)sparse;
        (void)total;
    }

    bool compute_codewords(std::vector<int> &codes)
    {
        uint32_t available[33];
        memset(available, 0, sizeof available);
        int k;
        for (k = 0; k < entries; ++k)
            if (lengths[k] > 0) break;
        if (k == entries) return true;
        codes[k] = 0;
        for (int i = 1; i <= lengths[k]; ++i) available[i] = 1u << (32 - i);
        for (int i = k + 1; i < entries; ++i) {
            int z = lengths[i];
            if (z <= 0) continue;
            while (z > 0 && available[z] == 0) --z;
            if (z == 0) return false;
            uint32_t res = available[z];
            available[z] = 0;
            codes[i] = (int)bit_reverse32(res);
            if (z != lengths[i])
                for (int y = lengths[i]; y > z; --y) available[y] = res + (1u << (32 - y));
        }
        return true;
    }

    void generate_table(const std::vector<int> &codes)  // Huffman.cs:24-105
    {
        int count = 0, last_valid = -1, max_len = 0;
        std::vector<HuffNode> list;
        list.reserve(entries);
        for (int i = 0; i < entries; ++i) {
            int len = lengths[i];
            if (len != 0) { ++count; last_valid = i; }
            if (len > 0) {
                HuffNode n;
                n.value = i;
                n.length = len;
                n.bits = codes[i];
                n.mask = (int)((1u << (len & 31)) - 1u);
                list.push_back(n);
                if (len > max_len) max_len = len;
            }
        }
        if (count == 1 && lengths[last_valid] != 1) throw InvalidData("Invalid single entry.");
        int table_bits = max_len > 10 ? 10 : max_len;
        prefix_bits = table_bits;
        prefix.assign((size_t)1 << table_bits, 0u);
        for (const HuffNode &n : list) {
            if (n.length > table_bits) {
                overflow.push_back(n);
            } else {
                const int max_val = 1 << (table_bits - n.length);
                const uint32_t word = ((uint32_t)n.value << 6) | (uint32_t)n.length;  // entries < 2^24, length <= 32
                for (int j = 0; j < max_val; ++j) prefix[(size_t)((j << n.length) | n.bits)] = word;
            }
        }
    }

    static int lookup1_values(int entries, int dimensions)  // Codebook.cs:290-298
    {
        int r = (int)std::floor(std::exp(std::log((double)entries) / dimensions));
        if (std::floor(std::pow((double)r + 1, dimensions)) <= entries) ++r;
        return r;
    }

    void init_lookup(BitReader &p)  // Codebook.cs:220-288
    {
        if (map_type == 0) return;
        if (map_type > 2) throw InvalidData("invalid codebook lookup type");
        float min_value = vorbis_float32((uint32_t)p.read_bits(32));
        float delta_value = vorbis_float32((uint32_t)p.read_bits(32));
        int value_bits = (int)p.read_bits(4) + 1;
        bool sequence_p = p.read_bit();
        size_t count = (size_t)entries * dimensions;
        lookup.assign(count, 0.f);
        size_t mcount = map_type == 1 ? (size_t)lookup1_values(entries, dimensions) : count;
        std::vector<uint16_t> mult(mcount);
        for (size_t i = 0; i < mcount; ++i) mult[i] = (uint16_t)p.read_bits(value_bits);
        // (separately rounded multiply, add, add: the translation unit is built with -ffp-contract=off)
        // map type 1: dimension i of entry idx takes multiplicand (idx / mcount^i) % mcount -- the digits of idx in base
        // mcount, kept as an odometer instead of two divisions per value (Codebook.cs:262-270 wraps its uint `idxDiv` the
        // same way: a digit whose weight has overflowed 2^32 is computed from the wrapped weight)
        std::vector<float> scaled(mcount);
        for (size_t i = 0; i < mcount; ++i) scaled[i] = (float)mult[i] * delta_value;
        bool plain_digits = map_type == 1;
        if (map_type == 1) {  // the odometer equals the reference's arithmetic while mcount^(dimensions-1) fits 32 bits
            uint64_t wgt = 1;
            for (int i = 1; i < dimensions && plain_digits; ++i) {
                wgt *= mcount;
                if (wgt > 0xFFFFFFFFull) plain_digits = false;
            }
        }
        std::vector<uint32_t> digit((size_t)(dimensions > 0 ? dimensions : 1), 0u);
        for (int idx = 0; idx < entries; ++idx) {
            float last = 0.f;
            float *dst = lookup.data() + (size_t)idx * dimensions;  // (a book without dimensions has no values)
            if (map_type == 1 && plain_digits) {
                for (int i = 0; i < dimensions; ++i) {
                    const float sum = scaled[digit[i]] + min_value;
                    const float value = sum + last;
                    dst[i] = value;
                    if (sequence_p) last = value;
                }
                for (int i = 0; i < dimensions; ++i) {  // idx + 1 in base mcount
                    if (++digit[i] < (uint32_t)mcount) break;
                    digit[i] = 0;
                }
            } else {
                uint32_t idx_div = 1;
                for (int i = 0; i < dimensions; ++i) {
                    // (a weight that wrapped to 0 is the reference's DivideByZeroException: the value of digit 0 stands in)
                    const size_t moff = map_type == 1 ? (size_t)(idx_div ? ((uint32_t)idx / idx_div) % (uint32_t)mcount : 0)
                                                      : (size_t)idx * dimensions + i;
                    const float sum = scaled[moff] + min_value;
                    const float value = sum + last;
                    dst[i] = value;
                    if (sequence_p) last = value;
                    if (map_type == 1) idx_div *= (uint32_t)mcount;
                }
            }
        }
        // (ABI v5) the integer form of the table, if it has one
        bool integral = !lookup.empty();
        for (float v : lookup) {
            const float back = (float)(int16_t)(int32_t)v;
            integral = integral && fabsf(v) < 32768.0f && memcmp(&back, &v, sizeof v) == 0;
        }
        if (integral) {
            lookup_i16.resize(lookup.size());
            for (size_t i = 0; i < lookup.size(); ++i) lookup_i16[i] = (int16_t)(int32_t)lookup[i];
        }
        for (int idx = 0; idx < entries && dimensions > 0; ++idx) {
            double l1 = 0.0;
            for (int i = 0; i < dimensions; ++i) l1 += fabs((double)lookup[(size_t)idx * dimensions + i]);
            entry_l1 = std::max(entry_l1, l1);
        }
    }

    int decode_scalar(BitReader &p) const  // Codebook.cs:301-335
    {
        // fast path, same result as the general one below: at least 64 bits left, so the peek cannot come up
        // short and the skip cannot overrun
        if (p.pos + 64 <= p.total_bits && !prefix.empty()) {
            uint64_t w;
            memcpy(&w, p.data + (p.pos >> 3), 8);
            const uint32_t e = prefix[(size_t)((w >> (p.pos & 7)) & (((uint64_t)1 << prefix_bits) - 1))];
            if (e & 63u) {
                p.pos += e & 63u;
                return (int)(e >> 6);
            }
        }
        int n;
        uint64_t data = p.try_peek(prefix_bits, &n);
        if (n != 0 && !prefix.empty()) {
            const uint32_t e = prefix[(size_t)data];
            if (e & 63u) {
                p.skip((int)(e & 63u));
                return (int)(e >> 6);
            }
        }
        int d = (int)p.try_peek(max_bits, &n);
        if (n != 0) {
            for (const HuffNode &node : overflow)
                if (node.bits == (d & node.mask)) {
                    p.skip(node.length);
                    return node.value;
                }
        }
        return -1;
    }
};

// ---------------------------------------------------------------------------------------------
// Floor1.cs:39-219
// ---------------------------------------------------------------------------------------------
template <> inline const float *Codebook::values<float>() const { return lookup.data(); }
template <> inline const int16_t *Codebook::values<int16_t>() const { return lookup_i16.data(); }

struct Floor1 {
    std::vector<uint8_t> partition_class, class_dimensions, class_subclasses, class_masterbooks;
    std::vector<std::vector<int>> subclass_books;
    std::vector<int> x_list;
    int multiplier = 0, range = 0, y_bits = 0;

    void read(BitReader &p, int n_books)
    {
        static const int range_lookup[4] = {128, 64, 43, 32};
        static const int ybits_lookup[4] = {8, 7, 7, 6};
        int maximum_class = -1;
        partition_class.resize((size_t)p.read_bits(5));
        for (auto &pc : partition_class) {
            pc = (uint8_t)p.read_bits(4);
            if (pc > maximum_class) maximum_class = pc;
        }
        maximum_class += 1;
        class_dimensions.assign(maximum_class, 0);
        class_subclasses.assign(maximum_class, 0);
        class_masterbooks.assign(maximum_class, 0);
        subclass_books.assign(maximum_class, {});
        for (int i = 0; i < maximum_class; ++i) {
            class_dimensions[i] = (uint8_t)(p.read_bits(3) + 1);
            class_subclasses[i] = (uint8_t)p.read_bits(2);
            if (class_subclasses[i] > 0) class_masterbooks[i] = (uint8_t)p.read_bits(8);
            subclass_books[i].resize((size_t)1 << class_subclasses[i]);
            for (auto &b : subclass_books[i]) {
                int book = (int)p.read_bits(8) - 1;
                if (book >= n_books) throw InvalidData("floor1 subclass book out of range");
                b = book;
            }
        }
        int mult = (int)p.read_bits(2);
        range = range_lookup[mult] * 2;
        y_bits = ybits_lookup[mult];
        multiplier = mult + 1;
        int range_bits = (int)p.read_bits(4);
        x_list.clear();
        x_list.push_back(0);
        x_list.push_back(1 << range_bits);
        for (uint8_t cls : partition_class)
            for (int j = 0; j < class_dimensions[cls]; ++j) x_list.push_back((int)p.read_bits(range_bits));
        for (size_t i = 0; i < x_list.size(); ++i)
            for (size_t j = i + 1; j < x_list.size(); ++j)
                if (x_list[i] == x_list[j]) throw InvalidData("duplicate floor1 X value");  // Floor1.cs:140-141
    }

    // Floor1.Unpack :162-219 -> posts[64], returns PostCount
    int unpack(BitReader &p, const std::vector<Codebook> &books, int *posts) const
    {
        if (!p.read_bit()) return 0;
        int post_count = 2;
        posts[0] = (int)p.read_bits(y_bits);
        posts[1] = (int)p.read_bits(y_bits);
        for (size_t i = 0; i < partition_class.size(); ++i) {
            int cls = partition_class[i];
            int cdim = class_dimensions[cls];
            int cbits = class_subclasses[cls];
            int csub = (1 << cbits) - 1;
            uint32_t cval = 0;
            if (cbits > 0) {
                // the reference indexes its codebook array unchecked here (IndexOutOfRangeException)
                if (class_masterbooks[cls] >= books.size()) throw InvalidData("floor1 master book out of range");
                int v = books[class_masterbooks[cls]].decode_scalar(p);
                if (v == -1) return 0;  // bad value: bail, PostCount = 0
                cval = (uint32_t)v;
            }
            for (int j = 0; j < cdim; ++j) {
                int book_idx = subclass_books[cls][cval & (uint32_t)csub];
                cval >>= cbits;
                int post = 0;
                if (book_idx >= 0) {
                    post = books[book_idx].decode_scalar(p);
                    if (post == -1) return 0;
                }
                if (post_count < 64) posts[post_count] = post;  // Posts is int[64] (Floor1.cs:17)
                ++post_count;
            }
        }
        return post_count;
    }
};

// ---------------------------------------------------------------------------------------------
// Floor0.cs:37-80 (header) and :113-162 (Unpack)
// ---------------------------------------------------------------------------------------------
struct Floor0 {
    int order = 0, rate = 0, bark_map_size = 0, amp_bits = 0, amp_ofs = 0;
    std::vector<uint8_t> book_list;

    void read(BitReader &p, const std::vector<Codebook> &cbs)
    {
        order = (int)p.read_bits(8);
        rate = (int)p.read_bits(16);
        bark_map_size = (int)p.read_bits(16);
        amp_bits = (int)p.read_bits(6);
        amp_ofs = (int)p.read_bits(8);
        book_list.resize((size_t)p.read_bits(4) + 1);
        if (order < 1 || rate < 1 || bark_map_size < 1) throw InvalidData("invalid floor0 header");
        for (auto &b : book_list) {
            b = (uint8_t)p.read_bits(8);
            if (b >= cbs.size() || cbs[b].map_type == 0 || cbs[b].dimensions < 1) throw InvalidData("invalid floor0 book");
        }
    }

    // returns Data.Amp; coeff[order] = Data.Coeff
    float unpack(BitReader &p, const std::vector<Codebook> &cbs, float *coeff) const
    {
        for (int i = 0; i < order; ++i) coeff[i] = 0.f;
        const uint64_t amp_raw = p.read_bits(amp_bits);
        // C# `(1 << _ampBits) - 1` on int: the shift count is taken modulo 32 and the subtraction wraps
        const double amp_div = (double)(int32_t)((1u << (amp_bits & 31)) - 1u);
        float amp = (float)((double)(amp_raw * (uint64_t)amp_ofs) / amp_div);  // (float)(amp * _ampOfs / ampDiv)
        const uint32_t book_num = (uint32_t)p.read_bits(ilog((int)book_list.size()));
        if (book_num >= book_list.size()) return 0.f;
        const Codebook &book = cbs[book_list[book_num]];
        for (int i = 0; i < order;) {
            const int entry = book.decode_scalar(p);
            if (entry == -1) return 0.f;
            const float *lk = &book.lookup[(size_t)entry * book.dimensions];
            for (int j = 0; i < order && j < book.dimensions; ++j, ++i) coeff[i] = lk[j];
        }
        const int dim = book.dimensions;
        float last = 0.f;
        for (int j = 0; j < order;) {
            for (int k = 0; j < order && k < dim; ++j, ++k) coeff[j] += last;
            last = coeff[j - 1];
        }
        return amp;
    }
};

// ---------------------------------------------------------------------------------------------
// Residue0.cs / Residue1.cs / Residue2.cs
// ---------------------------------------------------------------------------------------------
struct Residue {
    int type = 0, begin = 0, end = 0, partition_size = 0, classifications = 0, class_book = 0, max_stages = 0;
    std::vector<uint8_t> cascade;
    std::vector<std::vector<uint8_t>> books;  // per class: book per stage (empty: null)
    std::vector<int> decode_map;

    void read(BitReader &p, int type_, const std::vector<Codebook> &cbs)
    {
        type = type_;
        begin = (int)p.read_bits(24);
        end = (int)p.read_bits(24);
        partition_size = (int)p.read_bits(24) + 1;
        classifications = (int)p.read_bits(6) + 1;
        class_book = (int)p.read_bits(8);
        cascade.assign(classifications, 0);
        int acc = 0;
        for (auto &c : cascade) {
            uint32_t low = (uint32_t)p.read_bits(4);
            uint32_t bits = low & 7u;
            if (low & 8u) bits |= (uint32_t)p.read_bits(5) << 3;
            c = (uint8_t)bits;
            acc += __builtin_popcount(bits);
        }
        std::vector<uint8_t> book_nums(acc);
        for (auto &b : book_nums) {
            b = (uint8_t)p.read_bits(8);
            if (b >= cbs.size() || cbs[b].map_type == 0) throw InvalidData("residue book without value mapping");
            // a value book without dimensions: `partitionSize / dimensions` (Residue0.cs:213) is the reference's
            // DivideByZeroException, and Residue1's `i += dimensions` would never advance
            if (cbs[b].dimensions < 1) throw InvalidData("residue book with zero dimensions");
        }
        if (class_book >= (int)cbs.size()) throw InvalidData("residue class book out of range");
        const Codebook &cb = cbs[class_book];
        int partvals = 1;
        for (int i = 0; i < cb.dimensions; ++i) {
            partvals *= classifications;
            if (partvals > cb.entries) throw InvalidData("residue classbook too small");
        }
        books.assign(classifications, {});
        acc = 0;
        int maxstage = 0;
        for (int j = 0; j < classifications; ++j) {
            int stages = ilog(cascade[j]);
            if (stages <= 0) continue;
            books[j].assign(stages, 0);
            if (stages > maxstage) maxstage = stages;
            for (int k = 0; k < stages; ++k)
                if (cascade[j] & (1 << k)) books[j][k] = book_nums[acc++];
        }
        max_stages = maxstage;
        class_dim = cb.dimensions;
        decode_map.assign((size_t)partvals * cb.dimensions, 0);
        for (int j = 0; j < partvals; ++j) {
            int val = j, mult = partvals / classifications;
            for (int k = 0; k < cb.dimensions; ++k) {
                int deco = val / mult;
                val -= deco * mult;
                mult /= classifications;
                decode_map[(size_t)j * cb.dimensions + k] = deco;
            }
        }
        finish_setup();
    }

    // per (classification, stage): the value book, or -1 (flattened `books`: one load in the partition loop)
    std::vector<int16_t> stage_book;
    // per (class word, stage): which of the word's partitions have a book at that stage (bit k: partition k of the word).
    // Later stages code few partitions: the loop below visits those instead of asking every partition of every stage.
    std::vector<uint32_t> word_stage_mask;
    int class_dim = 0;  // dimensions of the class book (partitions per class word); the masks exist for <= 32
    void finish_setup()
    {
        stage_book.assign((size_t)classifications * 8, -1);
        for (int j = 0; j < classifications; ++j)
            for (size_t k = 0; k < books[j].size() && k < 8; ++k)
                if (cascade[j] & (1 << k)) stage_book[(size_t)j * 8 + k] = books[j][k];
        word_stage_mask.clear();
        if (class_dim >= 1 && class_dim <= 32) {
            const size_t words = decode_map.size() / (size_t)class_dim;
            word_stage_mask.assign(words * 8, 0u);
            for (size_t w = 0; w < words; ++w)
                for (int k = 0; k < class_dim; ++k)
                    for (int st = 0; st < 8; ++st)
                        if (stage_book[(size_t)decode_map[w * class_dim + k] * 8 + st] >= 0) word_stage_mask[w * 8 + st] |= 1u << k;
        }
    }

    // Residue1.WriteVectors (Residue1.cs:12-34) for a partition that lies inside the channel with room for a last
    // vector that overhangs it: no per-vector bounds test, the add loop specialised for the usual dimensions.
    // Returns true when the packet ran out (Codebook.DecodeScalar == -1).
    // T: float (what the reference accumulates, Residue*.cs) or int16_t (ABI v5: the same sums as integers -- exact when the setup
    // header guarantees integers, residue_integral below -- written straight into the int16 vector the device is given)
    template <int kDim, class T>
    static bool write_vectors_fast(const Codebook &cb, BitReader &p, T *dst, int partition_size)
    {
        const T *lookup = cb.values<T>();
        const int dim = kDim ? kDim : cb.dimensions;
        int i = 0;
        // The bulk of a packet: a 64-bit window of the bit stream in a register, refilled every few symbols -- the chain
        // from one symbol to the next is a shift and one table read instead of an address computation, an unaligned load,
        // a shift and the table read.  Same bits consumed, same entries as Codebook.DecodeScalar (Codebook.cs:301-335);
        // codes longer than the prefix table and the last bytes of the packet take decode_scalar itself.
        if (!cb.prefix.empty()) {
            const uint32_t *table = cb.prefix.data();
            const uint64_t mask = ((uint64_t)1 << cb.prefix_bits) - 1;
            int64_t pos = p.pos;
            const int64_t safe_end = p.total_bits - 64;  // an 8-byte load at or below this bit position stays inside
            while (i < partition_size && pos <= safe_end) {
                uint64_t w;
                memcpy(&w, p.data + (pos >> 3), 8);
                int avail = 64 - (int)(pos & 7);
                w >>= (pos & 7);
                // symbols out of this window: each needs at most prefix_bits (<= 10) valid bits
                while (i < partition_size && avail >= 10) {
                    const uint32_t e = table[w & mask];
                    const int len = (int)(e & 63u);
                    if (len == 0) goto general;  // longer than the table: the overflow list
                    w >>= len;
                    avail -= len;
                    pos += len;
                    const T *lk = lookup + (size_t)(e >> 6) * dim;
                    if (kDim) {
#pragma GCC unroll 8
                        for (int j = 0; j < kDim; ++j) dst[i + j] += lk[j];
                    } else {
                        for (int j = 0; j < dim; ++j) dst[i + j] += lk[j];
                    }
                    i += dim;
                }
            }
        general:
            p.pos = pos;
        }
        for (; i < partition_size; i += dim) {
            const int entry = cb.decode_scalar(p);
            if (entry == -1) return true;
            const T *lk = lookup + (size_t)entry * dim;
            if (kDim) {
#pragma GCC unroll 8
                for (int j = 0; j < kDim; ++j) dst[i + j] += lk[j];
            } else {
                for (int j = 0; j < dim; ++j) dst[i + j] += lk[j];
            }
        }
        return false;
    }

    // WriteVectors: Residue0.cs:208-231 (type 0, sums the entry into ONE bin: quirk q9) and
    // Residue1.cs:12-34 (types 1 and 2)
    template <class T>
    bool write_vectors(const Codebook &cb, BitReader &p, T *chan, int chan_len, int offset) const
    {
#ifdef VPZH_STATS
        g_vpzh_stats[cb.dimensions < 16 ? cb.dimensions : 15] += 1;
        g_vpzh_stats[16] = partition_size;
        g_vpzh_stats[17] = type;
        g_vpzh_stats[20 + (cb.prefix_bits < 11 ? cb.prefix_bits : 11)] += 1;
        g_vpzh_stats[40 + (cb.max_bits < 20 ? cb.max_bits : 20)] += 1;
#endif
        if (type != 0) {
            const int dim = cb.dimensions;
            const int reach = (partition_size + dim - 1) / dim * dim;  // a last vector may overhang the partition
            if (offset + reach <= chan_len) {
                T *dst = chan + offset;
                switch (dim) {
                    case 1: return write_vectors_fast<1, T>(cb, p, dst, partition_size);
                    case 2: return write_vectors_fast<2, T>(cb, p, dst, partition_size);
                    case 4: return write_vectors_fast<4, T>(cb, p, dst, partition_size);
                    case 8: return write_vectors_fast<8, T>(cb, p, dst, partition_size);
                    default: return write_vectors_fast<0, T>(cb, p, dst, partition_size);
                }
            }
        }
        if (type == 0) {
            int steps = partition_size / cb.dimensions;
            for (int step = 0; step < steps; ++step) {
                int entry = cb.decode_scalar(p);
                if (entry == -1) return true;
                T r = 0;
                const T *lk = cb.values<T>() + (size_t)entry * cb.dimensions;
                for (int d = 0; d < cb.dimensions; ++d) { volatile T t = (T)(r + lk[d]); r = t; }  // (float: every sum rounded, as the reference's)
                if (offset + step < chan_len) { volatile T t = (T)(chan[offset + step] + r); chan[offset + step] = t; }
            }
            return false;
        }
        for (int i = 0; i < partition_size;) {
            int entry = cb.decode_scalar(p);
            if (entry == -1) return true;
            const T *lk = cb.values<T>() + (size_t)entry * cb.dimensions;
            if (offset + i + cb.dimensions > chan_len) throw InvalidData("residue vector overruns the block");
            for (int j = 0; j < cb.dimensions; ++j) chan[offset + i + j] = (T)(chan[offset + i + j] + lk[j]);
            i += cb.dimensions;
        }
        return false;
    }

    // Residue0.Decode :117-206.  buffer: `count` channels at stride `stride`.
    // (part_word_cache: the caller's scratch -- a Residue belongs to a setup that many streams share, so it is not written here)
    template <class T>
    void decode(BitReader &p, const std::vector<uint8_t> &do_not_decode, int block_size, T *buffer, int stride,
                const std::vector<Codebook> &cbs, std::vector<int> &part_word_cache) const
    {
        int half = block_size / 2;
        int b = begin < half ? begin : half;
        int e = end < half ? end : half;
        int n = e - b;
        if (n <= 0) return;
        const int count = (int)do_not_decode.size();
        int partition_count = n / partition_size;
        const Codebook &cb = cbs[class_book];
        int dim = cb.dimensions;
        if (dim < 1) throw InvalidData("residue class book without dimensions");  // DivideByZeroException there
        int partition_words = (partition_count + dim - 1) / dim;
        part_word_cache.assign((size_t)count * partition_words, 0);
        for (int stage = 0; stage < max_stages; ++stage) {
            for (int partition_idx = 0, entry_idx = 0; partition_idx < partition_count; ++entry_idx) {
                if (stage == 0) {
                    for (int ch = 0; ch < count; ++ch) {
                        if (do_not_decode[ch]) continue;
                        int idx = cb.decode_scalar(p);
                        if (idx >= 0 && idx < (int)decode_map.size() / dim) {
                            part_word_cache[(size_t)ch * partition_words + entry_idx] = idx;
                        } else {
                            partition_idx = partition_count;
                            stage = max_stages;
                            break;
                        }
                    }
                }
                if (count == 1 && partition_idx < partition_count && !do_not_decode[0] && !word_stage_mask.empty() && stage < 8) {
                    // one channel (every Residue2 packet): only the partitions of this word that have a book at this stage,
                    // in their order -- the bit stream is read exactly as by the loop below
                    const int cw = part_word_cache[(size_t)entry_idx];
                    const int left = partition_count - partition_idx;
                    uint32_t m = word_stage_mask[(size_t)cw * 8 + stage];
                    if (left < dim) m &= (1u << left) - 1u;
                    while (m) {
                        const int dim_idx = __builtin_ctz(m);
                        m &= m - 1;
                        const int idx = decode_map[(size_t)cw * dim + dim_idx];
                        const Codebook &book = cbs[stage_book[(size_t)idx * 8 + stage]];
                        if (write_vectors(book, p, buffer, stride, b + (partition_idx + dim_idx) * partition_size)) {
                            partition_idx = partition_count;
                            stage = max_stages;
                            break;
                        }
                    }
                    if (partition_idx < partition_count) partition_idx += left < dim ? left : dim;
                    continue;
                }
                for (int dim_idx = 0; partition_idx < partition_count && dim_idx < dim; ++dim_idx, ++partition_idx) {
                    int offset = b + partition_idx * partition_size;
                    for (int ch = 0; ch < count; ++ch) {
                        if (do_not_decode[ch]) continue;
                        int map_index = part_word_cache[(size_t)ch * partition_words + entry_idx] * dim;
                        int idx = decode_map[(size_t)map_index + dim_idx];
                        const int bk = stage < 8 ? stage_book[(size_t)idx * 8 + stage] : -1;  // (cascade bit + book list)
                        if (bk < 0) continue;
                        const Codebook &book = cbs[bk];
                        if (write_vectors(book, p, buffer + (size_t)ch * stride, stride, offset)) {
                            partition_idx = partition_count;
                            stage = max_stages;
                            break;
                        }
                    }
                }
            }
        }
    }
};

struct Mapping {  // Mapping.cs:19-95
    std::vector<uint8_t> coupling_angle, coupling_magnitude, mux, submap_floor, submap_residue;
    void read(BitReader &p, int channels, int n_floors, int n_residues)
    {
        int submaps = 1;
        if (p.read_bit()) submaps += (int)p.read_bits(4);
        int steps = 0;
        if (p.read_bit()) steps = (int)p.read_bits(8) + 1;
        int cbits = ilog(channels - 1);
        for (int j = 0; j < steps; ++j) {
            int mag = (int)p.read_bits(cbits), ang = (int)p.read_bits(cbits);
            if (mag == ang || mag > channels - 1 || ang > channels - 1)
                throw InvalidData("Invalid magnitude or angle in mapping header!");
            coupling_angle.push_back((uint8_t)ang);
            coupling_magnitude.push_back((uint8_t)mag);
        }
        if (p.read_bits(2) != 0) throw InvalidData("Reserved bits not 0 in mapping header.");
        mux.assign(channels, 0);
        if (submaps > 1)
            for (int c = 0; c < channels; ++c) {
                mux[c] = (uint8_t)p.read_bits(4);
                if (mux[c] > submaps) throw InvalidData("Invalid channel mux submap index in mapping header!");
            }
        for (int j = 0; j < submaps; ++j) {
            p.skip(8);
            int fl = (int)p.read_bits(8);
            if (fl >= n_floors) throw InvalidData("Invalid floor number in mapping header!");
            int rs = (int)p.read_bits(8);
            if (rs >= n_residues) throw InvalidData("Invalid residue number in mapping header!");
            submap_floor.push_back((uint8_t)fl);
            submap_residue.push_back((uint8_t)rs);
        }
        // Mapping.cs:89-93: `floors[_submapFloor[mux[c]]]` -- the check above lets mux == submapCount through
        // and the reference's constructor then dies with IndexOutOfRangeException; same outcome here
        for (int c = 0; c < channels; ++c)
            if (mux[c] >= submaps) throw InvalidData("Channel mux submap index out of range in mapping header!");
    }
};

struct Mode {  // Mode.cs:12-28
    bool block_flag = false;
    int mapping = 0;
};

// (the payload lies in the stream's one `payload` buffer: a vector per packet was ~700 heap blocks per stream, allocated by the thread
// that opens the container and -- in the multi-device dispatcher -- freed by the one that decodes it, i.e. through the allocator's
// cross-thread path, 16 threads at a time)
struct OggPacket {
    const uint8_t *data = nullptr;
    size_t size = 0;
    int64_t granule = -1;
    bool eos = false;
    bool resync = false;   // VorbisPacket.IsResync: the page that completes the packet was found after lost sync
};

// Ogg CRC (polynomial 0x04c11db7, no reflection), Ogg/Crc.cs.  Eight tables (slicing by 8): the byte-at-a-time form is a
// chain of dependent table reads, ~6 cycles per byte -- a third of a millisecond for a 190 KB file, as much as all its
// setup headers cost; eight bytes per step bring it below a tenth of that.
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t r = i << 24;
            for (int j = 0; j < 8; ++j) r = (r << 1) ^ ((r & 0x80000000u) ? 0x04c11db7u : 0u);
            t[0][i] = r;
        }
        for (int k = 1; k < 8; ++k)
            for (uint32_t i = 0; i < 256; ++i) t[k][i] = (t[k - 1][i] << 8) ^ t[0][t[k - 1][i] >> 24];
    }
};
const CrcTables &crc_tables()
{
    static const CrcTables tables;  // (initialised once, thread-safe: streams are opened from many threads)
    return tables;
}
void crc_init() { (void)crc_tables(); }
uint32_t crc_update(uint32_t crc, const uint8_t *d, size_t n)
{
    const CrcTables &T = crc_tables();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint32_t x = crc ^ (((uint32_t)d[i] << 24) | ((uint32_t)d[i + 1] << 16) | ((uint32_t)d[i + 2] << 8) | (uint32_t)d[i + 3]);
        crc = T.t[7][x >> 24] ^ T.t[6][(x >> 16) & 0xff] ^ T.t[5][(x >> 8) & 0xff] ^ T.t[4][x & 0xff] ^
              T.t[3][d[i + 4]] ^ T.t[2][d[i + 5]] ^ T.t[1][d[i + 6]] ^ T.t[0][d[i + 7]];
    }
    for (; i < n; ++i) crc = (crc << 8) ^ T.t[0][((crc >> 24) & 0xff) ^ d[i]];
    return crc;
}

}  // namespace

// What the setup header of a stream unpacks into (StreamDecoder.cs:262-321): codebooks with their decode tables, floors,
// residues, mappings, modes.  Files that come out of one encoder at one setting carry byte-identical setup headers, and a host
// that transcodes a library opens thousands of them: the unpacked form of the last few distinct (identification, setup)
// header pairs is kept, and a stream whose headers match one byte for byte SHARES it (read-only; a stream's scratch is its own)
// instead of building ~40 Huffman tables again -- no copy either: a thousand streams of one encoder decode out of one set of
// tables that stays in the cache.  VPZH_NO_SETUP_CACHE=1 gives every stream its own.
struct SetupBlob {
    std::vector<uint8_t> ident, setup;  // the key: both packets, compared in full
    std::vector<Codebook> books;
    std::vector<Floor1> floors;
    std::vector<Floor0> floors0;
    std::vector<uint8_t> floor_types;
    int max_floor0_order = 0;
    std::vector<Residue> residues;
    std::vector<Mapping> mappings;
    std::vector<Mode> modes;
    int mode_field_bits = 0;
};
struct SetupCache {
    static constexpr size_t kEntries = 16;
    std::mutex mu;
    std::vector<std::shared_ptr<const SetupBlob>> entries;
    size_t next = 0;  // round robin once full
    bool enabled = true;
    SetupCache()
    {
        const char *e = getenv("VPZH_NO_SETUP_CACHE");
        enabled = !(e && atoi(e));
    }
    std::shared_ptr<const SetupBlob> find(const uint8_t *ident, size_t n_ident, const uint8_t *setup, size_t n_setup)
    {
        if (!enabled) return nullptr;
        std::lock_guard<std::mutex> lock(mu);
        for (const auto &b : entries)
            if (b->setup.size() == n_setup && b->ident.size() == n_ident && memcmp(b->setup.data(), setup, n_setup) == 0 &&
                memcmp(b->ident.data(), ident, n_ident) == 0)
                return b;
        return nullptr;
    }
    std::shared_ptr<const SetupBlob> insert(std::shared_ptr<const SetupBlob> b)
    {
        if (!enabled) return b;
        std::lock_guard<std::mutex> lock(mu);
        for (const auto &e : entries)  // (threads that opened the same headers at the same time: one copy)
            if (e->setup == b->setup && e->ident == b->ident) return e;
        if (entries.size() < kEntries) {
            entries.push_back(b);
        } else {
            entries[next] = b;
            next = (next + 1) % kEntries;
        }
        return b;
    }
};
static SetupCache &setup_cache()
{
    static SetupCache c;  // (thread-safe initialisation)
    return c;
}

struct vpzh_stream {
    std::string error;
    int channels = 0, sample_rate = 0, size0 = 0, size1 = 0;
    // the unpacked setup header (floors indexed by floor number: unused entries in `floors` for type-0 floors and in `floors0` for
    // type-1 ones), possibly shared with other streams -- never written after parse_headers
    std::shared_ptr<const SetupBlob> su;
    std::vector<uint8_t> payload;        // every packet's bytes, back to back (OggPacket::data points in here)
    std::vector<int> residue_scratch;    // Residue::decode's part_word_cache
    std::vector<OggPacket> audio;
    int64_t last_granule = -1;
    int pages = 0, bad_crc = 0;
    int64_t residue_floats = 0;
    int64_t decode_failures = 0, first_failed_packet = -1;  // of the last vpzh_decode_range* call

    // ---- container: Ogg/PageReaderBase.cs:286-361 (sync, CRC), Ogg/PacketProvider.cs:427-560 (packets)
    // `want`: which logical stream of the container, counted by beginning-of-stream pages (0 = the first one; a
    // chained file starts its second stream after the first one's end-of-stream page, VorbisReader.FindNextStream).
    // Returns false when the container holds no such stream.
    bool demux(const uint8_t *d, size_t size, std::vector<OggPacket> &packets, int want = 0)
    {
        crc_init();
        payload.clear();
        payload.reserve(size + 8);  // (every packet's bytes come out of the container once: never reallocated, pointers stay)
        auto keep = [&](const uint8_t *a, size_t na, const uint8_t *b, size_t nb) -> const uint8_t * {
            const size_t at = payload.size();
            if (at + na + nb > payload.capacity()) throw InvalidData("container payload larger than the container");
            payload.insert(payload.end(), a, a + na);
            payload.insert(payload.end(), b, b + nb);
            return payload.data() + at;
        };
        size_t pos = 0;
        bool have_serial = false, finished = false;
        uint32_t serial = 0;
        int bos_seen = 0;
        // IsResync (Ogg/PageReaderBase.cs:297-342, StreamPageReader.cs:87-96): a page is a resync page when bytes
        // had to be skipped to find it, or when its sequence number does not follow the previous page of the stream
        bool skipped_bytes = false, page_resync = false;
        uint32_t last_seq = 0;
        bool resync_carry = false;  // a resync page whose first piece was dropped: the next packet carries the flag
        std::vector<uint8_t> pending;  // packet continued from earlier pages
        bool pending_valid = false;
        OggPacket *last_granule_candidate = nullptr;
        (void)last_granule_candidate;
        while (pos + 27 <= size) {
            if (memcmp(d + pos, "OggS", 4) != 0) { ++pos; skipped_bytes = true; continue; }
            const uint8_t *h = d + pos;
            int nseg = h[26];
            if (pos + 27 + nseg > size) break;
            size_t body = 0;
            for (int i = 0; i < nseg; ++i) body += h[27 + i];
            if (pos + 27 + nseg + body > size) break;
            uint32_t stored = (uint32_t)h[22] | ((uint32_t)h[23] << 8) | ((uint32_t)h[24] << 16) | ((uint32_t)h[25] << 24);
            uint32_t crc = crc_update(0, h, 22);
            static const uint8_t zero4[4] = {0, 0, 0, 0};
            crc = crc_update(crc, zero4, 4);
            crc = crc_update(crc, h + 26, 1 + nseg + body);
            if (crc != stored) { ++bad_crc; ++pos; skipped_bytes = true; continue; }  // resync byte-wise
            const bool found_after_skip = skipped_bytes;
            skipped_bytes = false;
            uint32_t ser = (uint32_t)h[14] | ((uint32_t)h[15] << 8) | ((uint32_t)h[16] << 16) | ((uint32_t)h[17] << 24);
            uint8_t flags = h[5];
            int64_t granule = 0;
            for (int i = 7; i >= 0; --i) granule = (granule << 8) | h[6 + i];
            const size_t page_len = 27 + nseg + body;
            if (!have_serial) {
                if (!(flags & 2)) { pos += page_len; continue; }  // wait for a beginning-of-stream page
                if (bos_seen++ != want) { pos += page_len; continue; }
                serial = ser;
                have_serial = true;
            }
            if (ser != serial || finished) { pos += page_len; continue; }  // pages of other logical streams
            ++pages;
            const uint32_t seq = (uint32_t)h[18] | ((uint32_t)h[19] << 8) | ((uint32_t)h[20] << 16) | ((uint32_t)h[21] << 24);
            page_resync = found_after_skip || (last_seq != 0 && last_seq + 1 != seq);
            last_seq = seq;
            if (flags & 4) finished = true;  // end-of-stream page: a later stream may reuse the serial number
            const uint8_t *seg = h + 27;
            const uint8_t *data = h + 27 + nseg;
            const bool continuation = flags & 1;
            // packet count as PageHeader.GetPacketCount: terminated packets + 1 if the page ends open
            int packet_count = 0;
            for (int i = 0; i < nseg; ++i) if (seg[i] < 255) ++packet_count;
            const bool page_continued = nseg > 0 && seg[nseg - 1] == 255;
            if (page_continued) ++packet_count;
            if ((!continuation || page_resync) && pending_valid) {  // broken continuation (PacketProvider.cs:473-478
                pending.clear();                                    // "can't merge across resync"): drop what we had
                pending_valid = false;
            }
            if (page_resync) resync_carry = true;
            int packet_idx = 0;
            size_t off = 0, cur = 0;
            bool first_is_continuation = continuation;
            for (int i = 0; i < nseg; ++i) {
                cur += seg[i];
                if (seg[i] < 255) {
                    OggPacket pk;
                    if (packet_idx == 0 && first_is_continuation) {
                        if (pending_valid) {
                            pk.data = keep(pending.data(), pending.size(), data + off, cur);
                            pk.size = pending.size() + cur;
                        } else {
                            // The tail of a packet whose head was lost.  The reference's sequential reader hands this
                            // piece to the decoder as a packet of its own (flagged IsResync); it is known garbage, so
                            // it is dropped here and the flag moves to the next packet of the page.
                            off += cur; cur = 0; ++packet_idx;
                            continue;
                        }
                        pending.clear();
                        pending_valid = false;
                    } else {
                        pk.data = keep(nullptr, 0, data + off, cur);
                        pk.size = cur;
                    }
                    // GranulePosition only on the last packet of the page (PacketProvider.cs:515-530)
                    pk.granule = (packet_idx == packet_count - 1) ? granule : -1;
                    pk.resync = resync_carry;
                    resync_carry = false;
                    packets.push_back(std::move(pk));
                    off += cur; cur = 0; ++packet_idx;
                }
            }
            if (page_continued) {
                if (packet_idx == 0 && first_is_continuation && pending_valid) {
                    pending.insert(pending.end(), data + off, data + off + cur);
                } else {
                    pending.assign(data + off, data + off + cur);
                    pending_valid = true;
                }
            }
            last_granule = granule;
            pos += page_len;
        }
        // IsEndOfStream: last packet completed in the last page (PacketProvider.cs:522-525)
        if (!packets.empty() && packets.back().granule != -1) packets.back().eos = true;
        stream_serial = serial;
        return have_serial;
    }
    uint32_t stream_serial = 0;

    void load_headers(std::vector<OggPacket> &pk)
    {
        if (pk.size() < 3) throw InvalidData("missing header packets");
        // identification, StreamDecoder.cs:213-240
        {
            BitReader p;
            p.init(pk[0].data, pk[0].size);
            static const uint8_t sig[11] = {0x01, 'v', 'o', 'r', 'b', 'i', 's', 0, 0, 0, 0};
            for (uint8_t c : sig) if (p.read_bits(8) != c) throw InvalidData("not a Vorbis identification header");
            channels = (int)p.read_bits(8);
            sample_rate = (int)p.read_bits(32);
            p.read_bits(32); p.read_bits(32); p.read_bits(32);
            int b0 = (int)p.read_bits(4), b1 = (int)p.read_bits(4);
            size0 = 1 << b0;
            size1 = 1 << b1;
            if (channels < 1) throw InvalidData("no channels");
            // (StreamDecoder.cs:226 takes any two exponents and fails later, inside its transforms; Vorbis I allows 64 ... 8192 with
            // the short size not above the long one, the synthesis library refuses anything else at vpz_decoder_create, and a size
            // of 1 would make every residue vector empty here)
            if (b0 < 6 || b1 > 13 || b0 > b1) throw InvalidData("block sizes outside Vorbis I's range");
        }
        {   // comments: only the signature is checked (StreamDecoder.cs:242-260)
            BitReader p;
            p.init(pk[1].data, pk[1].size);
            static const uint8_t sig[7] = {0x03, 'v', 'o', 'r', 'b', 'i', 's'};
            for (uint8_t c : sig) if (p.read_bits(8) != c) throw InvalidData("not a Vorbis comment header");
        }
        // setup, StreamDecoder.cs:262-321 -- or its unpacked form, if these very headers have been seen before
        if (std::shared_ptr<const SetupBlob> hit = setup_cache().find(pk[0].data, pk[0].size, pk[2].data, pk[2].size)) {
            su = std::move(hit);
            return;
        }
        auto blob = std::make_shared<SetupBlob>();
        std::vector<Codebook> &books = blob->books;
        std::vector<Floor1> &floors = blob->floors;
        std::vector<Floor0> &floors0 = blob->floors0;
        std::vector<uint8_t> &floor_types = blob->floor_types;
        int &max_floor0_order = blob->max_floor0_order;
        std::vector<Residue> &residues = blob->residues;
        std::vector<Mapping> &mappings = blob->mappings;
        std::vector<Mode> &modes = blob->modes;
        BitReader p;
        p.init(pk[2].data, pk[2].size);
        static const uint8_t sig[7] = {0x05, 'v', 'o', 'r', 'b', 'i', 's'};
        for (uint8_t c : sig) if (p.read_bits(8) != c) throw InvalidData("not a Vorbis setup header");
        books.resize((size_t)p.read_bits(8) + 1);
        for (auto &b : books) b.read(p);
        int times = (int)p.read_bits(6) + 1;
        p.skip(16 * times);
        int n_floors = (int)p.read_bits(6) + 1;
        floors.resize(n_floors);
        floors0.resize(n_floors);
        floor_types.assign(n_floors, 1);
        for (int i = 0; i < n_floors; ++i) {
            int type = (int)p.read_bits(16);
            if (type == 0) {
                floor_types[i] = 0;
                floors0[i].read(p, books);
                if (floors0[i].order > max_floor0_order) max_floor0_order = floors0[i].order;
            } else if (type == 1) {
                floors[i].read(p, (int)books.size());
            } else {
                throw InvalidData("Invalid floor type!");
            }
        }
        int n_res = (int)p.read_bits(6) + 1;
        residues.resize(n_res);
        for (auto &r : residues) {
            int type = (int)p.read_bits(16);
            if (type > 2) throw InvalidData("Invalid residue type!");
            r.read(p, type, books);
        }
        int n_map = (int)p.read_bits(6) + 1;
        mappings.resize(n_map);
        for (auto &m : mappings) {
            if (p.read_bits(16) != 0) throw InvalidData("Invalid mapping type!");
            m.read(p, channels, n_floors, n_res);
        }
        int n_modes = (int)p.read_bits(6) + 1;
        modes.resize(n_modes);
        for (auto &m : modes) {
            m.block_flag = p.read_bit();
            if (p.read_bits(32) != 0) throw InvalidData("Mode header had invalid window or transform type!");
            m.mapping = (int)p.read_bits(8);
            if (m.mapping >= n_map) throw InvalidData("Mode header had invalid mapping index!");
        }
        if (!p.read_bit()) throw InvalidData("Book packet did not end on correct bit!");
        blob->mode_field_bits = ilog(n_modes - 1);
        if (setup_cache().enabled) {
            blob->ident.assign(pk[0].data, pk[0].data + pk[0].size);
            blob->setup.assign(pk[2].data, pk[2].data + pk[2].size);
            su = setup_cache().insert(std::move(blob));  // (the cache's copy, if another thread was first with these headers)
        } else {
            su = std::move(blob);
        }
    }

    // residue floats one packet contributes to the batch
    int64_t packet_floats(const OggPacket &pk) const
    {
        BitReader p;
        p.init(pk.data, pk.size);
        if (p.read_bits(1) != 0) return 0;
        int mode_idx = (int)p.read_bits(su->mode_field_bits);
        if (mode_idx >= (int)su->modes.size()) return 0;
        return (int64_t)channels * ((su->modes[mode_idx].block_flag ? size1 : size0) / 2);
    }

    // IPacketGranuleCountProvider.GetPacketGranuleCount (StreamDecoder.cs:884-913): what a packet adds to the
    // sample position, from its mode bits alone -- PacketInfo.SampleCount = RightStart - LeftStart (Mode.cs:30-66)
    int packet_sample_count(const OggPacket &pk) const
    {
        BitReader p;
        p.init(pk.data, pk.size);
        if (p.read_bits(1) != 0) return 0;
        const int mode_idx = (int)p.read_bits(su->mode_field_bits);
        if (mode_idx >= (int)su->modes.size()) return 0;
        const bool bf = su->modes[mode_idx].block_flag;
        bool prev = true, next = true;
        if (bf) {
            prev = p.read_bit();
            next = p.read_bit();
        }
        if (p.is_short) return 0;
        const int size = bf ? size1 : size0;
        const int left_start = prev ? 0 : (size - size0) / 4;
        const int right_start = next ? size / 2 : (size * 3 - size0) / 4;
        return right_start - left_start;
    }
    // PacketProvider._pageEndGranules, kept per packet: position after packet i when counting starts at the
    // second audio packet (the first one only primes the overlap; PacketProvider.cs:283-287)
    std::vector<int64_t> cum_samples;

    std::vector<float> scratch_decode;      // per-handle scratch of decode_packet (one thread per handle)
    std::vector<int16_t> scratch_decode16;  // ... of its int16 form (vpzh_decode_range_i16)
    std::vector<float> &scratch_of(float *) { return scratch_decode; }
    std::vector<int16_t> &scratch_of(int16_t *) { return scratch_decode16; }
    std::vector<uint8_t> one_flag;

    // StreamDecoder.DecodeNextPacket :696-762 -> Mode.Decode -> Mapping.DecodePacket :98-163
    // (T: the residue's element type -- float, or int16_t for a stream whose residue is integral, ABI v5)
    template <class T>
    void decode_packet(const OggPacket &pk, int32_t stream_id, int64_t residue_off, vpz_packet *out, T *residue,
                       int16_t *posts, uint8_t *post_counts, float *f0_amp = nullptr, float *f0_coeff = nullptr,
                       int f0_stride = 0)
    {
        memset(out, 0, sizeof *out);
        out->stream = stream_id;
        out->granule = -1;
        out->residue_offset = residue_off;
        if (pk.eos) out->flags |= VPZ_PKT_EOS;
        if (pk.resync) out->flags |= VPZ_PKT_RESYNC;
        for (int c = 0; c < channels; ++c) post_counts[c] = 0;
        memset(posts, 0, sizeof(int16_t) * 64 * (size_t)channels);
        BitReader p;
        p.init(pk.data, pk.size);
        if (p.read_bits(1) != 0) { out->flags |= VPZ_PKT_NOT_DECODED; return; }
        int mode_idx = (int)p.read_bits(su->mode_field_bits);
        if ((unsigned)mode_idx >= su->modes.size()) throw InvalidData("Unused mode index.");
        const Mode &mode = su->modes[mode_idx];
        if (p.is_short) { out->flags |= VPZ_PKT_NOT_DECODED; return; }  // Mode.cs:32-36
        const int block_size = mode.block_flag ? size1 : size0;
        const int half = block_size / 2;
        if (mode.block_flag) {
            out->flags |= VPZ_PKT_BLOCK_FLAG;
            if (p.read_bit()) out->flags |= VPZ_PKT_PREV_FLAG;
            if (p.read_bit()) out->flags |= VPZ_PKT_NEXT_FLAG;
        }
        out->mapping = (uint8_t)mode.mapping;
        out->granule = pk.granule;
        const Mapping &map = su->mappings[mode.mapping];

        // floors, Mapping.cs:109-118
        std::vector<uint8_t> no_execute(channels);
        for (int ch = 0; ch < channels; ++ch) {
            const int fl = map.submap_floor[map.mux[ch]];
            if (su->floor_types[fl] == 0) {  // Floor0.Unpack; ExecuteChannel = Amp != 0
                float tmp[256];
                const float amp = su->floors0[fl].unpack(p, su->books, tmp);
                if (f0_amp) {
                    f0_amp[ch] = amp;
                    for (int i = 0; i < f0_stride; ++i) f0_coeff[(size_t)ch * f0_stride + i] = i < su->floors0[fl].order ? tmp[i] : 0.f;
                }
                post_counts[ch] = amp != 0.f ? 1 : 0;
                no_execute[ch] = amp == 0.f;
                continue;
            }
            int raw[64];
            memset(raw, 0, sizeof raw);
            int pc = su->floors[fl].unpack(p, su->books, raw);
            if (pc > 64) pc = 64;
            post_counts[ch] = (uint8_t)pc;
            for (int i = 0; i < 64; ++i) {
                int v = raw[i];
                posts[ch * 64 + i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
            no_execute[ch] = pc == 0;
        }
        // coupling fix-up, Mapping.cs:121-130
        for (size_t i = 0; i < map.coupling_angle.size(); ++i) {
            int mag = map.coupling_magnitude[i], ang = map.coupling_angle[i];
            if (!(no_execute[mag] && no_execute[ang])) { no_execute[mag] = 0; no_execute[ang] = 0; }
        }
        // residues, Mapping.cs:132-163.  decodeBuffer is allocated once per packet and reused by every
        // submap without clearing (reference behaviour, matters only for multi-submap residue 0/1).
        T *dst = residue;  // planar [ch][half] unless the interleaved shortcut below is taken
        memset(dst, 0, sizeof(T) * (size_t)channels * half);
        const int submaps = (int)map.submap_residue.size();
        if (submaps == 1 && channels > 1 && su->residues[map.submap_residue[0]].type == 2) {
            // The common stereo / multichannel case, one Residue2 submap over every channel: decode the
            // interleaved vector straight into the output and let the GPU de-interleave (Residue2.cs:42-51) --
            // same values as the general path below, without its scratch buffers.
            bool all_mux0 = true, any = false;
            for (int j = 0; j < channels; ++j) {
                all_mux0 &= map.mux[j] == 0;
                any |= !no_execute[j];
            }
            if (all_mux0) {
                if (any) {
                    one_flag.assign(1, 0);
                    su->residues[map.submap_residue[0]].decode(p, one_flag, block_size * channels, dst, half * channels, su->books, residue_scratch);
                }
                // (a packet whose channels are all silent is zeros in either layout: it keeps the stream's layout, so that
                // a batch has ONE input layout and the back end's fast paths -- whose loads are unconditional -- take it)
                out->flags |= VPZ_PKT_INTERLEAVED;
                return;
            }
        }
        std::vector<T> &decode_buffer = scratch_of(residue);
        decode_buffer.assign((size_t)channels * block_size, (T)0);
        for (int i = 0; i < submaps; ++i) {
            std::vector<uint8_t> dnd;
            std::vector<int> members;
            for (int j = 0; j < channels; ++j)
                if (map.mux[j] == i) { dnd.push_back(no_execute[j]); members.push_back(j); }
            const Residue &res = su->residues[map.submap_residue[i]];
            const int count = (int)dnd.size();
            if (res.type == 2) {  // Residue2.cs:12-52
                bool any = false;
                for (uint8_t f : dnd) if (!f) any = true;
                if (!any) {
                    for (int k = 0; k < count; ++k) memset(&decode_buffer[(size_t)k * block_size], 0, sizeof(T) * half);
                } else {
                    std::vector<T> tmp((size_t)half * count, (T)0);
                    std::vector<uint8_t> one(1, 0);
                    res.decode(p, one, block_size * count, tmp.data(), half * count, su->books, residue_scratch);
                    if (submaps == 1 && count == channels && channels > 1) {
                        // hand the Residue2 vector over as it is: the GPU de-interleaves (Residue2.cs:42-51)
                        memcpy(dst, tmp.data(), sizeof(T) * (size_t)half * channels);
                        out->flags |= VPZ_PKT_INTERLEAVED;
                        return;
                    }
                    if (count == 1) {
                        memcpy(&decode_buffer[0], tmp.data(), sizeof(T) * half);
                    } else {
                        for (int k = 0; k < count; ++k)
                            for (int b = 0; b < half; ++b) decode_buffer[(size_t)k * block_size + b] = tmp[(size_t)b * count + k];
                    }
                }
            } else {
                res.decode(p, dnd, block_size, decode_buffer.data(), block_size, su->books, residue_scratch);
            }
            for (int k = 0; k < count; ++k)
                memcpy(dst + (size_t)members[k] * half, &decode_buffer[(size_t)k * block_size], sizeof(T) * half);
        }
    }
};

extern "C" {

int vpzh_open_memory(const uint8_t *data, uint64_t size, vpzh_stream **out)
{
    return vpzh_open_memory_stream(data, size, 0, out);
}

int vpzh_open_memory_stream(const uint8_t *data, uint64_t size, int32_t stream_index, vpzh_stream **out)
{
    if (!data || !out || stream_index < 0) return VPZH_E_ARG;
    *out = nullptr;
    std::unique_ptr<vpzh_stream> s(new vpzh_stream());
    int rc = VPZH_OK;
    try {
        std::vector<OggPacket> packets;
        if (!s->demux(data, (size_t)size, packets, stream_index)) {
            if (stream_index > 0) return VPZH_E_NO_STREAM;  // FindNextStream() == false
            throw InvalidData("no logical stream in the container");
        }
        s->load_headers(packets);
        s->audio.assign(std::make_move_iterator(packets.begin() + 3), std::make_move_iterator(packets.end()));
        s->residue_floats = 0;
        for (const OggPacket &pk : s->audio) s->residue_floats += s->packet_floats(pk);
        s->cum_samples.assign(s->audio.size(), 0);
        for (size_t i = 1; i < s->audio.size(); ++i)
            s->cum_samples[i] = s->cum_samples[i - 1] + s->packet_sample_count(s->audio[i]);
    } catch (const Unsupported &e) {
        s->error = e.what();
        rc = VPZH_E_UNSUPPORTED;
    } catch (const std::exception &e) {
        s->error = e.what();
        rc = VPZH_E_INVALID_DATA;
    }
    *out = s.release();
    return rc;
}

void vpzh_close(vpzh_stream *s) { delete s; }

const char *vpzh_last_error(vpzh_stream *s) { return s ? s->error.c_str() : ""; }

int vpzh_get_info(vpzh_stream *s, vpzh_info *info)
{
    if (!s || !info) return VPZH_E_ARG;
    memset(info, 0, sizeof *info);
    info->channels = s->channels;
    info->sample_rate = s->sample_rate;
    info->block_size0 = s->size0;
    info->block_size1 = s->size1;
    info->floor_count = (int32_t)s->su->floors.size();
    info->residue_count = (int32_t)s->su->residues.size();
    info->mapping_count = (int32_t)s->su->mappings.size();
    info->mode_count = (int32_t)s->su->modes.size();
    info->codebook_count = (int32_t)s->su->books.size();
    info->audio_packets = (int64_t)s->audio.size();
    info->last_granule = s->last_granule;
    info->residue_floats = s->residue_floats;
    info->pages = s->pages;
    info->bad_crc_pages = s->bad_crc;
    info->stream_serial = (int32_t)s->stream_serial;
    return VPZH_OK;
}

int vpzh_get_floor_type(vpzh_stream *s, int index)
{
    if (!s || index < 0 || index >= (int)s->su->floor_types.size()) return VPZH_E_ARG;
    return s->su->floor_types[index];
}

int vpzh_get_floor0(vpzh_stream *s, int index, vpz_floor0_config *out)
{
    if (!s || !out || index < 0 || index >= (int)s->su->floors0.size() || s->su->floor_types[index] != 0) return VPZH_E_ARG;
    const Floor0 &f = s->su->floors0[index];
    out->order = f.order;
    out->rate = f.rate;
    out->bark_map_size = f.bark_map_size;
    out->amp_bits = f.amp_bits;
    out->amp_ofs = f.amp_ofs;
    return VPZH_OK;
}

int vpzh_max_floor0_order(vpzh_stream *s) { return s ? s->su->max_floor0_order : 0; }

int vpzh_get_floor1(vpzh_stream *s, int index, vpz_floor1_config *out)
{
    if (!s || !out || index < 0 || index >= (int)s->su->floors.size() || s->su->floor_types[index] != 1) return VPZH_E_ARG;
    const Floor1 &f = s->su->floors[index];
    memset(out, 0, sizeof *out);
    if (f.x_list.size() > VPZ_POSTS_STRIDE) return VPZH_E_INVALID_DATA;  // `Posts = new int[64]`, Floor1.cs:17
    out->x_count = (int32_t)f.x_list.size();
    out->multiplier = f.multiplier;
    for (size_t i = 0; i < f.x_list.size(); ++i) out->x_list[i] = f.x_list[i];
    return VPZH_OK;
}

static bool residue_tiles_its_partitions(const SetupBlob &su, const Residue &r);

int vpzh_get_mapping(vpzh_stream *s, int index, vpz_mapping_config *out)
{
    if (!s || !out || index < 0 || index >= (int)s->su->mappings.size()) return VPZH_E_ARG;
    const Mapping &m = s->su->mappings[index];
    memset(out, 0, sizeof *out);
    out->coupling_steps = (int32_t)m.coupling_angle.size();
    for (size_t i = 0; i < m.coupling_angle.size(); ++i) {
        out->coupling_magnitude[i] = m.coupling_magnitude[i];
        out->coupling_angle[i] = m.coupling_angle[i];
    }
    for (int c = 0; c < s->channels && c <= VPZ_MAX_CHANNELS; ++c) out->channel_floor[c] = m.submap_floor[m.mux[c]];
    // The residue's support (ABI v4): a residue decodes into [min(_begin, n), min(_end, n)) of its vector and nowhere else
    // (Residue0.cs:122-125), n = blocksize/2 for types 0 / 1 and blocksize/2 * (channels of the submap) for type 2, whose vector
    // interleaves them (Residue2.cs:31-34): bin = index / channels.  Smallest begin and largest end over the submaps -- the
    // reference reuses one decode buffer for every submap without clearing it (Mapping.cs:132-163), so a later submap's
    // channels can carry an earlier one's values: the union covers that too.
    const int sizes[2] = {s->size0, s->size1};
    for (int b = 0; b < 2; ++b) {
        const int half = sizes[b] / 2;
        int lo = half, hi = 0;
        for (size_t i = 0; i < m.submap_residue.size(); ++i) {
            int count = 0;
            for (int c = 0; c < s->channels; ++c) count += m.mux[c] == (int)i;
            if (count == 0 || m.submap_residue[i] >= s->su->residues.size()) continue;
            const Residue &r = s->su->residues[m.submap_residue[i]];
            if (!residue_tiles_its_partitions(*s->su, r)) { lo = 0; hi = half; break; }  // (a vector may overhang `end`: the whole block)
            const int64_t n = r.type == 2 ? (int64_t)half * count : half;
            const int64_t rb = std::min<int64_t>(r.begin, n), re = std::min<int64_t>(r.end, n);
            if (re <= rb) continue;
            const int per = r.type == 2 ? count : 1;
            lo = std::min(lo, (int)(rb / per));
            hi = std::max(hi, (int)((re + per - 1) / per));
        }
        if (hi <= lo) { lo = 0; hi = 1; }  // (no residue reaches this block size: one bin stands for "nothing", 0 would say "not stated")
        out->residue_begin[b] = lo;
        out->residue_end[b] = std::min(hi, half);
    }
    return VPZH_OK;
}

int vpzh_get_residue_type(vpzh_stream *s, int index)
{
    if (!s || index < 0 || index >= (int)s->su->residues.size()) return VPZH_E_ARG;
    return s->su->residues[index].type;
}

// Is every residue value of this stream an integer that fits 16 bits?  A residue is a sum, per bin, of at most one codebook value per
// cascade stage (Residue0.cs:144-205); libvorbis' residue books are integer lattices, so for its streams the answer is yes and
// the vector can travel as int16 -- exactly: sums of integers below 2^24 are the same in float32 -- at half the bytes.  Decided
// from the setup header alone, conservatively: every value book any residue names must hold integers only, and the worst case --
// the largest magnitude of any of a residue's books, times its stages -- must stay below 2^15.
// A residue whose value books do not tile its partitions -- a book of more dimensions than a partition has bins, or of a
// dimension that does not divide the partition size: the reference decodes such setups (Residue0.cs:171-203 steps by the book's
// dimensions whatever the partition size) -- lets one vector cover several of the following partitions and overhang the
// residue's `end`: neither the "two vectors per bin and stage" bound of residue_integral nor "nothing beyond [begin, end)" of
// vpzh_get_mapping holds for it.  libvorbis never writes one.
static bool residue_tiles_its_partitions(const SetupBlob &su, const Residue &r)
{
    for (size_t cl = 0; cl < r.books.size(); ++cl)
        for (size_t st = 0; st < r.books[cl].size(); ++st) {
            if (!(r.cascade[cl] & (1u << st))) continue;
            const size_t b = r.books[cl][st];
            if (b >= su.books.size()) return false;
            const int dim = su.books[b].dimensions;
            if (dim <= 0 || dim > r.partition_size || r.partition_size % dim != 0) return false;
        }
    return true;
}

static bool residue_integral(const SetupBlob &su)
{
    for (const Residue &r : su.residues) {
        if (!residue_tiles_its_partitions(su, r)) return false;
        double worst = 0.0;
        for (size_t cl = 0; cl < r.books.size(); ++cl)
            for (size_t st = 0; st < r.books[cl].size(); ++st) {
                if (!(r.cascade[cl] & (1u << st))) continue;  // (no book at this stage of the class)
                const size_t b = r.books[cl][st];
                if (b >= su.books.size()) return false;
                const Codebook &cb = su.books[b];
                if (cb.lookup_i16.empty()) return false;  // (a fraction, a -0.0 or a value beyond 16 bits in the table)
                worst = std::max(worst, cb.entry_l1);
            }
        // per stage a bin takes one vector's value -- two where a partition's last vector overhangs into the next partition
        // (Residue1.cs:12-34), the sum of an entry's dimensions for residue type 0 (Residue0.cs:208-231): entry_l1 covers all three
        if (worst * 2.0 * std::max(1, r.max_stages) >= 32768.0) return false;
    }
    return true;
}

int vpzh_residue_is_integral(vpzh_stream *s) { return s && s->su && residue_integral(*s->su) ? 1 : 0; }

// (residue16 != nullptr: the int16 form -- the same decode with the codebooks' integer tables, summed as integers straight into the
// caller's vector; vpzh_residue_is_integral() == 1 is what makes those sums the float ones)
static int decode_range_impl(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                             vpz_packet *packets, float *residue, int16_t *residue16, int16_t *posts, uint8_t *post_counts,
                             int64_t *residue_floats_used, float *f0_amp, float *f0_coeff, int32_t f0_stride)
{
    if (!s || !packets || (!residue && !residue16) || !posts || !post_counts || first < 0 || count < 0 ||
        first + count > (int64_t)s->audio.size())
        return VPZH_E_ARG;
    if (f0_amp && (!f0_coeff || f0_stride < s->su->max_floor0_order)) return VPZH_E_ARG;
    s->decode_failures = 0;
    s->first_failed_packet = -1;
    int64_t off = 0;
    const size_t C = (size_t)s->channels;
    for (int64_t k = 0; k < count; ++k) {
        const OggPacket &pk = s->audio[(size_t)(first + k)];
        const int64_t n = s->packet_floats(pk);
        try {
            float *amp_k = f0_amp ? f0_amp + (size_t)k * C : nullptr;
            float *coeff_k = f0_coeff ? f0_coeff + (size_t)k * C * (size_t)f0_stride : nullptr;
            if (residue16)
                s->decode_packet(pk, stream_id, residue_base + off, &packets[k], residue16 + off, posts + (size_t)k * 64 * C,
                                 post_counts + (size_t)k * C, amp_k, coeff_k, f0_stride);
            else
                s->decode_packet(pk, stream_id, residue_base + off, &packets[k], residue + off, posts + (size_t)k * 64 * C,
                                 post_counts + (size_t)k * C, amp_k, coeff_k, f0_stride);
        } catch (const std::exception &e) {
            // An exception out of DecodeNextPacket (StreamDecoder.cs:696-762: "Unused mode index.", a residue vector
            // overrun, ...) costs the reference exactly that packet: it is consumed, no decoder state has changed
            // yet -- not even `_eosFound |= isEndOfStream` (:647) -- and the next Read goes on with the following
            // packet.  Same here: the packet is handed over as "not decoded" without its EOS flag, the batch goes on,
            // and vpzh_decode_failures reports where it happened.
            if (s->decode_failures++ == 0) {
                s->first_failed_packet = k;
                s->error = e.what();
            }
            const uint8_t resync = packets[k].flags & VPZ_PKT_RESYNC;
            memset(&packets[k], 0, sizeof packets[k]);
            packets[k].stream = stream_id;
            packets[k].granule = -1;
            packets[k].residue_offset = residue_base + off;
            packets[k].flags = (uint8_t)(VPZ_PKT_NOT_DECODED | resync);
            for (size_t c = 0; c < C; ++c) post_counts[(size_t)k * C + c] = 0;
        }
        off += n;
    }
    if (residue_floats_used) *residue_floats_used = off;
    return VPZH_OK;
}

int vpzh_decode_range_ex(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                         vpz_packet *packets, float *residue, int16_t *posts, uint8_t *post_counts,
                         int64_t *residue_floats_used, float *f0_amp, float *f0_coeff, int32_t f0_stride)
{
    if (!residue) return VPZH_E_ARG;
    return decode_range_impl(s, first, count, stream_id, residue_base, packets, residue, nullptr, posts, post_counts, residue_floats_used,
                             f0_amp, f0_coeff, f0_stride);
}

int vpzh_decode_range_i16(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                          vpz_packet *packets, int16_t *residue, int16_t *posts, uint8_t *post_counts,
                          int64_t *residue_values_used, float *f0_amp, float *f0_coeff, int32_t f0_stride)
{
    if (!residue || !s || !vpzh_residue_is_integral(s)) return VPZH_E_ARG;
    return decode_range_impl(s, first, count, stream_id, residue_base, packets, nullptr, residue, posts, post_counts, residue_values_used,
                             f0_amp, f0_coeff, f0_stride);
}

// Threads a call uses when the caller does not say: the cores this process may run on (its affinity mask, not the machine's
// count), divided among the processes torchrun started on this node -- the rule of the synthesis library's host pool.
int vpzh_default_threads(void)
{
    cpu_set_t set;
    int n = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
    // ... of which a container may be allowed less CPU TIME than its affinity mask shows cores (cgroup quota: a 256-core
    // host that gives the job 16 CPUs' worth -- 256 runnable threads would only thrash inside that share)
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 0;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
    if (const char *lw = getenv("LOCAL_WORLD_SIZE")) n /= std::max(1, atoi(lw));
    return std::max(1, n);
}

int vpzh_decode_many(int32_t n, int32_t channels, const uint8_t *const *data, const uint64_t *size, int32_t threads, int32_t stream_id0,
                     const int64_t *packet_base, const int64_t *packet_room, const int64_t *residue_base,
                     const int64_t *residue_room, int64_t residue_origin, vpz_packet *packets, float *residue, int16_t *posts,
                     uint8_t *post_counts, int64_t *failed_packets)
{
    return vpzh_decode_many_progress(n, channels, data, size, threads, stream_id0, packet_base, packet_room, residue_base, residue_room,
                                     residue_origin, packets, residue, posts, post_counts, failed_packets, nullptr);
}

int vpzh_decode_many_progress(int32_t n, int32_t channels, const uint8_t *const *data, const uint64_t *size, int32_t threads, int32_t stream_id0,
                              const int64_t *packet_base, const int64_t *packet_room, const int64_t *residue_base,
                              const int64_t *residue_room, int64_t residue_origin, vpz_packet *packets, float *residue,
                              int16_t *posts, uint8_t *post_counts, int64_t *failed_packets, int32_t *done)
{
    if (n < 0 || channels < 1 || channels > VPZ_MAX_CHANNELS ||
        (n > 0 && (!data || !size || !packet_base || !packet_room || !residue_base || !residue_room || !packets || !residue ||
                   !posts || !post_counts)))
        return VPZH_E_ARG;
    if (failed_packets) *failed_packets = 0;
    if (n == 0) return VPZH_OK;
    int workers = threads > 0 ? threads : vpzh_default_threads();
    workers = std::max(1, std::min(workers, (int)n));
    std::atomic<int32_t> next{0};
    std::atomic<int64_t> failed{0};
    std::atomic<int> status{VPZH_OK};
    // one container at a time per thread (the reference's model: a decoder per stream, a stream per thread); every stream
    // writes its own slices of the batch arrays
    auto work = [&]() {
        for (;;) {
            const int32_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= n) return;
            vpzh_stream *s = nullptr;
            int rc = VPZH_E_ARG;
            try {  // (a worker's exception -- bad_alloc -- must not end the process: it costs the stream)
                rc = vpzh_open_memory(data[k], size[k], &s);
                // (the slices were sized by the caller from a probe of the file: a container that holds more than its slice has
                // room for is refused, nothing of it is written)
                if (rc == VPZH_OK && ((int64_t)s->audio.size() > packet_room[k] || s->residue_floats > residue_room[k])) rc = VPZH_E_ARG;
                // ... and so is one with another channel count than the batch was laid out for: its post records would land
                // at packet_base * 64 * ITS channel count, beyond the records the caller sized (untrusted file contents)
                if (rc == VPZH_OK && s->channels != channels) rc = VPZH_E_ARG;
                if (rc == VPZH_OK) {
                    const size_t C = (size_t)channels;
                    const int64_t pb = packet_base[k];
                    rc = vpzh_decode_range_ex(s, 0, (int64_t)s->audio.size(), stream_id0 + k, residue_base[k] - residue_origin,
                                              packets + pb, residue + residue_base[k], posts + (size_t)pb * 64 * C,
                                              post_counts + (size_t)pb * C, nullptr, nullptr, nullptr, 0);
                    failed.fetch_add(s->decode_failures, std::memory_order_relaxed);
                }
            } catch (...) {
                rc = VPZH_E_INVALID_DATA;
            }
            if (rc != VPZH_OK) status.store(rc, std::memory_order_relaxed);
            if (s) vpzh_close(s);
            // (release: whoever sees the flag sees the stream's slices)
            if (done) __atomic_store_n(&done[k], rc == VPZH_OK ? 1 : -1, __ATOMIC_RELEASE);
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < workers; ++t) pool.emplace_back(work);
    } catch (...) {  // (thread creation failed: the calling thread does what is left)
    }
    work();
    for (std::thread &t : pool) t.join();
    if (failed_packets) *failed_packets = failed.load();
    return status.load();
}

int64_t vpzh_decode_failures(vpzh_stream *s, int64_t *first_failed_packet)
{
    if (!s) return 0;
    if (first_failed_packet) *first_failed_packet = s->first_failed_packet;
    return s->decode_failures;
}

int64_t vpzh_total_samples(vpzh_stream *s)
{
    // PacketProvider.GetGranuleCount (:35-49): the counted length, capped by the last page's granule position
    if (!s || s->cum_samples.empty()) return 0;
    int64_t total = s->cum_samples.back();
    if (s->last_granule >= 0 && total > s->last_granule) total = s->last_granule;
    return total;
}

int vpzh_seek(vpzh_stream *s, int64_t sample_position, int64_t *first_packet, int64_t *roll_forward)
{
    // PacketProvider.SeekTo(granulePos, preRoll = 1) (:56-84, GetTargetPageInfo :86-160): the packet whose
    // span [start, end) holds the position, then one packet back for the pre-roll.
    if (!s || !first_packet || !roll_forward) return VPZH_E_ARG;
    const std::vector<int64_t> &cum = s->cum_samples;
    if (sample_position < 0 || cum.size() < 2 || sample_position > cum.back()) {
        s->error = "The requested seek position extends beyond the stream.";  // SeekOutOfRangeException
        return VPZH_E_ARG;
    }
    size_t k = (size_t)(std::upper_bound(cum.begin() + 1, cum.end(), sample_position) - cum.begin());
    if (k >= cum.size()) k = cum.size() - 1;  // position == end of the stream: the last packet, rolled to its end
    *first_packet = (int64_t)k - 1;
    *roll_forward = sample_position - cum[k - 1];
    return VPZH_OK;
}

int vpzh_decode_range(vpzh_stream *s, int64_t first, int64_t count, int32_t stream_id, int64_t residue_base,
                      vpz_packet *packets, float *residue, int16_t *posts, uint8_t *post_counts,
                      int64_t *residue_floats_used)
{
    return vpzh_decode_range_ex(s, first, count, stream_id, residue_base, packets, residue, posts, post_counts,
                                residue_floats_used, nullptr, nullptr, 0);
}

int vpzh_decode_all(vpzh_stream *s, int32_t stream_id, int64_t residue_base, vpz_packet *packets, float *residue,
                    int16_t *posts, uint8_t *post_counts)
{
    if (!s) return VPZH_E_ARG;
    return vpzh_decode_range(s, 0, (int64_t)s->audio.size(), stream_id, residue_base, packets, residue, posts,
                             post_counts, nullptr);
}

}  // extern "C"
