// Profiling driver of the CPU front end (vorbis_front.cpp): opens a file `reps` times and entropy-decodes it, prints the
// split open (container + headers) / decode and the rate per host thread.  Build with -pg for gprof:
//   g++ -O2 -g -pg -std=c++17 -I include tools/front_prof.cpp vorbispizza_amd/host/vorbis_front.cpp -o /tmp/front_prof
//   /tmp/front_prof tests/golden/3test.ogg 30 && gprof /tmp/front_prof gmon.out | head -40
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vorbispizza_front.h"

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "tests/golden/3test.ogg";
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    double t_open = 0, t_dec = 0;
    int64_t samples = 0;
    std::vector<vpz_packet> pk;
    std::vector<float> res;
    std::vector<int16_t> posts;
    std::vector<uint8_t> counts;
    for (int r = 0; r < reps; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        vpzh_stream *s = nullptr;
        if (vpzh_open_memory(data.data(), data.size(), &s) != 0) return 2;
        vpzh_info info;
        vpzh_get_info(s, &info);
        pk.resize(info.audio_packets);
        res.resize(info.residue_floats);
        posts.resize((size_t)info.audio_packets * info.channels * 64);
        counts.resize((size_t)info.audio_packets * info.channels);
        auto t1 = std::chrono::steady_clock::now();
        if (vpzh_decode_all(s, 0, 0, pk.data(), res.data(), posts.data(), counts.data()) != 0) return 3;
        auto t2 = std::chrono::steady_clock::now();
        t_open += std::chrono::duration<double>(t1 - t0).count();
        t_dec += std::chrono::duration<double>(t2 - t1).count();
        samples = info.last_granule * info.channels;
        vpzh_close(s);
    }
    printf("%s: open %.3f ms, entropy decode %.3f ms per file; %.1f Msamples/s per thread (decode only %.1f)\n", path,
           t_open / reps * 1e3, t_dec / reps * 1e3, samples / ((t_open + t_dec) / reps) / 1e6, samples / (t_dec / reps) / 1e6);
    return 0;
}
