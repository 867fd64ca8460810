// Floor0.Apply (Floor0.cs:164-225) -- the LSP floor the reference itself calls "virtually unused".
// One workgroup per (packet, channel) record that has a type-0 floor: the spectrum in the planar temp is
// multiplied in place; floor1_unwrap_kernel gives the record the one-post curve "index 255" (table[255] == 1.0),
// so the synthesis kernels downstream need no floor-0 variant.  Compiled with -ffp-contract=off.  The LSP
// product is ill conditioned next to a root (w - c_j cancels), where a 1-ulp difference between two cosf
// implementations shows up as 1e-4 of the output; cos / sqrt / exp / divide are therefore evaluated in
// double and rounded once, which reproduces a correctly rounded host libm (the reference's MathF).
#include "synth_desc.hpp"
#include "vpz_internal.hpp"

namespace vpz {

struct Floor0Dev {
    int32_t order, bark_map_size, amp_ofs, reserved;
    int64_t bark_off[2];  // offset of the short / long bark map (n+1 ints each) in the maps buffer
};

struct Floor0Rec {        // one record to process
    int64_t spec_off;     // float offset of this channel's spectrum in the temp
    int32_t rec;          // channel record (index into amp / coeff rows)
    int32_t floor;        // index into Floor0Dev table
    int32_t half;         // blocksize / 2
    int32_t is_long;
};

__global__ __launch_bounds__(256) void floor0_apply_kernel(const Floor0Rec *__restrict__ recs,
                                                          const Floor0Dev *__restrict__ floors,
                                                          const int32_t *__restrict__ bark_maps,
                                                          const float *__restrict__ amp,
                                                          const float *__restrict__ coeff, int coeff_stride,
                                                          float *__restrict__ spec)
{
    __shared__ float s_c[256];
    const Floor0Rec r = recs[blockIdx.x];
    const Floor0Dev f = floors[r.floor];
    float *x = spec + r.spec_off;
    const float a = amp[r.rec];
    if (a <= 0.0f) {  // :169-173
        for (int i = threadIdx.x; i < r.half; i += 256) x[i] = 0.0f;
        return;
    }
    for (int j = threadIdx.x; j < f.order; j += 256)
        s_c[j] = 2.0f * (float)cos((double)coeff[(size_t)r.rec * coeff_stride + j]);
    __syncthreads();
    const int32_t *bark = bark_maps + f.bark_off[r.is_long];
    const float wdel = (float)(3.14159265358979323846 / (double)f.bark_map_size);
    const float amp_ofs = (float)f.amp_ofs;
    for (int i = threadIdx.x; i < r.half; i += 256) {
        const int k = bark[i];
        float p = .5f, q = .5f;
        const float w = 2.0f * (float)cos((double)(wdel * (float)k));
        int j;
        for (j = 1; j < f.order; j += 2) {
            q *= w - s_c[j - 1];
            p *= w - s_c[j];
        }
        if (j == f.order) {  // odd order
            q *= w - s_c[j - 1];
            p *= p * (4.0f - w * w);
            q *= q;
        } else {
            p *= p * (2.0f - w);
            q *= q * (2.0f + w);
        }
        q = (float)((double)a / (double)(float)sqrt((double)(p + q))) - amp_ofs;
        q = (float)exp((double)(q * 0.11512925f));
        x[i] *= q;
    }
}

hipError_t launch_floor0_apply(const void *recs, int n_recs, const void *floors, const int32_t *bark_maps,
                               const float *amp, const float *coeff, int coeff_stride, float *spec,
                               hipStream_t stream)
{
    if (n_recs <= 0) return hipSuccess;
    hipLaunchKernelGGL(floor0_apply_kernel, dim3(n_recs), dim3(256), 0, stream,
                       static_cast<const Floor0Rec *>(recs), static_cast<const Floor0Dev *>(floors), bark_maps, amp,
                       coeff, coeff_stride, spec);
    return hipGetLastError();
}

size_t floor0_dev_size() { return sizeof(Floor0Dev); }
size_t floor0_rec_size() { return sizeof(Floor0Rec); }
void fill_floor0_dev(void *dst, int order, int bark_map_size, int amp_ofs, int64_t off_short, int64_t off_long)
{
    Floor0Dev *d = static_cast<Floor0Dev *>(dst);
    d->order = order;
    d->bark_map_size = bark_map_size;
    d->amp_ofs = amp_ofs;
    d->reserved = 0;
    d->bark_off[0] = off_short;
    d->bark_off[1] = off_long;
}
void fill_floor0_rec(void *dst, int64_t spec_off, int rec, int floor, int half, int is_long)
{
    Floor0Rec *r = static_cast<Floor0Rec *>(dst);
    r->spec_off = spec_off;
    r->rec = rec;
    r->floor = floor;
    r->half = half;
    r->is_long = is_long;
}

}  // namespace vpz
