// Device front door of `call variants`: observation BCFs (format v15) -> the SoA columns of vlr_batch, with
// the inflated record stream and the columns born in HBM.
//
// What it replaces in the reference (file:line under /root/reference/src):
//   calling/variants/calling.rs:297-339              bcf::Reader per sample over the observation files (htslib: BGZF inflate, record split)
//   calling/variants/preprocessing/mod.rs:818-919    read_observations: INFO integer vectors -> u16 words -> bincode -> ReadObservation
//   utils/mod.rs:449-474                             MiniLogProb {F16, F32}
// and what vlr_ingest.cpp does for the same rows on the host (decode_into, parse_bcf_record): the two paths are compared column by
// column and byte by byte in tests/test_gpu_ingest_device.py.
//
// Stages per chunk of a sample file (all on one stream of the reader):
//   1. compressed BGZF members  --H2D-->  vlr_inflate_kernel (vlr_inflate.hip)  -->  inflated stream in HBM
//   2. record boundaries: records are length-prefixed, so the starts form a serial chain.  rec_anchor_kernel guesses, for every 64 KiB
//      segment of the stream, the first record start at or behind the segment boundary (a header plausibility test on every byte
//      offset, 64 offsets per step); rec_walk_kernel walks each segment from its anchor with one lane per segment and checks that it
//      lands exactly on the next segment's anchor.  The first anchor is known (end of the BCF header / of the previous chunk), so by
//      induction every start is exact if all checks pass; if one fails, the serial walk (one lane, the whole chunk) replaces it.
//   3. rec_scan_kernel: one lane per record walks the typed INFO entries and leaves a RecDesc (payload offset, count and integer
//      type of every observation vector; n_obs; the size of the record's cold part).
//   4. host: observation offsets of the merged table (locus-major over the sample files) from the n_obs of all files.
//   5. rec_decode_kernel: one wave per record; lane k walks the tag chain of vector k (MiniLogProb elements are 3 or 4 words long),
//      the enum and bit vectors are decoded by all lanes; columns, flags and third-allele evidence go straight to the merged layout.
//   6. rec_cold_kernel: fixed fields, ID, alleles, FILTER and the non-vector INFO entries of every record are copied into a compact
//      "cold record" the host parses with the same code as a whole record (strings, EVENT / MATEID, IMPRECISE, priors).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vlr.h"
#include "vlr_gpuio.h"

extern "C" void vlr_set_error(const char* msg);

namespace {
int dfail(int code, const char* fmt, const char* a = "", long long b = 0) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    vlr_set_error(buf);
    return code;
}
#define VLR_HIP_OK(call)                                                                                        \
    do {                                                                                                        \
        const hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return dfail(VLR_ERR_HIP, "%s (hip error %lld)", #call, (long long)e_);       \
    } while (0)

// the members of a BGZF byte string (SAM spec 4.1): DEFLATE payload and ISIZE of each; false: not BGZF / truncated
bool bgzf_members(const uint8_t* p, size_t n, std::vector<vlr::InflateBlock>& out, uint64_t& total) {
    size_t off = 0;
    total = 0;
    while (off < n) {
        if (n - off < 28 || p[off] != 0x1f || p[off + 1] != 0x8b || p[off + 2] != 8 || !(p[off + 3] & 4)) return false;
        const uint32_t xlen = (uint32_t)p[off + 10] | ((uint32_t)p[off + 11] << 8);
        if (n - off < 12 + (size_t)xlen) return false;
        uint32_t bsize = 0;
        bool found = false;
        for (size_t q = off + 12; q + 4 <= off + 12 + xlen;) {
            const uint32_t slen = (uint32_t)p[q + 2] | ((uint32_t)p[q + 3] << 8);
            if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= off + 12 + xlen) { bsize = ((uint32_t)p[q + 4] | ((uint32_t)p[q + 5] << 8)) + 1; found = true; break; }
            q += 4 + slen;
        }
        if (!found || bsize < 12 + xlen + 8 || n - off < bsize) return false;
        vlr::InflateBlock b;
        b.src = off + 12 + xlen;
        b.clen = bsize - (12 + xlen) - 8;
        memcpy(&b.isize, p + off + bsize - 4, 4);
        memcpy(&b.crc, p + off + bsize - 8, 4);
        b.pad = 0;
        b.dst = total;
        total += b.isize;
        out.push_back(b);
        off += bsize;
    }
    return true;
}
}  // namespace

extern "C" int vlr_bgzf_inflate(int device, const void* bgzf, int64_t n_bytes, void* out, int64_t out_capacity, int64_t* out_bytes) {
    if (!bgzf || n_bytes < 0 || !out_bytes || (!out && out_capacity > 0)) return dfail(VLR_ERR_INVALID_ARGUMENT, "vlr_bgzf_inflate: bad argument");
    std::vector<vlr::InflateBlock> blocks;
    uint64_t total = 0;
    if (!bgzf_members((const uint8_t*)bgzf, (size_t)n_bytes, blocks, total)) return dfail(VLR_ERR_INVALID_ARGUMENT, "vlr_bgzf_inflate: not a sequence of BGZF members");
    *out_bytes = (int64_t)total;
    if ((int64_t)total > out_capacity) return dfail(VLR_ERR_INVALID_ARGUMENT, "vlr_bgzf_inflate: output buffer too small (%s%lld bytes needed)", "", (long long)total);
    if (blocks.empty()) return VLR_OK;
    {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { (void)hipGetLastError(); return dfail(VLR_ERR_NO_DEVICE, "no HIP device (vlr_bgzf_inflate has no host path)"); }
    }
    VLR_HIP_OK(hipSetDevice(device));
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    vlr::InflateBlock* d_blocks = nullptr;
    int* d_status = nullptr;
    const size_t pad = vlr::kInflateInputSlack;
    int rc = VLR_OK;
    std::vector<int> status(blocks.size(), -1);
    auto run = [&]() -> int {
        VLR_HIP_OK(hipMalloc(&d_comp, (size_t)n_bytes + pad));
        VLR_HIP_OK(hipMalloc(&d_out, (size_t)total + 16));
        VLR_HIP_OK(hipMalloc(&d_blocks, blocks.size() * sizeof(vlr::InflateBlock)));
        VLR_HIP_OK(hipMalloc(&d_status, blocks.size() * sizeof(int)));
        VLR_HIP_OK(hipMemcpy(d_comp, bgzf, (size_t)n_bytes, hipMemcpyHostToDevice));
        VLR_HIP_OK(hipMemset(d_comp + n_bytes, 0, pad));
        VLR_HIP_OK(hipMemcpy(d_blocks, blocks.data(), blocks.size() * sizeof(vlr::InflateBlock), hipMemcpyHostToDevice));
        const int lrc = vlr_launch_inflate_kernel(d_comp, d_blocks, (int)blocks.size(), d_out, d_status, nullptr);
        if (lrc != 0) return dfail(VLR_ERR_HIP, "inflate kernel launch failed (hip error %s%lld)", "", lrc);
        VLR_HIP_OK(hipDeviceSynchronize());
        VLR_HIP_OK(hipMemcpy(status.data(), d_status, blocks.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < blocks.size(); ++i)
            if (status[i] != 0) return dfail(VLR_ERR_INVALID_ARGUMENT, "vlr_bgzf_inflate: corrupt DEFLATE stream (code %s%lld)", "", (long long)status[i] * 1000000 + (long long)i);
        if (total) VLR_HIP_OK(hipMemcpy(out, d_out, (size_t)total, hipMemcpyDeviceToHost));
        return VLR_OK;
    };
    rc = run();
    (void)hipFree(d_comp); (void)hipFree(d_out); (void)hipFree(d_blocks); (void)hipFree(d_status);
    return rc;
}

// ================================================================================================ kernels
namespace vlr {
namespace {

constexpr uint32_t kSeg = 65536;                       // bytes of the inflated stream per anchor / walk lane
constexpr uint64_t kNone = ~0ull;

// (records start at arbitrary byte offsets of the inflated stream; gfx950 global loads take unaligned addresses, and the compiler emits
// one load for these copies — byte-wise loads made the decode kernel request-bound in L2)
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// the fixed part of a BCF2 record header at offset o of the stream looks like one (necessary conditions of every record this
// reader accepts; NOT sufficient — the walk verifies every guess)
__device__ __forceinline__ bool header_plausible(const uint8_t* base, uint64_t o, uint64_t avail, int n_contigs, int n_hdr_samples, bool with_id) {
    if (o + 33 > avail) return false;
    const uint8_t* p = base + o;
    const uint32_t ls = ld32(p), li = ld32(p + 4);
    if (ls < 26 || ls >= (1u << 24) || li >= (1u << 28)) return false;
    const int32_t chrom = (int32_t)ld32(p + 8), pos = (int32_t)ld32(p + 12);
    if (chrom < 0 || (n_contigs > 0 && chrom >= n_contigs) || pos < -1) return false;
    const uint32_t nai = ld32(p + 24), nfs = ld32(p + 28);
    if ((nai >> 16) < 1 || (int)(nfs & 0xffffffu) != n_hdr_samples) return false;
    if (with_id && (p[32] & 15u) != 7u) return false;
    return true;
}

// anchor[i]: first plausible record start at or behind i * kSeg whose successor (if buffered) is plausible too; anchor[0] = 0 is known
__global__ __launch_bounds__(64) void rec_anchor_kernel(const uint8_t* __restrict__ base, uint64_t avail, int n_seg, int n_contigs, int n_hdr_samples,
                                                       uint64_t* __restrict__ anchor) {
    const int seg = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (seg >= n_seg) return;
    if (seg == 0) { if (lane == 0) anchor[0] = 0; return; }
    uint64_t found = kNone;
    for (uint64_t o0 = (uint64_t)seg * kSeg; o0 + 33 <= avail; o0 += 64) {
        const uint64_t o = o0 + (uint64_t)lane;
        bool ok = header_plausible(base, o, avail, n_contigs, n_hdr_samples, true);
        if (ok) {
            const uint64_t nxt = o + 8 + (uint64_t)ld32(base + o) + (uint64_t)ld32(base + o + 4);
            if (nxt + 33 <= avail) ok = header_plausible(base, nxt, avail, n_contigs, n_hdr_samples, true);
        }
        const unsigned long long m = __ballot(ok);
        if (m != 0) { found = o0 + (uint64_t)(__ffsll((long long)m) - 1); break; }
    }
    if (lane == 0) anchor[seg] = found;
}

// first record start at or behind offset 0 of a stream that begins INSIDE a record (a shard of a file, vlr_obs_reader_open_device_shard):
// the first offset whose header is plausible and whose two successors are as well.  A guess like every anchor: the verified walk of
// the split behind it and the neighbouring shard's landing (vlr_obs_reader_shard_assign) confirm it.
__global__ __launch_bounds__(64) void rec_first_kernel(const uint8_t* __restrict__ base, uint64_t avail, int n_contigs, int n_hdr_samples, uint64_t* __restrict__ first) {
    const int lane = (int)threadIdx.x;
    uint64_t found = kNone;
    for (uint64_t o0 = 0; o0 + 33 <= avail; o0 += 64) {
        const uint64_t o = o0 + (uint64_t)lane;
        bool ok = header_plausible(base, o, avail, n_contigs, n_hdr_samples, true);
        uint64_t q = o;
        for (int hop = 0; ok && hop < 2; ++hop) {
            q = q + 8 + (uint64_t)ld32(base + q) + (uint64_t)ld32(base + q + 4);
            if (q + 33 <= avail) ok = header_plausible(base, q, avail, n_contigs, n_hdr_samples, true);
            else break;
        }
        const unsigned long long m = __ballot(ok);
        if (m != 0) { found = o0 + (uint64_t)(__ffsll((long long)m) - 1); break; }
    }
    if (lane == 0) *first = found;
}

// one lane per segment: the complete records that start in [anchor, next boundary)
__global__ void rec_walk_kernel(const uint8_t* __restrict__ base, uint64_t avail, int n_seg, const uint64_t* __restrict__ anchor,
                                uint32_t* __restrict__ count, uint64_t* __restrict__ landing, uint8_t* __restrict__ land_complete) {
    const int seg = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (seg >= n_seg) return;
    uint64_t o = anchor[seg];
    const uint64_t lim = (uint64_t)(seg + 1) * kSeg;
    uint32_t c = 0;
    bool complete = false;
    if (o != kNone) {
        for (;;) {
            complete = false;
            if (o + 8 > avail) break;
            const uint64_t nxt = o + 8 + (uint64_t)ld32(base + o) + (uint64_t)ld32(base + o + 4);
            if (nxt > avail) break;
            complete = true;
            if (o >= lim) break;   // the landing: first start at or behind the next boundary (its record is complete)
            c += 1;
            o = nxt;
        }
    }
    count[seg] = c; landing[seg] = o; land_complete[seg] = complete ? 1 : 0;
}

// second pass of the verified walk: starts[seg_base[seg] + j] for the records of the segment (indices above n_keep are dropped)
__global__ void rec_starts_kernel(const uint8_t* __restrict__ base, uint64_t avail, int n_seg, const uint64_t* __restrict__ anchor,
                                  const uint64_t* __restrict__ seg_base, const uint32_t* __restrict__ count, uint64_t n_keep, uint64_t* __restrict__ starts) {
    const int seg = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (seg >= n_seg) return;
    uint64_t o = anchor[seg];
    const uint64_t b = seg_base[seg];
    const uint32_t c = count[seg];
    for (uint32_t j = 0; j <= c; ++j) {   // (j == c: the landing = start of the next segment's first record, or the end of the records)
        if (b + j <= n_keep && (j < c || b + j == n_keep)) starts[b + j] = o;
        if (j < c) o = o + 8 + (uint64_t)ld32(base + o) + (uint64_t)ld32(base + o + 4);
    }
}

// the serial walk (fallback when a guessed anchor was wrong): one lane, every record
__global__ void rec_walk_serial_kernel(const uint8_t* __restrict__ base, uint64_t avail, uint64_t max_records, uint64_t* __restrict__ starts, uint64_t* __restrict__ n_out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t o = 0, n = 0;
    while (n < max_records && o + 8 <= avail) {
        const uint64_t nxt = o + 8 + (uint64_t)ld32(base + o) + (uint64_t)ld32(base + o + 4);
        if (nxt > avail) break;
        starts[n++] = o;
        o = nxt;
    }
    starts[n] = o;
    *n_out = n;
}

struct DTyped { uint32_t type, n, off; };
// BCF2 typed value at rec[q] (bcf_typed of vlr_ingest.cpp): descriptor, optional length scalar, payload
__device__ __forceinline__ bool d_typed(const uint8_t* rec, uint32_t& q, uint32_t end, DTyped& t) {
    if (q >= end) return false;
    const uint32_t b = rec[q++];
    t.type = b & 15u; t.n = b >> 4;
    if (t.n == 15) {
        if (q >= end) return false;
        const uint32_t b2 = rec[q++];
        const uint32_t lt = b2 & 15u;
        if ((b2 >> 4) != 1) return false;
        if (lt == 1) { if (end - q < 1) return false; t.n = (uint32_t)(int32_t)(int8_t)rec[q]; q += 1; }
        else if (lt == 2) { if (end - q < 2) return false; t.n = (uint32_t)(int32_t)(int16_t)ld16(rec + q); q += 2; }
        else if (lt == 3) { if (end - q < 4) return false; t.n = ld32(rec + q); q += 4; }
        else return false;
    }
    const uint32_t size = t.type == 1 ? 1u : t.type == 2 ? 2u : (t.type == 3 || t.type == 5) ? 4u : t.type == 7 ? 1u : 0u;
    const uint64_t bytes = (uint64_t)t.n * size;
    if ((uint64_t)(end - q) < bytes) return false;
    t.off = q;
    q += (uint32_t)bytes;
    return true;
}
// u16 word k of an INFO integer vector (read_observations: i32 -> u16, preprocessing/mod.rs:836-842)
__device__ __forceinline__ uint32_t vword(const uint8_t* v, int stride, uint32_t k) {
    if (stride == 4) return ld16(v + 4 * (size_t)k);
    if (stride == 2) return ld16(v + 2 * (size_t)k);
    return (uint32_t)(uint16_t)(int16_t)(int8_t)v[k];
}
__device__ __forceinline__ uint32_t vbyte(const uint8_t* v, int stride, uint32_t j) {  // byte j of the little-endian word stream
    if (stride == 4) return v[4 * (size_t)(j >> 1) + (j & 1u)];
    if (stride == 2) return v[j];
    const uint32_t w = (uint32_t)(uint16_t)(int16_t)(int8_t)v[j >> 1];
    return (w >> (8 * (j & 1u))) & 0xffu;
}
__device__ __forceinline__ uint64_t vlen(const uint8_t* v, int stride, uint32_t nw) {  // the u64 element count in front of a bincode Vec
    if (nw < 4) return ~0ull;
    return (uint64_t)vword(v, stride, 0) | ((uint64_t)vword(v, stride, 1) << 16) | ((uint64_t)vword(v, stride, 2) << 32) | ((uint64_t)vword(v, stride, 3) << 48);
}

// one lane per record: walk the shared part, leave the descriptor
__global__ void rec_scan_kernel(const uint8_t* __restrict__ base, const uint64_t* __restrict__ starts, int64_t n_rec, const int8_t* __restrict__ field_of_key, int n_keys,
                                RecDesc* __restrict__ desc, RecHost* __restrict__ host) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rec) return;
    const uint64_t start = starts[r];
    const uint64_t rec_len = starts[r + 1] - start;
    const uint8_t* rec = base + start;
    RecDesc d;
    d.start = start; d.l_shared = 0; d.n_obs = 0; d.cold_bytes = 0; d.n_cold_info = 0;
    for (int i = 0; i < kColdSegs; ++i) { d.seg_off[i] = 0; d.seg_len[i] = 0; }
    for (int i = 0; i < kGpuVec; ++i) { d.voff[i] = 0; d.vn[i] = 0; d.vstride[i] = 0; }
    for (int i = 0; i < 6; ++i) d.pad[i] = 0;
    uint32_t status = 0;
    bool cold_present[kColdSegs] = {false, false, false, false, false, false};
    do {
        if (rec_len < 32) { status |= REC_TRUNCATED; break; }
        const uint32_t ls = ld32(rec);
        d.l_shared = ls;
        if ((uint64_t)8 + ls > rec_len || ls < 24) { status |= REC_TRUNCATED; break; }
        const uint32_t end = 8 + ls;
        const uint32_t nai = ld32(rec + 24);
        const uint32_t n_allele = nai >> 16, n_info = nai & 0xffffu;
        uint32_t q = 32;
        DTyped t;
        if (!d_typed(rec, q, end, t)) { status |= REC_BAD_ID; break; }
        bool bad = false;
        for (uint32_t a = 0; a < n_allele; ++a)
            if (!d_typed(rec, q, end, t) || (t.type != 7 && t.n != 0)) { bad = true; break; }
        if (bad) { status |= REC_BAD_ALLELE; break; }
        if (!d_typed(rec, q, end, t)) { status |= REC_BAD_FILTER; break; }
        d.seg_off[0] = 32; d.seg_len[0] = q - 32;
        for (uint32_t k = 0; k < n_info; ++k) {
            const uint32_t entry = q;
            DTyped key, val;
            if (!d_typed(rec, q, end, key) || key.n != 1 || key.type < 1 || key.type > 3 || !d_typed(rec, q, end, val)) { bad = true; break; }
            const int32_t ki = key.type == 1 ? (int32_t)(int8_t)rec[key.off] : key.type == 2 ? (int32_t)(int16_t)ld16(rec + key.off) : (int32_t)ld32(rec + key.off);
            const int f = (ki >= 0 && ki < n_keys) ? (int)field_of_key[ki] : -1;
            if (f < 0) continue;
            if (f < FD_N_VEC) {
                if (val.type < 1 || val.type > 3) continue;
                d.voff[f] = val.off; d.vn[f] = val.n; d.vstride[f] = (uint8_t)(val.type == 3 ? 4 : val.type);
            } else {
                const int slot = 1 + (f - FD_N_VEC);
                d.seg_off[slot] = entry; d.seg_len[slot] = q - entry;
                cold_present[slot] = true;
            }
        }
        if (bad) { status |= REC_BAD_INFO; break; }
        for (int f = 0; f <= FD_MAX_MAPQ; ++f)
            if (d.vstride[f] == 0) status |= REC_MISSING_FIELD;
        if (status) break;
        const uint64_t n = vlen(rec + d.voff[FD_PROB_MAPPING], d.vstride[FD_PROB_MAPPING], d.vn[FD_PROB_MAPPING]);
        if (n > (1ull << 28)) { status |= REC_BAD_LENGTHS; break; }
        // (a count the vector cannot hold — three words per element at least — is refused here, before the table is sized by it)
        if ((uint64_t)d.vn[FD_PROB_MAPPING] < 4ull + 3ull * n) { status |= REC_BAD_VECTOR; break; }
        d.n_obs = (uint32_t)n;
    } while (false);
    uint32_t cold = 8 + 24, nci = 0;
    for (int i = 0; i < kColdSegs; ++i) { cold += d.seg_len[i]; if (i > 0 && cold_present[i]) nci += 1; }
    d.cold_bytes = cold; d.n_cold_info = nci;
    desc[r] = d;
    RecHost h;
    h.n_obs = status ? 0u : d.n_obs; h.cold_bytes = cold; h.status = status;
    h.flags = (d.vstride[FD_HP_ART] != 0 && d.n_obs > 0) ? 1u : 0u;
    host[r] = h;
}

// half -> float bits, the host decoder's half_to_float (exact; NaN payloads kept)
__device__ __forceinline__ uint32_t half_bits(uint32_t h) {
    const uint32_t s = (h >> 15) << 31, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return s;
        const int k = __clz((int)m) - 21;  // shifts until bit 10 is set
        const uint32_t mm = m << k;
        return s | ((uint32_t)(127 - 15 - k + 1) << 23) | ((mm & 0x3ffu) << 13);
    }
    if (e == 31) return s | 0x7f800000u | (m << 13);
    return s | ((e + 112u) << 23) | (m << 13);
}

// one wave per record.  Lanes 0..10 each walk one variable-stride vector (elements are 2..9 bytes of the little-endian word stream:
// [Option tag u8] [MiniLogProb: u32 tag + f16 | f32] or [u8] or [u32]); then all lanes assemble the flags word of every observation
// from the fixed-stride enum vectors and the bit vectors.
__global__ __launch_bounds__(64) void rec_decode_kernel(const uint8_t* __restrict__ base, const RecDesc* __restrict__ desc, int64_t n_rec, const uint32_t* __restrict__ obs_offset,
                                                       int n_samples, int sample, DeviceCols cols, RecHost* __restrict__ host) {
    const int64_t r = (int64_t)blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (r >= n_rec) return;
    const RecDesc& d = desc[r];
    if (host[r].status != 0) return;
    const uint8_t* rec = base + d.start;
    const uint32_t n = d.n_obs;
    const size_t ob = (size_t)obs_offset[(size_t)r * (size_t)n_samples + (size_t)sample];
    uint32_t status = 0;
    // ---- chains, in lockstep (round 5).  Where element i + 1 starts depends on element i; round 4 let eleven lanes walk their vectors
    // through one generic step with a branch per element kind and typed-integer width — the lanes diverge, the wave ran every path of
    // every step (~1 000 static instructions, 40 % of them scalar control) and the kernel was bound by scalar issue: without the chains
    // it takes 0.1 ms instead of 1.5 ms per 32 768 records, without their stores or with the vectors copied to LDS first just as long.
    // Now the step is the same straight-line code for every lane: three unaligned loads at an address scaled by the lane's integer
    // width, the three widths and the four element kinds told apart by selects.  (The same step on an LDS copy of the vectors was 1.8 x
    // faster alone, but 12 kB of LDS per wave keep the kernel off the CUs the call kernel fills: end to end 0.64 M against 0.88 M.)
    // lane -> (field, kind): kind 0 MiniLogProb, 1 Option<MiniLogProb>, 2 Option<u8>, 3 Option<u32>
    if (lane < 11) {
        const int field = lane == 0 ? FD_PROB_MAPPING : lane == 1 ? FD_PROB_ALT : lane == 2 ? FD_PROB_REF : lane == 3 ? FD_PROB_MISSED : lane == 4 ? FD_PROB_SAMPLE_ALT
                        : lane == 5 ? FD_PROB_DOUBLE_OVERLAP : lane == 6 ? FD_PROB_HIT_BASE : lane == 7 ? FD_HP_ART : lane == 8 ? FD_HP_VAR : lane == 9 ? FD_HP_LEN : FD_THIRD;
        const uint32_t kind = lane < 7 ? 0u : lane < 9 ? 1u : lane == 9 ? 2u : 3u;
        const int stride = d.vstride[field];
        const bool present = stride != 0;
        const uint8_t* v = rec + d.voff[field];
        const uint32_t nbytes = present ? 2u * d.vn[field] : 0u;
        uint32_t* out = lane < 9 ? reinterpret_cast<uint32_t*>(cols.col[lane]) + ob : lane == 9 ? cols.flags + ob : reinterpret_cast<uint32_t*>(cols.third) + ob;
        const uint32_t none = kind <= 1 ? 0x7fc00000u : kind == 2 ? 0u : 0xffffffffu;
        // an absent optional field is None everywhere (the mandatory ones were checked by the scan); a vector whose length word differs
        // from the record's observation count writes nothing: the record is an error
        bool alive = present;
        if (present && vlen(v, stride, d.vn[field]) != (uint64_t)n) { status |= REC_BAD_LENGTHS; alive = false; }
        const bool writes = alive || !present;
        uint32_t j = 8;
        for (uint32_t i = 0; i < n; ++i) {
            // bytes j .. j + 8 of the little-endian word stream: five words = five int32 / int16 / int8 elements of the file
            const uint32_t jj = alive ? j : 0u;
            const uint8_t* q = v + (stride == 4 ? 4 * (size_t)(jj >> 1) : stride == 2 ? (size_t)jj : (size_t)(jj >> 1));
            const uint64_t A = ld64(q), C = ld64(q + 8);
            const uint32_t W = ld32(q + 16);
            const bool odd = (jj & 1u) != 0;
            // int32: the words are the low halves
            const uint64_t W4 = (A & 0xffffull) | ((A >> 16) & 0xffff0000ull) | ((C & 0xffffull) << 32) | (((C >> 32) & 0xffffull) << 48);
            const uint64_t lo4 = odd ? (W4 >> 8) | ((uint64_t)(W & 0xffu) << 56) : W4;
            const uint32_t b84 = odd ? (W >> 8) & 0xffu : W & 0xffu;
            // int8: word k = byte k sign-extended
            const uint64_t e0 = A & 0xffull, e1 = (A >> 8) & 0xffull, e2 = (A >> 16) & 0xffull, e3 = (A >> 24) & 0xffull, e4 = (A >> 32) & 0xffull;
            const uint64_t W1 = (e0 | ((e0 & 0x80ull) ? 0xff00ull : 0ull)) | ((e1 | ((e1 & 0x80ull) ? 0xff00ull : 0ull)) << 16) | ((e2 | ((e2 & 0x80ull) ? 0xff00ull : 0ull)) << 32) |
                                ((e3 | ((e3 & 0x80ull) ? 0xff00ull : 0ull)) << 48);
            const uint32_t w41 = (uint32_t)(e4 | ((e4 & 0x80ull) ? 0xff00ull : 0ull));
            const uint64_t lo1 = odd ? (W1 >> 8) | ((uint64_t)(w41 & 0xffu) << 56) : W1;
            const uint32_t b81 = odd ? (w41 >> 8) & 0xffu : w41 & 0xffu;
            uint64_t lo = stride == 4 ? lo4 : stride == 2 ? A : lo1;
            uint32_t b8 = stride == 4 ? b84 : stride == 2 ? (uint32_t)C & 0xffu : b81;
            {   // zero beyond the end of the stream (jj <= nbytes)
                const uint32_t left = nbytes - jj;
                lo = left < 8 ? lo & ((1ull << (8 * left)) - 1ull) : lo;
                b8 = left < 9 ? 0u : b8;
            }
            const bool some = kind == 0 || ((uint32_t)lo & 0xffu) != 0;
            const uint64_t t = kind == 0 ? lo : (lo >> 8) | ((uint64_t)b8 << 56);   // the eight bytes behind the Option tag
            const uint32_t tag = (uint32_t)t, pay = (uint32_t)(t >> 32);
            // MiniLogProb: u32 tag, then f16 (tag 0, the host decoder's half_to_float: exact, NaN payloads kept) or f32 (tag 1)
            const uint32_t h = pay & 0xffffu, hs = (h >> 15) << 31, he = (h >> 10) & 0x1fu, hm = h & 0x3ffu;
            const int hk = __clz((int)(hm | 1u)) - 21;
            const uint32_t sub = hm == 0 ? hs : hs | ((uint32_t)(127 - 15 - hk + 1) << 23) | (((hm << hk) & 0x3ffu) << 13);
            const uint32_t half = he == 0 ? sub : he == 31 ? hs | 0x7f800000u | (hm << 13) : hs | ((he + 112u) << 23) | (hm << 13);
            const uint32_t val_m = tag == 0 ? half : pay;
            const uint32_t o = kind == 0 ? 0u : 1u;
            const uint32_t size = kind <= 1 ? o + (some ? (tag == 0 ? 6u : 8u) : 0u) : kind == 2 ? 1u + (some ? 1u : 0u) : 1u + (some ? 4u : 0u);
            const uint32_t val_s = kind <= 1 ? val_m : kind == 2 ? (VLR_F_HP_LEN_VALID | (((uint32_t)t & 0xffu) << VLR_F_HP_LEN_SHIFT)) : (uint32_t)t;
            const uint32_t val = some ? val_s : none;
            if (alive && kind <= 1 && some && tag > 1) status |= REC_BAD_VECTOR;
            if (alive && j + size > nbytes) { status |= REC_BAD_VECTOR; alive = false; }
            if (writes && (alive || !present)) out[i] = present ? val : none;
            j = alive ? j + size : j;
        }
    }
    __syncthreads();   // the HOMOPOLYMER_INDEL_LEN bits of lane 9 are in cols.flags now
    // ---- flags
    {
        const bool hp_len = d.vstride[FD_HP_LEN] != 0;
        const uint8_t* ev[4]; int es[4];
        const int ef[4] = {FD_STRAND, FD_ORIENT, FD_READPOS, FD_ALTLOCUS};
        bool ok = true;
        for (int k = 0; k < 4; ++k) {
            ev[k] = rec + d.voff[ef[k]]; es[k] = d.vstride[ef[k]];
            if (vlen(ev[k], es[k], d.vn[ef[k]]) != (uint64_t)n) { status |= REC_BAD_LENGTHS; ok = false; }
            else if ((uint64_t)d.vn[ef[k]] < 4ull + 2ull * n) { status |= REC_BAD_VECTOR; ok = false; }
        }
        // bv::BitVec<u8>: Option tag, u64 blocks, bytes, u64 bits
        const int bf[3] = {FD_SOFTCLIPPED, FD_PAIRED, FD_MAX_MAPQ};
        const uint32_t bflag[3] = {VLR_F_SOFTCLIPPED, VLR_F_PAIRED, VLR_F_MAX_MAPQ};
        const uint8_t* bv[3]; int bs[3]; bool bsome[3];
        for (int k = 0; k < 3; ++k) {
            bv[k] = rec + d.voff[bf[k]]; bs[k] = d.vstride[bf[k]];
            const uint32_t nbytes = 2u * d.vn[bf[k]];
            auto u64at = [&](uint32_t j) { uint64_t x = 0; for (int t = 0; t < 8; ++t) x |= (uint64_t)vbyte(bv[k], bs[k], j + (uint32_t)t) << (8 * t); return x; };
            bsome[k] = false;
            if (nbytes < 9) { status |= REC_BAD_VECTOR; ok = false; continue; }
            const bool some = vbyte(bv[k], bs[k], 0) != 0;
            const uint64_t a = u64at(1);
            if (!some) { if (a != (uint64_t)n && !(a == 0 && n == 0)) { status |= REC_BAD_LENGTHS; ok = false; } continue; }
            if (a > (uint64_t)nbytes || 9ull + a + 8ull > (uint64_t)nbytes) { status |= REC_BAD_VECTOR; ok = false; continue; }
            const uint64_t nbits = u64at(9u + (uint32_t)a);
            if (nbits != (uint64_t)n || a * 8 < nbits) { status |= REC_BAD_LENGTHS; ok = false; continue; }
            bsome[k] = true;
        }
        if (ok) {
            for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
                uint32_t e[4];
                for (int k = 0; k < 4; ++k) {
                    if (es[k] == 4) {   // the two words of the u32 in one 8-byte load
                        const uint64_t x = ld64(ev[k] + 4 * (size_t)(4 + 2 * i));
                        e[k] = (uint32_t)(x & 0xffffull) | (uint32_t)((x >> 32) & 0xffffull) << 16;
                    } else e[k] = vword(ev[k], es[k], 4 + 2 * i) | (vword(ev[k], es[k], 5 + 2 * i) << 16);
                }
                uint32_t fl = (e[0] & 3u) << VLR_F_STRAND_SHIFT;
                const uint32_t orient = e[1] == 0 ? VLR_ORIENT_F1R2 : e[1] == 1 ? VLR_ORIENT_F2R1 : e[1] == 8 ? VLR_ORIENT_NONE : VLR_ORIENT_OTHER;
                fl |= orient << VLR_F_ORIENT_SHIFT;
                if (e[2] == 0) fl |= VLR_F_READPOS_MAJOR;
                fl |= (e[3] & 3u) << VLR_F_ALTLOCUS_SHIFT;
                for (int k = 0; k < 3; ++k)
                    if (bsome[k] && ((vbyte(bv[k], bs[k], 9 + (i >> 3)) >> (i & 7u)) & 1u)) fl |= bflag[k];
                if (hp_len) fl |= cols.flags[ob + i];
                cols.flags[ob + i] = fl;
            }
        }
    }
    if (status) atomicOr(&host[r].status, status);
}

// one wave per record: header with patched lengths, then the kept segments
__global__ __launch_bounds__(64) void rec_cold_kernel(const uint8_t* __restrict__ base, const RecDesc* __restrict__ desc, int64_t n_rec, const uint64_t* __restrict__ cold_off,
                                                     uint8_t* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x;
    const int lane = (int)threadIdx.x;
    if (r >= n_rec) return;
    const RecDesc& d = desc[r];
    const uint8_t* rec = base + d.start;
    uint8_t* o = out + cold_off[r];
    const uint32_t ls = d.cold_bytes - 8;
    if (lane < 32) {
        uint32_t v;
        if (lane < 4) v = (ls >> (8 * lane)) & 0xffu;
        else if (lane < 8) v = 0;                                  // l_indiv
        else if (lane >= 24 && lane < 28) {                        // n_allele << 16 | n_info
            const uint32_t nai = (ld32(rec + 24) & 0xffff0000u) | d.n_cold_info;
            v = (nai >> (8 * (lane - 24))) & 0xffu;
        } else v = rec[lane];                                      // CHROM, POS, rlen, QUAL, n_fmt / n_sample
        o[lane] = (uint8_t)v;
    }
    uint32_t at = 32;
    for (int s = 0; s < kColdSegs; ++s) {
        const uint32_t len = d.seg_len[s], off = d.seg_off[s];
        for (uint32_t k = (uint32_t)lane; k < len; k += 64) o[at + k] = rec[off + k];
        at += len;
    }
}

// ---- observation summaries for the calls writer: one WAVE per pileup (round 5; round 4 ran the host's loop with one lane per pileup and
// kept 64 distinct keys in registers, which the synthetic pileups — almost one key per observation — overflowed).  The wave computes the
// packed key of every kept observation (sample_fields of vlr_ingest.cpp, the same decisions), counts equal keys by comparing every
// observation with every other one (keys in LDS, broadcast reads: 2 x 100 x 100 compares for a tumor-normal record), orders the distinct
// keys like generalized_cigar(keep_order = false) (utils/mod.rs:122-156: stable by count descending, then stable by class — one rank
// over (class, -count, first appearance)) and writes the OBS TEXT: the host copies it into the record.
__device__ __forceinline__ bool d_relative_eq(double a, double b, double eps) {
    if (a == b) return true;
    if (isinf(a) || isinf(b)) return false;
    const double d = fabs(a - b);
    return d <= eps || d <= fmax(fabs(a), fabs(b)) * eps;
}
__device__ __forceinline__ uint32_t d_kr_letter(double bf, double eps) {  // utils/mod.rs:158-167
    if (bf <= 1.0) return d_relative_eq(bf, 1.0, eps) ? 'E' : 'N';
    if (bf <= 3.0) return 'B';
    if (bf <= 20.0) return 'P';
    if (bf <= 150.0) return 'S';
    return 'V';
}
__device__ __forceinline__ uint32_t d_lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32u : c; }

// the packed key of one kept observation and the letter it adds to SAOBS (to_alt) or SROBS
struct ObsKey { uint64_t key; uint32_t letter; bool to_alt; };
__device__ __forceinline__ ObsKey d_obs_key(float pa_f, float pr_f, uint32_t f, int32_t third, const SumConsts& K) {
    const double pa = (double)pa_f, pr = (double)pr_f;
    const double d = pa - pr;
    double bf_alt, bf_ref;
    uint32_t kl_alt, kl_ref;
    if (fabs(d) >= 1e-15 && fabs(d) < 700.0) {
        const double ad = fabs(d);
        const uint32_t k = ad <= K.ln3 ? 'B' : ad <= K.ln20 ? 'P' : ad <= K.ln150 ? 'S' : 'V';
        bf_alt = d > 0 ? 2.0 : 0.5; bf_ref = d > 0 ? 0.5 : 2.0;
        kl_alt = d > 0 ? k : 'N'; kl_ref = d > 0 ? 'N' : k;
    } else if (fabs(d) >= 700.0) {
        // exp(+-d) is 0 / inf or 1e-304 / 1e304: the order and the letters of the host's exponentials
        bf_alt = d > 0 ? 2.0 : 0.5; bf_ref = d > 0 ? 0.5 : 2.0;
        kl_alt = d > 0 ? 'V' : 'N'; kl_ref = d > 0 ? 'N' : 'V';
    } else {
        // |d| < 1e-15 (or NaN): exp(d) = 1 + d rounded to nearest (the next term is below 1e-30)
        bf_alt = 1.0 + d; bf_ref = 1.0 - d;
        kl_alt = d_kr_letter(bf_alt, K.eps); kl_ref = d_kr_letter(bf_ref, K.eps);
    }
    const bool maxq = (f & VLR_F_MAX_MAPQ) != 0;
    uint32_t s0, s1 = 0;
    if (bf_alt > bf_ref) { s0 = 'A'; s1 = kl_alt; }
    else if (bf_ref > bf_alt) { s0 = 'R'; s1 = kl_ref; }
    else s0 = 'E';
    if (!maxq) { s0 = d_lower(s0); if (s1) s1 = d_lower(s1); }
    const uint32_t orient = (f >> VLR_F_ORIENT_SHIFT) & 3u, strand = (f >> VLR_F_STRAND_SHIFT) & 3u, altloc = (f >> VLR_F_ALTLOCUS_SHIFT) & 3u;
    const bool hp_err = (f & VLR_F_HP_LEN_VALID) && ((f >> VLR_F_HP_LEN_SHIFT) & 0xffu) != 0;
    ObsKey o;
    o.key = (uint64_t)s0 | ((uint64_t)s1 << 8) | ((uint64_t)((f & VLR_F_PAIRED) ? 1 : 0) << 16) |
            ((uint64_t)(altloc > 2 ? 2 : altloc) << 17) | ((uint64_t)strand << 19) | ((uint64_t)orient << 21) |
            ((uint64_t)((f & VLR_F_READPOS_MAJOR) ? 1 : 0) << 23) | ((uint64_t)((f & VLR_F_SOFTCLIPPED) ? 1 : 0) << 24) |
            ((uint64_t)(hp_err ? 1 : 0) << 25) | ((uint64_t)(uint32_t)(third + 1) << 32);
    o.to_alt = pa > pr;
    const uint32_t c0 = o.to_alt ? kl_alt : kl_ref;
    o.letter = maxq ? c0 : d_lower(c0);
    return o;
}
// N B P S V E, lower case + 6
__device__ __forceinline__ uint32_t d_letter_index(uint32_t c) {
    const uint32_t u = c & ~32u;
    const uint32_t i = u == 'N' ? 0u : u == 'B' ? 1u : u == 'P' ? 2u : u == 'S' ? 3u : u == 'V' ? 4u : 5u;
    return i + ((c & 32u) ? 6u : 0u);
}
__device__ __forceinline__ uint32_t d_letter_char(uint32_t i) {
    const uint32_t j = i >= 6 ? i - 6 : i;
    const uint32_t u = j == 0 ? 'N' : j == 1 ? 'B' : j == 2 ? 'P' : j == 3 ? 'S' : j == 4 ? 'V' : 'E';
    return i >= 6 ? u + 32u : u;
}
__device__ __forceinline__ uint32_t d_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ uint8_t* d_put_dec(uint8_t* p, uint32_t v) {
    const uint32_t n = d_digits(v);
    for (uint32_t i = n; i-- > 0;) { p[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
    return p + n;
}

// bytes of one OBS item: count, one or two score letters, third-allele evidence or '.', seven flag characters
__device__ __forceinline__ uint32_t d_item_len(uint64_t key, uint32_t cnt) {
    const uint32_t th = (uint32_t)(key >> 32);
    return d_digits(cnt) + 1u + (((key >> 8) & 0xffu) ? 1u : 0u) + (th ? d_digits(th - 1u) : 1u) + 7u;
}

// Three launches: obs_text_kernel counts, orders and leaves the distinct keys of pileup p in OUTPUT order in item_key / item_cnt
// [obs_offset[p], + n_item) with the byte length of its text in text_len[p]; span_alloc_kernel places the texts (one atomic per 64
// pileups: a bump cursor taken once per pileup made 131 000 same-address atomics per request the bound of the kernel);
// obs_write_kernel writes the bytes.
__global__ __launch_bounds__(64) void obs_text_kernel(DeviceCols cols, const uint32_t* __restrict__ obs_offset, const uint8_t* __restrict__ locus_flags, int n_samples,
                                                      SumConsts K, PileSum* __restrict__ hdr, uint64_t* __restrict__ item_key, uint32_t* __restrict__ item_cnt,
                                                      uint32_t* __restrict__ text_len, float* __restrict__ run_pm, uint32_t* __restrict__ run_len,
                                                      uint32_t* __restrict__ cursor, uint32_t lds_obs) {
    // dynamic LDS sized by the launcher for the largest pileup of the table (lds_obs = its observations, at most kSumMaxObs, + 8 for the
    // sentinels of the unrolled loops, a multiple of 8): a 100x table takes 4 kB per wave instead of 25
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint64_t* s_key = (uint64_t*)s_dyn;                  // keys of the kept observations; then the order values of the distinct keys
    uint64_t* s_hkey = s_key + lds_obs;                  // distinct keys in first-appearance order
    uint32_t* s_a = (uint32_t*)(s_hkey + lds_obs);       // prob_mapping of the kept observations (bits); then the counts of the distinct keys
    uint32_t* s_b = s_a + lds_obs;                       // run heads
    uint32_t* s_lc = s_b + lds_obs;                      // SAOBS [0, 12) / SROBS [12, 24) letters: count, first appearance, order
    uint32_t* s_lf = s_lc + 24;
    uint32_t* s_perm = s_lf + 24;
    const int64_t p = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint64_t below = (1ull << lane) - 1ull;
    const uint32_t b = obs_offset[p], e = obs_offset[p + 1], n = e - b;
    if (n + 8u > lds_obs) {   // (wave-uniform; more than kSumMaxObs observations) the host counts this table from the columns
        if (lane == 0) {
            PileSum h;
            memset(&h, 0, sizeof h);
            h.overflow = 1;
            hdr[p] = h;
            text_len[p] = 0xffffffffu;
        }
        return;
    }
    const bool drop_nonstd = (locus_flags[p / n_samples] & VLR_LOCUS_REMOVE_NONSTANDARD) != 0;   // pileup.rs:26-43
    if (lane < 24) { s_lc[lane] = 0u; s_lf[lane] = 0xffffffffu; }
    __syncthreads();
    // ---- keys of the kept observations, in observation order
    uint32_t kept = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t i = c0 + lane;
        bool keep = false;
        uint64_t key = 0;
        uint32_t pmb = 0;
        if (i < n) {
            const uint32_t f = cols.flags[b + i];
            keep = !(drop_nonstd && ((f >> VLR_F_ORIENT_SHIFT) & 3u) == VLR_ORIENT_OTHER);
            if (keep) {
                pmb = __float_as_uint(cols.col[0][b + i]);
                const ObsKey k = d_obs_key(cols.col[1][b + i], cols.col[2][b + i], f, cols.third[b + i], K);
                key = k.key;
                const uint32_t code = (k.to_alt ? 0u : 12u) + d_letter_index(k.letter);
                atomicAdd(&s_lc[code], 1u);
                atomicMin(&s_lf[code], i);
            }
        }
        const uint64_t m = __ballot(keep);
        if (keep) { const uint32_t at = kept + (uint32_t)__popcll(m & below); s_key[at] = key; s_a[at] = pmb; }
        kept += (uint32_t)__popcll(m);
    }
    __syncthreads();
    // ---- runs of equal prob_mapping (the host compares the floats: NaN starts a run every time, there and here)
    uint32_t R = 0;
    for (uint32_t c0 = 0; c0 < kept; c0 += 64) {
        const uint32_t k = c0 + lane;
        bool head = false;
        if (k < kept) head = k == 0 || !(__uint_as_float(s_a[k]) == __uint_as_float(s_a[k - 1]));
        const uint64_t m = __ballot(head);
        if (head) s_b[R + (uint32_t)__popcll(m & below)] = k;
        R += (uint32_t)__popcll(m);
    }
    __syncthreads();
    // (one run is the rule — prob_mapping is the MAPQ-adjusted mean of the pileup — and sits in the header; only further runs take
    //  entries of the run arrays)
    uint32_t run_off = 0;
    if (R > 1) {
        if (lane == 0) run_off = atomicAdd(&cursor[1], R - 1u);
        run_off = (uint32_t)__shfl((int)run_off, 0);
    }
    for (uint32_t r = 1u + lane; r < R; r += 64) {
        const uint32_t k0 = s_b[r], k1 = r + 1 < R ? s_b[r + 1] : kept;
        run_pm[run_off + r - 1u] = __uint_as_float(s_a[k0]);
        run_len[run_off + r - 1u] = k1 - k0;
    }
    const uint32_t run0_pm = R ? s_a[0] : 0u, run0_len = R ? (R > 1 ? s_b[1] : kept) : 0u;
    __syncthreads();
    // ---- distinct keys with their counts, in first-appearance order (what Counter::most_common sees).  Every observation is compared
    // with every other one: the keys come from LDS as broadcast reads, eight per step (behind the last key stand eight sentinels no key
    // equals), and a lane carries the observations of two chunks through one sweep.
    if (lane < 8) s_key[kept + lane] = ~0ull;
    __syncthreads();
    const uint32_t kept8 = (kept + 7u) & ~7u;
    uint32_t n_dist = 0;
    for (uint32_t c0 = 0; c0 < kept; c0 += 128) {
        const uint32_t k0 = c0 + lane, k1 = c0 + 64 + lane;
        const bool valid0 = k0 < kept, valid1 = k1 < kept;
        const uint64_t my0 = valid0 ? s_key[k0] : 0ull, my1 = valid1 ? s_key[k1] : 0ull;
        uint32_t cnt0 = 0, before0 = 0, cnt1 = 0, before1 = 0;
        for (uint32_t j = 0; j < kept8; j += 8) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint64_t kj = s_key[j + u];
                const bool eq0 = kj == my0, eq1 = kj == my1;
                cnt0 += eq0 ? 1u : 0u; before0 += (eq0 && j + u < k0) ? 1u : 0u;
                cnt1 += eq1 ? 1u : 0u; before1 += (eq1 && j + u < k1) ? 1u : 0u;
            }
        }
        const bool head0 = valid0 && before0 == 0, head1 = valid1 && before1 == 0;
        const uint64_t m0 = __ballot(head0), m1 = __ballot(head1);
        // (q <= k: s_hkey / s_a are written behind what this sweep read; s_a is free since the runs)
        if (head0) { const uint32_t q = n_dist + (uint32_t)__popcll(m0 & below); s_hkey[q] = my0; s_a[q] = cnt0; }
        n_dist += (uint32_t)__popcll(m0);
        if (head1) { const uint32_t q = n_dist + (uint32_t)__popcll(m1 & below); s_hkey[q] = my1; s_a[q] = cnt1; }
        n_dist += (uint32_t)__popcll(m1);
    }
    __syncthreads();
    // ---- output order: class (the writer's aux of the first score character), count descending, first appearance
    for (uint32_t q = lane; q < n_dist + 8u; q += 64) {
        uint64_t ord = ~0ull;   // (sentinels: never below a real order value)
        if (q < n_dist) {
            const uint32_t s0 = (uint32_t)(s_hkey[q] & 0xffu), cnt = s_a[q];
            const uint64_t aux = s0 == 'N' ? 2u : s0 == 'E' ? 1u : 0u;
            ord = (aux << 56) | ((uint64_t)(0xffffffu - (cnt < 0xffffffu ? cnt : 0xffffffu)) << 32) | (uint64_t)q;
        }
        s_key[q] = ord;
    }
    __syncthreads();
    const uint32_t dist8 = (n_dist + 7u) & ~7u;
    uint32_t my_len = 0;
    for (uint32_t c0 = 0; c0 < n_dist; c0 += 128) {
        const uint32_t q0 = c0 + lane, q1 = c0 + 64 + lane;
        const uint64_t my0 = q0 < n_dist ? s_key[q0] : 0ull, my1 = q1 < n_dist ? s_key[q1] : 0ull;
        uint32_t rank0 = 0, rank1 = 0;
        for (uint32_t j = 0; j < dist8; j += 8) {
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint64_t oj = s_key[j + u];
                rank0 += oj < my0 ? 1u : 0u;
                rank1 += oj < my1 ? 1u : 0u;
            }
        }
        for (int h = 0; h < 2; ++h) {
            const uint32_t q = h ? q1 : q0, rank = h ? rank1 : rank0;
            if (q < n_dist) {
                const uint64_t key = s_hkey[q];
                item_key[b + rank] = key;
                item_cnt[b + rank] = s_a[q];
                my_len += d_item_len(key, s_a[q]);
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) my_len += (uint32_t)__shfl_xor((int)my_len, d);
    const uint32_t total = my_len;
    // ---- header: SAOBS / SROBS letters in first-appearance order (the host sorts and formats the at most twelve items)
    if (lane < 24) {
        const uint32_t side = lane < 12 ? 0u : 12u;
        const uint32_t myf = s_lf[lane];
        uint32_t pos = 0;
        for (uint32_t j = 0; j < 12; ++j) pos += (s_lc[side + j] > 0u && s_lf[side + j] < myf) ? 1u : 0u;
        if (s_lc[lane]) s_perm[side + pos] = lane - side;
    }
    __syncthreads();
    if (lane == 0) {
        PileSum h;
        memset(&h, 0, sizeof h);
        for (uint32_t j = 0; j < 12; ++j) { h.alt_n += s_lc[j] ? 1 : 0; h.ref_n += s_lc[12 + j] ? 1 : 0; }
        for (uint32_t t = 0; t < h.alt_n; ++t) { const uint32_t j = s_perm[t]; h.alt_letter[t] = (uint8_t)d_letter_char(j); h.alt_cnt[t] = s_lc[j]; }
        for (uint32_t t = 0; t < h.ref_n; ++t) { const uint32_t j = s_perm[12 + t]; h.ref_letter[t] = (uint8_t)d_letter_char(j); h.ref_cnt[t] = s_lc[12 + j]; }
        h.n_item = n_dist;   // (obs_off / obs_len / overflow: obs_write_kernel, once the text has its place)
        h.run_off = run_off; h.n_run = R; h.run0_pm = __uint_as_float(run0_pm); h.run0_len = run0_len; h.kept = kept;
        hdr[p] = h;
        text_len[p] = total;
    }
}

// text lengths -> offsets: lane = pileup / list, one bump of the cursor per wave.  len 0xffffffff: not formatted (left alone); a text
// that finds no room becomes 0xffffffff.  cursor[2] counts those.  Texts start on 16-byte steps.
__global__ __launch_bounds__(64) void span_alloc_kernel(uint32_t* __restrict__ len, uint32_t* __restrict__ off, int stride, int64_t n, uint32_t cap, uint32_t* __restrict__ cursor) {
    const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t l = p < n ? len[p * stride] : 0u;
    const bool skip = l == 0xffffffffu;
    const uint32_t need = (skip || l == 0u) ? 0u : (l + 15u) & ~15u;
    uint32_t inc = need;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, d);
        if ((int)lane >= d) inc += t;
    }
    const uint32_t sum = (uint32_t)__shfl((int)inc, 63);
    uint32_t base = 0;
    if (sum) {
        if (lane == 0) base = atomicAdd(&cursor[0], sum);
        base = (uint32_t)__shfl((int)base, 0);
    }
    const uint32_t at = base + inc - need;
    const bool full = !skip && need && (uint64_t)at + l > (uint64_t)cap;
    const uint64_t lost = __ballot(skip || full);
    if (lane == 0 && lost) atomicAdd(&cursor[2], (uint32_t)__popcll(lost));
    if (p < n) { off[p * stride] = (skip || full) ? 0u : at; if (full) len[p * stride] = 0xffffffffu; }
}

__global__ __launch_bounds__(64) void obs_write_kernel(const uint32_t* __restrict__ obs_offset, PileSum* __restrict__ hdr, const uint64_t* __restrict__ item_key,
                                                       const uint32_t* __restrict__ item_cnt, const uint32_t* __restrict__ text_len, const uint32_t* __restrict__ text_off,
                                                       uint8_t* __restrict__ text) {
    const int64_t p = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t len = text_len[p], off = text_off[p];
    if (len == 0xffffffffu) {   // more observations than the kernel ranks, or no room for the text: the columns
        if (lane == 0) { hdr[p].obs_off = 0u; hdr[p].obs_len = 0u; hdr[p].overflow = 1u; }
        return;
    }
    const uint32_t b = obs_offset[p], n_item = hdr[p].n_item;
    uint32_t done = 0;
    for (uint32_t c0 = 0; c0 < n_item; c0 += 64) {
        const uint32_t r = c0 + lane;
        const bool valid = r < n_item;
        const uint64_t key = valid ? item_key[b + r] : 0ull;
        const uint32_t cnt = valid ? item_cnt[b + r] : 0u;
        const uint32_t v = valid ? d_item_len(key, cnt) : 0u;
        uint32_t inc = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d);
            if ((int)lane >= d) inc += t;
        }
        if (valid) {
            uint8_t* o = text + off + done + inc - v;
            o = d_put_dec(o, cnt);
            *o++ = (uint8_t)(key & 0xffu);
            if ((key >> 8) & 0xffu) *o++ = (uint8_t)((key >> 8) & 0xffu);
            const uint32_t th = (uint32_t)(key >> 32);
            if (th) o = d_put_dec(o, th - 1u);
            else *o++ = '.';
            const uint32_t al = (uint32_t)(key >> 17) & 3u, st = (uint32_t)(key >> 19) & 3u, orr = (uint32_t)(key >> 21) & 3u;
            *o++ = ((key >> 16) & 1u) ? 'p' : 's';
            *o++ = al == 0 ? '#' : al == 1 ? '*' : '.';
            *o++ = st == 0 ? '+' : st == 1 ? '-' : st == 2 ? '*' : '.';
            *o++ = orr == 0 ? '>' : orr == 1 ? '<' : orr == 2 ? '*' : '!';
            *o++ = ((key >> 23) & 1u) ? '^' : '*';
            *o++ = ((key >> 24) & 1u) ? '$' : '.';
            *o++ = ((key >> 25) & 1u) ? '*' : '.';
        }
        done += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) { hdr[p].obs_off = off; hdr[p].obs_len = len; hdr[p].overflow = 0u; }
}


// ---- FORMAT/AFD text (Call::write_final_record, calling/variants/mod.rs:473-559; sample_fields of vlr_ingest.cpp): one wave per list —
// entries ranked by allele frequency (stable: rank = entries that are smaller, or equal and earlier), "%.3f=%.2f" of (vaf, PHRED) with
// append_fixed's arithmetic (x = v * 10^digits with its exact error from an fma: correctly rounded like printf, ties to even),
// lengths scanned in output order, bytes written.  inf / nan are printf's words; a finite value of 1e12 or more, a NaN allele frequency,
// more than kAfdMax entries or a full text buffer leave the list to the host (length 0xffffffff).
constexpr int kAfdMax = 1024;
struct FixedNum { uint64_t n; uint32_t kind; bool neg; };   // kind 0: digits of n; 1: inf; 2: nan; 3: not formatted here
__device__ __forceinline__ FixedNum d_fixed(double v, int digits) {
    FixedNum f;
    f.neg = (__double_as_longlong(v) < 0);
    f.n = 0;
    if (v != v) { f.kind = 2; return f; }
    const double a = fabs(v);
    if (isinf(a)) { f.kind = 1; return f; }
    if (!(a < 1e12)) { f.kind = 3; return f; }
    const double sc = digits == 3 ? 1000.0 : digits == 2 ? 100.0 : digits == 1 ? 10.0 : 1.0;
    const double x = a * sc, err = fma(a, sc, -x);   // a * sc = x + err exactly
    double k = floor(x);
    const double frac = x - k;
    bool up;
    if (frac > 0.5) up = true;
    else if (frac < 0.5) up = false;
    else up = err > 0.0 || (err == 0.0 && (((uint64_t)k) & 1ull));
    if (up) k += 1.0;
    f.n = (uint64_t)k;
    f.kind = 0;
    return f;
}
// (64-bit divisions are long instruction sequences on this target: the number of decimals is a template parameter — every division is
//  by a constant — and values below 2^32, all that occur in practice, take 32-bit arithmetic)
__device__ __forceinline__ uint32_t d_digits64(uint64_t v) {
    if ((v >> 32) == 0) return d_digits((uint32_t)v);
    uint32_t n = 10;   // (v >= 2^32 > 10^9)
    v /= 1000000000ull;
    while (v >= 10ull) { v /= 10ull; ++n; }
    return n;
}
template <int DIGITS>
__device__ __forceinline__ uint32_t d_fixed_len(const FixedNum& f) {
    if (f.kind == 1 || f.kind == 2) return 3u + (f.neg ? 1u : 0u);
    constexpr uint32_t P = DIGITS == 3 ? 1000u : DIGITS == 2 ? 100u : DIGITS == 1 ? 10u : 1u;
    const uint32_t ipd = (f.n >> 32) == 0 ? d_digits((uint32_t)f.n / P) : d_digits64(f.n / (uint64_t)P);
    return (f.neg ? 1u : 0u) + ipd + (DIGITS ? 1u + (uint32_t)DIGITS : 0u);
}
template <int DIGITS>
__device__ __forceinline__ uint8_t* d_put_fixed(uint8_t* o, const FixedNum& f) {
    if (f.neg) *o++ = '-';
    if (f.kind == 1) { o[0] = 'i'; o[1] = 'n'; o[2] = 'f'; return o + 3; }
    if (f.kind == 2) { o[0] = 'n'; o[1] = 'a'; o[2] = 'n'; return o + 3; }
    constexpr uint32_t P = DIGITS == 3 ? 1000u : DIGITS == 2 ? 100u : DIGITS == 1 ? 10u : 1u;
    uint32_t fr;
    if ((f.n >> 32) == 0) {
        const uint32_t n32 = (uint32_t)f.n;
        fr = n32 % P;
        o = d_put_dec(o, n32 / P);
    } else {
        uint64_t ip = f.n / (uint64_t)P;
        fr = (uint32_t)(f.n % (uint64_t)P);
        const uint32_t nd = d_digits64(ip);
        for (uint32_t i = nd; i-- > 0;) { o[i] = (uint8_t)('0' + (uint32_t)(ip % 10ull)); ip /= 10ull; }
        o += nd;
    }
    if (DIGITS) {
        *o++ = '.';
        for (int i = DIGITS; i-- > 0;) { o[i] = (uint8_t)('0' + fr % 10u); fr /= 10u; }
        o += DIGITS;
    }
    return o;
}

// afd_rank_kernel: output position of every entry (rank16[p * capacity + i]) and the byte length of the list's text (span[2 p + 1];
// 0xffffffff: left to the host); span_alloc_kernel places the texts; afd_write_kernel writes them.
__global__ __launch_bounds__(64) void afd_rank_kernel(const int32_t* __restrict__ count, const double* __restrict__ vaf, const double* __restrict__ lnprob, int capacity,
                                                      double ln10, uint16_t* __restrict__ rank16, uint32_t* __restrict__ span) {
    // dynamic LDS for min(capacity, kAfdMax) entries (rounded up to an even number by the launcher)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_afd[];
    double* s_v = (double*)s_afd;
    const int64_t p = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    int nn = count[p];
    nn = nn < capacity ? nn : capacity;
    const uint32_t n = nn > 0 ? (uint32_t)nn : 0u;
    const double* v = vaf + (size_t)p * (size_t)capacity;
    const double* lp = lnprob + (size_t)p * (size_t)capacity;
    bool host = n > (uint32_t)kAfdMax;
    uint32_t my_len = 0;
    if (!host) {
        for (uint32_t i = lane; i < n; i += 64) s_v[i] = v[i];
        __syncthreads();
        for (uint32_t i = lane; i < n; i += 64) {
            const double my = s_v[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; ++j) { const double o = s_v[j]; rank += (o < my || (o == my && j < i)) ? 1u : 0u; }
            const double ph = -10.0 * lp[i] / ln10 + 0.0;
            const FixedNum a = d_fixed(my, 3), b = d_fixed(ph, 2);
            if (my != my || a.kind == 3 || b.kind == 3) host = true;
            rank16[(size_t)p * (size_t)capacity + i] = (uint16_t)rank;
            my_len += (rank ? 1u : 0u) + d_fixed_len<3>(a) + 1u + d_fixed_len<2>(b);
        }
        host = __any(host) != 0;
    }
    for (int d = 32; d >= 1; d >>= 1) my_len += (uint32_t)__shfl_xor((int)my_len, d);
    if (lane == 0) { span[2 * p] = 0u; span[2 * p + 1] = host ? 0xffffffffu : my_len; }
}

__global__ __launch_bounds__(64) void afd_write_kernel(const int32_t* __restrict__ count, const double* __restrict__ vaf, const double* __restrict__ lnprob, int capacity,
                                                       double ln10, const uint16_t* __restrict__ rank16, const uint32_t* __restrict__ span, uint8_t* __restrict__ text) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_afd[];
    uint32_t* s_len = (uint32_t*)s_afd;    // lengths in output order -> offsets
    const int64_t p = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint32_t off = span[2 * p], len = span[2 * p + 1];
    if (len == 0xffffffffu || len == 0u) return;
    int nn = count[p];
    nn = nn < capacity ? nn : capacity;
    const uint32_t n = (uint32_t)nn;
    const double* v = vaf + (size_t)p * (size_t)capacity;
    const double* lp = lnprob + (size_t)p * (size_t)capacity;
    const uint16_t* rk = rank16 + (size_t)p * (size_t)capacity;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t rank = rk[i];
        s_len[rank] = (rank ? 1u : 0u) + d_fixed_len<3>(d_fixed(v[i], 3)) + 1u + d_fixed_len<2>(d_fixed(-10.0 * lp[i] / ln10 + 0.0, 2));
    }
    __syncthreads();
    uint32_t total = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += 64) {
        const uint32_t r = c0 + lane;
        const uint32_t x = r < n ? s_len[r] : 0u;
        uint32_t inc = x;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d);
            if ((int)lane >= d) inc += t;
        }
        if (r < n) s_len[r] = total + inc - x;
        total += (uint32_t)__shfl((int)inc, 63);
    }
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t rank = rk[i];
        uint8_t* o = text + off + s_len[rank];
        if (rank) *o++ = ',';
        o = d_put_fixed<3>(o, d_fixed(v[i], 3));
        *o++ = '=';
        o = d_put_fixed<2>(o, d_fixed(-10.0 * lp[i] / ln10 + 0.0, 2));
    }
}

}  // namespace
}  // namespace vlr

// ================================================================================================ one sample file on the device
struct vlr_dev_file {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t feed_stream = nullptr;   // inflate kernels (+ compaction copies): run beside the decode of the previous chunk
    hipStream_t up_stream = nullptr;     // H2D of the compressed members of the feeds
    hipStream_t copy_stream = nullptr;   // column copies the caller does not wait for (vlr_dev_file_copy_detached)
    // feeds in flight on the feed stream, oldest first (round 5: up to kFeedSlots, so that the inflate of request k + 2 is enqueued while
    // request k + 1 is still inflating and the reader never waits for more than the oldest one)
    static constexpr int kFeedSlots = 4;
    struct Feed {
        hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;   // around the inflate kernel (vlr_dev_file_inflate_seconds); status copy complete
        size_t wr_end = 0, n_blocks = 0;
        // member list up, member status down: PAGE-LOCKED — a copy to or from pageable memory is synchronous, and the status copy behind
        // the inflate kernel kept the enqueueing call waiting for the whole inflate (0.09 s of the reader's 0.14 s per 200 000 records)
        int* h_status = nullptr;
        vlr::InflateBlock* h_blocks = nullptr;
        size_t h_cap = 0;
        // the feed's own device buffers — compressed bytes, member list, member status — and the event behind its upload: the upload of
        // feed k + 1 runs on the upload stream WHILE feed k inflates (one shared buffer on one stream put upload, inflate, upload, inflate
        // in a row: 5.5 ms of PCIe and 6.8 ms of kernels per request of 32 768 tumor-normal records that never overlapped)
        uint8_t* d_comp = nullptr; size_t comp_cap = 0;
        vlr::InflateBlock* d_blocks = nullptr; int* d_status = nullptr; size_t blocks_cap = 0;
        hipEvent_t up = nullptr;
    };
    Feed feeds[kFeedSlots];
    int feed_head = 0, feed_n = 0;   // ring: slots [feed_head, feed_head + feed_n) are pending
    double inflate_s = 0.0;
    uint8_t* buf = nullptr;       // inflated stream: bytes [rd, wr) are buffered, [rd, ready) inflated and checked
    size_t cap = 0, rd = 0, wr = 0, ready = 0;
    uint8_t* spare = nullptr;     // the other half of the ping-pong (compaction never copies inside one allocation)
    size_t spare_cap = 0;
    // ordering of a compaction against the decode stream: the copy may only overwrite `spare` once the kernels that
    // still read it (it was `buf` one compaction ago) are through, and the kernels enqueued after it must see the copied bytes
    hipEvent_t ev_readers = nullptr, ev_compact = nullptr;
    bool compact_pending = false;
    // split
    uint64_t *d_anchor = nullptr, *d_landing = nullptr, *d_segbase = nullptr; uint32_t* d_count = nullptr; uint8_t* d_landc = nullptr; size_t seg_cap = 0;
    uint64_t* d_starts = nullptr; uint64_t* d_nout = nullptr; vlr::RecDesc* d_desc = nullptr; vlr::RecHost* d_host = nullptr; size_t rec_cap = 0;
    int8_t* d_fok = nullptr; int fok_n = 0;
    std::vector<uint64_t> h_starts;
    vlr::RecHost* h_host = nullptr; size_t h_host_cap = 0;   // page-locked
    uint64_t* d_cold_off = nullptr; uint8_t* d_cold = nullptr; size_t cold_off_cap = 0, cold_cap = 0;
    int64_t n_split = 0;
};

namespace {
std::mutex& park_mutex() { static std::mutex m; return m; }
std::vector<vlr_dev_file*>& parked() { static auto* v = new std::vector<vlr_dev_file*>(); return *v; }   // (never destroyed: no HIP calls at exit)
// streams, events and every device buffer of a reader object back to the runtime (objects that are not parked, failed creations, trim)
size_t dev_file_bytes(const vlr_dev_file* f) { size_t n = f->cap + f->spare_cap; for (auto& fd : f->feeds) n += fd.comp_cap; return n; }

void dev_file_free(vlr_dev_file* f) {
    (void)hipSetDevice(f->device);
    if (f->feed_stream) (void)hipStreamDestroy(f->feed_stream);
    if (f->up_stream) (void)hipStreamDestroy(f->up_stream);
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    if (f->ev_readers) (void)hipEventDestroy(f->ev_readers);
    if (f->ev_compact) (void)hipEventDestroy(f->ev_compact);
    for (auto& fd : f->feeds) {
        for (hipEvent_t e : {fd.ev0, fd.ev1, fd.done, fd.up})
            if (e) (void)hipEventDestroy(e);
        if (fd.h_status) (void)hipHostFree(fd.h_status);
        if (fd.h_blocks) (void)hipHostFree(fd.h_blocks);
        for (void* q : {(void*)fd.d_comp, (void*)fd.d_blocks, (void*)fd.d_status})
            if (q) (void)hipFree(q);
    }
    if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
    void* all[] = {f->buf, f->spare, f->d_anchor, f->d_landing, f->d_segbase, f->d_count, f->d_landc, f->d_starts, f->d_nout,
                   f->d_desc, f->d_host, f->d_fok, f->d_cold_off, f->d_cold};
    for (void* p : all)
        if (p) (void)hipFree(p);
    if (f->h_host) (void)hipHostFree(f->h_host);
    delete f;
}

template <typename T>
int dev_grow(T*& p, size_t& cap, size_t need, size_t slack_num = 5, size_t slack_den = 4) {
    if (need <= cap) return VLR_OK;
    const size_t ncap = need * slack_num / slack_den + 64;
    T* q = nullptr;
    if (hipMalloc(&q, ncap * sizeof(T)) != hipSuccess) return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of device memory (%s%lld bytes)", "", (long long)(ncap * sizeof(T)));
    if (p) (void)hipFree(p);
    p = q; cap = ncap;
    return VLR_OK;
}
}  // namespace

extern "C" {

int vlr_dev_file_create(int device, vlr_dev_file** out) {
    if (!out) return dfail(VLR_ERR_INVALID_ARGUMENT, "vlr_dev_file_create: null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return dfail(VLR_ERR_NO_DEVICE, "no HIP device (the device reader has no host fallback)");
    if (device < 0 || device >= n) return dfail(VLR_ERR_INVALID_ARGUMENT, "device index out of range");
    VLR_HIP_OK(hipSetDevice(device));
    {   // a parked object of an earlier reader: its streams, events and (grown) buffers are taken over
        std::lock_guard<std::mutex> g(park_mutex());
        auto& park = parked();
        for (size_t i = 0; i < park.size(); ++i)
            if (park[i]->device == device) {
                vlr_dev_file* f = park[i];
                park.erase(park.begin() + (long)i);
                f->rd = f->wr = f->ready = 0; f->feed_head = f->feed_n = 0; f->n_split = 0; f->inflate_s = 0.0; f->compact_pending = false;
                f->fok_n = -1;   // (the key table of the new file is uploaded at its first split)
                *out = f;
                return VLR_OK;
            }
    }
    vlr_dev_file* f = new vlr_dev_file();
    f->device = device;
    // the feed stream (upload + inflate of the NEXT request) yields to everything that works on the current chunk: the decode kernels
    // of this reader and the caller's evaluation wait for nobody behind a prefetch
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // (numerically: lowest priority first)
    if (hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, prio_hi) != hipSuccess || hipStreamCreateWithPriority(&f->feed_stream, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipStreamCreateWithPriority(&f->up_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) {
        (void)hipGetLastError();
        dev_file_free(f);   // (whichever of the two streams exists is destroyed with it)
        return dfail(VLR_ERR_HIP, "hipStreamCreate failed");
    }
    if (hipMalloc(&f->d_nout, 8) != hipSuccess) { (void)hipGetLastError(); dev_file_free(f); return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of device memory"); }
    *out = f;
    return VLR_OK;
}

void vlr_dev_file_destroy(vlr_dev_file* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->up_stream) (void)hipStreamSynchronize(f->up_stream);
    if (f->feed_stream) (void)hipStreamSynchronize(f->feed_stream);
    if (f->stream) (void)hipStreamSynchronize(f->stream);
    if (f->copy_stream) (void)hipStreamSynchronize(f->copy_stream);
    {   // parked for the next reader of this process: allocating and freeing half a gigabyte of device buffers per file costs
        // milliseconds and hipFree synchronises the device.  Bounded in count AND in bytes (VLR_INGEST_PARK_MB, default 2048): a
        // long-lived process does not sit on the inflate windows of every file it ever read; vlr_ingest_device_trim() returns the rest.
        std::lock_guard<std::mutex> g(park_mutex());
        size_t held = dev_file_bytes(f);
        for (vlr_dev_file* q : parked()) held += dev_file_bytes(q);
        size_t budget = (size_t)2048 << 20;
        if (const char* ev = getenv("VLR_INGEST_PARK_MB")) budget = (size_t)std::max(0L, atol(ev)) << 20;
        if (parked().size() < 8 && held <= budget) { parked().push_back(f); return; }
    }
    dev_file_free(f);
}

// every reader object parked by vlr_dev_file_destroy goes back to the device (vlr_ingest_device_trim, include/vlr.h)
void vlr_dev_file_trim() {
    std::vector<vlr_dev_file*> all;
    {
        std::lock_guard<std::mutex> g(park_mutex());
        all.swap(parked());
    }
    for (vlr_dev_file* f : all) dev_file_free(f);
}

double vlr_dev_file_inflate_seconds(vlr_dev_file* f, int reset) { if (!f) return 0.0; const double v = f->inflate_s; if (reset) f->inflate_s = 0.0; return v; }
uint64_t vlr_dev_file_buffered(const vlr_dev_file* f) { return f ? (uint64_t)(f->wr - f->rd) : 0; }
void* vlr_dev_file_stream(vlr_dev_file* f) { return f ? (void*)f->stream : nullptr; }
int vlr_dev_file_sync(vlr_dev_file* f) { VLR_HIP_OK(hipSetDevice(f->device)); VLR_HIP_OK(hipStreamSynchronize(f->stream)); return VLR_OK; }

// Enqueue on the feed stream: compressed members up, inflate behind the buffered bytes.  `comp` and `blocks` must stay valid until
// vlr_dev_file_feed_wait.  The buffered bytes [rd, wr) are only read by kernels already enqueued on the decode stream: compaction
// copies them into the other allocation (never inside one), so the feed may run beside those kernels.
int vlr_dev_file_feed(vlr_dev_file* f, const uint8_t* comp, size_t comp_bytes, const vlr::InflateBlock* blocks, int n_blocks, uint64_t inflated_bytes) {
    return vlr_dev_file_feed_pieces(f, &comp, &comp_bytes, 1, blocks, n_blocks, inflated_bytes);
}

// the same with the compressed bytes in n_pieces host pieces that are uploaded one behind the other (the segments of a page-locked
// staging ring: the copies are DMA from there and the call returns at once; from pageable memory the calling thread stages them)
int vlr_dev_file_feed_pieces(vlr_dev_file* f, const uint8_t* const* piece, const size_t* piece_bytes, int n_pieces, const vlr::InflateBlock* blocks, int n_blocks,
                             uint64_t inflated_bytes) {
    if (!f || n_blocks <= 0) return VLR_OK;
    size_t comp_bytes = 0;
    for (int i = 0; i < n_pieces; ++i) comp_bytes += piece_bytes[i];
    VLR_HIP_OK(hipSetDevice(f->device));
    if (f->feed_n == vlr_dev_file::kFeedSlots) { const int rc = vlr_dev_file_feed_wait_oldest(f); if (rc != VLR_OK) return rc; }
    hipStream_t st = f->feed_stream;
    const size_t live = f->wr - f->rd;
    if (f->wr + inflated_bytes + 64 > f->cap) {   // compact into the other buffer (grown if needed)
        const size_t need = live + (size_t)inflated_bytes + 64;
        if (need > f->spare_cap) {
            VLR_HIP_OK(hipStreamSynchronize(f->stream));   // (kernels of two chunks ago may still read the allocation that is replaced)
            if (f->spare) (void)hipFree(f->spare);
            f->spare = nullptr; f->spare_cap = 0;
            const size_t ncap = need + need / 4 + (1u << 20);
            if (hipMalloc(&f->spare, ncap) != hipSuccess) return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of device memory (%s%lld bytes)", "", (long long)ncap);
            f->spare_cap = ncap;
        }
        if (!f->ev_readers) { (void)hipEventCreateWithFlags(&f->ev_readers, hipEventDisableTiming); (void)hipEventCreateWithFlags(&f->ev_compact, hipEventDisableTiming); }
        if (!f->ev_readers || !f->ev_compact) return dfail(VLR_ERR_HIP, "device reader: hipEventCreate failed%s%lld", "", 0LL);
        // (a second compaction within one call, or a feed cut into several pieces, reaches this point while split / decode kernels of
        //  the chunk in front are still enqueued on the decode stream: they read what is about to become the copy's target)
        VLR_HIP_OK(hipEventRecord(f->ev_readers, f->stream));
        VLR_HIP_OK(hipStreamWaitEvent(st, f->ev_readers, 0));
        if (live) VLR_HIP_OK(hipMemcpyAsync(f->spare, f->buf + f->rd, live, hipMemcpyDeviceToDevice, st));
        VLR_HIP_OK(hipEventRecord(f->ev_compact, st));
        f->compact_pending = true;   // the next kernels on the decode stream wait for the copy (vlr_dev_file_split)
        std::swap(f->buf, f->spare); std::swap(f->cap, f->spare_cap);
        // (positions move with the bytes: the copy runs behind the pending inflates on the feed stream, which wrote the old positions)
        for (int i = 0; i < f->feed_n; ++i) f->feeds[(f->feed_head + i) % vlr_dev_file::kFeedSlots].wr_end -= f->rd;
        f->ready -= f->rd;
        f->rd = 0; f->wr = live;
    }
    vlr_dev_file::Feed& fd = f->feeds[(f->feed_head + f->feed_n) % vlr_dev_file::kFeedSlots];
    if ((size_t)n_blocks > fd.h_cap) {
        if (fd.h_status) (void)hipHostFree(fd.h_status);
        if (fd.h_blocks) (void)hipHostFree(fd.h_blocks);
        fd.h_status = nullptr; fd.h_blocks = nullptr; fd.h_cap = 0;
        const size_t ncap = (size_t)n_blocks + (size_t)n_blocks / 2 + 256;
        if (hipHostMalloc(&fd.h_status, ncap * sizeof(int), hipHostMallocDefault) != hipSuccess || hipHostMalloc(&fd.h_blocks, ncap * sizeof(vlr::InflateBlock), hipHostMallocDefault) != hipSuccess)
            return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of page-locked memory%s%lld", "", 0LL);
        fd.h_cap = ncap;
    }
    memcpy(fd.h_blocks, blocks, (size_t)n_blocks * sizeof(vlr::InflateBlock));
    {   // the slot's device buffers (free: its last feed has been waited for); the kernel may read kInflateInputSlack bytes beyond the last member
        int rc = dev_grow(fd.d_comp, fd.comp_cap, comp_bytes + vlr::kInflateInputSlack);
        if (rc) return rc;
        if ((size_t)n_blocks > fd.blocks_cap) {
            size_t c1 = fd.blocks_cap, c2 = fd.blocks_cap;
            if ((rc = dev_grow(fd.d_blocks, c1, (size_t)n_blocks))) return rc;
            if ((rc = dev_grow(fd.d_status, c2, (size_t)n_blocks))) return rc;
            fd.blocks_cap = c1 < c2 ? c1 : c2;
        }
    }
    if (!fd.ev0) { (void)hipEventCreate(&fd.ev0); (void)hipEventCreate(&fd.ev1); (void)hipEventCreateWithFlags(&fd.done, hipEventDisableTiming); (void)hipEventCreateWithFlags(&fd.up, hipEventDisableTiming); }
    if (!fd.done || !fd.up) return dfail(VLR_ERR_HIP, "device reader: hipEventCreate failed%s%lld", "", 0LL);
    {   // upload on its own stream: beside the inflate kernels of the feeds in front of this one
        hipStream_t us = f->up_stream;
        size_t at = 0;
        for (int i = 0; i < n_pieces; ++i) {
            if (piece_bytes[i]) VLR_HIP_OK(hipMemcpyAsync(fd.d_comp + at, piece[i], piece_bytes[i], hipMemcpyHostToDevice, us));
            at += piece_bytes[i];
        }
        VLR_HIP_OK(hipMemsetAsync(fd.d_comp + comp_bytes, 0, vlr::kInflateInputSlack, us));
        VLR_HIP_OK(hipMemcpyAsync(fd.d_blocks, fd.h_blocks, (size_t)n_blocks * sizeof(vlr::InflateBlock), hipMemcpyHostToDevice, us));
        VLR_HIP_OK(hipEventRecord(fd.up, us));
        VLR_HIP_OK(hipStreamWaitEvent(st, fd.up, 0));
    }
    if (fd.ev0) (void)hipEventRecord(fd.ev0, st);
    const int lrc = vlr_launch_inflate_kernel(fd.d_comp, fd.d_blocks, n_blocks, f->buf + f->wr, fd.d_status, st);
    if (lrc != 0) return dfail(VLR_ERR_HIP, "inflate kernel launch failed (hip error %s%lld)", "", lrc);
    if (fd.ev1) (void)hipEventRecord(fd.ev1, st);
    VLR_HIP_OK(hipMemcpyAsync(fd.h_status, fd.d_status, (size_t)n_blocks * sizeof(int), hipMemcpyDeviceToHost, st));
    VLR_HIP_OK(hipEventRecord(fd.done, st));
    fd.n_blocks = (size_t)n_blocks;
    f->wr += (size_t)inflated_bytes;
    fd.wr_end = f->wr;
    f->feed_n += 1;
    return VLR_OK;
}

// the oldest feed in flight: its bytes are inflated and checked afterwards
int vlr_dev_file_feed_wait_oldest(vlr_dev_file* f) {
    if (!f || f->feed_n == 0) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    vlr_dev_file::Feed& fd = f->feeds[f->feed_head];
    VLR_HIP_OK(hipEventSynchronize(fd.done));
    f->feed_head = (f->feed_head + 1) % vlr_dev_file::kFeedSlots;
    f->feed_n -= 1;
    f->ready = fd.wr_end;
    if (fd.ev0 && fd.ev1) { float ms = 0.0f; if (hipEventElapsedTime(&ms, fd.ev0, fd.ev1) == hipSuccess) f->inflate_s += (double)ms * 1e-3; }
    for (size_t i = 0; i < fd.n_blocks; ++i)
        if (fd.h_status[i] != 0) return dfail(VLR_ERR_INVALID_ARGUMENT, "%s in a BGZF member (inflate status %lld)", fd.h_status[i] == vlr::INFL_CRC_MISMATCH ? "CRC32 checksum mismatch" : "corrupt DEFLATE stream", (long long)fd.h_status[i]);
    return VLR_OK;
}

// every feed in flight
int vlr_dev_file_feed_wait(vlr_dev_file* f) {
    if (!f) return VLR_OK;
    while (f->feed_n > 0) { const int rc = vlr_dev_file_feed_wait_oldest(f); if (rc != VLR_OK) return rc; }
    return VLR_OK;
}

// feeds, oldest first, until `bytes` behind the read position are inflated (or nothing is in flight any more)
int vlr_dev_file_wait_ready(vlr_dev_file* f, uint64_t bytes) {
    if (!f) return VLR_OK;
    while (f->feed_n > 0 && (uint64_t)(f->ready - f->rd) < bytes) { const int rc = vlr_dev_file_feed_wait_oldest(f); if (rc != VLR_OK) return rc; }
    return VLR_OK;
}
int vlr_dev_file_feeds_in_flight(const vlr_dev_file* f) { return f ? f->feed_n : 0; }
uint64_t vlr_dev_file_ready(const vlr_dev_file* f) { return f ? (uint64_t)(f->ready - f->rd) : 0; }

int vlr_dev_file_skip(vlr_dev_file* f, uint64_t bytes) {
    if (f->rd + bytes > f->wr) return dfail(VLR_ERR_INVALID_ARGUMENT, "device reader: skip beyond the buffered bytes");
    f->rd += (size_t)bytes;
    if (f->ready < f->rd) f->ready = f->rd;   // (skipped before its feed completed: nothing of it is read)
    return VLR_OK;
}

int vlr_dev_file_split(vlr_dev_file* f, int64_t max_records, int n_contigs, int n_hdr_samples, const int8_t* field_of_key, int n_keys, int64_t* n_records,
                       const vlr::RecHost** rec_host, int* used_serial_walk) {
    VLR_HIP_OK(hipSetDevice(f->device));
    // (the inflated and checked bytes only: the caller waits for as many feeds as it wants — vlr_dev_file_wait_ready — and at least one
    //  here when nothing is ready)
    { const int rcw = vlr_dev_file_wait_ready(f, 8); if (rcw != VLR_OK) return rcw; }
    *n_records = 0; *rec_host = nullptr;
    if (used_serial_walk) *used_serial_walk = 0;
    f->n_split = 0;
    const uint64_t avail = f->ready - f->rd;
    if (avail < 8 || max_records <= 0) return VLR_OK;
    if (f->compact_pending) {   // bytes that were buffered before a compaction reach this allocation by a copy on the feed stream
        VLR_HIP_OK(hipStreamWaitEvent(f->stream, f->ev_compact, 0));
        f->compact_pending = false;
    }
    const uint8_t* base = f->buf + f->rd;
    const int n_seg = (int)((avail + vlr::kSeg - 1) / vlr::kSeg);
    int rc;
    if ((size_t)n_seg > f->seg_cap) {
        VLR_HIP_OK(hipStreamSynchronize(f->stream));
        size_t c[5] = {f->seg_cap, f->seg_cap, f->seg_cap, f->seg_cap, f->seg_cap};
        if ((rc = dev_grow(f->d_anchor, c[0], (size_t)n_seg)) || (rc = dev_grow(f->d_landing, c[1], (size_t)n_seg)) || (rc = dev_grow(f->d_segbase, c[2], (size_t)n_seg)) ||
            (rc = dev_grow(f->d_count, c[3], (size_t)n_seg)) || (rc = dev_grow(f->d_landc, c[4], (size_t)n_seg))) return rc;
        f->seg_cap = c[0];
    }
    if ((size_t)max_records + 1 > f->rec_cap) {
        VLR_HIP_OK(hipStreamSynchronize(f->stream));
        size_t c[3] = {f->rec_cap, f->rec_cap, f->rec_cap};
        if ((rc = dev_grow(f->d_starts, c[0], (size_t)max_records + 1)) || (rc = dev_grow(f->d_desc, c[1], (size_t)max_records + 1)) || (rc = dev_grow(f->d_host, c[2], (size_t)max_records + 1))) return rc;
        f->rec_cap = c[0];
    }
    if ((size_t)max_records > f->h_host_cap) {
        if (f->h_host) (void)hipHostFree(f->h_host);
        f->h_host = nullptr; f->h_host_cap = 0;
        const size_t ncap = (size_t)max_records + (size_t)max_records / 4 + 64;
        if (hipHostMalloc(&f->h_host, ncap * sizeof(vlr::RecHost), hipHostMallocDefault) != hipSuccess) return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of page-locked memory");
        f->h_host_cap = ncap;
    }
    if (f->fok_n != n_keys || !f->d_fok) {
        VLR_HIP_OK(hipStreamSynchronize(f->stream));
        if (f->d_fok) (void)hipFree(f->d_fok);
        f->d_fok = nullptr;
        VLR_HIP_OK(hipMalloc(&f->d_fok, (size_t)(n_keys > 0 ? n_keys : 1)));
        if (n_keys > 0) VLR_HIP_OK(hipMemcpy(f->d_fok, field_of_key, (size_t)n_keys, hipMemcpyHostToDevice));
        f->fok_n = n_keys;
    }
    // ---- record starts
    hipLaunchKernelGGL(vlr::rec_anchor_kernel, dim3((unsigned)n_seg), dim3(64), 0, f->stream, base, avail, n_seg, n_contigs, n_hdr_samples, f->d_anchor);
    hipLaunchKernelGGL(vlr::rec_walk_kernel, dim3((unsigned)((n_seg + 63) / 64)), dim3(64), 0, f->stream, base, avail, n_seg, f->d_anchor, f->d_count, f->d_landing, f->d_landc);
    std::vector<uint64_t> anchor((size_t)n_seg), landing((size_t)n_seg), segbase((size_t)n_seg);
    std::vector<uint32_t> count((size_t)n_seg);
    std::vector<uint8_t> landc((size_t)n_seg);
    VLR_HIP_OK(hipMemcpyAsync(anchor.data(), f->d_anchor, (size_t)n_seg * 8, hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipMemcpyAsync(landing.data(), f->d_landing, (size_t)n_seg * 8, hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipMemcpyAsync(count.data(), f->d_count, (size_t)n_seg * 4, hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipMemcpyAsync(landc.data(), f->d_landc, (size_t)n_seg, hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipStreamSynchronize(f->stream));
    bool verified = true;
    uint64_t total = 0;
    int last_seg = -1;   // last segment whose records count
    for (int i = 0; i < n_seg; ++i) {
        if (i > 0 && anchor[(size_t)i] != landing[(size_t)i - 1]) {
            if (!landc[(size_t)i - 1]) break;   // the record at the previous landing is not complete: the buffered records end there
            verified = false;
            break;
        }
        segbase[(size_t)i] = total;
        total += count[(size_t)i];
        last_seg = i;
        if (total >= (uint64_t)max_records) break;
    }
    const char* force = getenv("VLR_INGEST_SERIAL_WALK");
    if (force && atoi(force) != 0) verified = false;
    uint64_t n = 0;
    if (verified) {
        n = total < (uint64_t)max_records ? total : (uint64_t)max_records;
        const int used = last_seg + 1;
        if (n > 0) {
            VLR_HIP_OK(hipMemcpyAsync(f->d_segbase, segbase.data(), (size_t)used * 8, hipMemcpyHostToDevice, f->stream));
            hipLaunchKernelGGL(vlr::rec_starts_kernel, dim3((unsigned)((used + 63) / 64)), dim3(64), 0, f->stream, base, avail, used, f->d_anchor, f->d_segbase, f->d_count, n, f->d_starts);
        }
    } else {
        if (used_serial_walk) *used_serial_walk = 1;
        hipLaunchKernelGGL(vlr::rec_walk_serial_kernel, dim3(1), dim3(64), 0, f->stream, base, avail, (uint64_t)max_records, f->d_starts, f->d_nout);
        VLR_HIP_OK(hipMemcpyAsync(&n, f->d_nout, 8, hipMemcpyDeviceToHost, f->stream));
        VLR_HIP_OK(hipStreamSynchronize(f->stream));
    }
    if (n == 0) return VLR_OK;
    // ---- INFO scan
    hipLaunchKernelGGL(vlr::rec_scan_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, f->stream, base, f->d_starts, (int64_t)n, f->d_fok, n_keys, f->d_desc, f->d_host);
    f->h_starts.resize((size_t)n + 1);
    VLR_HIP_OK(hipMemcpyAsync(f->h_starts.data(), f->d_starts, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipMemcpyAsync(f->h_host, f->d_host, (size_t)n * sizeof(vlr::RecHost), hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipStreamSynchronize(f->stream));
    VLR_HIP_OK(hipGetLastError());
    f->n_split = (int64_t)n;
    *n_records = (int64_t)n;
    *rec_host = f->h_host;
    return VLR_OK;
}

// The buffered bytes begin inside a record (the window of a file's shard): move the read position to the first record start.
// *skipped: the bytes in front of it; VLR_ERR_INVALID_ARGUMENT when no start is found in the buffered bytes.
int vlr_dev_file_anchor_first(vlr_dev_file* f, int n_contigs, int n_hdr_samples, uint64_t* skipped) {
    VLR_HIP_OK(hipSetDevice(f->device));
    { const int rcw = vlr_dev_file_feed_wait(f); if (rcw != VLR_OK) return rcw; }
    const uint64_t avail = f->wr - f->rd;
    uint64_t found = vlr::kNone;
    if (avail >= 33) {
        hipLaunchKernelGGL(vlr::rec_first_kernel, dim3(1), dim3(64), 0, f->stream, f->buf + f->rd, avail, n_contigs, n_hdr_samples, (uint64_t*)f->d_nout);
        VLR_HIP_OK(hipMemcpyAsync(&found, f->d_nout, 8, hipMemcpyDeviceToHost, f->stream));
        VLR_HIP_OK(hipStreamSynchronize(f->stream));
    }
    if (found == vlr::kNone) return dfail(VLR_ERR_INVALID_ARGUMENT, "device reader: no record start in the shard's window");
    f->rd += (size_t)found;
    if (skipped) *skipped = found;
    return VLR_OK;
}
// record starts of the last split (n + 1 offsets from the read position: the last one is the end of the last complete record)
const uint64_t* vlr_dev_file_starts(const vlr_dev_file* f, int64_t* n) {
    if (n) *n = f->n_split;
    return f->h_starts.data();
}

int vlr_dev_file_decode(vlr_dev_file* f, int64_t n, const uint32_t* d_obs_offset, int n_samples, int sample, const vlr::DeviceCols* cols) {
    if (n <= 0) return VLR_OK;
    if (n > f->n_split) return dfail(VLR_ERR_INVALID_ARGUMENT, "device reader: decode beyond the split records");
    VLR_HIP_OK(hipSetDevice(f->device));
    hipLaunchKernelGGL(vlr::rec_decode_kernel, dim3((unsigned)n), dim3(64), 0, f->stream, f->buf + f->rd, f->d_desc, n, d_obs_offset, n_samples, sample, *cols, f->d_host);
    VLR_HIP_OK(hipGetLastError());
    return VLR_OK;
}

int vlr_dev_file_cold(vlr_dev_file* f, int64_t n, const uint64_t* cold_off, uint8_t* host_out) {
    if (n <= 0) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    const uint64_t total = cold_off[n];
    if ((size_t)n + 1 > f->cold_off_cap || (size_t)total > f->cold_cap) VLR_HIP_OK(hipStreamSynchronize(f->stream));
    int rc;
    if ((rc = dev_grow(f->d_cold_off, f->cold_off_cap, (size_t)n + 1)) || (rc = dev_grow(f->d_cold, f->cold_cap, (size_t)total + 64))) return rc;
    VLR_HIP_OK(hipMemcpyAsync(f->d_cold_off, cold_off, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, f->stream));
    hipLaunchKernelGGL(vlr::rec_cold_kernel, dim3((unsigned)n), dim3(64), 0, f->stream, f->buf + f->rd, f->d_desc, n, f->d_cold_off, f->d_cold);
    VLR_HIP_OK(hipMemcpyAsync(host_out, f->d_cold, (size_t)total, hipMemcpyDeviceToHost, f->stream));
    return VLR_OK;
}

int vlr_dev_file_errors(vlr_dev_file* f, int64_t n, uint32_t* status_or, int64_t* first_bad) {
    *status_or = 0; *first_bad = -1;
    if (n <= 0) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipMemcpyAsync(f->h_host, f->d_host, (size_t)n * sizeof(vlr::RecHost), hipMemcpyDeviceToHost, f->stream));
    VLR_HIP_OK(hipStreamSynchronize(f->stream));
    for (int64_t r = 0; r < n; ++r)
        if (f->h_host[r].status) { *status_or |= f->h_host[r].status; if (*first_bad < 0) *first_bad = r; }
    return VLR_OK;
}

int vlr_dev_file_copy(vlr_dev_file* f, void* dst, const void* src, size_t bytes, int to_device) {
    if (!bytes) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, f->stream));
    return VLR_OK;
}

int vlr_dev_file_summaries(vlr_dev_file* f, const vlr::DeviceCols* cols, const uint32_t* d_obs_offset, const uint8_t* d_locus_flags, int64_t n_loci, int n_samples,
                           uint32_t max_pileup_obs, const vlr::SumConsts* k, vlr::PileSum* d_hdr, uint8_t* d_text, uint32_t text_cap, float* d_run_pm,
                           uint32_t* d_run_len, uint64_t* d_item_key, uint32_t* d_item_cnt, uint32_t* d_text_len, uint32_t* d_text_off, uint32_t* d_cursor) {
    const int64_t P = n_loci * n_samples;
    if (P <= 0) return VLR_OK;
    if (P > 0x7fffffff) return VLR_ERR_INVALID_ARGUMENT;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipMemsetAsync(d_cursor, 0, 16, f->stream));
    // LDS per wave: 24 bytes per observation of the largest pileup (larger ones than kSumMaxObs are left to the columns) + the letter tables
    const uint32_t lds_obs = (std::min<uint32_t>(std::max<uint32_t>(max_pileup_obs, 1u), (uint32_t)vlr::kSumMaxObs) + 8u + 7u) & ~7u;
    const size_t lds = (size_t)lds_obs * 24 + 72 * 4;
    hipLaunchKernelGGL(vlr::obs_text_kernel, dim3((unsigned)P), dim3(64), lds, f->stream, *cols, d_obs_offset, d_locus_flags, n_samples, *k,
                       d_hdr, d_item_key, d_item_cnt, d_text_len, d_run_pm, d_run_len, d_cursor, lds_obs);
    hipLaunchKernelGGL(vlr::span_alloc_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, f->stream, d_text_len, d_text_off, 1, P, text_cap, d_cursor);
    hipLaunchKernelGGL(vlr::obs_write_kernel, dim3((unsigned)P), dim3(64), 0, f->stream, d_obs_offset, d_hdr, d_item_key, d_item_cnt, d_text_len, d_text_off, d_text);
    VLR_HIP_OK(hipGetLastError());
    return VLR_OK;
}

int vlr_launch_afd_text(const int32_t* d_count, const double* d_vaf, const double* d_lnprob, int64_t n_lists, int capacity, uint8_t* d_text, uint32_t text_cap,
                        uint32_t* d_span, uint16_t* d_rank, uint32_t* d_cursor, void* stream) {
    if (n_lists <= 0) return VLR_OK;
    if (n_lists > 0x7fffffff || capacity < 1) return VLR_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    VLR_HIP_OK(hipMemsetAsync(d_cursor, 0, 16, st));
    const uint32_t lds_n = ((uint32_t)std::min(capacity, vlr::kAfdMax) + 1u) & ~1u;
    const double ln10 = std::log(10.0);
    hipLaunchKernelGGL(vlr::afd_rank_kernel, dim3((unsigned)n_lists), dim3(64), (size_t)lds_n * 8, st, d_count, d_vaf, d_lnprob, capacity, ln10, d_rank, d_span);
    hipLaunchKernelGGL(vlr::span_alloc_kernel, dim3((unsigned)((n_lists + 63) / 64)), dim3(64), 0, st, d_span + 1, d_span, 2, n_lists, text_cap, d_cursor);
    hipLaunchKernelGGL(vlr::afd_write_kernel, dim3((unsigned)n_lists), dim3(64), (size_t)lds_n * 4, st, d_count, d_vaf, d_lnprob, capacity, ln10, d_rank, d_span, d_text);
    VLR_HIP_OK(hipGetLastError());
    return VLR_OK;
}

int vlr_dev_copy_to_host(int device, void* dst, const void* src, size_t bytes) {
    if (!bytes) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(device));
    VLR_HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return VLR_OK;
}

// device -> host copy on the file's copy stream, NOT waited for by vlr_dev_file_sync: *event_out (owned by the caller:
// vlr_dev_event_wait / vlr_dev_event_destroy) completes when the bytes are there.  The source must be final (the caller synchronised the
// streams that wrote it).
int vlr_dev_file_copy_detached(vlr_dev_file* f, void* dst, const void* src, size_t bytes, void** event_out) {
    *event_out = nullptr;
    VLR_HIP_OK(hipSetDevice(f->device));
    if (!f->copy_stream) VLR_HIP_OK(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    hipEvent_t ev = nullptr;
    VLR_HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (bytes) VLR_HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, f->copy_stream));
    VLR_HIP_OK(hipEventRecord(ev, f->copy_stream));
    *event_out = (void*)ev;
    return VLR_OK;
}
// more bytes of the same group: enqueued on the copy stream without an event of their own (a vlr_dev_file_copy_detached behind them marks
// the end of the group)
int vlr_dev_file_copy_more(vlr_dev_file* f, void* dst, const void* src, size_t bytes) {
    VLR_HIP_OK(hipSetDevice(f->device));
    if (!f->copy_stream) VLR_HIP_OK(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    if (bytes) VLR_HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, f->copy_stream));
    return VLR_OK;
}
// what the copy stream has been given so far, as a mark the file's main stream can wait for before it overwrites the sources (*mark is
// created on first use and owned by the caller: vlr_dev_event_destroy)
int vlr_dev_file_copy_mark(vlr_dev_file* f, void** mark) {
    VLR_HIP_OK(hipSetDevice(f->device));
    if (!f->copy_stream) VLR_HIP_OK(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    if (!*mark) { hipEvent_t ev = nullptr; VLR_HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); *mark = (void*)ev; }
    VLR_HIP_OK(hipEventRecord((hipEvent_t)*mark, f->copy_stream));
    return VLR_OK;
}
int vlr_dev_file_wait_mark(vlr_dev_file* f, void* mark) {
    if (!mark) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipStreamWaitEvent(f->stream, (hipEvent_t)mark, 0));
    return VLR_OK;
}
int vlr_dev_event_wait(int device, void* event) {
    if (!event) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(device));
    VLR_HIP_OK(hipEventSynchronize((hipEvent_t)event));
    return VLR_OK;
}
void vlr_dev_event_destroy(int device, void* event) {
    if (!event) return;
    (void)hipSetDevice(device);
    (void)hipEventSynchronize((hipEvent_t)event);
    (void)hipEventDestroy((hipEvent_t)event);
}

int vlr_dev_slab_alloc(int device, size_t bytes, void** d, void** h) {
    *d = nullptr;
    if (h) *h = nullptr;
    VLR_HIP_OK(hipSetDevice(device));
    if (hipMalloc(d, bytes) != hipSuccess) return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of device memory (%s%lld bytes)", "", (long long)bytes);
    if (h == nullptr) return VLR_OK;   // device side only
    if (hipHostMalloc(h, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipFree(*d); *d = nullptr; return dfail(VLR_ERR_OUT_OF_MEMORY, "device reader: out of page-locked memory (%s%lld bytes)", "", (long long)bytes); }
    return VLR_OK;
}
void vlr_dev_slab_free(int device, void* d, void* h) {
    (void)hipSetDevice(device);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
}

int vlr_dev_file_consume(vlr_dev_file* f, int64_t n) {
    if (n < 0 || n > f->n_split) return dfail(VLR_ERR_INVALID_ARGUMENT, "device reader: consume beyond the split records");
    if (n > 0) f->rd += (size_t)f->h_starts[(size_t)n];
    f->n_split = 0;
    return VLR_OK;
}

}  // extern "C"
