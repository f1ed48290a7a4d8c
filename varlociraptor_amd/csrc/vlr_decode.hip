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

#include <cstdio>
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
    // ---- chains
    // lane -> (field, kind): kind 0 MiniLogProb, 1 Option<MiniLogProb>, 2 Option<u8>, 3 Option<u32>
    if (lane < 11) {
        const int field = lane == 0 ? FD_PROB_MAPPING : lane == 1 ? FD_PROB_ALT : lane == 2 ? FD_PROB_REF : lane == 3 ? FD_PROB_MISSED : lane == 4 ? FD_PROB_SAMPLE_ALT
                        : lane == 5 ? FD_PROB_DOUBLE_OVERLAP : lane == 6 ? FD_PROB_HIT_BASE : lane == 7 ? FD_HP_ART : lane == 8 ? FD_HP_VAR : lane == 9 ? FD_HP_LEN : FD_THIRD;
        const int kind = lane < 7 ? 0 : lane < 9 ? 1 : lane == 9 ? 2 : 3;
        const int stride = d.vstride[field];
        const uint8_t* v = rec + d.voff[field];
        const uint32_t nbytes = stride ? 2u * d.vn[field] : 0u;
        uint32_t* out = lane < 9 ? reinterpret_cast<uint32_t*>(cols.col[lane]) + ob : lane == 9 ? cols.flags + ob : reinterpret_cast<uint32_t*>(cols.third) + ob;
        const uint32_t none = kind <= 1 ? 0x7fc00000u : kind == 2 ? 0u : 0xffffffffu;
        if (stride == 0) {   // absent optional field: None everywhere (the mandatory ones were checked by the scan)
            for (uint32_t i = 0; i < n; ++i) out[i] = none;
        } else if (vlen(v, stride, d.vn[field]) != (uint64_t)n) {
            status |= REC_BAD_LENGTHS;
        } else {
            uint32_t j = 8;
            for (uint32_t i = 0; i < n; ++i) {
                // bytes j .. j + 8 of the little-endian word stream (zero beyond its end)
                uint64_t lo;
                uint32_t b8;
                if (stride == 4) {          // u16 words in the low halves of five consecutive int32 elements: one 16-byte and one 4-byte load
                    const uint8_t* q = v + 4 * (size_t)(j >> 1);
                    const uint64_t a = ld64(q), c = ld64(q + 8);
                    const uint32_t w4 = ld32(q + 16);
                    const uint64_t W = (a & 0xffffull) | ((a >> 16) & 0xffff0000ull) | ((c & 0xffffull) << 32) | ((c >> 32 & 0xffffull) << 48);
                    lo = (j & 1u) ? (W >> 8) | ((uint64_t)(w4 & 0xffu) << 56) : W;
                    b8 = (j & 1u) ? (w4 >> 8) & 0xffu : w4 & 0xffu;
                } else if (stride == 2) {   // the words are the bytes
                    lo = ld64(v + j);
                    b8 = v[j + 8];
                } else {
                    lo = 0;
                    for (int t = 0; t < 8; ++t) lo |= (uint64_t)vbyte(v, stride, j + (uint32_t)t) << (8 * t);
                    b8 = vbyte(v, stride, j + 8);
                }
                {
                    const uint32_t left = nbytes - j;   // (j < nbytes here: the previous element ended inside the stream, or n would be 0)
                    if (left < 8) lo &= (1ull << (8 * left)) - 1ull;
                    if (left < 9) b8 = 0;
                }
                uint32_t b[9];
#pragma unroll
                for (int t = 0; t < 8; ++t) b[t] = (uint32_t)(lo >> (8 * t)) & 0xffu;
                b[8] = b8;
                const uint32_t o = kind == 0 ? 0u : 1u;
                const bool some = kind == 0 ? true : b[0] != 0;
                uint32_t size, val = none;
                if (kind <= 1) {
                    const uint32_t tag = b[o] | (b[o + 1] << 8) | (b[o + 2] << 16) | (b[o + 3] << 24);
                    size = o + (some ? 4u + (tag == 0 ? 2u : 4u) : 0u);
                    if (some) {
                        if (tag > 1) status |= REC_BAD_VECTOR;
                        val = tag == 0 ? half_bits(b[o + 4] | (b[o + 5] << 8)) : (b[o + 4] | (b[o + 5] << 8) | (b[o + 6] << 16) | (b[o + 7] << 24));
                    }
                } else if (kind == 2) {
                    size = 1u + (some ? 1u : 0u);
                    if (some) val = VLR_F_HP_LEN_VALID | (b[1] << VLR_F_HP_LEN_SHIFT);
                } else {
                    size = 1u + (some ? 4u : 0u);
                    if (some) val = b[1] | (b[2] << 8) | (b[3] << 16) | (b[4] << 24);
                }
                if (j + size > nbytes) { status |= REC_BAD_VECTOR; break; }
                out[i] = val;
                j += size;
            }
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

// ---- observation summaries for the calls writer: one lane per pileup, the host's per-observation loop (sample_fields) as it stands
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

__global__ void obs_summary_kernel(DeviceCols cols, const uint32_t* __restrict__ obs_offset, const uint8_t* __restrict__ locus_flags, int64_t n_pileups, int n_samples,
                                   SumConsts K, PileSum* __restrict__ hdr, uint64_t* __restrict__ ent_key, uint32_t* __restrict__ ent_cnt,
                                   float* __restrict__ run_pm, uint32_t* __restrict__ run_len, uint32_t* __restrict__ cursor) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pileups) return;
    const uint32_t b = obs_offset[p], e = obs_offset[p + 1];
    const bool drop_nonstd = (locus_flags[p / n_samples] & VLR_LOCUS_REMOVE_NONSTANDARD) != 0;   // pileup.rs:26-43
    uint64_t keys[kSumMaxKeys];
    uint32_t cnts[kSumMaxKeys];
    uint32_t nk = 0, kept = 0, n_run = 0, overflow = 0;
    PileSum h;
    h.alt_n = 0; h.ref_n = 0; h.pad[0] = h.pad[1] = 0;
    for (int i = 0; i < kSumLetters; ++i) { h.alt_letter[i] = 0; h.ref_letter[i] = 0; h.alt_cnt[i] = 0; h.ref_cnt[i] = 0; }
    // the runs go straight to the bump array: reserve the worst case (one run per observation) only when a second run appears —
    // first count them
    {
        float last = 0.0f;
        bool have = false;
        for (uint32_t i = b; i < e; ++i) {
            const uint32_t f = cols.flags[i];
            if (drop_nonstd && ((f >> VLR_F_ORIENT_SHIFT) & 3u) == VLR_ORIENT_OTHER) continue;
            const float pm = cols.col[0][i];
            const bool same = have && pm == last;   // (exactly the test of the second pass: the reservation must match what it writes)
            if (!same) { n_run += 1; last = pm; have = true; }
        }
    }
    const uint32_t run_off = n_run ? atomicAdd(&cursor[1], n_run) : 0u;
    uint32_t r_at = 0, r_len = 0;
    float r_pm = 0.0f;
    bool r_have = false;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t f = cols.flags[i];
        const uint32_t orient = (f >> VLR_F_ORIENT_SHIFT) & 3u;
        if (drop_nonstd && orient == VLR_ORIENT_OTHER) continue;
        kept += 1;
        const float pmf = cols.col[0][i];
        {   // (the host compares doubles converted from these floats: pm != last_pm — NaN starts a run every time there, and here)
            const bool same = r_have && pmf == r_pm;
            if (!same) {
                if (r_have) { run_pm[run_off + r_at] = r_pm; run_len[run_off + r_at] = r_len; r_at += 1; }
                r_pm = pmf; r_len = 0; r_have = true;
            }
            r_len += 1;
        }
        const double pa = (double)cols.col[1][i], pr = (double)cols.col[2][i];
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
        const uint32_t strand = (f >> VLR_F_STRAND_SHIFT) & 3u, altloc = (f >> VLR_F_ALTLOCUS_SHIFT) & 3u;
        const bool hp_err = (f & VLR_F_HP_LEN_VALID) && ((f >> VLR_F_HP_LEN_SHIFT) & 0xffu) != 0;
        const uint64_t key = (uint64_t)s0 | ((uint64_t)s1 << 8) | ((uint64_t)((f & VLR_F_PAIRED) ? 1 : 0) << 16) |
                             ((uint64_t)(altloc > 2 ? 2 : altloc) << 17) | ((uint64_t)strand << 19) | ((uint64_t)orient << 21) |
                             ((uint64_t)((f & VLR_F_READPOS_MAJOR) ? 1 : 0) << 23) | ((uint64_t)((f & VLR_F_SOFTCLIPPED) ? 1 : 0) << 24) |
                             ((uint64_t)(hp_err ? 1 : 0) << 25) | ((uint64_t)(uint32_t)(cols.third[i] + 1) << 32);
        if (!overflow) {
            uint32_t k = 0;
            while (k < nk && keys[k] != key) ++k;
            if (k == nk) {
                if (nk == kSumMaxKeys) overflow = 1;
                else { keys[nk] = key; cnts[nk] = 1; nk += 1; }
            } else cnts[k] += 1;
        }
        {
            const bool to_alt = pa > pr;
            const uint32_t c0 = to_alt ? kl_alt : kl_ref;
            const uint8_t c = (uint8_t)(maxq ? c0 : d_lower(c0));
            uint8_t* letters = to_alt ? h.alt_letter : h.ref_letter;
            uint32_t* lc = to_alt ? h.alt_cnt : h.ref_cnt;
            uint8_t& n = to_alt ? h.alt_n : h.ref_n;
            uint32_t q = 0;
            while (q < n && letters[q] != c) ++q;
            if (q == n) { if (n < kSumLetters) { letters[n] = c; lc[n] = 1; n += 1; } else overflow = 1; }
            else lc[q] += 1;
        }
    }
    if (r_have) { run_pm[run_off + r_at] = r_pm; run_len[run_off + r_at] = r_len; r_at += 1; }
    const uint32_t ent_off = (nk && !overflow) ? atomicAdd(&cursor[0], nk) : 0u;
    if (!overflow)
        for (uint32_t k = 0; k < nk; ++k) { ent_key[ent_off + k] = keys[k]; ent_cnt[ent_off + k] = cnts[k]; }
    h.ent_off = ent_off; h.n_ent = overflow ? 0u : nk; h.run_off = run_off; h.n_run = r_at; h.kept = kept; h.overflow = overflow;
    hdr[p] = h;
}

}  // namespace
}  // namespace vlr

// ================================================================================================ one sample file on the device
struct vlr_dev_file {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t feed_stream = nullptr;   // H2D of compressed members + inflate kernel: runs beside the decode of the previous chunk
    hipStream_t copy_stream = nullptr;   // column copies the caller does not wait for (vlr_dev_file_copy_detached)
    bool feed_pending = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // around the inflate kernel of the feed in flight (measurement: vlr_dev_file_inflate_seconds)
    double inflate_s = 0.0;
    uint8_t* buf = nullptr;       // inflated stream: bytes [rd, wr) are buffered
    size_t cap = 0, rd = 0, wr = 0;
    uint8_t* spare = nullptr;     // the other half of the ping-pong (compaction never copies inside one allocation)
    size_t spare_cap = 0;
    uint8_t* d_comp = nullptr; size_t comp_cap = 0;
    vlr::InflateBlock* d_blocks = nullptr; int* d_status = nullptr; size_t blocks_cap = 0;
    std::vector<int> h_status; size_t pending_blocks = 0;
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
size_t dev_file_bytes(const vlr_dev_file* f) { return f->cap + f->spare_cap + f->comp_cap; }

void dev_file_free(vlr_dev_file* f) {
    (void)hipSetDevice(f->device);
    if (f->feed_stream) (void)hipStreamDestroy(f->feed_stream);
    if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
    if (f->ev0) (void)hipEventDestroy(f->ev0);
    if (f->ev1) (void)hipEventDestroy(f->ev1);
    if (f->stream) { (void)hipStreamSynchronize(f->stream); (void)hipStreamDestroy(f->stream); }
    void* all[] = {f->buf, f->spare, f->d_comp, f->d_blocks, f->d_status, f->d_anchor, f->d_landing, f->d_segbase, f->d_count, f->d_landc, f->d_starts, f->d_nout,
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
                f->rd = f->wr = 0; f->feed_pending = false; f->pending_blocks = 0; f->n_split = 0; f->inflate_s = 0.0;
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
    if (hipStreamCreateWithPriority(&f->stream, hipStreamNonBlocking, prio_hi) != hipSuccess || hipStreamCreateWithPriority(&f->feed_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) {
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
    if (!f || n_blocks <= 0) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    if (f->feed_pending) { const int rc = vlr_dev_file_feed_wait(f); if (rc != VLR_OK) return rc; }
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
        if (live) VLR_HIP_OK(hipMemcpyAsync(f->spare, f->buf + f->rd, live, hipMemcpyDeviceToDevice, st));
        std::swap(f->buf, f->spare); std::swap(f->cap, f->spare_cap);
        f->rd = 0; f->wr = live;
    }
    {   // compressed bytes and member list; the kernel may read kInflateInputSlack bytes beyond the last member
        int rc = dev_grow(f->d_comp, f->comp_cap, comp_bytes + vlr::kInflateInputSlack);
        if (rc) return rc;
        if ((size_t)n_blocks > f->blocks_cap) {
            size_t c1 = f->blocks_cap, c2 = f->blocks_cap;
            if ((rc = dev_grow(f->d_blocks, c1, (size_t)n_blocks))) return rc;
            if ((rc = dev_grow(f->d_status, c2, (size_t)n_blocks))) return rc;
            f->blocks_cap = c1 < c2 ? c1 : c2;
        }
    }
    VLR_HIP_OK(hipMemcpyAsync(f->d_comp, comp, comp_bytes, hipMemcpyHostToDevice, st));
    VLR_HIP_OK(hipMemsetAsync(f->d_comp + comp_bytes, 0, vlr::kInflateInputSlack, st));
    VLR_HIP_OK(hipMemcpyAsync(f->d_blocks, blocks, (size_t)n_blocks * sizeof(vlr::InflateBlock), hipMemcpyHostToDevice, st));
    if (!f->ev0) { (void)hipEventCreate(&f->ev0); (void)hipEventCreate(&f->ev1); }
    if (f->ev0) (void)hipEventRecord(f->ev0, st);
    const int lrc = vlr_launch_inflate_kernel(f->d_comp, f->d_blocks, n_blocks, f->buf + f->wr, f->d_status, st);
    if (lrc != 0) return dfail(VLR_ERR_HIP, "inflate kernel launch failed (hip error %s%lld)", "", lrc);
    if (f->ev1) (void)hipEventRecord(f->ev1, st);
    f->h_status.resize((size_t)n_blocks);
    VLR_HIP_OK(hipMemcpyAsync(f->h_status.data(), f->d_status, (size_t)n_blocks * sizeof(int), hipMemcpyDeviceToHost, st));
    f->pending_blocks = (size_t)n_blocks;
    f->wr += (size_t)inflated_bytes;
    f->feed_pending = true;
    return VLR_OK;
}

int vlr_dev_file_feed_wait(vlr_dev_file* f) {
    if (!f || !f->feed_pending) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipStreamSynchronize(f->feed_stream));
    f->feed_pending = false;
    if (f->ev0 && f->ev1) { float ms = 0.0f; if (hipEventElapsedTime(&ms, f->ev0, f->ev1) == hipSuccess) f->inflate_s += (double)ms * 1e-3; }
    for (size_t i = 0; i < f->pending_blocks; ++i)
        if (f->h_status[i] != 0) return dfail(VLR_ERR_INVALID_ARGUMENT, "%s in a BGZF member (inflate status %lld)", f->h_status[i] == vlr::INFL_CRC_MISMATCH ? "CRC32 checksum mismatch" : "corrupt DEFLATE stream", (long long)f->h_status[i]);
    f->pending_blocks = 0;
    return VLR_OK;
}

int vlr_dev_file_skip(vlr_dev_file* f, uint64_t bytes) {
    if (f->rd + bytes > f->wr) return dfail(VLR_ERR_INVALID_ARGUMENT, "device reader: skip beyond the buffered bytes");
    f->rd += (size_t)bytes;
    return VLR_OK;
}

int vlr_dev_file_split(vlr_dev_file* f, int64_t max_records, int n_contigs, int n_hdr_samples, const int8_t* field_of_key, int n_keys, int64_t* n_records,
                       const vlr::RecHost** rec_host, int* used_serial_walk) {
    VLR_HIP_OK(hipSetDevice(f->device));
    { const int rcw = vlr_dev_file_feed_wait(f); if (rcw != VLR_OK) return rcw; }
    *n_records = 0; *rec_host = nullptr;
    if (used_serial_walk) *used_serial_walk = 0;
    f->n_split = 0;
    const uint64_t avail = f->wr - f->rd;
    if (avail < 8 || max_records <= 0) return VLR_OK;
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
                           const vlr::SumConsts* k, vlr::PileSum* d_hdr, uint64_t* d_ent_key, uint32_t* d_ent_cnt, float* d_run_pm, uint32_t* d_run_len, uint32_t* d_cursor) {
    const int64_t P = n_loci * n_samples;
    if (P <= 0) return VLR_OK;
    VLR_HIP_OK(hipSetDevice(f->device));
    VLR_HIP_OK(hipMemsetAsync(d_cursor, 0, 8, f->stream));
    hipLaunchKernelGGL(vlr::obs_summary_kernel, dim3((unsigned)((P + 63) / 64)), dim3(64), 0, f->stream, *cols, d_obs_offset, d_locus_flags, P, n_samples, *k,
                       d_hdr, d_ent_key, d_ent_cnt, d_run_pm, d_run_len, d_cursor);
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
