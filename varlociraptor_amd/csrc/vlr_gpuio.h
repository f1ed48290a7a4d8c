// Device front door: what vlr_ingest.cpp (host side of the reader) and the kernels of vlr_inflate.hip /
// vlr_decode.hip share.  Internal: the C ABI of the reader stays vlr_obs_reader_* (include/vlr.h).
#pragma once
#include <cstddef>
#include <cstdint>

namespace vlr {

// one BGZF member (SAM spec 4.1): its raw DEFLATE payload inside the compressed bytes and where its inflated bytes go
struct InflateBlock {
    uint64_t src;    // byte offset of the DEFLATE stream (after the 18-byte member header) in the compressed buffer
    uint64_t dst;    // byte offset of the member's first inflated byte in the output buffer
    uint32_t clen;   // DEFLATE bytes (BSIZE + 1 - 18 - 8)
    uint32_t isize;  // inflated bytes (ISIZE, <= 65536)
    uint32_t crc;    // CRC32 of the inflated bytes (member trailer), checked on the device after the inflate (vlr_crc_kernel)
    uint32_t pad;
};
constexpr size_t kInflateInputSlack = 32768;
enum InflateStatus : int {
    INFL_OK = 0, INFL_BAD_BLOCK_TYPE = 1, INFL_BAD_STORED = 2, INFL_BAD_CODE_LENGTHS = 3, INFL_OVERSUBSCRIBED = 4, INFL_BAD_SYMBOL = 5,
    INFL_BAD_DISTANCE = 6, INFL_OUTPUT_OVERRUN = 7, INFL_INPUT_OVERRUN = 8, INFL_SIZE_MISMATCH = 9, INFL_CRC_MISMATCH = 10,
};

// what the INFO scan leaves per record for the decode and cold-copy kernels
constexpr int kGpuVec = 18;    // the vector fields (FD_N_VEC of VlrObsField below)
constexpr int kColdSegs = 6;   // [0] ID, alleles, FILTER; [1..5] the INFO entries IMPRECISE, EVENT, MATEID, HETEROZYGOSITY, SOMATIC_EFFECTIVE_MUTATION_RATE
struct RecDesc {
    uint64_t start;                 // offset of the record in the inflated stream
    uint32_t l_shared;
    uint32_t n_obs;                 // u64 length of PROB_MAPPING
    uint32_t cold_bytes;            // size of the record's cold copy (8 + 24 + segments)
    uint32_t n_cold_info;
    uint32_t seg_off[kColdSegs], seg_len[kColdSegs];   // from the record start; length 0 = absent
    uint32_t voff[kGpuVec], vn[kGpuVec];               // payload offset from the record start, element count
    uint8_t vstride[kGpuVec];                          // bytes per element (1: int8, sign-extended to the u16 word; 2; 4); 0 = absent
    uint8_t pad[6];
};
// what the host reads back per record
struct RecHost { uint32_t n_obs, cold_bytes, status, flags; };  // flags bit 0: homopolymer fields present (is_homopolymer_indel)
enum RecStatus : uint32_t {
    REC_OK = 0, REC_TRUNCATED = 1u << 0, REC_BAD_ID = 1u << 1, REC_BAD_ALLELE = 1u << 2, REC_BAD_FILTER = 1u << 3, REC_BAD_INFO = 1u << 4,
    REC_MISSING_FIELD = 1u << 5, REC_BAD_LENGTHS = 1u << 6, REC_BAD_VECTOR = 1u << 7,
};
struct DeviceCols { float* col[9]; uint32_t* flags; int32_t* third; };

// Per-pileup summary of the observations for the calls writer (Call::write_final_record, calling/variants/mod.rs:233-360): what
// sample_fields of vlr_ingest.cpp derives per observation — the OBS text itself (generalized_cigar over the packed observation keys,
// utils/mod.rs:122-156: distinct keys counted, ordered by class, count and first appearance, written as text by obs_text_kernel), the
// Kass-Raftery letters of the alt- and ref-supporting observations (SAOBS, SROBS: at most twelve distinct items, formatted on the
// host), the kept observations and the run-lengths of prob_mapping over them (DP is a sum of exp(prob_mapping) in observation order:
// the host repeats the additions with its own exp) — so that the host writes records without touching the columns.
constexpr int kSumLetters = 12;    // N B P S V E in two cases
constexpr int kSumMaxObs = 1024;   // kept observations of one pileup the kernel ranks in LDS (more: `overflow`, the host counts from the columns)
constexpr int kSumItemMax = 40;    // bytes of one OBS item at most (count, two score letters, third-allele evidence, seven flag characters)
struct PileSum {
    uint32_t obs_off, obs_len;    // OBS text [obs_off, obs_off + obs_len) of the text buffer (empty pileup: length 0, the writer's ".")
    uint32_t n_item;              // distinct observation keys (items of the text)
    uint32_t run_off, n_run;      // runs of equal prob_mapping over the kept observations: the first one below, runs 1 .. n_run - 1 at
    float run0_pm;                //   [run_off, run_off + n_run - 1) of the run arrays
    uint32_t run0_len;
    uint32_t kept, overflow;
    uint8_t alt_n, ref_n, pad[2];
    uint8_t alt_letter[kSumLetters], ref_letter[kSumLetters];
    uint32_t alt_cnt[kSumLetters], ref_cnt[kSumLetters];
};
struct SumConsts { double ln3, ln20, ln150, eps; };

}  // namespace vlr

// INFO fields of an observation record (format v15), in the order both decoders index them
enum VlrObsField {
    FD_PROB_MAPPING, FD_PROB_REF, FD_PROB_ALT, FD_PROB_MISSED, FD_PROB_SAMPLE_ALT, FD_PROB_DOUBLE_OVERLAP, FD_PROB_HIT_BASE,
    FD_STRAND, FD_ORIENT, FD_READPOS, FD_ALTLOCUS, FD_SOFTCLIPPED, FD_PAIRED, FD_MAX_MAPQ, FD_HP_ART, FD_HP_VAR, FD_HP_LEN, FD_THIRD,
    FD_N_VEC,
    FD_IMPRECISE = FD_N_VEC, FD_EVENT, FD_MATEID, FD_HET, FD_SOM, FD_N
};
static_assert(FD_N_VEC == vlr::kGpuVec, "field table");

extern "C" {
// vlr_inflate.hip: n_blocks BGZF members, one wave each.  d_comp must be readable kInflateInputSlack bytes beyond the last member (the
// prefetch window, and the distance a corrupt stream can run before the kernel's next position check stops it).
int vlr_launch_inflate_kernel(const uint8_t* d_comp, const vlr::InflateBlock* d_blocks, int n_blocks, uint8_t* d_out, int* d_status, void* stream);

// vlr_decode.hip: one sample file's side of the device reader (buffers and stream owned by the object; see the .hip for the stages)
struct vlr_dev_file;
int vlr_dev_file_create(int device, vlr_dev_file** out);
void vlr_dev_file_destroy(vlr_dev_file* f);
// destroyed reader objects are parked for the next reader (their device buffers stay allocated, bounded by VLR_INGEST_PARK_MB): free them all
void vlr_dev_file_trim();
// bytes of complete-or-not record data currently buffered behind the read position
uint64_t vlr_dev_file_buffered(const vlr_dev_file* f);
// append the inflated bytes of n_blocks members (src offsets relative to comp) behind the buffered ones: enqueued on the file's feed
// stream, beside whatever the decode stream still runs and behind earlier feeds (up to four in flight); comp stays valid until the feed
// has been waited for, blocks is copied
int vlr_dev_file_feed(vlr_dev_file* f, const uint8_t* comp, size_t comp_bytes, const vlr::InflateBlock* blocks, int n_blocks, uint64_t inflated_bytes);
// (the compressed bytes in n_pieces host pieces, uploaded one behind the other: segments of a page-locked staging ring)
int vlr_dev_file_feed_pieces(vlr_dev_file* f, const uint8_t* const* piece, const size_t* piece_bytes, int n_pieces, const vlr::InflateBlock* blocks, int n_blocks,
                             uint64_t inflated_bytes);
// wait for every feed in flight / for the oldest one / for feeds, oldest first, until `bytes` behind the read position are inflated and
// checked (the split only looks at those); feeds still in flight; inflated bytes behind the read position
int vlr_dev_file_feed_wait(vlr_dev_file* f);
int vlr_dev_file_feed_wait_oldest(vlr_dev_file* f);
int vlr_dev_file_wait_ready(vlr_dev_file* f, uint64_t bytes);
int vlr_dev_file_feeds_in_flight(const vlr_dev_file* f);
uint64_t vlr_dev_file_ready(const vlr_dev_file* f);
// skip `bytes` (the BCF header) at the read position
// seconds the inflate kernels of this file ran (HIP events on the feed stream), summed since the last reset
double vlr_dev_file_inflate_seconds(vlr_dev_file* f, int reset);
int vlr_dev_file_skip(vlr_dev_file* f, uint64_t bytes);
// split the buffered bytes into records (at most max_records), scan their INFO entries; *n_records complete records found,
// rec_host[0..n) (array owned by the object, valid until the next split) their counts.  Synchronises the file's stream.
int vlr_dev_file_split(vlr_dev_file* f, int64_t max_records, int n_contigs, int n_hdr_samples, const int8_t* field_of_key, int n_keys, int64_t* n_records,
                       const vlr::RecHost** rec_host, int* used_serial_walk);
// a shard's window begins inside a record: read position to the first record start (a guess the split's verified walk confirms)
int vlr_dev_file_anchor_first(vlr_dev_file* f, int n_contigs, int n_hdr_samples, uint64_t* skipped);
// record starts of the last split: n + 1 offsets from the read position (host copy, valid until the next split)
const uint64_t* vlr_dev_file_starts(const vlr_dev_file* f, int64_t* n);
// decode records [0, n) into the merged columns: record r of this file goes to observation offset d_obs_offset[r * n_samples + sample]
int vlr_dev_file_decode(vlr_dev_file* f, int64_t n, const uint32_t* d_obs_offset, int n_samples, int sample, const vlr::DeviceCols* cols);
// cold copies of records [0, n) -> host buffer (cold_off[r] = prefix sum of RecHost.cold_bytes, cold_off[n] bytes in all); asynchronous
int vlr_dev_file_cold(vlr_dev_file* f, int64_t n, const uint64_t* cold_off, uint8_t* host_out);
// consume records [0, n): the bytes behind them stay buffered for the next split (kernels already enqueued keep their pointers)
int vlr_dev_file_consume(vlr_dev_file* f, int64_t n);
int vlr_dev_file_sync(vlr_dev_file* f);
// record-level error bits of the last decode (REC_*), and the first record that has any
int vlr_dev_file_errors(vlr_dev_file* f, int64_t n, uint32_t* status_or, int64_t* first_bad);
void* vlr_dev_file_stream(vlr_dev_file* f);
// asynchronous copy on the file's stream (to_device != 0: host -> device)
int vlr_dev_file_copy(vlr_dev_file* f, void* dst, const void* src, size_t bytes, int to_device);
// observation summaries of n_pileups pileups (locus-major): hdr[n_pileups], the OBS text (placed by a cursor: cursor[0] = text bytes, in
// 16-byte steps) and the prob_mapping runs behind the first one of a pileup (cursor[1]); cursor[2] = pileups left to the columns;
// cursor[0..4) zeroed by the call.  Everything is enqueued on the file's stream; d_text holds text_cap bytes (a pileup whose text does
// not fit is marked `overflow`), run_* / item_* at least n_obs elements, text_len / text_off n_pileups.  max_pileup_obs: observations of
// the largest pileup (sizes the kernel's LDS).
int vlr_dev_file_summaries(vlr_dev_file* f, const vlr::DeviceCols* cols, const uint32_t* d_obs_offset, const uint8_t* d_locus_flags, int64_t n_loci, int n_samples,
                           uint32_t max_pileup_obs, const vlr::SumConsts* k, vlr::PileSum* d_hdr, uint8_t* d_text, uint32_t text_cap, float* d_run_pm,
                           uint32_t* d_run_len, uint64_t* d_item_key, uint32_t* d_item_cnt, uint32_t* d_text_len, uint32_t* d_text_off, uint32_t* d_cursor);
// FORMAT/AFD text of n_lists lists (vlr_results.afd_text, include/vlr.h): one wave per list; span[2 p] = offset, span[2 p + 1] = length
// (0xffffffff: left to the arrays); rank: n_lists * capacity entries of scratch; cursor[0] text bytes (16-byte steps), cursor[2] lists not
// formatted — cursor[0..4) zeroed by the call.
int vlr_launch_afd_text(const int32_t* d_count, const double* d_vaf, const double* d_lnprob, int64_t n_lists, int capacity, uint8_t* d_text, uint32_t text_cap,
                        uint32_t* d_span, uint16_t* d_rank, uint32_t* d_cursor, void* stream);
// synchronous device -> host copy outside any stream (lazy fetch of a table's columns)
int vlr_dev_copy_to_host(int device, void* dst, const void* src, size_t bytes);
// device -> host copy nobody waits for until vlr_dev_event_wait(*event_out) (the caller owns the event)
int vlr_dev_file_copy_detached(vlr_dev_file* f, void* dst, const void* src, size_t bytes, void** event_out);
// more copies of one group on the copy stream (no event of their own); a mark of what the copy stream has been given so far (created on
// first use, owned by the caller) and the main stream waiting for one before it overwrites the sources
int vlr_dev_file_copy_more(vlr_dev_file* f, void* dst, const void* src, size_t bytes);
int vlr_dev_file_copy_mark(vlr_dev_file* f, void** mark);
int vlr_dev_file_wait_mark(vlr_dev_file* f, void* mark);
int vlr_dev_event_wait(int device, void* event);
void vlr_dev_event_destroy(int device, void* event);
// column storage of a table: `bytes` of device memory and as many page-locked host bytes
int vlr_dev_slab_alloc(int device, size_t bytes, void** d, void** h);
void vlr_dev_slab_free(int device, void* d, void* h);
}
